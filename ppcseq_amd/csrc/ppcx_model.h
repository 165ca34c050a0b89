// ppcx_model.h -- the negative-binomial hierarchical log density of ppcseq, restructured for a
// streaming per-gene reduction on CDNA4.
//
// What is computed is exactly the `target` of inst/stan/negBinomial_MPI.stan (reference file:line):
//   transforms :183-197,:203      priors :210-223      likelihood lp_reduce :58-120 via map_rect :226-240
//   coefficient assembly merge_coefficients :122-139 and X*alpha :205 (fused away: eta is formed per cell)
//
// How it is computed (the MI355X-first part, see DESIGN.md "log-likelihood kernel"). With log phi = -sigma_raw exactly,
// t = eta + sigma_raw, u = exp(t), w = 1 + u, x = y + phi, xf = x/phi = 1 + y/phi and rho = xf/w = x/(phi w):
//   * Stirling at x AND at phi, leading terms cancelled analytically (dlt, dps = the "Stirling excess" of phi,
//     ppcx_math.h):
//       NB2log(y|eta,phi) + lgamma(y+1) = y eta - y + (y + phi) ln rho - 1/2 ln xf + lg_tail(1/x) - dlt
//       d/deta = y - x u/w = phi (rho - 1)
//       d/dphi = psi(x) - psi(phi) - ln w + 1 - x/(phi w) = ln rho - dg_tail(1/x) + dps + 1 - rho
//     so a cell needs ONE logarithm (ln rho), one reciprocal (of w xf: it yields 1/w and 1/xf) and the two tails;
//     sum_s ln xf is the logarithm of a running product (one multiply per cell, renormalised every few cells);
//     the tails are degree-4 polynomials in 1/x^2 valid for every x >= 8 (ppcx_math.h stirling_tails);
//   * sum_s y eta, sum_s y are per-gene SUFFICIENT STATISTICS (SyE, SyX, Sy) precomputed once, so the large
//     cancelling terms never go through the per-cell loop; sum_s lgamma(y+1) is a per-gene constant of the data (Lg1);
//   * cells with y <= 7 do not use Stirling at x: lgamma(y + phi) - lgamma(phi) = sum_{k<y} ln(phi + k) depends on the
//     count and on phi only, so the gene adds it once per k = 0..6, weighted by the number M_k of such cells with y > k
//     (counted at upload); per cell only - (y + phi) ln w and rho remain: one reciprocal, one logarithm;
//   * for genes without slope terms (g >= K, X[,1] == 1) exp(t) factorises into E_s * A_g with
//     E_s = exp(exposure_s) staged in LDS and A_g = exp(intercept_g + sigma_raw_g): no per-cell exp;
//   * the row sweep evaluates only the cells with y >= 8 (then x >= 8 whatever phi is: one regime, no test); the cells
//     with 0 <= y <= 7 are kept in a per-gene list built at upload and evaluated by a second, short loop;
//   * excluded cells (to_exclude, R/utilities.R:321-359, subtracted at .stan:105-115) are stored as count = -1, are in
//     neither loop, and are left out of the sufficient statistics.
#pragma once
#include "ppcx_math.h"

namespace ppcx {

constexpr int kMaxC = 8;        // design-matrix columns supported by the kernels

struct Dims {
  int G, S, C, K, D;
  int off_intercept, off_alpha1, off_alpha2, off_sigma_raw, off_tail;  // Stan declaration order (.stan:183-197)
  int x0_is_one;                // X[,1] == 1 (model.matrix intercept column, R/utilities.R:887-900)
  int x1_binary;                // C >= 2 and every slope column of X in {0, 1} (model.matrix of factors: `~ Label`, a multi-level
                                // factor, `~ a + b`): e^t of a gene with slopes is E_s A_g times exp(slope_c) of the sample's
                                // columns -- no per-cell exp for the checked genes either (C == 2: A_g or A1_g by the group)
  int Gt, Kt, g0, k0;           // gene shard: totals of the whole problem and this shard's first gene / checked gene
  double lambda_mu_mu;
};

PPCX_HD Dims make_dims(int G, int S, int C, int K, double lambda_mu_mu) {
  Dims d;
  d.G = G; d.S = S; d.C = C; d.K = K;
  d.off_intercept = 3;
  d.off_alpha1 = 3 + G;
  d.off_alpha2 = d.off_alpha1 + K;
  d.off_sigma_raw = d.off_alpha2 + (C > 2 ? C - 2 : 0) * K;
  d.off_tail = d.off_sigma_raw + G;
  d.D = d.off_tail + 3;
  d.x0_is_one = 1; d.x1_binary = 0; d.lambda_mu_mu = lambda_mu_mu;
  d.Gt = G; d.Kt = K; d.g0 = 0; d.k0 = 0;
  return d;
}
// Index of local coordinate i in the unconstrained vector of the WHOLE problem (Stan order). Used only as
// the Philox stream id, so that a gene-sharded run draws exactly what the unsharded run draws.
PPCX_HD int global_flat(const Dims& d, int i) {
  if (d.Gt == d.G) return i;
  const int n2 = d.C > 2 ? d.C - 2 : 0;
  if (i < d.off_intercept) return i;
  if (i < d.off_alpha1) return 3 + d.g0 + (i - d.off_intercept);
  if (i < d.off_alpha2) return 3 + d.Gt + d.k0 + (i - d.off_alpha1);
  if (i < d.off_sigma_raw) return 3 + d.Gt + d.Kt + n2 * d.k0 + (i - d.off_alpha2);
  const int sr_t = 3 + d.Gt + d.Kt + n2 * d.Kt;
  if (i < d.off_tail) return sr_t + d.g0 + (i - d.off_sigma_raw);
  return sr_t + d.Gt + (i - d.off_tail);
}
// flat index of the k-th hyper-parameter, k = 0..5 = lambda_mu, lambda_sigma, lambda_skew,
// sigma_slope, sigma_intercept, sigma_sigma
PPCX_HD int hyper_index(const Dims& d, int k) { return k < 3 ? k : d.off_tail + (k - 3); }
// flat index of coefficient c (0 = intercept, 1 = alpha_sub_1, >= 2 = alpha_2 row c-2) of gene g
PPCX_HD int coef_index(const Dims& d, int c, int g) {
  return c == 0 ? d.off_intercept + g : (c == 1 ? d.off_alpha1 + g : d.off_alpha2 + (c - 2) + (d.C - 2) * g);
}

struct Hyper {                  // constrained hyper-parameters + derived constants
  double lambda_mu, lambda_sigma, lambda_skew, sigma_slope, sigma_intercept, sigma_sigma;
  double xi, inv_om, log_om, inv_ss, inv_ss2, log_ss;
};
PPCX_HD Hyper make_hyper(const double* u6, double lambda_mu_mu) {
  Hyper h;
  h.lambda_mu = u6[0] + lambda_mu_mu;        // <offset = lambda_mu_mu>  (.stan:183)
  h.lambda_sigma = fast_exp(u6[1]);          // <lower = 0>              (.stan:184)
  h.lambda_skew = u6[2];
  h.sigma_slope = -fast_exp(u6[3]);          // <upper = 0>              (.stan:195)
  h.sigma_intercept = u6[4];
  h.sigma_sigma = fast_exp(u6[5]);           // <lower = 0>              (.stan:197)
  h.xi = h.lambda_mu + lambda_mu_mu;         // offset enters twice by construction (.stan:219)
  h.inv_om = fast_rcp(h.lambda_sigma); h.log_om = u6[1];       // runs on the step kernel's critical path: no division
  h.inv_ss = fast_rcp(h.sigma_sigma); h.inv_ss2 = h.inv_ss * h.inv_ss; h.log_ss = u6[5];
  return h;
}

// everything a gene's lanes need that does not depend on the sample
template <int CM>
struct GeneParams {
  double coef[CM];              // intercept, alpha_sub_1, alpha_2...   (zero beyond K, .stan:133-135)
  double sigma_raw, phi, invphi;  // phi = exp(-sigma_raw) (.stan:203), invphi = 1/phi
  double A, A1;                 // A = exp(intercept + sigma_raw), A1 = A exp(slope) (two-group designs)
  double dlt, dps;              // Stirling excess of phi (ppcx_math.h stirling_excess)
};

// per-lane partial sums over the cells of one gene (see the header comment for the algebra)
template <int CM>
struct CellAcc {
  double SA;                    // sum y ln(arg); arg = rho for the cells of the row sweep (y >= 8), 1/w for the list cells
  double SL;                    // sum ln(arg)
  double TL;                    // sum lg_tail(1/x) over the cells of the row sweep (their - dlt is left to the gene's
                                // epilogue, which knows their number)
  double TD;                    // sum dg_tail(1/x) over the cells of the row sweep (- dps likewise)
  double Px; int Pxe;           // running product of xf over the cells of the row sweep: mantissa and binary exponent
  double Sr;                    // sum rho
  double Tx[CM];                // sum X_sc rho  (paths with a per-cell design row only)
  PPCX_HD void zero() { SA = SL = TL = TD = Sr = 0.0; Px = 1.0; Pxe = 0;
#pragma unroll
    for (int c = 0; c < CM; ++c) Tx[c] = 0.0; }
  PPCX_HD void renorm() {       // keep the running product's exponent in range: called every few cells
#if defined(__HIP_DEVICE_COMPILE__)
    Pxe += __builtin_amdgcn_frexp_exp(Px);
    Px = __builtin_amdgcn_frexp_mant(Px);
#else
    int e; Px = frexp(Px, &e); Pxe += e;
#endif
  }
};
constexpr int kRenormEvery = 8;  // cells between renormalisations: a factor stays below 2^120 for phi > 2^-89

#if defined(__HIP_DEVICE_COMPILE__)
#define PPCX_WAVE_ANY(p) (__any(p) != 0)
#define PPCX_WAVE_ALL(p) (__all(p) != 0)
// placed in the rarely taken side of a wave-uniform branch: keeps hipcc from turning the branch into selects that
// execute both sides for every cell (it does so for short bodies: the renormalisation, the longer Stirling tails)
#define PPCX_KEEP_BRANCH() asm volatile("" ::: "memory")
// makes a value opaque at this point, so that what is computed from it in a rarely taken branch stays in that branch
// instead of being hoisted into registers that the hot loop then has to carry
#define PPCX_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define PPCX_WAVE_ANY(p) (p)
#define PPCX_WAVE_ALL(p) (p)
#define PPCX_KEEP_BRANCH() ((void)0)
#define PPCX_OPAQUE(x) ((void)0)
#endif

#if defined(__HIP_DEVICE_COMPILE__) && !defined(PPCX_NO_ASM_CELL)
// The common regime of cell_eval on the device, written as three register-only blocks of gfx950 instructions (no
// memory operation and no wait inside: the table reads between them are the compiler's). hipcc's own code for the same
// C++ (cell_eval below, which the host emulation runs) spends 62 vector issue slots on it -- every Horner step of the
// polynomials becomes a register copy plus v_fmac, and the joins of the regime branches add more copies; this is 41:
// one v_fma per Horner step with the coefficient in an SGPR pair (a VOP3 instruction may read one SGPR operand, so the
// first step's second coefficient sits in a VGPR). v_rcp_f64 is a transcendental-unit instruction: the instruction after it
// must not read its result (one wait state on gfx940-class chips), hence the independent product update there.
template <int CM, int TAIL>
__device__ __forceinline__ double cell_big_dev(int y, double e, double A, const GeneParams<CM>& gp, const double* tab,
                                               CellAcc<CM>& a) {
  double yd, w, xf, r2, rx, rho;
  asm("v_cvt_f64_u32_e32 %[yd], %[y]\n\t"
      "v_fma_f64 %[w], %[e], %[A], 1.0\n\t"
      "v_fma_f64 %[xf], %[yd], %[ip], 1.0\n\t"
      "v_mul_f64 %[r2], %[w], %[xf]\n\t"             // m = w xf
      "v_rcp_f64_e32 %[rx], %[r2]\n\t"               // q ~ 1/m
      "s_nop 0\n\t"                                  // transcendental result: one wait state before its first reader
      "v_fma_f64 %[r2], -%[r2], %[rx], 1.0\n\t"      // 1 - m q
      "v_fma_f64 %[rx], %[rx], %[r2], %[rx]\n\t"     // q (one Newton step)
      "v_mul_f64 %[r2], %[rx], %[xf]\n\t"            // 1/w = q xf
      "v_mul_f64 %[rx], %[rx], %[w]\n\t"             // 1/xf = q w
      "v_mul_f64 %[rho], %[xf], %[r2]\n\t"           // rho = xf/w
      "v_mul_f64 %[rx], %[rx], %[ip]\n\t"            // 1/x = (1/xf)(1/phi)
      "v_mul_f64 %[r2], %[rx], %[rx]"
      : [yd] "=&v"(yd), [w] "=&v"(w), [xf] "=&v"(xf), [r2] "=&v"(r2), [rx] "=&v"(rx), [rho] "=&v"(rho)
      : [y] "v"(y), [e] "v"(e), [A] "v"(A), [ip] "v"(gp.invphi));
  const int j = (int)(dbl_bits(rho) >> (52 - kLogTabBits)) & (kLogTabSize - 1);
  const double cinv = tab[j], logc = tab[kLogTabSize + j];
  __builtin_amdgcn_sched_barrier(0);     // the table reads are issued here, before the tails: their latency hides behind those
  a.Sr += rho;
  a.Px *= xf;
  double t, d;
  if (TAIL == 4) {                       // every x >= 8
    const double F3 = -5.94317590856362882e-04, G3 = -4.15672846375406482e-03;
    asm("v_fma_f64 %[t], %[r2], %[F4], %[F3]\n\t"
        "v_fma_f64 %[d], %[r2], %[G4], %[G3]\n\t"
        "v_fma_f64 %[t], %[r2], %[t], %[F2]\n\t"
        "v_fma_f64 %[d], %[r2], %[d], %[G2]\n\t"
        "v_fma_f64 %[t], %[r2], %[t], %[F1]\n\t"
        "v_fma_f64 %[d], %[r2], %[d], %[G1]\n\t"
        "v_fma_f64 %[t], %[r2], %[t], %[F0]\n\t"
        "v_fma_f64 %[d], %[r2], %[d], %[G0]"
        : [t] "=&v"(t), [d] "=&v"(d)
        : [r2] "v"(r2), [F3] "v"(F3), [G3] "v"(G3),
          [F4] "s"(7.72651446721163817e-04), [F2] "s"(7.93645716111539040e-04), [F1] "s"(-2.77777776791245188e-03),
          [F0] "s"(8.33333333333302478e-02), [G4] "s"(6.82627523986508236e-03), [G2] "s"(3.96819926156938719e-03),
          [G1] "s"(-8.33333322714054948e-03), [G0] "s"(8.33333333333001886e-02));
  } else if (TAIL == 2) {                // x >= kTailX2: degree 2 (ppcx_math.h stirling_tails_short)
    const double F1 = kStirlingF2[1], G1 = kStirlingG2[1];
    asm("v_fma_f64 %[t], %[r2], %[F2], %[F1]\n\t"
        "v_fma_f64 %[d], %[r2], %[G2], %[G1]\n\t"
        "v_fma_f64 %[t], %[r2], %[t], %[F0]\n\t"
        "v_fma_f64 %[d], %[r2], %[d], %[G0]"
        : [t] "=&v"(t), [d] "=&v"(d)
        : [r2] "v"(r2), [F1] "v"(F1), [G1] "v"(G1),
          [F2] "s"(kStirlingF2[2]), [F0] "s"(kStirlingF2[0]), [G2] "s"(kStirlingG2[2]), [G0] "s"(kStirlingG2[0]));
  } else {                               // x >= kTailX1: degree 1
    const double F0 = kStirlingF1[0], G0 = kStirlingG1[0];
    asm("v_fma_f64 %[t], %[r2], %[F1], %[F0]\n\t"
        "v_fma_f64 %[d], %[r2], %[G1], %[G0]"
        : [t] "=&v"(t), [d] "=&v"(d)
        : [r2] "v"(r2), [F0] "v"(F0), [G0] "v"(G0), [F1] "s"(kStirlingF1[1]), [G1] "s"(kStirlingG1[1]));
  }
  a.TL = fma(rx, t, a.TL);
  a.TD = fma(r2, d, a.TD);
  a.TD = fma(0.5, rx, a.TD);
  double l;
  {
    double m, ed; int ex;
    const double L4 = -0.25;
    asm("v_frexp_mant_f64_e32 %[m], %[rho]\n\t"
        "v_frexp_exp_i32_f64_e32 %[ex], %[rho]\n\t"
        "v_fma_f64 %[m], %[m], %[cinv], -1.0\n\t"      // r = m/c - 1
        "v_cvt_f64_i32_e32 %[ed], %[ex]\n\t"
        "v_fma_f64 %[p], %[m], %[L5], %[L4]\n\t"       // log1p(r) = r (1 - r/2 + r^2/3 - r^3/4 + r^4/5)
        "v_fma_f64 %[p], %[m], %[p], %[L3]\n\t"
        "v_fma_f64 %[p], %[m], %[p], -0.5\n\t"
        "v_fma_f64 %[p], %[m], %[p], 1.0\n\t"
        "v_fma_f64 %[p], %[m], %[p], %[logc]\n\t"
        "v_fma_f64 %[p], %[ed], %[LN2], %[p]"            // l = ln rho
        : [m] "=&v"(m), [ex] "=&v"(ex), [ed] "=&v"(ed), [p] "=&v"(l)
        : [rho] "v"(rho), [cinv] "v"(cinv), [logc] "v"(logc), [L4] "v"(L4),
          [L5] "s"(0.2), [L3] "s"(1.0 / 3.0), [LN2] "s"(6.93147180559945286227e-01));
  }
  a.SA = fma(yd, l, a.SA);
  a.SL += l;
  return rho;
}
#endif

// One cell of the row sweep (count y >= 8, hence x = y + phi >= 8 for every phi) with e^t = e A: adds the cell to the
// sums and returns rho = x/(phi w). Straight-line code: one logarithm, one reciprocal, the two tails.
// TAIL: degree of the Stirling-tail polynomials -- 4 for every x >= 8; 2 / 1 for cells known to have x >= kTailX2 / kTailX1
// (whole passes of genes whose smallest row-sweep count is that large: the host orders the genes accordingly)
template <int CM, int TAIL = 4>
PPCX_HD double cell_eval(int y, double e, double A, const GeneParams<CM>& gp, const double* tab, CellAcc<CM>& a) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PPCX_NO_ASM_CELL)
  return cell_big_dev<CM, TAIL>(y, e, A, gp, tab, a);
#else
  const double yd = (double)y;
  const double w = fma(e, A, 1.0);
  const double xf = fma(yd, gp.invphi, 1.0);
  // 1/w and 1/xf from ONE hardware reciprocal: q = 1/(w xf), 1/w = q xf, 1/xf = q w (v_rcp_f64 is quarter rate)
  const double q = fast_rcp(w * xf), rw = q * xf, rx = (q * w) * gp.invphi;
  const double rho = xf * rw;
  const double l = table_log(rho, tab);
  a.SA = fma(yd, l, a.SA);
  a.SL += l;
  a.Sr += rho;
  a.Px *= xf;
  const double r2 = rx * rx;
  double t, dd;
  if (TAIL == 4) {
    t = fma(r2, 7.72651446721163817e-04, -5.94317590856362882e-04);     // stirling_tails (ppcx_math.h), fused with the sums
    t = fma(r2, t, 7.93645716111539040e-04);
    t = fma(r2, t, -2.77777776791245188e-03);
    t = fma(r2, t, 8.33333333333302478e-02);
    dd = fma(r2, 6.82627523986508236e-03, -4.15672846375406482e-03);
    dd = fma(r2, dd, 3.96819926156938719e-03);
    dd = fma(r2, dd, -8.33333322714054948e-03);
    dd = fma(r2, dd, 8.33333333333001886e-02);
  } else if (TAIL == 2) {
    t = fma(r2, fma(r2, kStirlingF2[2], kStirlingF2[1]), kStirlingF2[0]);
    dd = fma(r2, fma(r2, kStirlingG2[2], kStirlingG2[1]), kStirlingG2[0]);
  } else {
    t = fma(r2, kStirlingF1[1], kStirlingF1[0]);
    dd = fma(r2, kStirlingG1[1], kStirlingG1[0]);
  }
  a.TL = fma(rx, t, a.TL);
  a.TD = fma(0.5, rx, fma(r2, dd, a.TD));
  return rho;
#endif
}

// One cell of the low-count list (0 <= y <= 7). lgamma(y + phi) - lgamma(phi) = sum_{k<y} ln(phi + k) depends on the
// count and on phi, not on the sample: the gene adds it once per value of k, weighted by the number of its list
// cells with y > k (low_terms below). What is left per cell is the sample part,
//   - (y + phi) ln w   and   rho = (1 + y/phi)/w :
// one reciprocal, one logarithm (of 1/w), no product, no tails.
template <int CM>
PPCX_HD double cell_eval_low(int y, double e, double A, const GeneParams<CM>& gp, const double* tab, CellAcc<CM>& a) {
  const double yd = (double)y;
  const double w = fma(e, A, 1.0);
  const double q = fast_rcp(w);
  const double l = table_log(q, tab);                    // - ln w
  a.SA = fma(yd, l, a.SA);
  a.SL += l;
  const double rho = fma(yd, gp.invphi, 1.0) * q;
  a.Sr += rho;
  return rho;
}
// The count part of the gene's list cells, term k = 0..6 (M = number of list cells with y > k):
//   sum over the cells of [lgamma(y + phi) - lgamma(phi) + y sigma_raw + y] = sum_k M_k [ln(1 + k/phi) + 1]
//   (sigma_raw = - ln phi; the "+ y" undoes the "- y" that gene_close subtracts for every cell, a Stirling term)
//   sum over the cells of [psi(y + phi) - psi(phi)]                        = sum_k M_k (1/phi) / (1 + k/phi)
PPCX_HD void low_terms(int k, double M, double invphi, const double* tab, double* lik, double* dph) {
  const double z = fma((double)k, invphi, 1.0);
  *lik = fma(M, table_log(z, tab) + 1.0, *lik);
  *dph = fma(M * invphi, fast_rcp(z), *dph);
}

// What the log-likelihood kernel hands to the close kernel per gene (sums over the gene's non-excluded cells):
//   lik = sum [ (y + phi) ln rho - 1/2 ln xf + lg_tail - dlt ]   (the exact-recurrence cells accordingly)
//   dph = sum [ ln rho - dg_tail + dps ]  = sum [ psi(x) - psi(phi) - ln w ]
//   Sr  = sum rho,  Tx[c] = sum X_sc rho
template <int CM>
struct GeneSumsV { double lik, dph, Sr, Tx[CM]; };
template <int CM> struct GeneSums { static constexpr int N = 3 + CM; };

// a lane's share of the gene: fold the accumulators of its cells into the hand-over sums. The Stirling excess of phi
// that every cell of the row sweep owes (n_hi = their number in the whole gene) is added by ONE lane of the gene.
template <int CM>
PPCX_HD void cell_acc_close(const GeneParams<CM>& gp, CellAcc<CM>& a, const double* tab, double n_hi_if_first_lane,
                            double low_lik, double low_dph, GeneSumsV<CM>* o) {
  a.renorm();
  a.TL = fma(-n_hi_if_first_lane, gp.dlt, a.TL);
  a.TD = fma(-n_hi_if_first_lane, gp.dps, a.TD);
  const double lPx = fma((double)a.Pxe, 6.93147180559945286227e-01, table_log(a.Px, tab));
  o->lik = ((a.SA + gp.phi * a.SL) - 0.5 * lPx + a.TL) + low_lik;
  o->dph = (a.SL - a.TD) + low_dph;
  o->Sr = a.Sr;
#pragma unroll
  for (int c = 0; c < CM; ++c) o->Tx[c] = a.Tx[c];
}

// Result of closing one gene: its log-density contribution, the gradient of its own coordinates and
// its six contributions to the hyper-parameter gradient sums.
template <int CM>
struct GeneOut {
  double lp;
  double g_coef[CM], g_sigma_raw;
  double h[6];                  // d/d{lambda_mu, lambda_sigma, lambda_skew, sigma_slope, sigma_intercept, sigma_sigma} (constrained scale)
};

// Sy, SyE, SyX are the per-gene sufficient statistics; SX[c] = sum of X_sc and ncell = number of non-excluded cells.
template <int CM>
PPCX_HD void gene_close(const Dims& d, const Hyper& hy, int g, bool has_slopes, const GeneParams<CM>& gp,
                        const GeneSumsV<CM>& a, double Sy, double SyE, const double* SyX /*CM*/, const double* SX /*CM*/,
                        double ncell, double Lg1, GeneOut<CM>* o) {
  const double SQRT1_2 = 0.70710678118654752440, SQRT_2_OVER_PI = 0.79788456080286535588;
  // ----- likelihood -----
  double lik = (SyE - Sy) + (a.lik - Lg1);
#pragma unroll
  for (int c = 0; c < CM; ++c) {
    o->g_coef[c] = 0.0;
    if (c < d.C && (c == 0 || has_slopes)) {
      lik += gp.coef[c] * SyX[c];
      o->g_coef[c] = gp.phi * (a.Tx[c] - SX[c]);           // sum X_sc (y - x u/w) = phi sum X_sc (rho - 1)
    }
  }
  o->g_sigma_raw = gp.phi * ((a.Sr - ncell) - a.dph);      // -phi d/dphi
  // ----- gene-level priors (.stan:219-223) -----
  const double icpt = gp.coef[0];
  const double z = (icpt - hy.xi) * hy.inv_om;
  double lerfc, ratio;
  log_erfc_and_ratio(-hy.lambda_skew * z * SQRT1_2, &lerfc, &ratio);
  ratio *= SQRT_2_OVER_PI;
  const double dz = -z + hy.lambda_skew * ratio;
  double lp = lik - hy.log_om - 0.5 * z * z + lerfc;
  o->g_coef[0] += dz * hy.inv_om;
  o->h[0] = -dz * hy.inv_om;
  o->h[1] = -hy.inv_om - dz * z * hy.inv_om;
  o->h[2] = z * ratio;
  const double r = gp.sigma_raw - (hy.sigma_slope * icpt + hy.sigma_intercept);
  const double rs = r * hy.inv_ss2;
  lp += -hy.log_ss - 0.5 * r * rs;
  o->g_sigma_raw += -rs;
  o->g_coef[0] += hy.sigma_slope * rs;
  o->h[3] = icpt * rs;
  o->h[4] = rs;
  o->h[5] = (-1.0 + r * rs) * hy.inv_ss;
  if (g < d.K) {
    if (d.C >= 2) {                           // alpha_sub_1 ~ double_exponential(0,1)   (.stan:220)
      const double al = gp.coef[1];             // CM >= 2 always
      lp += -fabs(al);
      o->g_coef[1] += (al > 0.0) ? -1.0 : (al < 0.0 ? 1.0 : 0.0);
    }
#pragma unroll
    for (int c = 2; c < CM; ++c) if (c < d.C) {  // alpha_2 ~ normal(0,2.5)                 (.stan:221)
      lp += -0.5 * gp.coef[c] * gp.coef[c] * (1.0 / 6.25);
      o->g_coef[c] += -gp.coef[c] * (1.0 / 6.25);
    }
  }
  o->lp = lp;
}

// Hyper priors, Jacobians and the chain rule to the unconstrained scale (.stan:183-197,:210-216).
// hsum[6] = sums over genes of GeneOut::h; lp_genes = sum over genes of GeneOut::lp.
PPCX_HD double hyper_close(const Dims& d, const Hyper& hy, const double* u6, double lp_genes, const double* hsum,
                           double* g6) {
  double lp = lp_genes;
  lp += u6[1] + u6[3] + u6[5];                                  // Jacobians
  const double dm = hy.lambda_mu - d.lambda_mu_mu;
  lp += -0.125 * dm * dm - 0.125 * hy.lambda_sigma * hy.lambda_sigma - 0.5 * hy.lambda_skew * hy.lambda_skew;
  lp += -0.125 * hy.sigma_intercept * hy.sigma_intercept - 0.125 * hy.sigma_slope * hy.sigma_slope
        - 0.125 * hy.sigma_sigma * hy.sigma_sigma;
  g6[0] = hsum[0] - 0.25 * dm;
  g6[1] = (hsum[1] - 0.25 * hy.lambda_sigma) * hy.lambda_sigma + 1.0;
  g6[2] = hsum[2] - hy.lambda_skew;
  g6[3] = (hsum[3] - 0.25 * hy.sigma_slope) * hy.sigma_slope + 1.0;
  g6[4] = hsum[4] - 0.25 * hy.sigma_intercept;
  g6[5] = (hsum[5] - 0.25 * hy.sigma_sigma) * hy.sigma_sigma + 1.0;
  return lp;
}

}  // namespace ppcx
