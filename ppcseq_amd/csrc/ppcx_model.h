// ppcx_model.h -- the negative-binomial hierarchical log density of ppcseq, restructured for a
// streaming per-gene reduction on CDNA4.
//
// What is computed is exactly the `target` of inst/stan/negBinomial_MPI.stan (reference file:line):
//   transforms :183-197,:203      priors :210-223      likelihood lp_reduce :58-120 via map_rect :226-240
//   coefficient assembly merge_coefficients :122-139 and X*alpha :205 (fused away: eta is formed per cell)
//
// How it is computed (the MI355X-first part). With log phi = -sigma_raw exactly, t = eta + sigma_raw,
// w = 1 + exp(t), q = 1/w, l = ln w:
//       NB2log(y|eta,phi) + lgamma(y+1) = y eta - y - (y + phi) l + [lgamma(y+phi) - lgamma(phi) + y sigma_raw + y]
//       d/deta = y - (y + phi)(1 - q) = phi (rho - 1),   rho = (1 + y/phi) q
//       d/dphi = - l + 1 - rho + [psi(y+phi) - psi(phi)]
//   * the brackets depend on the count and the dispersion only: summed over a gene's cells they are two functions of sigma_raw
//     per gene, tabulated at upload (ppcx_disp.h, round 5; rounds 2-4 evaluated them per cell with Stirling tails at y + phi and
//     at phi, leading terms cancelled analytically -- that cell, ppcx_disp.h disp_cell, now runs at the table nodes and outside
//     the tables' range only);
//   * a cell needs ONE reciprocal (v_rcp_f64 + a Newton step) and ONE logarithm (table-driven) and four accumulations
//     (sum q, sum y q, sum l, sum y l); sum rho is formed once per gene;
//   * sum_s y eta, sum_s y are per-gene SUFFICIENT STATISTICS (SyE, SyX, Sy) precomputed once, so the large cancelling terms
//     never go through the per-cell loop; sum_s lgamma(y+1) is a per-gene constant of the data (Lg1);
//   * exp(t) factorises into E_s * A_g with E_s = exp(exposure_s) staged in LDS and A_g = exp(intercept_g + sigma_raw_g), times
//     exp(slope_c) for the sample's indicator columns of a gene with slopes: no per-cell exp; a continuous covariate costs one
//     exp(X_s . slopes) per cell on top of it (ppcx_gene.h sweep_cells MODE 3);
//   * the cells of a gene whose w lie within four binades are evaluated on 2^-k w in [1, 16) (the windowed cell below);
//   * excluded cells (to_exclude, R/utilities.R:321-359, subtracted at .stan:105-115) are stored as count = -1, skipped by
//     the sweep, and left out of the sufficient statistics and the tables.
#pragma once
#include "ppcx_math.h"
#include "ppcx_disp.h"

namespace ppcx {

constexpr int kMaxC = 16;       // design-matrix columns supported by the kernels (instantiations for up to 2, 4, 8 and 16)

struct Dims {
  int G, S, C, K, D;
  int off_intercept, off_alpha1, off_alpha2, off_sigma_raw, off_tail;  // Stan declaration order (.stan:183-197)
  int x0_is_one;                // X[,1] == 1 (model.matrix intercept column, R/utilities.R:887-900)
  int x1_binary;                // C >= 2 and every slope column of X in {0, 1} (model.matrix of factors: `~ Label`, a multi-level
                                // factor, `~ a + b`): e^t of a gene with slopes is E_s A_g times exp(slope_c) of the sample's
                                // columns -- no per-cell exp for the checked genes either (C == 2: A_g or A1_g by the group)
  int raw_consts;               // the model has genes whose linear predictor is formed per cell (a continuous covariate, or
                                // X[,1] != 1): the position itself is kept among a coordinate's constants too (coord_consts)
  int Gt, Kt, g0, k0;           // gene shard: totals of the whole problem and this shard's first gene / checked gene
  int gstride;                  // ... and the distance of its consecutive genes in the whole problem: 1 = a contiguous range;
                                // N = every N-th gene (the reference deals genes to its shards round-robin, R/utilities.R:125-136)
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
  d.x0_is_one = 1; d.x1_binary = 0; d.raw_consts = 0; d.lambda_mu_mu = lambda_mu_mu;
  d.Gt = G; d.Kt = K; d.g0 = 0; d.k0 = 0; d.gstride = 1;
  return d;
}
// Index of local coordinate i in the unconstrained vector of the WHOLE problem (Stan order). Used only as
// the Philox stream id, so that a gene-sharded run draws exactly what the unsharded run draws.
PPCX_HD int global_flat(const Dims& d, int i) {
  if (d.Gt == d.G) return i;
  const int n2 = d.C > 2 ? d.C - 2 : 0, st = d.gstride;
  if (i < d.off_intercept) return i;
  if (i < d.off_alpha1) return 3 + d.g0 + st * (i - d.off_intercept);
  if (i < d.off_alpha2) return 3 + d.Gt + d.k0 + st * (i - d.off_alpha1);
  if (i < d.off_sigma_raw) {                    // alpha_2: n2 entries per checked gene, gene-major
    const int j = i - d.off_alpha2, kl = j / n2, c = j - kl * n2;
    return 3 + d.Gt + d.Kt + n2 * (d.k0 + st * kl) + c;
  }
  const int sr_t = 3 + d.Gt + d.Kt + n2 * d.Kt;
  if (i < d.off_tail) return sr_t + d.g0 + st * (i - d.off_sigma_raw);
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
};

// per-lane partial sums over the cells of one gene (see the header comment for the algebra); l = ln w >= 0, q = 1/w
template <int CM>
struct CellAcc {
  double SA;                    // sum y l
  double SL;                    // sum l
  double Sq;                    // sum q
  double SYq;                   // sum y q   (sum rho = Sq + SYq / phi, rho = (1 + y/phi) q, is formed once per gene)
  double Tx[CM];                // sum X_sc rho  (genes with slopes only)
  PPCX_HD void zero() { SA = SL = Sq = SYq = 0.0;
#pragma unroll
    for (int c = 0; c < CM; ++c) Tx[c] = 0.0; }
};

#if defined(__HIP_DEVICE_COMPILE__)
#define PPCX_WAVE_ANY(p) (__any(p) != 0)
#define PPCX_WAVE_ALL(p) (__all(p) != 0)
// placed in the rarely taken side of a wave-uniform branch: keeps hipcc from turning the branch into selects that
// execute both sides for every cell
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

// One cell with e^t = e A: the sample part of the NB2-log likelihood -- w = 1 + e A, q = 1/w (one v_rcp_f64 + Newton step),
// l = ln w (table-driven: ppcx_math.h table_log) -- and its four accumulations. Everything that depends on the count and the
// dispersion only comes from the gene's table (ppcx_disp.h). SLOPES: the pass holds genes with slope coordinates: rho is
// also formed per cell and returned (the caller weights it with the sample's design row); sum rho itself is formed once per
// gene from sum q and sum y q on every route, so that a gene's sums do not depend on the genes it shares a pass with.
// In two halves, so that the row sweep can run the first halves of a trip's four cells -- which end with the request of the
// table entries -- before the second halves, which need them: four LDS round trips in flight behind sixty instructions instead
// of one behind two.
//   cell_front: w, q, the accumulations of q, the table index and its two reads;   cell_back: ln w and its accumulations.
// On the device both are register-only blocks of gfx950 instructions (no memory operation and no wait inside: the table reads
// between them are the compiler's): hipcc's own code for the same C++ spends a register copy per Horner step (v_fmac) and
// joins; here a Horner step is one v_fma with the coefficient in an SGPR pair (a VOP3 instruction may read one SGPR operand).
// v_rcp_f64 is a transcendental-unit instruction: the instruction after it must not read its result, hence the two v_frexp
// behind it. 21 vector instructions per cell (rounds 2-4: 41).
struct CellMid { double yd, m, cinv, logc; int ex; };
template <int CM, bool SLOPES>
PPCX_HD double cell_front(int y, double e, double A, const GeneParams<CM>& gp, const double* tab, CellAcc<CM>& a, CellMid& c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PPCX_NO_ASM_CELL)
  double w, q, t;
  asm("v_cvt_f64_u32_e32 %[yd], %[y]\n\t"
      "v_fma_f64 %[w], %[e], %[A], 1.0\n\t"
      "v_rcp_f64_e32 %[q], %[w]\n\t"
      "v_frexp_mant_f64_e32 %[m], %[w]\n\t"
      "v_frexp_exp_i32_f64_e32 %[ex], %[w]\n\t"
      "v_fma_f64 %[t], -%[w], %[q], 1.0\n\t"         // 1 - w q
      "v_fma_f64 %[q], %[q], %[t], %[q]"             // q (one Newton step)
      : [yd] "=&v"(c.yd), [w] "=&v"(w), [q] "=&v"(q), [t] "=&v"(t), [m] "=&v"(c.m), [ex] "=&v"(c.ex)
      : [y] "v"(y), [e] "v"(e), [A] "v"(A));
#else
  c.yd = (double)y;
  const double w = fma(e, A, 1.0);
  const double q = fast_rcp(w);
  c.m = frexp(w, &c.ex);
#endif
  const int j = (int)(dbl_bits(w) >> (52 - kLogTabBits)) & (kLogTabSize - 1);
  c.cinv = tab[j]; c.logc = tab[kLogTabSize + j];
  a.Sq += q; a.SYq = fma(c.yd, q, a.SYq);
  return SLOPES ? fma(c.yd, gp.invphi, 1.0) * q : 0.0;
}
template <int CM>
PPCX_HD void cell_back(const CellMid& c, CellAcc<CM>& a) {
  double l;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PPCX_NO_ASM_CELL)
  double ed, r;
  const double L4 = -0.25;
  asm("v_fma_f64 %[r], %[m], %[cinv], -1.0\n\t"      // r = m/c - 1
      "v_cvt_f64_i32_e32 %[ed], %[ex]\n\t"
      "v_fma_f64 %[p], %[r], %[L5], %[L4]\n\t"       // log1p(r) = r (1 - r/2 + r^2/3 - r^3/4 + r^4/5)
      "v_fma_f64 %[p], %[r], %[p], %[L3]\n\t"
      "v_fma_f64 %[p], %[r], %[p], -0.5\n\t"
      "v_fma_f64 %[p], %[r], %[p], 1.0\n\t"
      "v_fma_f64 %[p], %[r], %[p], %[logc]\n\t"
      "v_fma_f64 %[p], %[ed], %[LN2], %[p]"            // l = ln w
      : [r] "=&v"(r), [ed] "=&v"(ed), [p] "=&v"(l)
      : [m] "v"(c.m), [ex] "v"(c.ex), [cinv] "v"(c.cinv), [logc] "v"(c.logc), [L4] "v"(L4),
        [L5] "s"(0.2), [L3] "s"(1.0 / 3.0), [LN2] "s"(6.93147180559945286227e-01));
#else
  const double r = fma(c.m, c.cinv, -1.0);             // table_log (ppcx_math.h) on the parts cell_front has taken
  double p = fma(r, 0.2, -0.25);
  p = fma(r, p, 1.0 / 3.0);
  p = fma(r, p, -0.5);
  p = fma(r, p, 1.0);
  l = fma((double)c.ex, 6.93147180559945286227e-01, fma(r, p, c.logc));
#endif
  a.SA = fma(c.yd, l, a.SA);
  a.SL += l;
}
// the whole cell (partial trips, passes with excluded cells, per-cell linear predictors)
template <int CM, bool SLOPES>
PPCX_HD double cell_eval(int y, double e, double A, const GeneParams<CM>& gp, const double* tab, CellAcc<CM>& a) {
  CellMid c;
  const double rho = cell_front<CM, SLOPES>(y, e, A, gp, tab, a, c);
  cell_back<CM>(c, a);
  return rho;
}

// ---- the windowed cell (round 5) --------------------------------------------------------------------------------------
// All cells of a gene share A, so their w = 1 + E_s A lie within a factor max E / min E of each other (times exp |slope| for
// a gene with slopes): with 2^k <= w_min and w_max < 2^(k + 4) -- GeneWindow::ok -- the cells are evaluated on w' = 2^-k w in
// [1, 16): w' = fma(E_s, 2^-k A, 2^-k) costs the same multiply-add, the table index comes from the bits of w' (two exponent
// bits + eight mantissa bits: ppcx_math.h window table), r = fma(w', 1/c, -1), ln w' = ln c + log1p(r) and q' = 1/w' = 2^k q.
// The gene adds k ln 2 (sum y + phi n) once and scales the q-sums by 2^-k (exact). Per cell that removes v_frexp_mant,
// v_frexp_exp, the conversion of the exponent and its multiply-add: 17 vector instructions instead of 21.
// A gene whose cells span more than four binades (exposures more than a factor 8 apart, a large slope), and every gene with a
// per-cell linear predictor, takes the general cell above.
struct GeneWindow { int k; bool ok; double scale; };     // scale = 2^-k
PPCX_HD GeneWindow gene_window(double w_min, double w_max) {
  GeneWindow g;
  const int e0 = (int)((dbl_bits(w_min) >> 52) & 0x7ff), e1 = (int)((dbl_bits(w_max) >> 52) & 0x7ff);
  g.k = e0 - 1023;
  g.ok = w_min >= 1.0 && e1 - e0 < kWinBinades && e1 < 0x7ff;          // (false for NaN / inf)
  g.scale = bits_dbl((unsigned long long)(g.ok ? 2046 - e0 : 1023) << 52);
  if (!g.ok) g.k = 0;
  return g;
}
struct CellMidWin { double yd, w, cinv, logc; };
// e^t 2^-k = e A with A already scaled; one = 2^-k
template <int CM, bool SLOPES>
PPCX_HD double cell_front_win(int y, double e, double A, double one, const GeneParams<CM>& gp, const double* wt, CellAcc<CM>& a, CellMidWin& c) {
  double q;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PPCX_NO_ASM_CELL)
  double t;
  asm("v_fma_f64 %[w], %[e], %[A], %[one]\n\t"
      "v_rcp_f64_e32 %[q], %[w]\n\t"
      "v_cvt_f64_u32_e32 %[yd], %[y]\n\t"            // (the instruction behind v_rcp_f64 must not read its result)
      "v_fma_f64 %[t], -%[w], %[q], 1.0\n\t"         // 1 - w q
      "v_fma_f64 %[q], %[q], %[t], %[q]"             // q (one Newton step)
      : [yd] "=&v"(c.yd), [w] "=&v"(c.w), [q] "=&v"(q), [t] "=&v"(t)
      : [y] "v"(y), [e] "v"(e), [A] "v"(A), [one] "v"(one));
#else
  c.yd = (double)y;
  c.w = fma(e, A, one);
  q = fast_rcp(c.w);
#endif
  const int j = (int)(dbl_bits(c.w) >> (52 - 8)) & (kWinTabSize - 1);
  const WinEntry we = window_entry(wt, j);
  c.cinv = we.cinv; c.logc = we.logc;
  a.Sq += q; a.SYq = fma(c.yd, q, a.SYq);
  return SLOPES ? fma(c.yd, gp.invphi, 1.0) * q : 0.0;
}
template <int CM>
PPCX_HD void cell_back_win(const CellMidWin& c, CellAcc<CM>& a) {
  double l;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PPCX_NO_ASM_CELL)
  double r;
  const double L4 = -0.25;
  asm("v_fma_f64 %[r], %[w], %[cinv], -1.0\n\t"      // r = w/c - 1
      "v_fma_f64 %[p], %[r], %[L5], %[L4]\n\t"       // log1p(r) = r (1 - r/2 + r^2/3 - r^3/4 + r^4/5)
      "v_fma_f64 %[p], %[r], %[p], %[L3]\n\t"
      "v_fma_f64 %[p], %[r], %[p], -0.5\n\t"
      "v_fma_f64 %[p], %[r], %[p], 1.0\n\t"
      "v_fma_f64 %[p], %[r], %[p], %[logc]"          // l = ln w'
      : [r] "=&v"(r), [p] "=&v"(l)
      : [w] "v"(c.w), [cinv] "v"(c.cinv), [logc] "v"(c.logc), [L4] "v"(L4), [L5] "s"(0.2), [L3] "s"(1.0 / 3.0));
#else
  const double r = fma(c.w, c.cinv, -1.0);
  double p = fma(r, 0.2, -0.25);
  p = fma(r, p, 1.0 / 3.0);
  p = fma(r, p, -0.5);
  p = fma(r, p, 1.0);
  l = fma(r, p, c.logc);
#endif
  a.SA = fma(c.yd, l, a.SA);
  a.SL += l;
}
template <int CM, bool SLOPES>
PPCX_HD double cell_eval_win(int y, double e, double A, double one, const GeneParams<CM>& gp, const double* wt, CellAcc<CM>& a) {
  CellMidWin c;
  const double rho = cell_front_win<CM, SLOPES>(y, e, A, one, gp, wt, a, c);
  cell_back_win<CM>(c, a);
  return rho;
}

// What the log-likelihood kernel hands to the close kernel per gene (sums over the gene's non-excluded cells):
//   lik = sum [ - (y + phi) ln w ] + Fh_g(sigma_raw)         (Fh, Dh: the gene's table, ppcx_disp.h)
//   dph = sum [ - ln w ] + Dh_g(sigma_raw)  = sum [ psi(y + phi) - psi(phi) - ln w ]
//   Sr  = sum rho,  Tx[c] = sum X_sc rho,   rho = (1 + y/phi)/w
template <int CM>
struct GeneSumsV { double lik, dph, Sr, Tx[CM]; };
template <int CM> struct GeneSums { static constexpr int N = 3 + CM; };

// a lane's share of the gene: fold the accumulators of its cells into the hand-over sums (the table's part is added by
// the lanes that evaluate it, ppcx_gene.h lane_gene_sums). scale: 2^-k of the gene's window (1 for the general cell): the cells'
// q and rho were those of w' = 2^-k w, 2^k times too large (the logarithms' k ln 2 per cell is added once per gene by the caller).
template <int CM>
PPCX_HD void cell_acc_close(const GeneParams<CM>& gp, const CellAcc<CM>& a, double scale, GeneSumsV<CM>* o) {
  o->lik = -fma(gp.phi, a.SL, a.SA);
  o->dph = -a.SL;
  o->Sr = scale * fma(gp.invphi, a.SYq, a.Sq);
#pragma unroll
  for (int c = 0; c < CM; ++c) o->Tx[c] = scale * a.Tx[c];
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
