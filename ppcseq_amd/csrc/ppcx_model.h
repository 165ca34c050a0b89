// ppcx_model.h -- the negative-binomial hierarchical log density of ppcseq, restructured for a
// streaming per-gene reduction on CDNA4.
//
// What is computed is exactly the `target` of inst/stan/negBinomial_MPI.stan (reference file:line):
//   transforms :183-197,:203      priors :210-223      likelihood lp_reduce :58-120 via map_rect :226-240
//   coefficient assembly merge_coefficients :122-139 and X*alpha :205 (fused away: eta is formed per cell)
//
// How it is computed (the MI355X-first part, see DESIGN.md "lp/grad kernel"):
//   * log phi = -sigma_raw exactly, so with t = eta + sigma_raw, u = exp(t), w = 1 + u:
//       NB2log(y|eta,phi) = y*t - (y+phi)*log(w) + [lgamma(y+phi) - lgamma(phi)] - lgamma(y+1)
//       d/deta = y - (y+phi)*u/w           d/dphi = psi(y+phi) - psi(phi) + 1 - log(w) - (y+phi)/(phi*w)
//   * sum_s y*t and sum_s y are per-gene SUFFICIENT STATISTICS (Sy, SyE, SyX) precomputed once, so the
//     large cancelling terms never go through the per-cell loop;
//   * sum_s lgamma(y+1) is a per-gene constant of the data (Lg1), subtracted at gene level so the
//     large lgamma terms cancel inside each gene instead of across the whole matrix;
//   * for genes without slope terms (g >= K, X[,1] == 1) exp(t) factorises into E_s * A_g with
//     E_s = exp(exposure_s) staged in LDS and A_g = exp(intercept_g + sigma_raw_g): no per-cell exp;
//   * excluded cells (to_exclude, R/utilities.R:321-359, subtracted at .stan:105-115) are stored as
//     count = -1 and skipped, and are left out of the sufficient statistics.
#pragma once
#include "ppcx_math.h"

namespace ppcx {

constexpr int kMaxC = 8;        // design-matrix columns supported by the kernels

struct Dims {
  int G, S, C, K, D;
  int off_intercept, off_alpha1, off_alpha2, off_sigma_raw, off_tail;  // Stan declaration order (.stan:183-197)
  int x0_is_one;                // X[,1] == 1 (model.matrix intercept column, R/utilities.R:887-900)
  int x1_binary;                // C == 2 and X[,2] in {0, 1} (a two-group design, `~ Label`): e^t of a gene with a slope is
                                // E_s A_g or E_s A1_g by the sample's group -- no per-cell exp for the checked genes either
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

// per-lane partial sums over the cells of one gene
template <int CM>
struct CellAcc {
  double T1, SP, T2u, T3, T4;   // sum x*log w, sum log w, sum x*u/w, sum dlgamma, sum ddigamma
  double T2x[CM];               // sum X_sc * x*u/w  (generic path only)
  PPCX_HD void zero() { T1 = SP = T2u = T3 = T4 = 0.0;
#pragma unroll
    for (int c = 0; c < CM; ++c) T2x[c] = 0.0; }
};

#if defined(__HIP_DEVICE_COMPILE__)
#define PPCX_WAVE_ANY(p) (__any(p) != 0)
#define PPCX_WAVE_ALL(p) (__all(p) != 0)
#else
#define PPCX_WAVE_ANY(p) (p)
#define PPCX_WAVE_ALL(p) (p)
#endif

// One valid cell (y >= 0) once u = exp(t) is known. The per-lane work is free of selects and divisions in
// the common case: the three regimes of lgamma(y+phi) - lgamma(phi) and its digamma counterpart are chosen per
// WAVEFRONT (device) / per cell (host emulation): every lane has y+phi >= 32 (4-term Stirling tails), every lane
// >= 8 (7-term tails), or some lane has y+phi < 8. Such a lane has y <= 7, so the differences are the exact
// recurrences  log prod_{k<y}(phi+k)  and  P'/P  -- no Stirling series, no cancellation; the other lanes of
// that wavefront use the 7-term tails. Every regime costs one logarithm and one reciprocal per cell.
// The host orders the genes by their smallest count (ppcx_capi.hip, gene_order) so that the lanes of a
// wavefront mostly agree on the regime.
PPCX_HD void cell_eval(int y, double u, double phi, double lgphi, double dgphi, const double* tab,
                       double* T1, double* SP, double* T3, double* T4, double* xsig) {
  const double x = (double)y + phi;
  const double w = 1.0 + u;
  const double sp = table_log(w, tab);
  *T1 = fma(x, sp, *T1);
  *SP += sp;
  double dl, dd;                 // lgamma(y+phi) - lgamma(phi), digamma(y+phi) - digamma(phi)
  // 1/w and 1/arg from ONE hardware reciprocal: q = 1/(w arg), 1/w = q arg, 1/arg = q w (v_rcp_f64 is quarter rate)
  if (PPCX_WAVE_ALL(x >= 32.0)) {
    const double q = fast_rcp(w * x), rx = q * w;
    *xsig = x * (u * (q * x));
    double lg, dg;
    lgamma_digamma_stirling4(x, table_log(x, tab), rx, &lg, &dg);
    dl = lg - lgphi; dd = dg - dgphi;
  } else if (PPCX_WAVE_ALL(x >= 8.0)) {
    const double q = fast_rcp(w * x), rx = q * w;
    *xsig = x * (u * (q * x));
    double lg, dg;
    lgamma_digamma_stirling(x, table_log(x, tab), rx, &lg, &dg);
    dl = lg - lgphi; dd = dg - dgphi;
  } else {
    const bool small = x < 8.0;
    double P = 1.0, dP = 0.0;    // P = prod_{k<y}(phi+k), dP = dP/dphi  (y = 0: P = 1, dP = 0)
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      if (small && k < y) {
        const double f = phi + (double)k;
        dP = fma(dP, f, P);
        P = P * f;
      }
    }
    const double arg = small ? P : x;
    const double q = fast_rcp(w * arg), ra = q * w;
    *xsig = x * (u * (q * arg));
    const double la = table_log(arg, tab);
    double lg, dg;
    lgamma_digamma_stirling(x, la, ra, &lg, &dg);        // meaningful for the lanes with x >= 8 only
    dl = small ? la : lg - lgphi;
    dd = small ? dP * ra : dg - dgphi;
  }
  *T3 += dl;
  *T4 += dd;
}

// everything a gene's lanes need that does not depend on the sample
template <int CM>
struct GeneParams {
  double coef[CM];              // intercept, alpha_sub_1, alpha_2...   (zero beyond K, .stan:133-135)
  double sigma_raw, phi, lgphi, dgphi, A, A1;   // A = exp(intercept + sigma_raw), A1 = A exp(slope) (two-group designs)
};

// Result of closing one gene: its log-density contribution, the gradient of its own coordinates and
// its six contributions to the hyper-parameter gradient sums.
template <int CM>
struct GeneOut {
  double lp;
  double g_coef[CM], g_sigma_raw;
  double h[6];                  // d/d{lambda_mu, lambda_sigma, lambda_skew, sigma_slope, sigma_intercept, sigma_sigma} (constrained scale)
};

// Sy, SyE, SyX are the per-gene sufficient statistics; ncell = number of non-excluded cells.
template <int CM>
PPCX_HD void gene_close(const Dims& d, const Hyper& hy, int g, bool has_slopes, const GeneParams<CM>& gp,
                        const CellAcc<CM>& a, double Sy, double SyE, const double* SyX /*CM*/, double ncell,
                        double Lg1, GeneOut<CM>* o) {
  const double SQRT1_2 = 0.70710678118654752440, SQRT_2_OVER_PI = 0.79788456080286535588;
  // ----- likelihood -----
  double lik = SyE + gp.sigma_raw * Sy - a.T1 + (a.T3 - Lg1);
#pragma unroll
  for (int c = 0; c < CM; ++c) {
    o->g_coef[c] = 0.0;
    if (c < d.C && (c == 0 || has_slopes)) {
      lik += gp.coef[c] * SyX[c];
      o->g_coef[c] = SyX[c] - a.T2x[c];
    }
  }
  o->g_sigma_raw = -gp.phi * (a.T4 + ncell - a.SP) + (Sy + ncell * gp.phi) - a.T2u;
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
