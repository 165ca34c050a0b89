// ppcx_disp.h -- the part of the NB2-log likelihood that depends on the counts and the dispersion ONLY, tabulated per gene.
//
// neg_binomial_2_log_lpmf (inst/stan/negBinomial_MPI.stan:97-103) with log phi = -sigma_raw exactly, t = eta + sigma_raw,
// w = 1 + e^t:
//     NB2log(y | eta, phi) + lgamma(y + 1) = y eta - y - (y + phi) ln w + [lgamma(y + phi) - lgamma(phi) + y sigma_raw + y]
//     d/dphi                               = - ln w + 1 - (1 + y/phi)/w   + [psi(y + phi) - psi(phi)]
// The brackets do not depend on the sample's linear predictor. Summed over the gene's non-excluded cells they are two
// functions of ONE variable per gene,
//     Fh_g(sigma) = sum_s lgamma(y_s + phi) - lgamma(phi) + y_s sigma + y_s        Dh_g(sigma) = sum_s psi(y_s + phi) - psi(phi),
// analytic in the strip |Im sigma| < pi (the nearest singularities are where y + e^{-sigma} is a non-positive integer).
// Rounds 1-4 evaluated them cell by cell in every gradient evaluation -- Stirling tails at y + phi, a running product of
// 1 + y/phi: 21 of a cell's 41 instructions; exact recurrences hoisted per gene for y <= 7 only. Round 5 hoists them for EVERY
// count: at upload (and whenever the exclusions change) each gene gets a table of kDispPanels panels of width kDispWidth on
// sigma_raw in [kDispLo, kDispLo + kDispPanels kDispWidth), each holding the degree-kDispDeg polynomials that interpolate
// Fh_g and Dh_g at the panel's Chebyshev nodes (Bernstein ellipse parameter 25: truncation below 2e-16 of the functions'
// size, scripts/fit/dispersion_table.py). A gradient evaluation reads one panel per (chain, gene) -- 176 bytes -- and runs
// two Horner recurrences per GENE; what is left per CELL is the sample part: ln w and 1/w.
// Outside the tabulated range (phi < 3.4e-4 or phi > 2981: early warm-up excursions) the gene's lanes evaluate the same
// functions directly from the row (disp_row), with the code that builds the tables.
//
// The count matrix is still streamed once per gradient evaluation: the table replaces arithmetic, not the stream.
#pragma once
#include "ppcx_math.h"

namespace ppcx {

constexpr int kDispDeg = 10, kDispN = kDispDeg + 1;     // degree; nodes = coefficients per panel and function
constexpr int kDispStride = 12;                         // doubles per (gene, panel, function): 11 coefficients + 1 pad (16-byte requests)
constexpr int kDispPanels = 32;
constexpr double kDispLo = -8.0, kDispWidth = 0.5, kDispInvWidth = 2.0;
constexpr long kDispGeneDoubles = (long)kDispPanels * 2 * kDispStride;    // 768 doubles = 6 KB per gene
// table layout: tab[((g * kDispPanels + panel) * 2 + f) * kDispStride + k], f = 0: Fh, 1: Dh; coefficient k of x^k,
// x = 2 (sigma - panel start) / width - 1 in [-1, 1)

// ---- direct evaluation (table build, out-of-range positions) ------------------------------------------------------
// ln(1 + z), z >= 0, accurate for small z too (Kahan's correction of the rounding of 1 + z); one-off / rare paths only
PPCX_HD double log1p_acc(double z) {
  const double u = 1.0 + z, d = u - 1.0;
  if (d == 0.0) return z;
  const double l = fast_log(u);
  return z < 4.0 ? l * (z / d) : l;
}
// the Stirling excess of phi (ppcx_math.h stirling_excess) without the LDS log table
PPCX_HD void stirling_excess_acc(double phi, double lnphi, double* dlt, double* dps) {
  const bool small = phi < 8.0;
  const double xs = small ? phi + 8.0 : phi;
  double lgt, dgt;
  stirling_tails(1.0 / xs, &lgt, &dgt);
  *dlt = lgt; *dps = dgt;
  if (small) {
    double P = phi, dP = 1.0;
#pragma unroll
    for (int k = 1; k < 8; ++k) { const double f = phi + (double)k; dP = fma(dP, f, P); P = P * f; }
    const double lxs = fast_log(xs), lP = fast_log(P);
    *dlt = (xs - 0.5) * lxs - 8.0 - lP - (phi - 0.5) * lnphi + lgt;
    *dps = (lnphi - lxs) + dgt + dP / P;
  }
}
struct DispPoint { double sigma, phi, invphi, dlt, dps; };
// sigma_lo: what sigma lacks to the point that is meant (a table node, which is not a double: disp_node_sigma); 0 for a position
PPCX_HD DispPoint disp_point(double sigma, double sigma_lo = 0.0) {
  DispPoint p;
  p.sigma = sigma;
  const double ph = fast_exp(-sigma);
  p.phi = fma(-sigma_lo, ph, ph);                // exp(-(sigma + sigma_lo)), sigma_lo ~ 1e-16
  p.invphi = 1.0 / p.phi;
  stirling_excess_acc(p.phi, -sigma - sigma_lo, &p.dlt, &p.dps);
  return p;
}
// one cell: Fh = lgamma(y + phi) - lgamma(phi) + y sigma + y,  Dh = psi(y + phi) - psi(phi)
//   y < 8 : lgamma(y + phi) - lgamma(phi) = sum_{k<y} ln(phi + k)  =>  Fh = ln prod_{k<y} (1 + k/phi) + y,  Dh = sum_{k<y} 1/(phi + k)
//   y >= 8: Stirling at y + phi and at phi, leading terms cancelled analytically (ppcx_math.h):
//           Fh = (y + phi - 1/2) ln(1 + y/phi) + lg_tail(1/(y + phi)) - dlt(phi),  Dh = ln(1 + y/phi) - dg_tail(1/(y + phi)) + dps(phi)
PPCX_HD void disp_cell(int y, const DispPoint& p, double* F, double* D) {
  const double yd = (double)y;
  if (y < 8) {
    double P = 1.0, ds = 0.0;
    for (int k = 0; k < y; ++k) { P *= fma((double)k, p.invphi, 1.0); ds += 1.0 / (p.phi + (double)k); }
    *F = fast_log(P) + yd; *D = ds;
    return;
  }
  const double x = yd + p.phi;
  const double lx = log1p_acc(yd * p.invphi);
  double lgt, dgt;
  stirling_tails(1.0 / x, &lgt, &dgt);
  *F = (x - 0.5) * lx + (lgt - p.dlt);
  *D = lx - (dgt - p.dps);
}
// compensated running sums (Neumaier): the cells' terms are of one sign and size, the table should not lose what they carry
struct DispSum {
  double s = 0.0, c = 0.0;
  PPCX_HD void add(double v) { const double t = s + v; c += fabs(s) >= fabs(v) ? (s - t) + v : (v - t) + s; s = t; }
  PPCX_HD double value() const { return s + c; }
};
// cells start, start + stride, ... of one row (excluded cells, count -1, are skipped)
PPCX_HD void disp_row(const int* row, int S, int start, int stride, const DispPoint& p, double* F, double* D) {
  DispSum f, d;
  for (int s = start; s < S; s += stride) {
    const int y = row[s];
    if (y < 0) continue;
    double fc, dc;
    disp_cell(y, p, &fc, &dc);
    f.add(fc); d.add(dc);
  }
  *F = f.value(); *D = d.value();
}

// The out-of-range path of the log-likelihood kernel: NOT inlined there -- inlined, the compiler moves the materialisation of
// every constant of fast_exp / fast_log / the tails (forty vector registers) in front of the kernel's pass loop, where the sweep
// pays for them with spills.
#if defined(__HIPCC__)
__host__ __device__ __attribute__((noinline))
#endif
inline void disp_row_at(const int* row, int S, int start, int stride, double sigma, double* F, double* D) {
  const DispPoint pt = disp_point(sigma);
  disp_row(row, S, start, stride, pt, F, D);
}

// ---- nodes -> coefficients -------------------------------------------------------------------------------------------
// Chebyshev points of the first kind x_k = cos(pi (2k + 1) / (2N)); c_j = (2/N) sum_k f_k cos(j pi (2k + 1) / (2N)) (c_0 halved);
// then the monomial coefficients a_m = sum_j c_j [x^m] T_j. Two steps on purpose: the discrete cosine transform is orthogonal
// (the c_j carry absolute errors of the size of the node values' rounding), and in the conversion the large entries of [x^m] T_j
// meet coefficients that have already decayed like 25^-j -- the one-step inverse Vandermonde product would cancel to 1e-13.
struct DispFit { double xnode[kDispN], xnode_lo[kDispN]; double dct[kDispN][kDispN]; double mono[kDispN][kDispN]; };
inline void disp_fit_init(DispFit& f) {         // host
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int k = 0; k < kDispN; ++k) {
    const long double x = cosl(pi * (2 * k + 1) / (2.0L * kDispN));
    f.xnode[k] = (double)x; f.xnode_lo[k] = (double)(x - (long double)f.xnode[k]);
  }
  for (int j = 0; j < kDispN; ++j)
    for (int k = 0; k < kDispN; ++k) f.dct[j][k] = (double)((j == 0 ? 1.0L : 2.0L) / kDispN * cosl(pi * j * (2 * k + 1) / (2.0L * kDispN)));
  long double T[kDispN][kDispN];
  for (int j = 0; j < kDispN; ++j) for (int m = 0; m < kDispN; ++m) T[j][m] = 0.0L;
  T[0][0] = 1.0L;
  if (kDispN > 1) T[1][1] = 1.0L;
  for (int j = 2; j < kDispN; ++j) for (int m = 0; m < kDispN; ++m) T[j][m] = (m > 0 ? 2.0L * T[j - 1][m - 1] : 0.0L) - T[j - 2][m];
  for (int j = 0; j < kDispN; ++j) for (int m = 0; m < kDispN; ++m) f.mono[j][m] = (double)T[j][m];      // integers: exact
}
// node k of a panel as a double and what it lacks to the node itself. The node is where the interpolation takes the value to be:
// a function value at the ROUNDED node instead is off by f' times the rounding of sigma (1e-15), which a table of Dh ~ 1/phi shows
// as 1e-15 of its value.
PPCX_HD double disp_node_sigma(const DispFit& f, int panel, int k, double* lo) {
  const double a = fma((double)panel, kDispWidth, kDispLo) + 0.5 * kDispWidth;      // the panel's centre: exact
  const double b = 0.5 * kDispWidth * f.xnode[k];                                     // exact (a power of two times a double)
  const double s = a + b;
  const double bb = s - a;
  const double err = (a - (s - bb)) + (b - bb);                                       // TwoSum: a + b = s + err
  *lo = err + 0.5 * kDispWidth * f.xnode_lo[k];
  return s;
}
// out[0 .. kDispStride): coefficients of x^0 .. x^kDispDeg, then zero padding
PPCX_HD void disp_fit_panel(const DispFit& f, const double* fv /* kDispN node values */, double* out) {
  double c[kDispN];
  for (int j = 0; j < kDispN; ++j) {
    DispSum s;
    for (int k = 0; k < kDispN; ++k) s.add(f.dct[j][k] * fv[k]);
    c[j] = s.value();
  }
  for (int m = 0; m < kDispN; ++m) {
    DispSum s;
    for (int j = kDispN - 1; j >= m; --j) s.add(c[j] * f.mono[j][m]);     // smallest terms first
    out[m] = s.value();
  }
  for (int m = kDispN; m < kDispStride; ++m) out[m] = 0.0;
}
// the whole table of one gene (host: emulation harness, CPU comparator; the device builds it with ppcx_disp_build_kernel)
inline void disp_build_gene_host(const DispFit& f, const int* row, int S, double* out /* kDispGeneDoubles */) {
  for (int p = 0; p < kDispPanels; ++p) {
    double fv[kDispN], dv[kDispN];
    for (int k = 0; k < kDispN; ++k) {
      double lo;
      const double sg = disp_node_sigma(f, p, k, &lo);
      const DispPoint pt = disp_point(sg, lo);
      disp_row(row, S, 0, 1, pt, &fv[k], &dv[k]);
    }
    disp_fit_panel(f, fv, out + ((long)p * 2 + 0) * kDispStride);
    disp_fit_panel(f, dv, out + ((long)p * 2 + 1) * kDispStride);
  }
}

// ---- lookup ----------------------------------------------------------------------------------------------------------
struct DispRef { int panel; bool in; double x; };
PPCX_HD DispRef disp_ref(double sigma) {
  DispRef r;
  const double t = (sigma - kDispLo) * kDispInvWidth;
  r.in = t >= 0.0 && t < (double)kDispPanels;                  // false for NaN
  const int pi = r.in ? (int)t : 0;
  r.panel = pi;
  // sigma minus the panel's start is exact (both are multiples of sigma's last place), so x carries one rounding -- formed from
  // t it would carry the rounding of sigma - kDispLo, which the steepest tables (Dh ~ 1/phi) show as 1e-15 of their value
  r.x = fma(sigma - fma((double)pi, kDispWidth, kDispLo), 2.0 * kDispInvWidth, -1.0);
  return r;
}
PPCX_HD double disp_horner(const double* c, double x) {
  double p = c[kDispDeg];
#pragma unroll
  for (int k = kDispDeg - 1; k >= 0; --k) p = fma(p, x, c[k]);
  return p;
}

}  // namespace ppcx
