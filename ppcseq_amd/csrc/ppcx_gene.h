// ppcx_gene.h -- the per-gene bodies of the kernels:
//   log-likelihood kernel : lane_gene_sums (per lane) -> [L-lane butterfly] -> per-gene sums
//   close kernel          : gene_load -> gene_finish -> tree bookkeeping (coord_merge_dots / coord_store_slot /
//                           coord_top_dots)
//   update kernel         : coord_update (per coordinate: the command's coordinate work + coord_consts)
//   step kernel           : chain_step (scalar state machine)
// Shared by the gfx950 kernel (ppcx_kernels.hip) and the CPU emulation harness in tests/emul.
#pragma once
#include "ppcx_nuts.h"


namespace ppcx {

template <int CM>
struct GeneCtx {
  static constexpr int NCM = CM + 1;          // coordinates a gene can own: intercept, sigma_raw, CM-1 slopes
  int gg, ncoord;
  bool active, has_slopes, fast, two;           // two: fast path with the group-dependent constant A / A1
  int idx[NCM];
  double q[NCM];                  // the coordinates' positions (intercept, sigma_raw, slopes) at the end being evaluated
  GeneParams<CM> gp;
};

// Constants of the cell loop that depend on ONE coordinate, written next to the coordinate by whoever moves it (the
// update kernel, the gene kernel, the ADVI kernel), one thread per coordinate, instead of being recomputed by every lane of
// every gene in the log-likelihood kernel:  intercept, slopes: exp(q);  sigma_raw: phi = exp(-q) (.stan:203) and q itself --
// the log-likelihood kernel of a pipelined round reads constants only (they may belong to an anticipated position that is in
// none of the trajectory's ends yet), and it needs sigma_raw to address the gene's dispersion table (ppcx_disp.h).
PPCX_HD void coord_consts(const Dims& d, const VecRef& v, int i, double q) {
  if (i >= d.off_sigma_raw && i < d.off_tail) {
    v.at(V_C0, i) = fast_exp(-q); v.at(V_C2, i) = q;         // 1/phi: the reader takes fast_rcp(phi) itself (V_C1, V_C3 are unused)
  } else if (i >= d.off_intercept && i < d.off_sigma_raw) {
    v.at(V_C0, i) = fast_exp(q);
    // a per-cell linear predictor (generic_cells) needs the coefficients themselves: kept beside exp(q), so that such a model's
    // cells too read constants only and its rounds can be pipelined (round 5; before, they read the trajectory's end and the
    // model ran the three-launch round)
    if (d.raw_consts) v.at(V_C2, i) = q;
  }
}

// load the gene's (already drifted) coordinates -- coefficients, sigma_raw, phi -- for the close kernel
// which coordinates gene g owns and which cell path it takes (no memory access)
template <int CM>
PPCX_HD void gene_index(const Dims& d, int g, GeneCtx<CM>& x) {
  constexpr int NCM = CM + 1;
  x.active = g < d.G;
  x.gg = x.active ? g : 0;
  const int C = d.C;
  const int nslope = x.gg < d.K ? (C - 1 > 1 ? C - 1 : 1) : 0;   // alpha_sub_1 exists even for C == 1 (.stan:189)
  x.ncoord = x.active ? 2 + nslope : 0;
#pragma unroll
  for (int j = 0; j < NCM; ++j)
    x.idx[j] = j == 0 ? d.off_intercept + x.gg : (j == 1 ? d.off_sigma_raw + x.gg : coef_index(d, j - 1, x.gg));
  x.has_slopes = x.active && x.gg < d.K && C >= 2;
  x.two = x.has_slopes && d.x0_is_one && d.x1_binary;
  x.fast = d.x0_is_one && (!x.has_slopes || x.two);
}
// the gene's parameters from its coordinates' positions x.q[] (phi: the constant written with the position)
template <int CM>
PPCX_HD void gene_params(const Dims& d, const VecRef& v, GeneCtx<CM>& x) {
  x.gp.coef[0] = x.q[0];
#pragma unroll
  for (int cc = 1; cc < CM; ++cc) x.gp.coef[cc] = (x.has_slopes && cc < d.C) ? x.q[cc + 1] : 0.0;
  x.gp.sigma_raw = x.q[1];
  x.gp.phi = x.active ? v.at(V_C0, x.idx[1]) : 1.0;        // sigma = 1 ./ exp(sigma_raw)   (.stan:203)
  x.gp.invphi = 0.0;
}
template <int CM>
PPCX_HD void gene_load(const Dims& d, const Cmd& c, const VecRef& v, int g, GeneCtx<CM>& x) {
  constexpr int NCM = CM + 1;
  gene_index<CM>(d, g, x);
#pragma unroll
  for (int j = 0; j < NCM; ++j) x.q[j] = j < x.ncoord ? v.at(V_Q0 + 3 * c.dir, x.idx[j]) : 0.0;
  gene_params<CM>(d, v, x);
}

// ---------------------------------------------------------------------------------------------------------------
// The log-likelihood kernel's work on one gene, as seen by ONE of the L lanes that share the gene (sub = 0 .. L - 1):
//   (1) the row sweep: cells s = sub, sub + L, ... -- every count takes the same straight-line code (cell_eval: the sample
//       part, ln w and 1/w); excluded cells (-1) are passed over, in passes that hold a gene with such cells (MASKED; the
//       other passes do not look at the count). Four cells per trip, the counts of the next trip requested before the
//       current one is evaluated; L is a compile-time constant, so the four loads of a trip differ by immediate offsets
//       and a trip costs one address update;
//   (2) the count-and-dispersion part of the whole gene from its table (ppcx_disp.h): lane 0 evaluates Fh, lane 1 Dh (one
//       Horner recurrence each, side by side; a lone lane does both); the panel's coefficients are requested before the sweep.
//       A position outside the tabulated range is evaluated directly from the row by all the gene's lanes (disp_row).
// Genes with slopes in a two-group design take the same route with e^t = E_s A or E_s A1 by the sample's group; with more
// indicator columns (factor designs, C > 2; model.matrix of a multi-level factor or of `~ a + b`, R/utilities.R:887-900)
// e^t = E_s A prod exp(slope_c) over the sample's columns, from the per-coordinate constants exp(q) -- no exp per cell; any
// other gene with slopes, and every gene when X[,1] != 1, forms eta per cell (generic_cells: an exp per cell).
// `counts` must be readable 4 L entries past the end of the matrix, `sE` / `sX` (LDS) 4 L entries past S: the host and
// the kernel pad them.
// ---------------------------------------------------------------------------------------------------------------
struct CellData {                // the chain-independent inputs of the log-likelihood kernel (device pointers)
  const int* counts;             // G x S gene-major, excluded cells = -1
  const double* disp;            // [G][kDispPanels][2][kDispStride] the genes' dispersion tables (ppcx_disp.h)
  const unsigned char* gflags;   // [G] bit 0: the gene has excluded cells
  const double* Sy;              // [G] sum of the gene's non-excluded counts   } what a gene's window (ppcx_model.h GeneWindow)
  const double* ncell;           // [G] number of its non-excluded cells        } owes the sums once per gene
  double e_min, e_max;           // smallest and largest exp(exposure_s): the window's bounds
};

// What a pass needs of a gene before its sweep, all of it addressed by the gene alone: the log-likelihood kernel requests it a
// whole pass ahead (ppcx_kernels.hip loglik_passes), so that a pass starts with its constants in registers. (Every wavefront of
// the launch starts and ends its passes at the same time -- equal costs since round 5 -- so a round trip at the start of a pass is
// hidden by nobody; with the 17-instruction cell the sweep of 25 cells no longer dwarfs it.)
struct GenePre { double phi, sigma, a0, Sy, n; int flags; };
PPCX_HD GenePre gene_pre_load(const Dims& d, const VecRef& v, const CellData& m, int g) {
  GenePre r;
  r.phi = v.at(V_C0, d.off_sigma_raw + g);       // phi = exp(-sigma_raw) and sigma_raw of the position being evaluated (coord_consts)
  r.sigma = v.at(V_C2, d.off_sigma_raw + g);
  r.a0 = v.at(V_C0, d.off_intercept + g);        // exp(intercept)
  r.Sy = m.Sy[g]; r.n = m.ncell[g]; r.flags = m.gflags[g];
  return r;
}
// how the 2 x kDispStride coefficients of a gene's panel are dealt to its lanes: NL lanes per function, CH coefficients each
// (L >= 8: lanes 0-3 take Fh's chunks, lanes 4-7 Dh's, three coefficients each; L = 4: two lanes of six; L = 2: one lane per
// function; a lone lane both). A lane evaluates its chunk times x^(CH chunk) and adds it to its own partial sum: the gene's
// L-lane reduction, which follows anyway, adds the chunks up. Three coefficients per lane can be requested BEFORE the sweep and
// held across it (six registers); the eleven of a whole function could not (round 5 first form: requested after the sweep, a
// round trip per pass in the open).
template <int L> struct DispDeal {
  static constexpr int NL = L >= 8 ? 4 : (L >= 4 ? 2 : 1);
  static constexpr int CH = kDispStride / NL;
};

// MODE 0: plain gene (e^t = E_s A); 1: two-group design (A or A1 by the sample's group, sX1 = the group column);
// 2: more indicator columns (C > 2; sX1 = column 1 of X in LDS, column c at sX1 + (c - 1) S; ec[c] = exp(slope_c)):
// A times the ec of the sample's columns; 3: slope columns of any values (a continuous covariate: `~ group + age`,
// R/utilities.R:887-900): ec[c] = slope_c itself, e^t = E_s A exp(sum_c X_sc slope_c) -- one exp per cell on top of the
// factorised E_s A, in the same four-cell trips with their counts requested a trip ahead (the general cell: such a gene's w
// have no common window that the slopes' constants would give away)
// WIN: the windowed cell (ppcx_model.h GeneWindow): A, A1 arrive scaled by 2^-k, one = 2^-k, tab = the window table; else the
// general cell: one = 1 (not used), tab = the mantissa table
template <int CM, int L, int MODE, bool MASKED, bool WIN>
PPCX_HD void sweep_cells(int S, const int* row, const double* sE, const double* sX1, int sub, double A, double A1, double one,
                         const GeneParams<CM>& gp, const double* tab, CellAcc<CM>& acc, const double* ec = nullptr, int C = 2) {
  const int nmin = S / L;                                  // cells every lane of the gene has
  const int nlane = nmin + (sub < S - nmin * L ? 1 : 0);
  const int* p = row + sub;
  const double* q = sE + sub;
  const double* qx = sX1 + sub;
  int y0 = p[0], y1 = p[L], y2 = p[2 * L], y3 = p[3 * L];
  int k = 0;
#define PPCX_SWEEP_CELL(Y, E, XB, OFF, COND)                                                    \
  if ((COND) && (!MASKED || (Y) >= 0)) {                                                        \
    if (MODE == 1) { const double a1_ = (XB) != 0.0 ? A1 : A;                                   \
                     const double rho_ = WIN ? cell_eval_win<CM, true>(Y, E, a1_, one, gp, tab, acc) : cell_eval<CM, true>(Y, E, a1_, gp, tab, acc); \
                     acc.Tx[1] = fma(XB, rho_, acc.Tx[1]); }                                    \
    else if (MODE == 2) {                                                                       \
      double a_ = (XB) != 0.0 ? A1 : A; double xk_[CM];                                         \
      _Pragma("unroll") for (int cc = 2; cc < CM; ++cc) { xk_[cc] = cc < C ? qx[(cc - 1) * S + (OFF)] : 0.0; a_ = xk_[cc] != 0.0 ? a_ * ec[cc] : a_; } \
      const double rho_ = WIN ? cell_eval_win<CM, true>(Y, E, a_, one, gp, tab, acc) : cell_eval<CM, true>(Y, E, a_, gp, tab, acc); \
      acc.Tx[1] = fma(XB, rho_, acc.Tx[1]);                                                     \
      _Pragma("unroll") for (int cc = 2; cc < CM; ++cc) acc.Tx[cc] = fma(xk_[cc], rho_, acc.Tx[cc]); \
    }                                                                                           \
    else if (MODE == 3) {                                                                       \
      double t_ = (XB) * ec[1]; double xk_[CM];                                                 \
      _Pragma("unroll") for (int cc = 2; cc < CM; ++cc) { xk_[cc] = cc < C ? qx[(cc - 1) * S + (OFF)] : 0.0; t_ = fma(xk_[cc], ec[cc], t_); } \
      const double rho_ = cell_eval<CM, true>(Y, E, A * fast_exp(t_), gp, tab, acc);           \
      acc.Tx[1] = fma(XB, rho_, acc.Tx[1]);                                                     \
      _Pragma("unroll") for (int cc = 2; cc < CM; ++cc) acc.Tx[cc] = fma(xk_[cc], rho_, acc.Tx[cc]); \
    }                                                                                           \
    else if (WIN) (void)cell_eval_win<CM, false>(Y, E, A, one, gp, tab, acc);                   \
    else (void)cell_eval<CM, false>(Y, E, A, gp, tab, acc);                                     \
  }
  for (; k + 4 <= nmin; k += 4) {
    const double e0 = q[0], e1 = q[L], e2 = q[2 * L], e3 = q[3 * L];
    double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
    if (MODE != 0) { x0 = qx[0]; x1 = qx[L]; x2 = qx[2 * L]; x3 = qx[3 * L]; }
    p += 4 * L; q += 4 * L; qx += 4 * L;
    const int n0 = p[0], n1 = p[L], n2 = p[2 * L], n3 = p[3 * L];
    if (MODE == 0 && !MASKED && WIN) {
      // the common trip: the four first halves, then the four second halves (cell_front_win / cell_back_win, ppcx_model.h)
      CellMidWin c0, c1, c2, c3;
      (void)cell_front_win<CM, false>(y0, e0, A, one, gp, tab, acc, c0);
      (void)cell_front_win<CM, false>(y1, e1, A, one, gp, tab, acc, c1);
      (void)cell_front_win<CM, false>(y2, e2, A, one, gp, tab, acc, c2);
      (void)cell_front_win<CM, false>(y3, e3, A, one, gp, tab, acc, c3);
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
      cell_back_win<CM>(c0, acc); cell_back_win<CM>(c1, acc); cell_back_win<CM>(c2, acc); cell_back_win<CM>(c3, acc);
    } else if (MODE == 0 && !MASKED) {
      CellMid c0, c1, c2, c3;
      (void)cell_front<CM, false>(y0, e0, A, gp, tab, acc, c0);
      (void)cell_front<CM, false>(y1, e1, A, gp, tab, acc, c1);
      (void)cell_front<CM, false>(y2, e2, A, gp, tab, acc, c2);
      (void)cell_front<CM, false>(y3, e3, A, gp, tab, acc, c3);
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
      cell_back<CM>(c0, acc); cell_back<CM>(c1, acc); cell_back<CM>(c2, acc); cell_back<CM>(c3, acc);
    } else {
      PPCX_SWEEP_CELL(y0, e0, x0, -4 * L, true)            // (MODE 2 reads its further columns behind the advanced pointer)
      PPCX_SWEEP_CELL(y1, e1, x1, -3 * L, true)
      PPCX_SWEEP_CELL(y2, e2, x2, -2 * L, true)
      PPCX_SWEEP_CELL(y3, e3, x3, -L, true)
    }
    y0 = n0; y1 = n1; y2 = n2; y3 = n3;
  }
  if (k < nlane) {                                         // the last, partial trip
    const double e0 = q[0], e1 = q[L], e2 = q[2 * L], e3 = q[3 * L];
    double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
    if (MODE != 0) { x0 = qx[0]; x1 = qx[L]; x2 = qx[2 * L]; x3 = qx[3 * L]; }
    PPCX_SWEEP_CELL(y0, e0, x0, 0, true)
    PPCX_SWEEP_CELL(y1, e1, x1, L, k + 1 < nlane)
    PPCX_SWEEP_CELL(y2, e2, x2, 2 * L, k + 2 < nlane)
    PPCX_SWEEP_CELL(y3, e3, x3, 3 * L, k + 3 < nlane)
  }
#undef PPCX_SWEEP_CELL
}

// a gene whose linear predictor has to be formed per cell: t = exposure_s + X_s . coef + sigma_raw, u = exp(t)
template <int CM, int L>
PPCX_HD void generic_cells(const Dims& d, const Cmd& c, const VecRef& v, int g, bool has_slopes, const int* row,
                           const double* sExpo, const double* sX, int sub, GeneParams<CM>& gp, const double* tab,
                           CellAcc<CM>& acc) {
  const int S = d.S, C = d.C;
  // the position being evaluated, from the coordinates' constants (coord_consts; gp.sigma_raw is set by the caller from them)
  gp.coef[0] = v.at(V_C2, d.off_intercept + g);
#pragma unroll
  for (int cc = 1; cc < CM; ++cc) gp.coef[cc] = (has_slopes && cc < C) ? v.at(V_C2, coef_index(d, cc, g)) : 0.0;
  (void)c;
  for (int s = sub; s < S; s += L) {
    const int y = row[s];
    if (y >= 0) {
      double t = sExpo[s] + gp.sigma_raw;
#pragma unroll
      for (int cc = 0; cc < CM; ++cc) if (cc < C) t += sX[cc * S + s] * gp.coef[cc];
      const double u = fast_exp(t);
      const double rho = cell_eval<CM, true>(y, u, 1.0, gp, tab, acc);
#pragma unroll
      for (int cc = 0; cc < CM; ++cc) if (cc < C) acc.Tx[cc] = fma(sX[cc * S + s], rho, acc.Tx[cc]);
    }
  }
}

// one lane's share of gene g: the hand-over sums before the L-lane reduction
// GEN: which routes with an exp per cell are compiled in -- bit 0: slopes on columns of any values (a continuous covariate;
// sweep_cells MODE 3), bit 1: a design without the column of ones (generic_cells). 0: a model in which every gene factorises
// (X[,1] == 1 and slopes only on indicator columns), which leaves the registers to the sweep; the kernels have one instantiation
// per value (0, 1, 2), the host-side emulation compiles both routes (3)
template <int CM, int L, int GEN = 3>
PPCX_HD void lane_gene_sums(const Dims& d, const Cmd& c, const VecRef& v, const CellData& m, int g, const GenePre& pre, int sub,
                            const double* sE, const double* sExpo, const double* sX, const double* tab, const double* wtab,
                            GeneSumsV<CM>& o
#ifdef PPCX_TRACE
                            , unsigned long long* trace_stamps = nullptr
#endif
                            ) {
#if defined(PPCX_TRACE) && defined(__HIP_DEVICE_COMPILE__)
#define PPCX_GSTAMP(k) do { if (trace_stamps) trace_stamps[k] = __builtin_readcyclecounter(); } while (0)
#else
#define PPCX_GSTAMP(k) ((void)0)
#endif
  const int S = d.S;
  const bool has_slopes = g < d.K && d.C >= 2;
  const bool two = has_slopes && d.x0_is_one && d.x1_binary;
  // slopes on columns of any values: the sweep with an exp per cell (MODE 3); only a design without the column of ones still
  // forms the whole linear predictor per cell (generic_cells)
  const bool lin = (GEN & 1) && d.x0_is_one && has_slopes && !two;
  const bool generic = (GEN & 2) && !d.x0_is_one;
  GeneParams<CM> gp;
  gp.phi = pre.phi;
  const double sigma = pre.sigma;
  gp.sigma_raw = sigma;
  gp.invphi = fast_rcp(gp.phi);                             // once per gene and pass: cheaper than another constant in memory
  const double A = pre.a0 * gp.invphi;                      // exp(intercept + sigma_raw)
  const bool masked = (pre.flags & 1) != 0;
  const int* row = m.counts + (long)g * S;
  // the gene's table: the lane's chunk of the panel's coefficients (DispDeal), requested here, ahead of the sweep
  constexpr int NL = DispDeal<L>::NL, CH = DispDeal<L>::CH;
  const DispRef dr = disp_ref(sigma);
  const int fn = L == 1 ? 0 : sub / NL, jch = L == 1 ? 0 : sub - fn * NL;       // function (0: Fh, 1: Dh; >= 2: none), chunk
  const bool tab_lane = dr.in && (L == 1 || fn < 2);
  const double* pc = m.disp + (((long)g * kDispPanels + dr.panel) * 2 + (fn & 1)) * kDispStride + jch * CH;
  double cf[CH], cf2[L == 1 ? CH : 1];
#pragma unroll
  for (int k = 0; k < CH; ++k) cf[k] = 0.0;
  cf2[0] = 0.0;
  if (tab_lane) {
#pragma unroll
    for (int k = 0; k < CH; ++k) cf[k] = pc[k];
    if (L == 1) {
#pragma unroll
      for (int k = 0; k < CH; ++k) cf2[k] = pc[kDispStride + k];
    }
  }
  CellAcc<CM> acc; acc.zero();
  PPCX_GSTAMP(5);
  if (GEN & 2) {
    if (PPCX_WAVE_ANY(generic)) {
      if (generic) generic_cells<CM, L>(d, c, v, g, has_slopes, row, sExpo, sX, sub, gp, tab, acc);
    }
  }
  GeneWindow gw; gw.k = 0; gw.ok = false; gw.scale = 1.0;
  bool win = false;
  if (!(GEN & 2) || PPCX_WAVE_ANY(!generic)) {
    if (!generic) {
      const bool any_masked = PPCX_WAVE_ANY(masked);
      // the pass runs the windowed cell if every gene of it has a window (and none has excluded cells: those passes are few,
      // they keep the one general sweep that tests the counts' sign)
#define PPCX_SWEEP(MODE, A_LO, A_HI, A_, A1_, X1_, ...)                                                                  \
      gw = gene_window(fma(m.e_min, A_LO, 1.0), fma(m.e_max, A_HI, 1.0));                                                 \
      win = !any_masked && PPCX_WAVE_ALL(gw.ok);                                                                          \
      if (any_masked) sweep_cells<CM, L, MODE, true, false>(S, row, sE, X1_, sub, A_, A1_, 1.0, gp, tab, acc, ##__VA_ARGS__);   \
      else if (win) sweep_cells<CM, L, MODE, false, true>(S, row, sE, X1_, sub, (A_) * gw.scale, (A1_) * gw.scale, gw.scale, gp, wtab, acc, ##__VA_ARGS__); \
      else sweep_cells<CM, L, MODE, false, false>(S, row, sE, X1_, sub, A_, A1_, 1.0, gp, tab, acc, ##__VA_ARGS__);
      if ((GEN & 1) && PPCX_WAVE_ANY(lin)) {                     // a continuous covariate among the slope columns: exp(X_s . slopes) per
        double be[CM];                                     // cell; a plain gene of the same pass (the host puts genes with slopes
        be[0] = 0.0;                                       // first) runs it with slopes 0
#pragma unroll
        for (int cc = 1; cc < CM; ++cc) be[cc] = (lin && cc < d.C) ? v.at(V_C2, coef_index(d, cc, g)) : 0.0;
        if (any_masked) sweep_cells<CM, L, 3, true, false>(S, row, sE, sX + S, sub, A, A, 1.0, gp, tab, acc, be, d.C);
        else sweep_cells<CM, L, 3, false, false>(S, row, sE, sX + S, sub, A, A, 1.0, gp, tab, acc, be, d.C);
      } else if (CM > 2 && d.C > 2 && PPCX_WAVE_ANY(two)) {       // indicator columns, C > 2 (factor designs): e^t = E_s A times the
        double ec[CM];                                     // exp(slope_c) of the sample's columns; a plain gene of the same
        ec[0] = 1.0;                                       // pass (the host puts genes with slopes first) runs it with ec = 1
        double a_lo = A, a_hi = A;
#pragma unroll
        for (int cc = 1; cc < CM; ++cc) {
          ec[cc] = (two && cc < d.C) ? v.at(V_C0, coef_index(d, cc, g)) : 1.0;
          a_lo = ec[cc] < 1.0 ? a_lo * ec[cc] : a_lo; a_hi = ec[cc] > 1.0 ? a_hi * ec[cc] : a_hi;
        }
        const double A1 = A * ec[1];
        PPCX_SWEEP(2, a_lo, a_hi, A, A1, sX + S, ec, d.C)
      } else if (PPCX_WAVE_ANY(two)) {                     // e^t = E_s A or E_s A1 by the sample's group (X[,2] is 0 or 1)
        const double A1 = two ? A * v.at(V_C0, coef_index(d, 1, g)) : A;
        PPCX_SWEEP(1, (A1 < A ? A1 : A), (A1 > A ? A1 : A), A, A1, sX + S)
      } else {
        PPCX_SWEEP(0, A, A, A, A, sX)
      }
#undef PPCX_SWEEP
    }
  }
  PPCX_GSTAMP(6);
  if (!win) gw.scale = 1.0;
  cell_acc_close<CM>(gp, acc, gw.scale, &o);
  if ((GEN & 1) && lin) o.Tx[0] = o.Sr;                           // X[,1] == 1: sum X_s1 rho is sum rho
  if (win && sub == 0 && gw.k != 0) {                       // the cells' logarithms were those of 2^-k w: k ln 2 per cell, once per gene
    const double kd = (double)gw.k;
    const double klog2 = fma(kd, 6.93147180369123816490e-01, kd * 1.90821492927058770002e-10);
    o.lik -= klog2 * fma(gp.phi, pre.n, pre.Sy);
    o.dph -= klog2 * pre.n;
  }
  if (tab_lane) {
    double hv = cf[CH - 1];
#pragma unroll
    for (int k = CH - 2; k >= 0; --k) hv = fma(hv, dr.x, cf[k]);
    if (NL > 1) {                                           // times x^(CH chunk): chunk 0 .. NL - 1
      const double x3 = dr.x * dr.x * dr.x, xc = CH == 3 ? x3 : x3 * x3;
      double xp = (jch & 1) ? xc : 1.0;
      if (NL > 2) xp = (jch & 2) ? xp * (xc * xc) : xp;
      hv *= xp;
    }
    if (L == 1) {
      double h2 = cf2[CH - 1];
#pragma unroll
      for (int k = CH - 2; k >= 0; --k) h2 = fma(h2, dr.x, cf2[k]);
      o.lik += hv; o.dph += h2;
    } else if (fn == 0) o.lik += hv;
    else o.dph += hv;
  }
  if (PPCX_WAVE_ANY(!dr.in)) {                              // outside the tabulated range: the functions themselves, from the row
    PPCX_KEEP_BRANCH();
    if (!dr.in) {
      double F, D;
      disp_row_at(row, S, sub, L, sigma, &F, &D);
      o.lik += F; o.dph += D;
    }
  }
}

// close the gene with its reduced sums: gradient, second half kick, stores, partial sums part[0..9].
// The gene's data constants and its coordinates' momenta / metric arrive in registers (gene_finish loads them; the gene
// kernel of a pipelined round has requested them at its start, together with everything else it reads).
struct GeneData { double Sy, SyE, ncell, Lg1, SyX[kMaxC], SX[kMaxC]; };
template <int CM>
PPCX_HD void gene_data_load(const Dims& d, int gg, const double* Sy, const double* SyE, const double* SyXg, const double* SXg,
                            const double* ncell, const double* Lg1, GeneData& o) {
  o.Sy = Sy[gg]; o.SyE = SyE[gg]; o.ncell = ncell[gg]; o.Lg1 = Lg1[gg];
#pragma unroll
  for (int cc = 0; cc < CM; ++cc) {
    if (cc == 0 && d.x0_is_one) { o.SyX[0] = o.Sy; o.SX[0] = o.ncell; continue; }     // X[,1] == 1: the same sums, bit for bit
    o.SyX[cc] = (cc < d.C) ? SyXg[(long)cc * d.G + gg] : 0.0;
    o.SX[cc] = (cc < d.C) ? SXg[(long)cc * d.G + gg] : 0.0;
  }
}
template <int CM>
PPCX_HD void gene_finish_vals(const Dims& d, const Cmd& c, const VecRef& v, const GeneCtx<CM>& x, GeneSumsV<CM>& acc,
                              const GeneData& gd, const double* p_in, const double* minv, double* part, double* pn,
                              double* gn = nullptr) {
  constexpr int NCM = CM + 1;
  if (x.fast) acc.Tx[0] = acc.Sr;               // X[,1] == 1
  GeneOut<CM> go;
  gene_close<CM>(d, c.hy, x.gg, x.has_slopes, x.gp, acc, gd.Sy, gd.SyE, gd.SyX, gd.SX, gd.ncell, gd.Lg1, &go);
#pragma unroll
  for (int k = 0; k < 10; ++k) part[k] = 0.0;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < NCM; ++j) {
    const double gnew = j == 0 ? go.g_coef[0] : (j == 1 ? go.g_sigma_raw : go.g_coef[j >= 2 ? j - 1 : 0]);
    pn[j] = 0.0;
    if (gn) gn[j] = gnew;
    if (j < x.ncoord) {
      pn[j] = p_in[j] + 0.5 * c.eps * gnew;      // second half kick
      v.at(V_P0 + 3 * c.dir, x.idx[j]) = pn[j];
      v.at(V_G0 + 3 * c.dir, x.idx[j]) = gnew;
      part[PT_T1] += pn[j] * pn[j] * minv[j];
      bad = bad || !isfinite(gnew);
    }
  }
  if (x.active) {
    part[PT_LP] = go.lp;
#pragma unroll
    for (int k = 0; k < 6; ++k) part[PT_H0 + k] = go.h[k];
    part[PT_NONFINITE] = bad ? 1.0 : 0.0;
  }
}
template <int CM>
PPCX_HD void gene_finish(const Dims& d, const Cmd& c, const VecRef& v, const GeneCtx<CM>& x, GeneSumsV<CM>& acc,
                         const double* Sy, const double* SyE, const double* SyXg, const double* SXg, const double* ncell,
                         const double* Lg1, double* part, double* pn, double* minv, double* gn = nullptr) {
  constexpr int NCM = CM + 1;
  GeneData gd;
  gene_data_load<CM>(d, x.gg, Sy, SyE, SyXg, SXg, ncell, Lg1, gd);
  double p_in[NCM];
#pragma unroll
  for (int j = 0; j < NCM; ++j) {
    minv[j] = 1.0; p_in[j] = 0.0;
    if (j < x.ncoord) { minv[j] = v.at(V_MINV, x.idx[j]); p_in[j] = v.at(V_P0 + 3 * c.dir, x.idx[j]); }
  }
  gene_finish_vals<CM>(d, c, v, x, acc, gd, p_in, minv, part, pn, gn);
}

// update kernel, one gene-owned coordinate: pre-operations of the new command, then the first half kick and the drift of
// the next leapfrog (written in place into the end being advanced) and the constants of the new position. A command
// without a step (eps = 0: the evaluation of a given point) leaves position and momentum as they are.
PPCX_HD void coord_update(const Dims& d, const Cmd& nc, const VecRef& v, int i, double* draws, double* T0,
                          const CoordCache* cc = nullptr) {
  const CoordVals cv = coord_pre(nc, v, i, i, global_flat(d, i), true, draws, d.D, nc.k0, nc.k1, T0, cc);
  if (nc.type == CMD_FLUSH) return;
  double qn = cv.q;
  if (nc.eps != 0.0) {
    double ph;
    kick_drift(cv.q, cv.p, cv.g, nc.eps, cv.minv, &ph, &qn);
    v.at(V_P0 + 3 * nc.dir, i) = ph;
    v.at(V_Q0 + 3 * nc.dir, i) = qn;
  }
  coord_consts(d, v, i, qn);
}

// ---------------------------------------------------------------------------------------------------------------
// Pipelined rounds (two launches per leapfrog: ppcx_ls_kernel, ppcx_gene_kernel). The gene kernel does everything that
// belongs to ONE gene, one thread per gene: the per-coordinate work of the command (gene_coord_update: what coord_update
// does per coordinate), the close of the evaluated leaf (gene_finish + tree bookkeeping), and -- ahead of the state
// machine's decision -- the constants of the position the next leaf of the same subtree would evaluate (gene_spec_consts),
// so that the next log-likelihood launch can run beside the state machine instead of after it.
// ---------------------------------------------------------------------------------------------------------------
// the command's work on the gene's coordinates; leaves the new positions in x.q[]. consts: also the constants of the new
// positions (commands whose position the log-likelihood kernel has not evaluated ahead of time).
// cache: the coordinates' end states requested ahead (coord_prefetch_for) or null; p_out / minv_out: momentum after the
// first half kick and the metric, for a close that follows in the same thread (null: not wanted).
// A command with pre-operations other than a leaf's inside a transition (kPreCommon: proposal / sample copies, the saved
// near end) -- the first leaf of a transition, a step-size trial, the initial point: one command in thirty of a fit -- has ALL
// its pre-operations done here, in a pass of its own in front of everything else the thread does, through memory; the
// loop below then runs as for a command without pre-operations and reads what this pass stored. With the rare paths inside
// the loop the compiler moves what its unrolled iterations share of them -- the Philox key schedule, the metric's two
// divisions, every field of the command they read, ~70 scalars spilled to vector lanes -- in front of the loop, where every
// command pays for it: 130 of a wavefront's 920 vector instructions and 0.6 us of the launch (round 4,
// profiles/r04_sq_counters_gene.txt); a second copy of the loop for such commands costs the kernel its registers instead.
// Returns the pre-operations left for the loop (coord_pre's fmask). Must run before the coordinates are prefetched.
template <int CM>
PPCX_HD int gene_rare_pre(const Dims& d, const Cmd& c, const VecRef& v, const GeneCtx<CM>& x, double* draws, double* T0) {
  if ((c.pre_flags & ~kPreCommon) == 0) return ~0;
  constexpr int NCM = CM + 1;
#pragma unroll
  for (int j = 0; j < NCM; ++j) {
    if (j < x.ncoord) { const int i = x.idx[j]; (void)coord_pre<~0>(c, v, i, i, global_flat(d, i), true, draws, d.D, c.k0, c.k1, T0); }
  }
  return 0;
}
// fmask: what gene_rare_pre returned when the caller ran it (it has to, ahead of its prefetch, when it passes a cache);
// -1 = not run yet.
template <int CM, bool CACHED = false>
PPCX_HD void gene_coord_update(const Dims& d, const Cmd& c, const VecRef& v, GeneCtx<CM>& x, double* draws, double* T0,
                               bool consts, const CoordCache* cache = nullptr, double* p_out = nullptr,
                               double* minv_out = nullptr, bool store_p = true, int fmask = -1) {
  constexpr int NCM = CM + 1;
  if (!CACHED && fmask == -1) fmask = gene_rare_pre<CM>(d, c, v, x, draws, T0);
#pragma unroll
  for (int j = 0; j < NCM; ++j) {
    x.q[j] = 0.0;
    if (p_out) { p_out[j] = 0.0; minv_out[j] = 1.0; }
    if (j < x.ncoord) {
      const int i = x.idx[j];
      CoordVals cv;
      if (CACHED) { const CoordCache cj = cache[j]; cv = coord_pre<kPreCommon>(c, v, i, i, global_flat(d, i), true, draws, d.D, c.k0, c.k1, T0, &cj, fmask); }
      else cv = coord_pre<kPreCommon>(c, v, i, i, global_flat(d, i), true, draws, d.D, c.k0, c.k1, T0, nullptr, fmask);
      double qn = cv.q, ph = cv.p;
      if (c.type != CMD_FLUSH) {
        if (c.eps != 0.0) {
          kick_drift(cv.q, cv.p, cv.g, c.eps, cv.minv, &ph, &qn);
          if (store_p) v.at(V_P0 + 3 * c.dir, i) = ph;       // not when the close that follows overwrites it anyway
          v.at(V_Q0 + 3 * c.dir, i) = qn;
        }
        if (consts) coord_consts(d, v, i, qn);
      }
      x.q[j] = qn;
      if (p_out) { p_out[j] = ph; minv_out[j] = cv.minv; }
    }
  }
}
// constants of the position the NEXT leaf evaluates if the tree goes on (Cmd::next_dir): inside a subtree, or into a new
// doubling in the same direction, q + eps minv (pn + eps/2 gn) from the values the close has in registers; into a new
// doubling in the OTHER direction the same step from the other end of the trajectory, with the step's sign turned.
template <int CM>
PPCX_HD void gene_spec_consts(const Dims& d, const Cmd& c, const VecRef& v, const GeneCtx<CM>& x, const double* pn,
                              const double* gn, const double* minv) {
  constexpr int NCM = CM + 1;
  const bool turn = c.next_dir != c.dir;
  const int o = 1 - c.dir;
#pragma unroll
  for (int j = 0; j < NCM; ++j) {
    if (j < x.ncoord) {
      double ph, qn;
      if (turn) kick_drift(v.at(V_Q0 + 3 * o, x.idx[j]), v.at(V_P0 + 3 * o, x.idx[j]), v.at(V_G0 + 3 * o, x.idx[j]), -c.eps, minv[j], &ph, &qn);
      else kick_drift(x.q[j], pn[j], gn[j], c.eps, minv[j], &ph, &qn);
      coord_consts(d, v, x.idx[j], qn);
    }
  }
}

// Kernel B, serial part (one thread per chain): finish the hyper coordinates of the executed command,
// advance the state machine, run the hyper pre-operations + half kick + drift of the next command.
// `red` are the block partials reduced in a fixed order; `hv` the chain's hyper-coordinate vectors.
struct ChainIO {
  double* draws;                 // this chain's [n_keep][D] or null
  ChainOut out;
};
// How the six hyper coordinates are spread over the cooperating lanes. SerialLanes: one thread does all six
// (host emulation, tests). The device uses eight lanes of one wavefront (kernels.hip: WaveLanes): lane k < 6 owns
// hyper coordinate k, sums cross the lanes by xor shuffles, and every lane runs the scalar logic redundantly on
// identical inputs so no broadcast of the decisions is needed.
struct SerialLanes {
  static constexpr int kPerLane = 6;           // coordinates one lane owns (sizes the per-lane scratch arrays)
  PPCX_HD int k_begin() const { return 0; }
  PPCX_HD int k_end() const { return 6; }
  PPCX_HD bool leader() const { return true; }
  PPCX_HD double sum(double v) const { return v; }
  PPCX_HD double pick(const double* own, int k) const { return own[k]; }   // value held by the owner of coordinate k
};

// `red` holds the sums of the executed command over all gene coordinates (all shards); `rd` is scratch (LDS on the
// device) that ends up holding those sums plus the hyper coordinates' own terms.
template <class Lanes>
PPCX_HD void chain_step(const Lanes& ln, const Dims& d, ChainScalars& st, TreeArrays& ta, const Cmd& ex,
                        const double* red, bool have_parts, const VecRef& hv, const ChainIO& io, Reduced& rd, Cmd& nc) {
  // a lane's own coordinates sit in small per-lane arrays; with one coordinate per lane (the device) the slot is the
  // constant 0, so that the arrays stay in registers instead of becoming run-time-indexed stack objects
#define PPCX_SLOT(k, kb_) (Lanes::kPerLane == 1 ? 0 : (k) - (kb_))
  double lp = 0.0; bool finite = true;
  if (have_parts && ex.type != CMD_FLUSH) {
    double g6[6], hs[6];
    for (int k = 0; k < 6; ++k) hs[k] = red[PT_H0 + k];
    lp = hyper_close(d, ex.hy, ex.hyp_q, red[PT_LP], hs, g6);
    double T1h = 0.0, bad = 0.0;
    constexpr int NK = Lanes::kPerLane;
    const int kb = ln.k_begin();
    double pn_k[NK], mv_k[NK];                 // indexed by k - k_begin
    for (int k = ln.k_begin(); k < ln.k_end(); ++k) {   // second half kick of the hyper coordinates
      const double minv = hv.at(V_MINV, k);
      const double pn = hv.at(V_P0 + 3 * ex.dir, k) + 0.5 * ex.eps * g6[k];
      hv.at(V_P0 + 3 * ex.dir, k) = pn; hv.at(V_G0 + 3 * ex.dir, k) = g6[k];
      T1h += pn * pn * minv;
      bad += isfinite(g6[k]) ? 0.0 : 1.0;
      pn_k[PPCX_SLOT(k, kb)] = pn; mv_k[PPCX_SLOT(k, kb)] = minv;
    }
    T1h = ln.sum(T1h); bad = ln.sum(bad);
    finite = red[PT_NONFINITE] == 0.0 && bad == 0.0;
    if (ln.leader()) {
      rd.lp_genes = red[PT_LP];
      for (int k = 0; k < 6; ++k) rd.hsum[k] = hs[k];
      rd.T0 = (st.T0g_held ? st.T0g : red[PT_T0]) + st.T0h; rd.T1 = red[PT_T1] + T1h; rd.nonfinite = red[PT_NONFINITE];
    }
    if (ex.type == CMD_LEAF) {                 // tree terms of the hyper coordinates, level by level
      NodeVals nv[NK];
      for (int k = ln.k_begin(); k < ln.k_end(); ++k) nv[PPCX_SLOT(k, kb)] = NodeVals{pn_k[PPCX_SLOT(k, kb)], pn_k[PPCX_SLOT(k, kb)]};
      for (int l = 0; l < ex.n_merge; ++l) {
        double dots[6] = {0, 0, 0, 0, 0, 0};
        for (int k = ln.k_begin(); k < ln.k_end(); ++k) coord_merge_dots(hv, k, l, pn_k[PPCX_SLOT(k, kb)], mv_k[PPCX_SLOT(k, kb)], &nv[PPCX_SLOT(k, kb)], dots);
        for (int j = 0; j < 6; ++j) { const double t = ln.sum(dots[j]); if (ln.leader()) rd.dots[l][j] = red[PT_DOTS + 6 * l + j] + t; }
      }
      if (!ex.subtree_complete) {
        for (int k = ln.k_begin(); k < ln.k_end(); ++k) coord_store_slot(hv, k, ex.n_merge, pn_k[PPCX_SLOT(k, kb)], nv[PPCX_SLOT(k, kb)]);
      } else {
        double top[6] = {0, 0, 0, 0, 0, 0};
        for (int k = ln.k_begin(); k < ln.k_end(); ++k) coord_top_dots(hv, k, ex.dir, pn_k[PPCX_SLOT(k, kb)], mv_k[PPCX_SLOT(k, kb)], nv[PPCX_SLOT(k, kb)], top);
        for (int j = 0; j < 6; ++j) { const double t = ln.sum(top[j]); if (ln.leader()) rd.top[j] = red[PT_TOP + j] + t; }
      }
    }
  }
  ChainOut out = io.out;
  if (!ln.leader()) { out.lp = nullptr; out.stepsize = nullptr; out.treedepth = nullptr; out.n_leapfrog = nullptr; out.divergent = nullptr; out.accept = nullptr; }
  chain_advance(st, ta, ex, rd, lp, finite, out, nc);       // every lane: same inputs, same decisions
  nc.k0 = st.k0; nc.k1 = st.k1;
  if (nc.type != CMD_DONE) {
    constexpr int NK2 = Lanes::kPerLane;
    const int kb2 = ln.k_begin();
    double T0h = 0.0, hq[NK2];
    for (int j = 0; j < NK2; ++j) hq[j] = 0.0;
    for (int k = ln.k_begin(); k < ln.k_end(); ++k) {
      const int hcol = hyper_index(d, k);
      const CoordVals cv = coord_pre(nc, hv, k, hcol, global_flat(d, hcol), true, io.draws, d.D, st.k0, st.k1, &T0h);
      hq[PPCX_SLOT(k, kb2)] = cv.q;
      if (nc.type != CMD_FLUSH) {
        const double ph = cv.p + 0.5 * nc.eps * cv.g;
        hq[PPCX_SLOT(k, kb2)] = cv.q + nc.eps * cv.minv * ph;
        hv.at(V_Q0 + 3 * nc.dir, k) = hq[PPCX_SLOT(k, kb2)]; hv.at(V_P0 + 3 * nc.dir, k) = ph;
      }
    }
    st.T0h = ln.sum(T0h);
    if (nc.type != CMD_FLUSH) {
#pragma unroll
      for (int k = 0; k < 6; ++k) nc.hyp_q[k] = ln.pick(hq, k);
    }
    nc.hy = make_hyper(nc.hyp_q, d.lambda_mu_mu);
  }
}
#undef PPCX_SLOT

// The state machine's part of a pipelined round. `ex` is the chain's current command. If the log-likelihood launch that
// just ran had not evaluated it yet (its position was not the one anticipated), nothing is decided: the command is carried
// to the next round, now evaluated. Otherwise the chain steps, and the new command is marked as already evaluated when it
// is the continuation the gene kernel anticipated (spec: the model's cell paths read the anticipated constants only).
// Returns whether the chain stepped.
template <class Lanes>
PPCX_HD bool chain_step_pipelined(const Lanes& ln, const Dims& d, ChainScalars& st, TreeArrays& ta, const Cmd& ex,
                                  const double* red, const VecRef& hv, const ChainIO& io, Reduced& rd, Cmd& nc, bool spec) {
  if (st.phase != PH_START && cmd_evaluates(ex) && !ex.evaluated) {
    nc = ex; nc.evaluated = 1; nc.updated = 1;
    st.T0g = red[PT_T0]; st.T0g_held = 1;      // left by the round that applied the command (its slab is the one reduced here)
    return false;
  }
  chain_step(ln, d, st, ta, ex, red, st.phase != PH_START, hv, io, rd, nc);
  st.T0g_held = 0;
  nc.updated = 0;
  nc.evaluated = (spec && spec_continues(ex, nc)) ? 1 : 0;
  return true;
}

}  // namespace ppcx
