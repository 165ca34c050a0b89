// ppcx_kernels.hip -- gfx950 kernels of the NB hierarchical NUTS / posterior-predictive engine.
//
//   ppcx_loglik_kernel<CM,GEN>  streams the int32 count matrix once per gradient evaluation and reduces it to a handful
//                           of sums per gene. Replaces lp_reduce + map_rect + X*alpha of
//                           inst/stan/negBinomial_MPI.stan:58-120,:205,:226-240. CM = design columns (2, 4, 8, 16), GEN = the
//                           route with an exp per cell that is compiled in (0 none, 1 continuous covariate, 2 no column of ones).
//   ppcx_ls_kernel<CM,GEN>  the merged launch of a pipelined round: those workgroups beside the chains' state machines.
//   ppcx_gene_kernel<CM>    the other launch of a pipelined round: everything per gene (command, close, anticipated constants).
//   ppcx_disp_build_kernel  the genes' dispersion tables (ppcx_disp.h), once per model and per change of the exclusions.
//   ppcx_close_kernel<CM>   per gene: priors (.stan:219-223), gradient, second half kick of the leapfrog, NUTS tree
//                           bookkeeping of the gene's coordinates, block partial sums.
//   ppcx_step_kernel        reduction of the close kernel's block partials, hyper-parameters, NUTS / adaptation state
//                           machine (ppcx_nuts.h) and, in the same launch, the per-coordinate work of the next command.
//   ppcx_update_kernel      that per-coordinate work as a launch of its own (initialisation, ADVI).
//   ppcx_ppc_kernel         generated quantities (.stan:259-266) + credible-interval summary
//                           (R/utilities.R:685-703 / :733-784): NB draws straight into LDS, order
//                           statistics by bisection on the value, type-7 quantiles, mean, sd.
//   ppcx_gather_kernel      column gather of the retained draws.
//
// Work decomposition of the log-likelihood kernel: a gene is owned by L lanes of one wavefront
// (L in {1,2,4,...,64}); a wavefront walks a range of the host's gene order, 64 / L genes per pass; lanes stride over the
// gene's samples four cells per trip, read the per-sample constants from LDS, and combine with an L-lane butterfly.
// Per-block partial sums of the close kernel go to a slab that the step kernel reduces in a fixed order, so results are
// bitwise reproducible for a fixed number of lanes per gene.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "ppcx_gene.h"
#include "ppcx_kernels.h"

namespace ppcx {

// ChainState / Cmd travel between global memory, registers and LDS word by word (one 4-byte word per thread). The word type
// may alias any object: reading a struct with double members through a plain `int` lvalue is undefined under the strict
// aliasing rule that -O3 applies, whatever the copy looks like in practice.
typedef int __attribute__((may_alias)) word_t;
// 16-byte requests of arrays of doubles (LDS fills)
typedef double2 __attribute__((may_alias)) dpair_t;

// -----------------------------------------------------------------------------------------------------
// kernel A1: the log-likelihood kernel. Streams the count matrix once and leaves, per gene, the sums of
// GeneSumsV (ppcx_model.h): the likelihood part of the gene's log density, its d/dphi part, sum rho
// (+ sum X_sc rho for genes with slopes).
// -----------------------------------------------------------------------------------------------------
#ifndef PPCX_LOGLIK_OCC
#define PPCX_LOGLIK_OCC 2
#endif
#ifndef PPCX_LOGLIK_OCC_FAST
#define PPCX_LOGLIK_OCC_FAST 4                   // wavefronts per SIMD of the instantiation without the per-cell-eta path
#endif
__device__ __forceinline__ double wave_xor_add_rt(double v, int lane_xor_mask) {      // run-time mask: ds_bpermute
  const int src = (int)((threadIdx.x ^ (unsigned)lane_xor_mask) & 63u) << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
  return v + __hiloint2double(hi, lo);
}
// v of lane i moved by a DPP control word (data-parallel primitives: a VALU move, no LDS round trip); lanes the
// control leaves without a source, or outside row_mask, receive 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum over the L lanes that share a gene, every lane ending with the total: inside a row of 16 lanes by DPP moves (neighbours,
// pairs, the other quad by row_half_mirror, the other half-row by row_mirror -- vector-unit moves, no LDS round trips: nine
// dependent ds_bpermute round trips per pass were half a microsecond of nothing but latency), across rows by ds_bpermute
template <int L>
__device__ __forceinline__ double group_sum(double v) {
  if (L >= 2) v += dpp_take<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
  if (L >= 4) v += dpp_take<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
  if (L >= 8) v += dpp_take<0x141, 0xf>(v);    // row_half_mirror
  if (L >= 16) v += dpp_take<0x140, 0xf>(v);   // row_mirror
  if (L >= 32) v = wave_xor_add_rt(v, 16);
  if (L >= 64) v = wave_xor_add_rt(v, 32);
  return v;
}
// A resident launch (host: plan_launch): as many workgroups as the chip holds at once (4 per CU at 128 VGPRs), divided
// among the chains; every wavefront owns a contiguous range of the host's gene order -- bounds[j] .. bounds[j + 1] --
// chosen on the host so that all ranges cost the same, and walks it 64 / L genes at a time. All wavefronts start
// together and finish together: no partly filled last round of workgroups, one LDS fill per resident workgroup.
// Wavefronts are independent after the LDS fill (no barrier, no atomic), and a gene's sums depend on L only.
template <int CM, int LG, int GEN>
__device__ __forceinline__ void loglik_passes(const LoglikArgs& a, const Cmd& c, const VecRef& v, double* sums, int p0, int p1,
                                              const double* stab, const double* swin, const double* sE, const double* sExpo, const double* sX,
                                              int lane, bool any_generic) {
  constexpr int L = 1 << LG, GPW = 64 >> LG;     // lanes per gene, genes per wavefront and pass
  const Dims& d = a.d;
  const int sub = lane & (L - 1), gl = lane >> LG;
  // One pass ahead: while pass p is swept, the constants of pass p + 1's genes (gene_pre_load) and the gene indices of pass
  // p + 2 are on their way.
  int g_cur = a.order[p0 + gl < p1 ? p0 + gl : p1 - 1];
  int g_next = a.order[p0 + GPW + gl < p1 ? p0 + GPW + gl : p1 - 1];
  GenePre pre_cur = gene_pre_load(d, v, a.cd, g_cur);
#ifdef PPCX_TRACE
  unsigned long long* tr = nullptr;
  if (a.trace && lane == 0 && (blockIdx.x % 16) == 0 && blockIdx.x / 16 < kTraceBlocks)
    tr = a.trace + (((long)(blockIdx.x / 16) * 4 + (threadIdx.x >> 6)) * kTracePasses) * kTraceStamps;
  int tpass = 0;
#define PPCX_STAMP(k) do { if (tr && tpass < kTracePasses) tr[tpass * kTraceStamps + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PPCX_STAMP(k) ((void)0)
#endif
  for (int p = p0; p < p1; p += GPW) {
    PPCX_STAMP(0);
#ifdef PPCX_TRACE
    if (tr && tpass < kTracePasses) tr[tpass * kTraceStamps + 7] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef PPCX_PRIO_BALANCE
    // the SIMD serves its oldest wavefront first, so its four wavefronts (one from each quarter of the launch's workgroups)
    // finish one after the other and the last ones run alone, at a fraction of the issue rate: a wavefront with more passes
    // left goes first instead
    {
      const int rem = (p1 - p + GPW - 1) / GPW;
      if (rem >= 4) __builtin_amdgcn_s_setprio(3); else if (rem == 3) __builtin_amdgcn_s_setprio(2); else if (rem == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
#endif
    const bool act = p + gl < p1;                // lanes past the end of the range repeat its last gene and store nothing
    const int g = g_cur;
    const GenePre pre = pre_cur;
    if (p + GPW < p1) {
      g_cur = g_next;
      pre_cur = gene_pre_load(d, v, a.cd, g_next);
      const int pn = p + 2 * GPW + gl;
      g_next = a.order[pn < p1 ? pn : p1 - 1];
    }
    if (CM > 2 && d.x0_is_one && !PPCX_WAVE_ANY(g < d.K)) {
      // a pass of plain genes in a model with more than two design columns: the two-column instantiation of the gene's work
      // (its accumulators carry two slope sums, not CM: the registers the wider one needs cost the row sweep spills), and
      // the three sums of a plain gene
      GeneSumsV<2> o2;
      lane_gene_sums<2, L, 0>(d, c, v, a.cd, g, pre, sub, sE, sExpo, sX, stab, swin, o2);
      o2.lik = group_sum<L>(o2.lik); o2.dph = group_sum<L>(o2.dph); o2.Sr = group_sum<L>(o2.Sr);
      if (act && sub == 0) { const long G = d.G; sums[0 * G + g] = o2.lik; sums[1 * G + g] = o2.dph; sums[2 * G + g] = o2.Sr; }
      continue;
    }
    GeneSumsV<CM> o;
    PPCX_STAMP(1);
#ifdef PPCX_TRACE
    lane_gene_sums<CM, L, GEN>(d, c, v, a.cd, g, pre, sub, sE, sExpo, sX, stab, swin, o, (tr && tpass < kTracePasses) ? tr + tpass * kTraceStamps : nullptr);
#else
    lane_gene_sums<CM, L, GEN>(d, c, v, a.cd, g, pre, sub, sE, sExpo, sX, stab, swin, o);
#endif
    PPCX_STAMP(2);
    // sum X_sc rho is needed of genes with slopes only (and of every gene when X[,1] != 1): a pass without such genes
    // neither reduces nor stores it (the close kernel does not use those entries of a plain gene)
    const bool with_tx = any_generic && (!d.x0_is_one || PPCX_WAVE_ANY(g < d.K));
    // every lane of the gene ends with the gene totals
    o.lik = group_sum<L>(o.lik); o.dph = group_sum<L>(o.dph); o.Sr = group_sum<L>(o.Sr);
    PPCX_STAMP(3);
    if (with_tx) {
#pragma unroll
      for (int cc = 0; cc < CM; ++cc) if (cc < d.C) o.Tx[cc] = group_sum<L>(o.Tx[cc]);
    }
    if (act && sub == 0) {
      const long G = d.G;
      sums[0 * G + g] = o.lik; sums[1 * G + g] = o.dph; sums[2 * G + g] = o.Sr;
      if (with_tx) {
#pragma unroll
        for (int cc = 0; cc < CM; ++cc) if (cc < d.C) sums[(3 + cc) * G + g] = o.Tx[cc];
      }
    }
    PPCX_STAMP(4);
#ifdef PPCX_TRACE
    ++tpass;
#endif
  }
#undef PPCX_STAMP
}

// the body of a log-likelihood workgroup: range block jb of the chain in column `col` of the launch.
// PIPE: part of a pipelined round's merged launch (ppcx_ls_kernel) -- the command in a.cmds is then the chain's command
// BEFORE the state machine that runs beside this workgroup has looked at it.
template <int CM, int GEN, bool PIPE>
__device__ __forceinline__ void loglik_role(const LoglikArgs& a, int jb, int col, double* lds) {
  if (jb >= a.nbpc) return;
  constexpr int NS = GeneSums<CM>::N;
  const Dims& d = a.d;
  const int S = d.S, C = d.C;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  // What the workgroup stages in LDS and the wavefront's range are requested FIRST, before the chain's command is looked at
  // (three dependent scalar round trips: active list, command, its fields): the start of a launch is a chain of round trips
  // that every wavefront of the chip walks at the same time, and these need not be links of it. The first 256 (512) entries
  // travel in registers across the checks; longer per-sample arrays are completed afterwards.
  static_assert(2 * kLogTabSize == 2 * 256, "one 16-byte request per thread fills the table");
  const bool any_generic = !d.x0_is_one || (C >= 2 && d.K > 0);
  const bool evenS = (S & 1) == 0;             // then exp(exposure) is 16-byte aligned on both sides
  const int p0 = a.bounds[jb * 4 + wave], p1 = a.bounds[jb * 4 + wave + 1];
  const double2 f_tab = reinterpret_cast<const dpair_t*>(a.logtab)[tid];
  static_assert(2 * kWinTabSize == 4 * 2 * 256, "four 16-byte requests per thread fill the window table");
  double2 f_win[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) f_win[k] = reinterpret_cast<const dpair_t*>(a.wintab)[tid + 256 * k];
  double2 f_e = {0.0, 0.0};
  if (evenS) { if (tid < S / 2) f_e = reinterpret_cast<const dpair_t*>(a.sampleE)[tid]; }
  else if (tid < S) f_e.x = a.sampleE[tid];
  double f_x = 0.0;
  if (GEN != 2 && any_generic && tid < S) f_x = a.X[S + tid];      // the two-group path reads the group column only (C > 2: below)
  const int chain = a.active ? a.active[col] : col;
  const Cmd& c = a.cmds[chain];
  if (c.type == CMD_DONE || c.type == CMD_FLUSH) return;
  if (PIPE) {
    // evaluated and not a leaf: the gene kernel has closed it and anticipated nothing -- there is no position to evaluate
    // until the state machine has decided (a leaf that has been closed left the constants of the anticipated next leaf;
    // a command that has not been evaluated left its own)
    if (c.evaluated && c.type != CMD_LEAF) return;
  }
  double* stab = lds;                          // log table: 256 x 1/c then 256 x log c (4 KB)
  double* swin = lds + 2 * kLogTabSize;        // window table: 1024 x 1/c then 1024 x log c (16 KB)
  double* sE = swin + 2 * kWinTabSize;         // exp(exposure_s), readable kLdsPad entries past S (sweep_cells)
  // a design without the column of ones (GEN == 2) keeps the exposures and the whole of X; every other instantiation reads the
  // columns 1 .. C - 1 only: column c at sX + c S, column 0 and the exposures are not staged (the space of column 0 is sE's)
  double* sExpo = GEN == 2 ? sE + S + kLdsPad : nullptr;
  double* sX = GEN == 2 ? sExpo + S : sE + kLdsPad;      // S x C column-major, readable kLdsPad entries past its end
  const VecRef v{a.vecs + (long)chain * V_COUNT * a.Dpad, a.Dpad};
  double* sums = a.sums + (long)chain * NS * d.G;
  // the fill: every workgroup of the launch reads the same few KB at the same time, so as few requests as possible -- 16 bytes
  // per lane where the alignment allows
  reinterpret_cast<dpair_t*>(stab)[tid] = f_tab;
#pragma unroll
  for (int k = 0; k < 4; ++k) reinterpret_cast<dpair_t*>(swin)[tid + 256 * k] = f_win[k];
  if (evenS) {
    if (tid < S / 2) reinterpret_cast<dpair_t*>(sE)[tid] = f_e;
    for (int i = tid + 256; i < S / 2; i += 256) reinterpret_cast<dpair_t*>(sE)[i] = reinterpret_cast<const dpair_t*>(a.sampleE)[i];
  } else {
    if (tid < S) sE[tid] = f_e.x;
    for (int i = tid + 256; i < S; i += 256) sE[i] = a.sampleE[i];
  }
  if (any_generic) {
    if (GEN == 2) {
      for (int i = tid; i < S; i += 256) sExpo[i] = a.exposure[i];
      for (int i = tid; i < S * C; i += 256) sX[i] = a.X[i];
    } else {
      if (tid < S) sX[S + tid] = f_x;
      for (int i = tid + 256; i < S; i += 256) sX[S + i] = a.X[S + i];
      for (int i = 2 * S + tid; i < S * C; i += 256) sX[i] = a.X[i];      // further indicator columns (factor designs)
    }
  }
  __syncthreads();
  if (p0 >= p1) return;
  switch (a.lgL) {
    case 0: loglik_passes<CM, 0, GEN>(a, c, v, sums, p0, p1, stab, swin, sE, sExpo, sX, lane, any_generic); break;
    case 1: loglik_passes<CM, 1, GEN>(a, c, v, sums, p0, p1, stab, swin, sE, sExpo, sX, lane, any_generic); break;
    case 2: loglik_passes<CM, 2, GEN>(a, c, v, sums, p0, p1, stab, swin, sE, sExpo, sX, lane, any_generic); break;
    case 3: loglik_passes<CM, 3, GEN>(a, c, v, sums, p0, p1, stab, swin, sE, sExpo, sX, lane, any_generic); break;
    case 4: loglik_passes<CM, 4, GEN>(a, c, v, sums, p0, p1, stab, swin, sE, sExpo, sX, lane, any_generic); break;
    case 5: loglik_passes<CM, 5, GEN>(a, c, v, sums, p0, p1, stab, swin, sE, sExpo, sX, lane, any_generic); break;
    default: loglik_passes<CM, 6, GEN>(a, c, v, sums, p0, p1, stab, swin, sE, sExpo, sX, lane, any_generic); break;
  }
}

template <int CM, int GEN>
__global__ __launch_bounds__(256, GEN ? PPCX_LOGLIK_OCC : PPCX_LOGLIK_OCC_FAST) void ppcx_loglik_kernel(LoglikArgs a) {
  extern __shared__ double lds[];
  // workgroups are dealt to the 8 XCDs round-robin in dispatch order: ids chain * 8 + (jb & 7) inside every run of
  // 8 range blocks x chains put the chains of one range block on ONE XCD, so its L2 fetches the rows once
  const int nch = a.nchains;
  const int lin = blockIdx.x, run = lin / (8 * nch), r = lin - run * (8 * nch);
  loglik_role<CM, GEN, false>(a, run * 8 + (r & 7), r >> 3, lds);
}

// -----------------------------------------------------------------------------------------------------
// kernel A2: one thread per gene closes it -- priors, gradient, second half kick of the gene's coordinates,
// U-turn dot products and subtree slots -- and the workgroup leaves its partial sums in a slab.
// -----------------------------------------------------------------------------------------------------
// sum over the wavefront, left in lane 63: inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), then the
// row totals travel to the next row (row_bcast:15 into rows 1 and 3) and to the upper half (row_bcast:31 into rows 2, 3)
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v += dpp_take<0x111, 0xf>(v);
  v += dpp_take<0x112, 0xf>(v);
  v += dpp_take<0x114, 0xf>(v);
  v += dpp_take<0x118, 0xf>(v);
  v += dpp_take<0x142, 0xa>(v);
  v += dpp_take<0x143, 0xc>(v);
  return v;
}
template <int N>
__device__ __forceinline__ void block_accumulate(double* vals, double* wacc, int wave, int lane) {
#pragma unroll
  for (int k = 0; k < N; ++k) vals[k] = wave_sum_to_lane63(vals[k]);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < N; ++k) wacc[wave * PT_COUNT + k] = vals[k];
  }
}

// Sums of up to 16 quantities over the 64 lanes of a wavefront THROUGH LDS: row k = quantity k, one column per lane, rows padded
// to 65 doubles; lane 4 k + part adds columns 16 part .. 16 part + 15 of row k, the four parts meet by two cross-lane
// moves, and the lanes 4 k .. 4 k + 3 end with the total of quantity k. About 25 vector instructions and 32 LDS accesses for 16 sums,
// against 18 vector instructions PER sum of the cross-lane scan (wave_sum_to_lane63): the gene kernel spent a quarter of its 1100
// vector instructions per wavefront in those scans (profiles/r04_sq_counters_gene.txt). Fixed order, the scan's own (below): the same bits as before.
constexpr int kRowStride = 65, kPartStride = 16, kRowsPerWave = 16;     // 39 808 bytes of LDS per gene workgroup: four per CU (below)
template <int N>
__device__ __forceinline__ void wave_sums_lds(const double* vals, double* rows, double* out /* wacc + offset */, int lane) {
  static_assert(N >= 1 && N <= kRowsPerWave, "one batch");
#pragma unroll
  for (int k = 0; k < N; ++k) rows[k * kRowStride + lane + (kPartStride - 16) * (lane >> 4)] = vals[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int k = lane >> 2, part = lane & 3;
  double t = 0.0;
  if (k < N) {
    // the sixteen columns in the order of the cross-lane scan this replaces (wave_sum_to_lane63: neighbours, pairs of pairs,
    // ... -- a balanced tree), so that the sums, and with them every fit, keep the bits they had
    const double* r = rows + k * kRowStride + kPartStride * part;
    double b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = r[2 * j + 1] + r[2 * j];
    t = ((b[7] + b[6]) + (b[5] + b[4])) + ((b[3] + b[2]) + (b[1] + b[0]));
  }
  t += dpp_take<0xB1, 0xf>(t);                 // quad_perm [1,0,3,2]: parts 0 + 1, 2 + 3
  t += dpp_take<0x4E, 0xf>(t);                 // quad_perm [2,3,0,1]: the quad's total
  if (k < N && part == 0) out[k] = t;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();             // the rows are free again
}

template <int CM>
__global__ __launch_bounds__(256) void ppcx_close_kernel(CloseArgs a) {
  constexpr int NCM = CM + 1;
  constexpr int NS = GeneSums<CM>::N;
  __shared__ double wacc[4 * PT_COUNT];
  const int chain = blockIdx.y;
  const Cmd& c = a.cmds[chain];
  if (c.type == CMD_DONE || c.type == CMD_FLUSH) return;
  const Dims& d = a.d;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const VecRef v{a.vecs + (long)chain * V_COUNT * a.Dpad, a.Dpad};
  const double* sums = a.sums + (long)chain * NS * d.G;
  const int g = blockIdx.x * 256 + tid;
  const bool any_generic = !d.x0_is_one || (d.C >= 2 && d.K > 0);
  GeneCtx<CM> x;
  gene_load<CM>(d, c, v, g, x);
  GeneSumsV<CM> acc;
  acc.lik = acc.dph = acc.Sr = 0.0;
#pragma unroll
  for (int cc = 0; cc < CM; ++cc) acc.Tx[cc] = 0.0;
  if (x.active) {
    const long G = d.G;
    acc.lik = sums[0 * G + g]; acc.dph = sums[1 * G + g]; acc.Sr = sums[2 * G + g];
    if (any_generic) {
#pragma unroll
      for (int cc = 0; cc < CM; ++cc) if (cc < d.C) acc.Tx[cc] = sums[(3 + cc) * G + g];
    }
  }
  // the parked subtrees of the first levels this leaf closes are requested now: one round trip for all of them, behind the
  // gene's arithmetic, instead of one per level after it (a chain closing three levels kept the launch 4 us longer)
  constexpr int kPreLev = 3;
  double pre[kPreLev][NCM][3];
  const int n_pre = c.type == CMD_LEAF ? (c.n_merge < kPreLev ? c.n_merge : kPreLev) : 0;
#pragma unroll
  for (int lev = 0; lev < kPreLev; ++lev) {
#pragma unroll
    for (int j = 0; j < NCM; ++j) {
      pre[lev][j][0] = pre[lev][j][1] = pre[lev][j][2] = 0.0;
      if (lev < n_pre && j < x.ncoord) {
        coord_load_slot(v, x.idx[j], lev, &pre[lev][j][0], &pre[lev][j][1], &pre[lev][j][2]);
      }
    }
  }
  double pn[NCM], minv[NCM], part[10];
  gene_finish<CM>(d, c, v, x, acc, a.Sy, a.SyE, a.SyX, a.SX, a.ncell, a.Lg1, part, pn, minv);
  block_accumulate<10>(part, wacc, wave, lane);
  if (c.type == CMD_LEAF) {
    NodeVals nv[NCM];
#pragma unroll
    for (int j = 0; j < NCM; ++j) nv[j] = NodeVals{pn[j], pn[j]};
#pragma unroll
    for (int lev = 0; lev < kPreLev; ++lev) {
      if (lev < n_pre) {
        double dots[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_merge_dots_vals(pre[lev][j][0], pre[lev][j][1], pre[lev][j][2], pn[j], minv[j], &nv[j], dots);
        block_accumulate<6>(dots, wacc + PT_DOTS + 6 * lev, wave, lane);
      }
    }
    for (int lev = kPreLev; lev < c.n_merge; ++lev) {
      double dots[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_merge_dots(v, x.idx[j], lev, pn[j], minv[j], &nv[j], dots);
      block_accumulate<6>(dots, wacc + PT_DOTS + 6 * lev, wave, lane);
    }
    if (!c.subtree_complete) {
#pragma unroll
      for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_store_slot(v, x.idx[j], c.n_merge, pn[j], nv[j]);
    } else {
      double top[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_top_dots(v, x.idx[j], c.dir, pn[j], minv[j], nv[j], top);
      block_accumulate<6>(top, wacc + PT_TOP, wave, lane);
    }
  }
  __syncthreads();
  const int np = parts_used(c);
  double* slab = a.partials + ((long)chain * gridDim.x + blockIdx.x) * PT_COUNT;
  for (int k = tid; k < np; k += 256) {
    const bool used = k < 10 || (k >= PT_DOTS && k < PT_DOTS + 6 * c.n_merge) || (k >= PT_TOP && c.subtree_complete);
    slab[k] = used ? ((wacc[k] + wacc[PT_COUNT + k]) + wacc[2 * PT_COUNT + k]) + wacc[3 * PT_COUNT + k] : 0.0;
  }
}

// -----------------------------------------------------------------------------------------------------
// step kernel: one workgroup per chain.
//  phase REDUCE: fold the close kernel's slab (and the T0 slab of the previous update launch) into PT_COUNT sums in
//                a fixed order. For gene shards these per-shard sums are then added across shards (RCCL all-reduce
//                between processes, ppcx_sum_shards_kernel inside one process) before phase STEP runs.
//  phase STEP  : eight lanes of wavefront 0 run the NUTS / adaptation state machine (chain_step): lane k < 6 owns
//                hyper-parameter k, the scalar logic runs redundantly in registers, run-time-indexed arrays in LDS.
// -----------------------------------------------------------------------------------------------------
struct WaveLanes {                              // cooperating lanes 0..7 of one wavefront
  static constexpr int kPerLane = 1;
  int lane;
  __device__ __forceinline__ int k_begin() const { return lane; }
  __device__ __forceinline__ int k_end() const { return lane < 6 ? lane + 1 : lane; }
  __device__ __forceinline__ bool leader() const { return lane == 0; }
  __device__ __forceinline__ double sum(double v) const {      // all-reduce over the 8 lanes by DPP moves (no LDS round trips)
    v += dpp_take<0xB1, 0xf>(v);               // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_take<0x4E, 0xf>(v);               // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_take<0x141, 0xf>(v);              // row_half_mirror: lane i <-> lane 7 - i, the other quad's total
    return v;
  }
  __device__ __forceinline__ double pick(const double* own, int k) const { return __shfl(own[0], k, 8); }
};

__global__ __launch_bounds__(256) void ppcx_step_kernel(StepArgs a) {
  __shared__ double sm[3][8][32];
  __shared__ double sT0[256];
  __shared__ double red[PT_COUNT];
  __shared__ double hv[V_COUNT * 8];           // the six hyper coordinates of every per-coordinate vector
  __shared__ Cmd s_ex;
  __shared__ ChainState s_st;
  __shared__ Reduced s_rd;
  constexpr int NST = (int)(sizeof(ChainState) / sizeof(int)), NCMD = (int)(sizeof(Cmd) / sizeof(int)), NHV = V_COUNT * 8;
  static_assert(NST <= 4 * 256 && NCMD <= 256 && NHV <= 3 * 256 && PT_COUNT <= 96, "step kernel staging sizes");
  __shared__ Cmd s_nc;
  const int chain = blockIdx.y, tid = threadIdx.x;
  const bool lead = blockIdx.x == 0;           // with a.upd_vecs the grid has several workgroups per chain: all of them run the
                                               // step on the same inputs, the first one writes what the step leaves in memory
  const ChainState* st_in = a.states_in + chain;
  const bool done = st_in->sc.phase == PH_DONE;
  double* rg = a.red + (long)chain * PT_COUNT;
  // the chain's command, state and hyper vectors are requested now and parked in registers, so that their round trip
  // overlaps the reduction below
  int r_st[4], r_cmd = 0; double r_hv[3];
  {
    const word_t* s2 = reinterpret_cast<const word_t*>(st_in);
#pragma unroll
    for (int k = 0; k < 4; ++k) r_st[k] = tid + 256 * k < NST ? s2[tid + 256 * k] : 0;
    if (tid < NCMD) r_cmd = reinterpret_cast<const word_t*>(a.cmds_in + chain)[tid];
    const double* hvg = a.hyper_in + (long)chain * NHV;
#pragma unroll
    for (int k = 0; k < 3; ++k) r_hv[k] = tid + 256 * k < NHV ? hvg[tid + 256 * k] : 0.0;
  }
  if (a.phases & STEP_REDUCE) {
    const double* slab = a.partials + (long)chain * a.slab_stride * PT_COUNT;
    // one pass: thread (c, ch) sums rows ch, ch+8, ... of columns c, c+32, c+64, loads of several rows in flight; then
    // column v = sum over the eight row groups in a fixed order. All columns are loaded (stale ones included) so that
    // these loads do not wait for the command that says which sums it produced; the selection happens afterwards.
    const int c = tid & 31, ch = tid >> 5;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    {
      const bool in2 = c + 64 < PT_COUNT;
#pragma unroll 4
      for (int b = ch; b < a.nblocks_close; b += 8) {
        const double* row = slab + (long)b * PT_COUNT;
        const double v0 = row[c], v1 = row[c + 32], v2 = in2 ? row[c + 64] : 0.0;
        s0 += v0; s1 += v1; s2 += v2;
      }
    }
    const Cmd& exg = a.cmds_in[chain];
    const int np = (done || exg.type == CMD_DONE || exg.type == CMD_FLUSH) ? 0 : parts_used(exg);   // uniform
    sm[0][ch][c] = c < np ? s0 : 0.0; sm[1][ch][c] = c + 32 < np ? s1 : 0.0; sm[2][ch][c] = c + 64 < np ? s2 : 0.0;
    __syncthreads();
    if (tid < PT_COUNT) {
      double t = 0.0;
      if (tid < np && tid != PT_T0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sm[tid >> 5][k][tid & 31];
      }
      red[tid] = t;
    }
    // kinetic energy of freshly drawn momenta: only commands that drew momenta left something in the T0 slab
    const bool fresh = !done && (exg.pre_flags & (PRE_NEW_TRANSITION | PRE_EPS_TRY)) != 0;
    if (fresh) {
      const double* t0s = a.t0 + (long)chain * a.nblocks_update;
      double s = 0.0;
      for (int b = tid; b < a.nblocks_update; b += 256) s += t0s[b];
      sT0[tid] = s;
      __syncthreads();
      for (int stp = 128; stp > 0; stp >>= 1) { if (tid < stp) sT0[tid] += sT0[tid + stp]; __syncthreads(); }
      if (tid == 0) red[PT_T0] = sT0[0];
    }
    __syncthreads();
    if (!(a.phases & STEP_ADVANCE)) { if (lead) for (int i = tid; i < PT_COUNT; i += 256) rg[i] = red[i]; return; }
  } else {
    for (int i = tid; i < PT_COUNT; i += 256) red[i] = rg[i];     // sums completed by the shard exchange
    __syncthreads();
  }
  // ---- phase STEP
  if (done) {                                  // finished chain: carry its final state across the double buffer
    if (!lead) return;
    if (tid == 0) { a.states_out[chain] = *st_in; a.cmds_out[chain] = a.cmds_in[chain]; }
    const double* hi = a.hyper_in + (long)chain * V_COUNT * 8;
    double* ho = a.hyper_out + (long)chain * V_COUNT * 8;
    for (int i = tid; i < V_COUNT * 8; i += 256) ho[i] = hi[i];
    return;
  }
  {
    word_t* d2 = reinterpret_cast<word_t*>(&s_st);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (tid + 256 * k < NST) d2[tid + 256 * k] = r_st[k];
    if (tid < NCMD) reinterpret_cast<word_t*>(&s_ex)[tid] = r_cmd;
#pragma unroll
    for (int k = 0; k < 3; ++k) if (tid + 256 * k < NHV) hv[tid + 256 * k] = r_hv[k];
  }
  __syncthreads();
  const bool have_parts = s_st.sc.phase != PH_START;
  Cmd* nc_out = a.cmds_out + chain;
  if (tid < 8) {
    ChainScalars st = s_st.sc;                 // scalars in registers; the run-time-indexed arrays stay in LDS
    ChainIO io;
    io.draws = (lead && a.draws) ? a.draws + (long)chain * a.draws_chain_stride : nullptr;
    io.out.lp = (lead && a.out_lp) ? a.out_lp + (long)chain * a.n_keep : nullptr;
    io.out.stepsize = (lead && a.out_stepsize) ? a.out_stepsize + (long)chain * a.iter : nullptr;
    io.out.treedepth = (lead && a.out_treedepth) ? a.out_treedepth + (long)chain * a.iter : nullptr;
    io.out.n_leapfrog = (lead && a.out_n_leapfrog) ? a.out_n_leapfrog + (long)chain * a.iter : nullptr;
    io.out.divergent = (lead && a.out_divergent) ? a.out_divergent + (long)chain * a.iter : nullptr;
    io.out.accept = (lead && a.out_accept) ? a.out_accept + (long)chain * a.iter : nullptr;
    Cmd nc;
    chain_step(WaveLanes{tid}, a.d, st, s_st.ta, s_ex, red, have_parts, VecRef{hv, 8}, io, s_rd, nc);
    if (tid == 0) {
      s_st.sc = st;
      s_nc = nc;
      if (lead) {
        *nc_out = nc;
        if (st.phase == PH_DONE) a.done[chain] = 1 + st.error;
      }
    }
  }
  __syncthreads();
  if (lead) {
    double* hvo = a.hyper_out + (long)chain * V_COUNT * 8;
    for (int i = tid; i < V_COUNT * 8; i += 256) hvo[i] = hv[i];
    const word_t* s2 = reinterpret_cast<const word_t*>(&s_st); word_t* d2 = reinterpret_cast<word_t*>(a.states_out + chain);
    for (int i = tid; i < (int)(sizeof(ChainState) / sizeof(int)); i += 256) d2[i] = s2[i];
  }
  if (!a.upd_vecs) return;
  // ---- the per-coordinate work of the command just decided (what ppcx_update_kernel does after a separate step launch)
  const Dims& d = a.d;
  double T0 = 0.0;
  if (s_nc.type != CMD_DONE) {
    const VecRef v{a.upd_vecs + (long)chain * V_COUNT * a.upd_Dpad, a.upd_Dpad};
    double* draws = a.draws ? a.draws + (long)chain * a.draws_chain_stride : nullptr;
    for (int i = 3 + blockIdx.x * 256 + tid; i < d.off_tail; i += gridDim.x * 256) coord_update(d, s_nc, v, i, draws, &T0);
  }
  if ((s_nc.pre_flags & (PRE_NEW_TRANSITION | PRE_EPS_TRY)) == 0) return;
  __syncthreads();                             // sT0 was used by the reduction above
  sT0[tid] = T0;
  __syncthreads();
  for (int stp = 128; stp > 0; stp >>= 1) { if (tid < stp) sT0[tid] += sT0[tid + stp]; __syncthreads(); }
  if (tid == 0) a.upd_t0_out[(long)chain * gridDim.x + blockIdx.x] = sT0[0];
}



// -----------------------------------------------------------------------------------------------------
// Pipelined rounds: two launches per leapfrog instead of three, and the state machine off the critical path.
//   ppcx_ls_kernel   : ONE launch holds the log-likelihood workgroups of round r AND, in its first workgroups (one per
//                      chain), the state machine that digests round r - 1: it reduces the gene kernel's slab, advances
//                      the chain and writes the next command while the count matrix is being streamed. This works
//                      because the log-likelihood part reads only per-gene constants, and the gene kernel has already
//                      written those for the position the next leaf will most likely evaluate (same subtree direction,
//                      same step). When the state machine decides otherwise (a new transition, the tree turning round,
//                      the step-size search) the evaluation is void: the gene kernel then only applies the command and
//                      leaves its true constants, the next launch evaluates them, and the chain has lost one round.
//   ppcx_gene_kernel : everything that belongs to one gene, one thread per gene: the per-coordinate work of the command
//                      (what the step kernel's coordinate part does in the three-launch round), the close of the leaf,
//                      the constants of the anticipated next position.
// The three-launch round stays for gene shards (their sums cross processes between reduce and advance), ADVI,
// single evaluations and models with a per-cell linear predictor (whose cells read the positions themselves).
// -----------------------------------------------------------------------------------------------------
struct StepShared {
  double sm[3][8][32];
  double red[PT_COUNT];
  double hv[V_COUNT * 8];
  Cmd ex, nc;
  ChainState st;
  Reduced rd;
  int x_ok[kMaxRanks]; long long x_wait;       // direct exchange: has rank k's contribution arrived; ticks waited
};
// system-scope accesses to the peer-mapped exchange buffers (uncached memory: nothing may be served from a cache line that a
// peer has rewritten since)
__device__ __forceinline__ void sys_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned long long sys_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// Adds the ranks' partial sums in red[0 .. PT_COUNT) (LDS), in rank order on every rank -- identical bits everywhere, so the
// replicated state machines take identical decisions. Called by all 256 threads of a chain's state-machine workgroup, red
// complete and visible. Two slots by the parity of the sequence number: a rank can be one exchange ahead of a peer (it has
// published exchange n + 1 while the peer still reads exchange n), never two, because exchange n + 2 needs the peer's n + 1.
// Returns false when a peer has aborted or does not arrive within the timeout.
__device__ __forceinline__ bool xchg_sums(const XchgArgs& x, int chain, unsigned count, StepShared& s) {
  const int tid = threadIdx.x;
  const unsigned long long seq = ((unsigned long long)x.epoch << 32) | (unsigned long long)count;
  const int slot = (int)(count & 1u);
  if (tid < PT_COUNT) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(s.red[tid]);
    for (int k = 0; k < x.nranks; ++k)
      sys_store(reinterpret_cast<unsigned long long*>(x.recv[k] + xchg_recv_index(x, slot, x.rank, chain)) + tid, bits);
  }
  __threadfence_system();                      // the sums are in the peers' memory before the sequence number is
  __syncthreads();
  if (tid < x.nranks) {
    __hip_atomic_store(x.flags[tid] + xchg_flag_index(x, slot, x.rank, chain), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long* mine = x.flags[x.rank];
    const long long t0 = (long long)wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(mine + xchg_flag_index(x, slot, tid, chain), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      bool gone = (long long)wall_clock64() - t0 > x.timeout_ticks;
      for (int k = 0; k < x.nranks && !gone; ++k) gone = (unsigned)sys_load(mine + xchg_abort_index(x, k)) == x.epoch;
      if (gone) { ok = 0; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    s.x_ok[tid] = ok;
    if (tid == 0) s.x_wait = (long long)wall_clock64() - t0;
  }
  __syncthreads();
  bool ok = true;
  for (int k = 0; k < x.nranks; ++k) ok = ok && s.x_ok[k] != 0;
  if (ok && tid < PT_COUNT) {
    double t = 0.0;
    for (int k = 0; k < x.nranks; ++k)
      t += __longlong_as_double((long long)sys_load(reinterpret_cast<const unsigned long long*>(x.recv[x.rank] + xchg_recv_index(x, slot, k, chain)) + tid));
    s.red[tid] = t;
  }
  __syncthreads();
  return ok;
}
__device__ __forceinline__ void step_role_pipelined(const StepArgs& a, int chain, StepShared& s, bool spec) {
  constexpr int NST = (int)(sizeof(ChainState) / sizeof(int)), NCMD = (int)(sizeof(Cmd) / sizeof(int)), NHV = V_COUNT * 8;
  static_assert(NST <= 4 * 256 && NCMD <= 256 && NHV <= 3 * 256 && PT_COUNT <= 96, "step role staging sizes");
  const int tid = threadIdx.x;
#ifdef PPCX_TESTING
  long long tr_t[7] = {0, 0, 0, 0, 0, 0, 0};
#define PPCX_SM_STAMP(k) do { tr_t[k] = (long long)wall_clock64(); } while (0)
#else
#define PPCX_SM_STAMP(k) ((void)0)
#endif
  PPCX_SM_STAMP(0);
  const ChainState* st_in = a.states_in + chain;
  const bool done = st_in->sc.phase == PH_DONE;
  int r_st[4], r_cmd = 0; double r_hv[3];
  {
    const word_t* s2 = reinterpret_cast<const word_t*>(st_in);
#pragma unroll
    for (int k = 0; k < 4; ++k) r_st[k] = tid + 256 * k < NST ? s2[tid + 256 * k] : 0;
    if (tid < NCMD) r_cmd = reinterpret_cast<const word_t*>(a.cmds_in + chain)[tid];
    const double* hvg = a.hyper_in + (long)chain * NHV;
#pragma unroll
    for (int k = 0; k < 3; ++k) r_hv[k] = tid + 256 * k < NHV ? hvg[tid + 256 * k] : 0.0;
  }
  {
    // the gene kernel's slab, every column (the kinetic energy of fresh momenta arrives in column PT_T0 here); a carried
    // round reads a stale slab and ignores the sums
    const double* slab = a.partials + (long)chain * a.slab_stride * PT_COUNT;
    const int c = tid & 31, ch = tid >> 5;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const bool in2 = c + 64 < PT_COUNT;
#pragma unroll 4
    for (int b = ch; b < a.nblocks_close; b += 8) {
      const double* row = slab + (long)b * PT_COUNT;
      const double v0 = row[c], v1 = row[c + 32], v2 = in2 ? row[c + 64] : 0.0;
      s0 += v0; s1 += v1; s2 += v2;
    }
    const Cmd& exg = a.cmds_in[chain];
    const int np = (done || exg.type == CMD_DONE || exg.type == CMD_FLUSH) ? 0 : parts_used(exg);   // uniform
    s.sm[0][ch][c] = c < np ? s0 : 0.0; s.sm[1][ch][c] = c + 32 < np ? s1 : 0.0; s.sm[2][ch][c] = c + 64 < np ? s2 : 0.0;
    PPCX_SM_STAMP(1);                          // the state, the command, the hyper vectors and the slab have arrived
    __syncthreads();
    if (tid < PT_COUNT) {
      double t = 0.0;
      if (tid < np) {
#pragma unroll
        for (int k = 0; k < 8; ++k) t += s.sm[tid >> 5][k][tid & 31];
      }
      s.red[tid] = t;
    }
  }
  if (done) {                                  // finished chain: carry its final state across the double buffer
    if (tid == 0) { a.states_out[chain] = *st_in; a.cmds_out[chain] = a.cmds_in[chain]; }
    const double* hi = a.hyper_in + (long)chain * NHV;
    double* ho = a.hyper_out + (long)chain * NHV;
    for (int i = tid; i < NHV; i += 256) ho[i] = hi[i];
    return;
  }
  {
    word_t* d2 = reinterpret_cast<word_t*>(&s.st);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (tid + 256 * k < NST) d2[tid + 256 * k] = r_st[k];
    if (tid < NCMD) reinterpret_cast<word_t*>(&s.ex)[tid] = r_cmd;
#pragma unroll
    for (int k = 0; k < 3; ++k) if (tid + 256 * k < NHV) s.hv[tid + 256 * k] = r_hv[k];
  }
  __syncthreads();
  PPCX_SM_STAMP(2);                            // sums folded, everything staged in LDS
  // gene shards, one per rank: the other ranks' sums. Every rank's copy of this chain reaches this point the same number of times.
  bool x_ok = true;
  const bool x_on = a.x.nranks > 1;
  if (x_on) x_ok = xchg_sums(a.x, chain, s.st.xc.count + 1u, s);
  if (tid < 8) {
    // The chain's scalars and the new command stay where they are staged, in LDS (round 5; before, lane 0's register copy was
    // written back after the step). Under the merged launch's 128-register bound that copy lived in scratch: copying it in and
    // out and reloading spilled scalars cost 3-5 of the state machine's 12-16 us per round, which bounds the launches of one and
    // two chains (profiles/r05_sm_phase_trace.txt). The eight lanes run the scalar logic redundantly on identical values, so
    // their stores to the shared copy agree -- as they always have for the tree arrays beside it. Same arithmetic: with
    // -ffp-contract=on the chains are bit for bit those of the register copy; with the default contraction the backend fuses
    // multiply-adds across statements differently when an operand passes through LDS, and a chain's rounding changes like
    // under another seed.
    ChainScalars& st = s.st.sc;
    if (x_on && tid == 0) { s.st.xc.count += 1u; s.st.xc.ticks += s.x_wait; }
    if (!x_ok) {                               // a peer has left the fit or did not arrive: this chain ends with an error
      if (tid == 0) {
        st.error = 4; st.phase = PH_DONE;
        s.nc = s.ex; s.nc.type = CMD_DONE;
        a.done[chain] = 1 + st.error;
      }
    } else {
    ChainIO io;
    io.draws = a.draws ? a.draws + (long)chain * a.draws_chain_stride : nullptr;
    io.out.lp = a.out_lp ? a.out_lp + (long)chain * a.n_keep : nullptr;
    io.out.stepsize = a.out_stepsize ? a.out_stepsize + (long)chain * a.iter : nullptr;
    io.out.treedepth = a.out_treedepth ? a.out_treedepth + (long)chain * a.iter : nullptr;
    io.out.n_leapfrog = a.out_n_leapfrog ? a.out_n_leapfrog + (long)chain * a.iter : nullptr;
    io.out.divergent = a.out_divergent ? a.out_divergent + (long)chain * a.iter : nullptr;
    io.out.accept = a.out_accept ? a.out_accept + (long)chain * a.iter : nullptr;
    Cmd& nc = s.nc;
    PPCX_SM_STAMP(3);
    (void)chain_step_pipelined(WaveLanes{tid}, a.d, st, s.st.ta, s.ex, s.red, VecRef{s.hv, 8}, io, s.rd, nc, spec);
    PPCX_SM_STAMP(4);
    if (tid == 0 && st.phase == PH_DONE) a.done[chain] = 1 + st.error;
    }
  }
#ifdef PPCX_TESTING
  if (tid == 0) {                              // phases: loads + slab, fold + staging, (exchange), state machine, and the write-out of the previous round's
    PPCX_SM_STAMP(5);
    SmTrace& tr = s.st.tr;
    tr.t[0] += tr_t[1] - tr_t[0]; tr.t[1] += tr_t[2] - tr_t[1]; tr.t[2] += tr_t[3] - tr_t[2]; tr.t[3] += tr_t[4] - tr_t[3]; tr.t[4] += tr_t[5] - tr_t[4];
    tr.n += 1;
  }
#endif
  __syncthreads();
  double* hvo = a.hyper_out + (long)chain * NHV;
  for (int i = tid; i < NHV; i += 256) hvo[i] = s.hv[i];
  const word_t* s2 = reinterpret_cast<const word_t*>(&s.st); word_t* d2 = reinterpret_cast<word_t*>(a.states_out + chain);
  for (int i = tid; i < NST; i += 256) d2[i] = s2[i];
  const word_t* c2 = reinterpret_cast<const word_t*>(&s.nc); word_t* e2 = reinterpret_cast<word_t*>(a.cmds_out + chain);
  if (tid < NCMD) e2[tid] = c2[tid];
}

// Grid: runs of 8 x (chains of the launch) workgroups, as in ppcx_loglik_kernel: position r & 7 of a run is a range block,
// r >> 3 the chain's column, and the workgroups with equal r & 7 land on one XCD. In the first n_srun runs position 7 is not
// a range block but a state machine (of chain run * columns + column, if there is such a chain): state machines take the
// slot of a log-likelihood workgroup each, all on one XCD, and every workgroup of the launch is resident from its start.
template <int CM, int GEN>
__global__ __launch_bounds__(256, GEN ? PPCX_LOGLIK_OCC : PPCX_LOGLIK_OCC_FAST) void ppcx_ls_kernel(LoglikArgs a, StepArgs sa, int n_srun, int n_chains_total, int spec) {
  extern __shared__ double lds[];
  const int nch = a.nchains;
  const int lin = blockIdx.x, run = lin / (8 * nch), r = lin - run * (8 * nch);
  const int pos = r & 7, col = r >> 3;
  if (run < n_srun && pos == 7) {
    const int sc = run * nch + col;
    if (sc >= n_chains_total) return;
    __builtin_amdgcn_s_setprio(3);             // latency-bound and short: ahead of the log-likelihood wavefronts of its SIMDs
    step_role_pipelined(sa, sc, *reinterpret_cast<StepShared*>(lds), spec != 0);
    return;
  }
  const int jb = run < n_srun ? run * 7 + pos : n_srun * 7 + (run - n_srun) * 8 + pos;
  loglik_role<CM, GEN, true>(a, jb, col, lds);
}

// One gene's part of a pipelined round, for the lane that owns gene g (g >= G: a lane without a gene, which only takes part
// in the reductions): the command's work on the gene's coordinates, the close of the evaluated position, the anticipated
// constants. No workgroup barrier inside; the wavefront's partial sums go to wacc[wave][...]. Everything the lane reads is
// requested in one burst at the start: nothing else hides this kernel's latency.
// (Kept apart from the kernel's barriers because a fused one-launch round was built on it in round 3 -- the wavefront that
// swept a range of genes closed them itself after its chain's state machine, a workgroup of the same launch, had published the
// command through a flag -- and measured: 90 us per launch against 63 + 20 for the two launches)
template <int CM>
__device__ __forceinline__ void gene_wave_part(const GeneArgs& ga, const Cmd& c, int chain, int g,
                                               double* wacc, double* rows, int wave, int lane, bool do_update, bool do_close) {
  constexpr int NCM = CM + 1;
  constexpr int NS = GeneSums<CM>::N;
  const CloseArgs& a = ga.c;
  const Dims& d = a.d;
  const VecRef v{a.vecs + (long)chain * V_COUNT * a.Dpad, a.Dpad};
  const double* sums = a.sums + (long)chain * NS * d.G;
  const bool any_generic = !d.x0_is_one || (d.C >= 2 && d.K > 0);
  GeneCtx<CM> x;
  gene_index<CM>(d, g, x);
  // ---- a command with rare pre-operations: those first, through memory (gene_rare_pre)
  double T0 = 0.0;
  double* draws = ga.draws ? ga.draws + (long)chain * ga.draws_chain_stride : nullptr;
  const int fmask = do_update ? gene_rare_pre<CM>(d, c, v, x, draws, &T0) : ~0;
  // ---- everything this lane reads is requested here, in one round trip, before anything is stored
  CoordCache cache[NCM];
  double p_cur[NCM], minv[NCM];
#pragma unroll
  for (int j = 0; j < NCM; ++j) {
    p_cur[j] = 0.0; minv[j] = 1.0; x.q[j] = 0.0;
    if (j < x.ncoord) {
      if (do_update) cache[j] = coord_prefetch_for(c, v, x.idx[j]);
      else { x.q[j] = v.at(V_Q0 + 3 * c.dir, x.idx[j]); p_cur[j] = v.at(V_P0 + 3 * c.dir, x.idx[j]); minv[j] = v.at(V_MINV, x.idx[j]); }
    }
  }
  GeneSumsV<CM> acc;
  acc.lik = acc.dph = acc.Sr = 0.0;
#pragma unroll
  for (int cc = 0; cc < CM; ++cc) acc.Tx[cc] = 0.0;
  GeneData gd;
  double phi = 1.0;
  constexpr int kPreLev = 2;                   // tree levels whose slots travel with the burst of loads (a third: 18 registers; see the kernel)
  double pre[kPreLev][NCM][3];
  const int n_pre = (do_close && c.type == CMD_LEAF) ? (c.n_merge < kPreLev ? c.n_merge : kPreLev) : 0;
  if (do_close) {
    if (x.active) {
      const long G = d.G;
      acc.lik = sums[0 * G + g]; acc.dph = sums[1 * G + g]; acc.Sr = sums[2 * G + g];
      if (any_generic) {
#pragma unroll
        for (int cc = 0; cc < CM; ++cc) if (cc < d.C) acc.Tx[cc] = sums[(3 + cc) * G + g];
      }
      // phi of the position being closed: written with the constants the log-likelihood part evaluated (for a leaf
      // anticipated by the previous round the update below does not touch them)
      phi = v.at(V_C0, x.idx[1]);
    }
    gene_data_load<CM>(d, x.gg, a.Sy, a.SyE, a.SyX, a.SX, a.ncell, a.Lg1, gd);
#pragma unroll
    for (int lev = 0; lev < kPreLev; ++lev) {
#pragma unroll
      for (int j = 0; j < NCM; ++j) {
        pre[lev][j][0] = pre[lev][j][1] = pre[lev][j][2] = 0.0;
        if (lev < n_pre && j < x.ncoord) {
          coord_load_slot(v, x.idx[j], lev, &pre[lev][j][0], &pre[lev][j][1], &pre[lev][j][2]);
        }
      }
    }
  }
  // ---- the command's work on the gene's coordinates
  if (do_update) gene_coord_update<CM, true>(d, c, v, x, draws, &T0, !do_close, cache, p_cur, minv, !do_close, fmask);
  if (!do_close) {                             // the command's position has not been evaluated yet: nothing to close
    double t0v[1] = {T0};
    wave_sums_lds<1>(t0v, rows, wacc + wave * PT_COUNT, lane);
    return;
  }
  // ---- close the evaluated position
  x.gp.coef[0] = x.q[0];
#pragma unroll
  for (int cc = 1; cc < CM; ++cc) x.gp.coef[cc] = (x.has_slopes && cc < d.C) ? x.q[cc + 1] : 0.0;
  x.gp.sigma_raw = x.q[1];
  x.gp.phi = phi; x.gp.invphi = 0.0;
  double pn[NCM], gn[NCM], part[16];
  gene_finish_vals<CM>(d, c, v, x, acc, gd, p_cur, minv, part, pn, gn);
  part[PT_T0] = T0;
  double* wrow = wacc + wave * PT_COUNT;
  if (c.type != CMD_LEAF) { wave_sums_lds<10>(part, rows, wrow, lane); return; }
  {
    NodeVals nv[NCM];
#pragma unroll
    for (int j = 0; j < NCM; ++j) nv[j] = NodeVals{pn[j], pn[j]};
    // the ten sums of the leaf travel with the six U-turn products of the first level it closes (one batch of 16), the
    // further prefetched levels in a batch of their own
#pragma unroll
    for (int k = 0; k < 6; ++k) part[10 + k] = 0.0;
    if (n_pre > 0) {
#pragma unroll
      for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_merge_dots_vals(pre[0][j][0], pre[0][j][1], pre[0][j][2], pn[j], minv[j], &nv[j], part + 10);
    }
    wave_sums_lds<16>(part, rows, wrow, lane);   // wrow[0 .. 9] the leaf, wrow[PT_DOTS .. PT_DOTS + 5] level 0 (PT_DOTS = 10)
    static_assert(PT_DOTS == 10, "the first level's products follow the leaf's ten sums");
    if constexpr (kPreLev > 1) if (n_pre > 1) {
      constexpr int NB = 6 * (kPreLev - 1);
      double more[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) more[k] = 0.0;
#pragma unroll
      for (int lev = 1; lev < kPreLev; ++lev) {
        if (lev < n_pre) {
#pragma unroll
          for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_merge_dots_vals(pre[lev][j][0], pre[lev][j][1], pre[lev][j][2], pn[j], minv[j], &nv[j], more + 6 * (lev - 1));
        }
      }
      wave_sums_lds<NB>(more, rows, wrow + PT_DOTS + 6, lane);
    }
    for (int lev = kPreLev; lev < c.n_merge; ++lev) {
      double dots[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_merge_dots(v, x.idx[j], lev, pn[j], minv[j], &nv[j], dots);
      wave_sums_lds<6>(dots, rows, wrow + PT_DOTS + 6 * lev, lane);
    }
    if (!c.subtree_complete) {
#pragma unroll
      for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_store_slot(v, x.idx[j], c.n_merge, pn[j], nv[j]);
    } else {
      double top[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_top_dots(v, x.idx[j], c.dir, pn[j], minv[j], nv[j], top);
      wave_sums_lds<6>(top, rows, wrow + PT_TOP, lane);
    }
    // ahead of the state machine: the constants of the position the next leaf evaluates if the tree goes on
    if (ga.spec) gene_spec_consts<CM>(d, c, v, x, pn, gn, minv);
  }
}
// after a workgroup barrier: the four wavefronts' partial sums -> the workgroup's row of the slab
__device__ __forceinline__ void gene_block_finish(const Cmd& c, bool do_update, bool do_close, const double* wacc, double* slab, int tid) {
  if (!do_close) {
    if (tid == 0) slab[PT_T0] = ((wacc[0] + wacc[PT_COUNT]) + wacc[2 * PT_COUNT]) + wacc[3 * PT_COUNT];
    return;
  }
  const int np = parts_used(c);
  for (int k = tid; k < np; k += 256) {
    if (k == PT_T0 && !do_update) continue;    // left by the round that applied the command
    const bool used = k < 10 || (k >= PT_DOTS && k < PT_DOTS + 6 * c.n_merge) || (k >= PT_TOP && c.subtree_complete);
    slab[k] = used ? ((wacc[k] + wacc[PT_COUNT + k]) + wacc[2 * PT_COUNT + k]) + wacc[3 * PT_COUNT + k] : 0.0;
  }
}

// Register budget of the two-column instantiation: 128 vector registers, four wavefronts per SIMD -- what a log-likelihood
// wavefront holds. In a fit with chain groups this kernel starts while another group's log-likelihood wavefronts, four to a SIMD,
// own every register of the chip: a wavefront of 128 registers moves in as soon as ONE of them retires, a wavefront of 146 (152
// allocated: what the body took with three prefetched tree levels) only after two on the same SIMD have, and this kernel is what
// its group's next log-likelihood launch waits for. cfg3, 8 chains in three groups (round 4, five seeds, alternating builds):
// 2.52 s per fit with 146 registers, 2.41 s with 128 and 36 of them spilled, 2.35 s with two prefetched levels instead of three
// (kPreLev: 128 registers without a spill; alone on the chip as fast as before, 2.94 s on one stream, 1.03 s a lone chain).
// (Four columns: 230 registers; 168 or 128 with 120 / 202 spilled made the factor design's fits 7 % / 14 % slower.)
template <int CM>
__global__ __launch_bounds__(256, CM <= 2 ? 4 : (CM <= 4 ? 2 : 1)) void ppcx_gene_kernel(GeneArgs ga) {
  __shared__ double wacc[4 * PT_COUNT];
  __shared__ double s_rows[4 * kRowsPerWave * kRowStride];      // wave_sums_lds: 8.3 KB per wavefront
  const CloseArgs& a = ga.c;
  const int chain = blockIdx.y;
  // the chain's command, copied into registers HERE: read through the reference its fields would be requested where they are
  // first used -- after the barrier below, a scalar round trip in front of the loads whose addresses depend on them
  const Cmd c = a.cmds[chain];
  if (c.type == CMD_DONE) return;
  const bool do_update = !c.updated, do_close = c.evaluated && c.type != CMD_FLUSH;      // uniform over the launch's chain
  if (!do_update && !do_close) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  gene_wave_part<CM>(ga, c, chain, blockIdx.x * 256 + tid, wacc, s_rows + wave * kRowsPerWave * kRowStride, wave, lane, do_update, do_close);
  __syncthreads();
  gene_block_finish(c, do_update, do_close, wacc, a.partials + ((long)chain * gridDim.x + blockIdx.x) * PT_COUNT, tid);
}

// in-process gene shards: every shard ends with the sum over shards (fixed order => identical bits everywhere)
__global__ void ppcx_sum_shards_kernel(ShardSumArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  double s = 0.0;
  for (int k = 0; k < a.n_shards; ++k) s += a.bufs[k][i];
  for (int k = 0; k < a.n_shards; ++k) a.bufs[k][i] = s;
}

// -----------------------------------------------------------------------------------------------------
// update kernel: apply the command the step kernel just issued to every gene-owned coordinate -- proposal / sample
// copies, draw storage, Welford / metric updates, momentum refresh (Philox per coordinate), first half kick and
// drift of the next leapfrog -- and leave the kinetic energy of fresh momenta as per-workgroup partial sums.
// -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ppcx_update_kernel(UpdateArgs a) {
  __shared__ double sT0[256];
  const int chain = blockIdx.y, tid = threadIdx.x;
  const Cmd& nc = a.cmds[chain];
  const Dims& d = a.d;
  double T0 = 0.0;
  if (nc.type != CMD_DONE) {
    const VecRef v{a.vecs + (long)chain * V_COUNT * a.Dpad, a.Dpad};
    double* draws = a.draws ? a.draws + (long)chain * a.draws_chain_stride : nullptr;
    for (int i = 3 + blockIdx.x * 256 + tid; i < d.off_tail; i += gridDim.x * 256) coord_update(d, nc, v, i, draws, &T0);
  }
  // kinetic energy of freshly drawn momenta: only commands that draw momenta leave something (the step kernel's reduce
  // phase reads the slab for exactly those commands)
  if ((nc.pre_flags & (PRE_NEW_TRANSITION | PRE_EPS_TRY)) == 0) return;
  sT0[tid] = T0;
  __syncthreads();
  for (int stp = 128; stp > 0; stp >>= 1) { if (tid < stp) sT0[tid] += sT0[tid + stp]; __syncthreads(); }
  if (tid == 0) a.t0_out[(long)chain * gridDim.x + blockIdx.x] = sT0[0];
}

// -----------------------------------------------------------------------------------------------------
// ADVI (mean-field; rstan::vb of R/utilities.R:246-278,1487-1494). The variational parameters live in spare
// vectors of slot 0: mu = V_SQ, omega = V_SG, running squared gradients = V_WM / V_WM2, initial point = V_Q0.
// One launch (a) applies the stochastic-gradient step that uses the gradient the loglik/close/reduce kernels
// just evaluated at the previous draw, and (b) writes the next Monte-Carlo draws zeta_c = mu + exp(omega) eta_c
// into the evaluation slots (V_Q1 of slot c, hyper-parameters into that slot's command).
// -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ double advi_eta(uint32_t rid, uint32_t draw, uint32_t k0) {
  return coord_normal(rid, draw, 6u, 0u, k0, 0x41445649u);
}
__device__ __forceinline__ void advi_coord(const AdviArgs& a, double* mu, double* om, double* hm, double* ho,
                                            double mu0, double g, uint32_t rid) {
  if (a.op == ADVI_RESET) { *mu = mu0; *om = 0.0; *hm = 0.0; *ho = 0.0; }
  if (a.op == ADVI_STEP) {                     // Stan advi::stochastic_gradient_ascent / adapt_eta
    const double eta = advi_eta(rid, a.prev_draw, a.k0);
    const double gm = g, go = g * eta * exp(*om) + 1.0;
    if (isfinite(gm) && isfinite(go)) {        // Stan raises on a non-finite gradient; a kernel cannot: skip the draw
      *hm = a.first_iter ? gm * gm : 0.1 * gm * gm + 0.9 * *hm;
      *ho = a.first_iter ? go * go : 0.1 * go * go + 0.9 * *ho;
      *mu += a.eta_scaled * gm / (1.0 + sqrt(*hm));
      *om += a.eta_scaled * go / (1.0 + sqrt(*ho));
    }
  }
}
__global__ __launch_bounds__(256) void ppcx_advi_kernel(AdviArgs a) {
  __shared__ double s_g6[6];
  __shared__ double s_om[256];
  const Dims& d = a.d;
  const int tid = threadIdx.x;
  const VecRef v0{a.vecs, a.Dpad};
  if (a.op == ADVI_STEP && tid == 0 && blockIdx.x == 0) {   // hyper-parameter gradient of the evaluated draw
    // (only workgroup 0 owns the hyper-parameters; it rewrites cmds[] below, after the barrier)
    const Cmd& c = a.cmds[0];
    const double* r = a.red;
    double g6[6];
    (void)hyper_close(d, c.hy, c.hyp_q, r[PT_LP], r + PT_H0, g6);
    for (int k = 0; k < 6; ++k) s_g6[k] = g6[k];
  }
  __syncthreads();
  double om_sum = 0.0;
  for (int i = 3 + blockIdx.x * 256 + tid; i < d.off_tail; i += gridDim.x * 256) {
    double mu = v0.at(V_SQ, i), om = v0.at(V_SG, i), hm = v0.at(V_WM, i), ho = v0.at(V_WM2, i);
    const uint32_t rid = (uint32_t)global_flat(d, i);
    const double g = a.op == ADVI_STEP ? v0.at(V_G1, i) : 0.0;
    advi_coord(a, &mu, &om, &hm, &ho, v0.at(V_Q0, i), g, rid);
    if (a.op != ADVI_DRAW) { v0.at(V_SQ, i) = mu; v0.at(V_SG, i) = om; v0.at(V_WM, i) = hm; v0.at(V_WM2, i) = ho; }
    om_sum += om;
    const double sd = exp(om);
    for (int c = 0; c < a.n_slots; ++c) {
      const double z = mu + sd * advi_eta(rid, a.draw_base + c, a.k0);
      if (a.out_draws) a.out_draws[(long)(a.out_row0 + c) * d.D + i] = z;
      else {
        a.vecs[((long)c * V_COUNT + V_Q1) * a.Dpad + i] = z;
        coord_consts(d, VecRef{a.vecs + (long)c * V_COUNT * a.Dpad, a.Dpad}, i, z);
      }
    }
  }
  if (blockIdx.x == 0 && tid < 6) {            // the six hyper-parameters (slot 0 of the hyper vectors)
    const int k = tid, col = hyper_index(d, k);
    double* h = a.hyper;
    double mu = h[V_SQ * 8 + k], om = h[V_SG * 8 + k], hm = h[V_WM * 8 + k], ho = h[V_WM2 * 8 + k];
    const uint32_t rid = (uint32_t)global_flat(d, col);
    advi_coord(a, &mu, &om, &hm, &ho, h[V_Q0 * 8 + k], a.op == ADVI_STEP ? s_g6[k] : 0.0, rid);
    if (a.op != ADVI_DRAW) { h[V_SQ * 8 + k] = mu; h[V_SG * 8 + k] = om; h[V_WM * 8 + k] = hm; h[V_WM2 * 8 + k] = ho; }
    om_sum += om;
    const double sd = exp(om);
    for (int c = 0; c < a.n_slots; ++c) {
      const double z = mu + sd * advi_eta(rid, a.draw_base + c, a.k0);
      if (a.out_draws) a.out_draws[(long)(a.out_row0 + c) * d.D + col] = z;
      else a.cmds[c].hyp_q[k] = z;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0 && !a.out_draws) {
    for (int c = 0; c < a.n_slots; ++c) {
      Cmd& cm = a.cmds[c];
      cm.type = CMD_EVAL; cm.dir = 1; cm.eps = 0.0; cm.pre_flags = 0; cm.n_merge = 0; cm.subtree_complete = 0; cm.leaf_n = 0;
      cm.hy = make_hyper(cm.hyp_q, d.lambda_mu_mu);
    }
  }
  s_om[tid] = om_sum;
  __syncthreads();
  for (int stp = 128; stp > 0; stp >>= 1) { if (tid < stp) s_om[tid] += s_om[tid + stp]; __syncthreads(); }
  if (tid == 0) a.omega_part[blockIdx.x] = s_om[0];
}

// ELBO accumulation: adds log p(zeta_c) of the evaluated slots (acc[0]), their count (acc[1]) and the number of
// non-finite evaluations dropped (acc[2]); acc[3] = sum of omega (entropy term) from the last advi launch.
__global__ void ppcx_advi_elbo_kernel(AdviElboArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int c = 0; c < a.n_slots; ++c) {
    const Cmd& cm = a.cmds[c];
    const double* r = a.red + (long)c * PT_COUNT;
    double g6[6];
    const double lp = hyper_close(a.d, cm.hy, cm.hyp_q, r[PT_LP], r + PT_H0, g6);
    if (isfinite(lp)) { a.acc[0] += lp; a.acc[1] += 1.0; } else a.acc[2] += 1.0;
  }
  double s = 0.0;
  for (int b = 0; b < a.n_omega_parts; ++b) s += a.omega_part[b];
  a.acc[3] = s;
}

// -----------------------------------------------------------------------------------------------------
// posterior-predictive draws + credible intervals, one workgroup per (gene <= K, sample) cell
// -----------------------------------------------------------------------------------------------------
constexpr int kPpcThreads = 512;              // 8 wavefronts share the cell's draws in LDS (2 per SIMD; the kernel needs 233 VGPRs)
constexpr int kPpcWaves = kPpcThreads / 64;
// number of the workgroup's draws that are <= v (every thread gets the total; two barriers)
__device__ __forceinline__ int block_count_le(const int* vals, int n, int v, int* s_cnt, int tid) {
  int c = 0;
  for (int j = tid; j < n; j += kPpcThreads) c += vals[j] <= v ? 1 : 0;
#pragma unroll
  for (int msk = 1; msk < 64; msk <<= 1) c += __shfl_xor(c, msk, 64);
  if ((tid & 63) == 0) s_cnt[tid >> 6] = c;
  __syncthreads();
  int tot = 0;
#pragma unroll
  for (int w = 0; w < kPpcWaves; ++w) tot += s_cnt[w];
  __syncthreads();
  return tot;
}
// order statistics r and r+1 (0-based, r+1 < n or equal to r when r = n-1) of the non-negative draws by bisection on the
// value: the smallest v with #{draws <= v} >= r+1. Exact for integers; ~log2(max) counting passes instead of a sort.
__device__ __forceinline__ void block_select_pair(const int* vals, int n, int r, int vmax, int* s_cnt, int tid, int* v_r, int* v_r1) {
  int lo = 0, hi = vmax;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (block_count_le(vals, n, mid, s_cnt, tid) >= r + 1) hi = mid; else lo = mid + 1;
  }
  *v_r = lo;
  // the next order statistic: the same value if it occurs again at rank r+1, else the smallest draw above it
  int nxt = lo;
  if (r + 1 < n && block_count_le(vals, n, lo, s_cnt, tid) < r + 2) {
    int mn = 2147483647;
    for (int j = tid; j < n; j += kPpcThreads) { const int x = vals[j]; mn = (x > lo && x < mn) ? x : mn; }
#pragma unroll
    for (int msk = 1; msk < 64; msk <<= 1) { const int o = __shfl_xor(mn, msk, 64); mn = o < mn ? o : mn; }
    if ((tid & 63) == 0) s_cnt[tid >> 6] = mn;
    __syncthreads();
    nxt = s_cnt[0];
#pragma unroll
    for (int w = 1; w < kPpcWaves; ++w) nxt = min(nxt, s_cnt[w]);
    __syncthreads();
  }
  *v_r1 = nxt;
}

// The predictive draws of one cell by `stride` cooperating lanes (lane `first` takes draws first, first + stride, ...):
// ONE loop of attempts for the wavefront. Per turn a lane makes one attempt at the gamma variate of its current draw (or begins
// the draw) and, once it has it, one attempt at the Poisson variate; a lane whose draw is complete stores it and begins its next
// draw in the next turn. A draw costs a lane ~1.15 turns (the two samplers' rejection rates) instead of every draw waiting for
// the slowest lane's rejections. The parameters come from the transposed table T[g][c][draw] (ppcx_ppc_table_kernel).
__device__ __forceinline__ void cell_draws(const PpcArgs& a, const double* T, int cell, int* vals, int first, int stride,
                                           double* sum_out, int* vmax_out) {
  const Dims& d = a.d;
  const int n = a.n_gen;
  const int g = cell / d.S, s = cell - g * d.S;
  const double* Tg = T + (long)g * (d.C + 1) * a.n_draws;
  const double expo = a.exposure[s];
  double sum = 0.0; int vmax = 0;
  int j = first, phase = 0;                     // 0: begin the draw, 1: gamma attempts, 2: Poisson attempts
  GammaDraw gd; PoissonDraw pq; double scale = 0.0;
  while (PPCX_WAVE_ANY(j < n)) {
    if (j < n) {
      int val = 0; bool done = false;
      if (phase == 0) {                         // the draw's parameters, eta, the gamma stream
        long src = j;
        if (a.resample) {                       // R/utilities.R:760: sample(draws, n, replace = TRUE)
          const double u = coord_uniform((uint32_t)j, (uint32_t)cell, 5u, 0u, a.k0, 0x50504331u);
          src = (long)(u * (double)a.n_draws); if (src >= a.n_draws) src = a.n_draws - 1;
        }
        double eta = expo + a.X[s] * Tg[src];
        for (int cc = 1; cc < d.C; ++cc) eta += a.X[(long)cc * d.S + s] * Tg[(long)cc * a.n_draws + src];
        const double phi = Tg[(long)d.C * a.n_draws + src];
        if (nb2_invalid(eta, phi)) { val = 2147483647; done = true; }        // invalid draw: sorts last
        else { gamma_begin(gd, phi, a.k0, (uint32_t)cell, (uint32_t)j); scale = rng_div(rng_exp(eta), phi); phase = 1; }
      }
      if (phase == 1) {
        double gam;
        if (gamma_attempt(gd, &gam)) {
          const double lam = gam * scale;
          if (!(lam < 1073741824.0)) { val = 1073741823; done = true; }      // Stan raises above 2^30; we saturate
          else { poisson_begin(pq, lam, a.k0, (uint32_t)cell, (uint32_t)j); phase = 2; }
        }
      }
      if (phase == 2 && !done) {
        long long k;
        if (poisson_attempt(pq, &k)) { val = k > 2147483647LL ? 2147483647 : (int)k; done = true; }
      }
      if (done) {
        vals[j] = val;
        sum += (double)val; vmax = val > vmax ? val : vmax;
        if (a.counts_rng) a.counts_rng[(long)j * a.n_cells + cell] = val;
        j += stride; phase = 0;
      }
    }
  }
  *sum_out = sum; *vmax_out = vmax;
}

__global__ __launch_bounds__(kPpcThreads) void ppcx_ppc_kernel(PpcArgs a, const double* T) {
  extern __shared__ int ldsi[];
  __shared__ double sred[kPpcThreads];
  __shared__ int s_cnt[kPpcWaves];
  const Dims& d = a.d;
  const int tid = threadIdx.x;
  // a cell's draws live in LDS when they fit (one workgroup per cell), otherwise in this workgroup's slice of a global
  // scratch buffer, and the workgroup takes cells in turn (how_many_posterior_draws = draws_after_tail / threshold
  // reaches 100 000 at the reference's defaults with 200 samples, R/methods.R:166-167)
  int* vals = a.scratch ? a.scratch + (long)blockIdx.x * a.n_gen : ldsi;
  const int n = a.n_gen;
  for (int cell = blockIdx.x; cell < a.n_cells; cell += gridDim.x) {   // g * S + s
    double sum = 0.0;
    int vmax = 0;
    {
      // eight wavefronts share the cell here: every lane has few draws, and draw after draw (each wavefront waiting for its
      // slowest lane) measured faster than the one loop of attempts that the wavefront-per-cell kernel runs (145 vs 185 ms at
      // 10 500 draws per cell x 200 000 cells)
      const int g = cell / d.S, s = cell - g * d.S;
      const double* Tg = T + (long)g * (d.C + 1) * a.n_draws;
      const double expo = a.exposure[s];
      for (int j = tid; j < n; j += kPpcThreads) {
        long src = j;
        if (a.resample) {                        // R/utilities.R:760: sample(draws, n, replace = TRUE)
          const double u = coord_uniform((uint32_t)j, (uint32_t)cell, 5u, 0u, a.k0, 0x50504331u);
          src = (long)(u * (double)a.n_draws); if (src >= a.n_draws) src = a.n_draws - 1;
        }
        double eta = expo + a.X[s] * Tg[src];
        for (int cc = 1; cc < d.C; ++cc) eta += a.X[(long)cc * d.S + s] * Tg[(long)cc * a.n_draws + src];
        const int val = nb2_log_rng(eta, Tg[(long)d.C * a.n_draws + src], a.k0, (uint32_t)cell, (uint32_t)j);
        sum += (double)val;
        vmax = val > vmax ? val : vmax;
        if (a.counts_rng) a.counts_rng[(long)j * a.n_cells + cell] = val;
        vals[j] = val;
      }
    }
    // mean (fixed-order block reduction)
    sred[tid] = sum;
    __syncthreads();
    for (int st = kPpcThreads / 2; st > 0; st >>= 1) { if (tid < st) sred[tid] += sred[tid + st]; __syncthreads(); }
    const double mean = sred[0] / (double)n;
    __syncthreads();
    double ss = 0.0;
    for (int j = tid; j < n; j += kPpcThreads) { const double t = (double)vals[j] - mean; ss += t * t; }
    sred[tid] = ss;
    __syncthreads();
    for (int st = kPpcThreads / 2; st > 0; st >>= 1) { if (tid < st) sred[tid] += sred[tid + st]; __syncthreads(); }
    const double sd = n > 1 ? sqrt(sred[0] / (double)(n - 1)) : NAN;
    __syncthreads();
    // largest draw of the workgroup (upper end of the bisections)
#pragma unroll
    for (int msk = 1; msk < 64; msk <<= 1) { const int o = __shfl_xor(vmax, msk, 64); vmax = o > vmax ? o : vmax; }
    if ((tid & 63) == 0) s_cnt[tid >> 6] = vmax;
    __syncthreads();
    vmax = s_cnt[0];
#pragma unroll
    for (int w = 1; w < kPpcWaves; ++w) vmax = max(vmax, s_cnt[w]);
    __syncthreads();
    // type-7 quantiles (R quantile default; rstan::summary) from the two order statistics around (n-1) p
    double q[2];
    const double pr[2] = {a.p_lo, a.p_hi};
    for (int k = 0; k < 2; ++k) {
      double h = (double)(n - 1) * pr[k];
      PPCX_OPAQUE(h);                            // rounded here: the product must not be fused into h - lo below
      int lo = (int)floor(h);
      if (lo > n - 1) lo = n - 1;
      if (lo < 0) lo = 0;
      int v0, v1;
      block_select_pair(vals, n, lo, vmax, s_cnt, tid, &v0, &v1);
      q[k] = lo >= n - 1 ? (double)v0 : fma(h - (double)lo, (double)v1 - (double)v0, (double)v0);   // one rounding, as in the oracle
    }
    if (tid == 0) {
      double* o = a.ci + (long)cell * 4;
      o[0] = mean; o[1] = sd; o[2] = q[0]; o[3] = q[1];
    }
    __syncthreads();                             // the next cell reuses vals
  }
}

// -----------------------------------------------------------------------------------------------------
// posterior-predictive draws + credible intervals, ONE WAVEFRONT per (gene <= K, sample) cell (up to kPpcWaveMaxDraws
// predictive draws per cell; the workgroup-per-cell kernel above serves longer ones).
//   * the checked genes' parameters are first gathered into a transposed table T[g][c][draw] (ppcx_ppc_table_kernel:
//     intercept, slopes, phi = exp(-sigma_raw) x truncation_compensation -- the per-gene work, done once per draw instead
//     of once per draw and sample), so that a wavefront's lanes read consecutive draws of one parameter;
//   * the lanes' rejection samplers run in ONE loop of attempts (gamma_attempt / poisson_attempt, ppcx_math.h): a lane that
//     has accepted goes on to its next draw at once -- the wavefront pays ~1.15 turns per draw instead of the slowest lane's;
//   * mean, sd and the order statistics around the two type-7 quantiles are taken inside the wavefront (cross-lane moves, no
//     workgroup barrier): bisection on the value over the LDS-resident integers, as in the workgroup kernel.
// -----------------------------------------------------------------------------------------------------
constexpr int kPpcWaveMaxDraws = 4096;        // per cell: 16 KB of integers per wavefront, 64 KB per workgroup, two workgroups per CU
// T[g][c][j], c = 0 intercept, 1 .. C - 1 slopes, C phi: one thread per (draw j, gene g), 32 x 32 tiles through LDS so that
// both the reads (along g) and the writes (along j) are coalesced
__global__ __launch_bounds__(256) void ppcx_ppc_table_kernel(const double* draws, long n_draws, Dims d, double tc, double* T) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  const long j0 = (long)blockIdx.x * 32; const int g0 = blockIdx.y * 32;
  const int ncol = d.C + 1;
  for (int c = 0; c < ncol; ++c) {
    for (int r = ty; r < 32; r += 8) {
      const long j = j0 + r; const int g = g0 + tx;
      double v = 0.0;
      if (j < n_draws && g < d.K) {
        const double* u = draws + j * (long)d.D;
        if (c == 0) v = u[d.off_intercept + g];
        else if (c < d.C) v = u[coef_index(d, c, g)];
        else v = exp(-u[d.off_sigma_raw + g]) * tc;                  // sigma = 1 ./ exp(sigma_raw), .stan:203,:264
      }
      tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int g = g0 + r; const long j = j0 + tx;
      if (j < n_draws && g < d.K) T[((long)g * ncol + c) * n_draws + j] = tile[tx][r];
    }
    __syncthreads();
  }
}
__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
  for (int msk = 1; msk < 64; msk <<= 1) v += __shfl_xor(v, msk, 64);
  return v;
}
__device__ __forceinline__ int wave_count_le(const int* vals, int n, int v, int lane) {
  int c = 0;
  for (int j = lane; j < n; j += 64) c += vals[j] <= v ? 1 : 0;
  return wave_sum_int(c);
}
__device__ __forceinline__ void wave_select_pair(const int* vals, int n, int r, int vmax, int lane, int* v_r, int* v_r1) {
  int lo = 0, hi = vmax;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (wave_count_le(vals, n, mid, lane) >= r + 1) hi = mid; else lo = mid + 1;
  }
  *v_r = lo;
  int nxt = lo;
  if (r + 1 < n && wave_count_le(vals, n, lo, lane) < r + 2) {     // the next order statistic is the smallest draw above lo
    int mn = 2147483647;
    for (int j = lane; j < n; j += 64) { const int x = vals[j]; mn = (x > lo && x < mn) ? x : mn; }
#pragma unroll
    for (int msk = 1; msk < 64; msk <<= 1) { const int o = __shfl_xor(mn, msk, 64); mn = o < mn ? o : mn; }
    nxt = mn;
  }
  *v_r1 = nxt;
}
__device__ __forceinline__ double wave_sum_double(double v) {
#pragma unroll
  for (int msk = 1; msk < 64; msk <<= 1) v += __shfl_xor(v, msk, 64);
  return v;
}

__global__ __launch_bounds__(256) void ppcx_ppc_wave_kernel(PpcArgs a, const double* T) {
  extern __shared__ int ldsw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = a.n_gen;
  int* vals = ldsw + (long)wave * ((n + 1) & ~1);                            // [n] the cell's draws
  for (int cell = blockIdx.x * 4 + wave; cell < a.n_cells; cell += gridDim.x * 4) {   // g * S + s
    double sum = 0.0; int vmax = 0;
    cell_draws(a, T, cell, vals, lane, 64, &sum, &vmax);
    // ---- mean, sd, type-7 quantiles (R quantile default; rstan::summary), all inside the wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the lanes' integers are in LDS before any lane reads another's
    __builtin_amdgcn_wave_barrier();
    const double mean = wave_sum_double(sum) / (double)n;
    double ss = 0.0;
    for (int j = lane; j < n; j += 64) { const double t = (double)vals[j] - mean; ss += t * t; }
    const double sd = n > 1 ? sqrt(wave_sum_double(ss) / (double)(n - 1)) : NAN;
#pragma unroll
    for (int msk = 1; msk < 64; msk <<= 1) { const int o = __shfl_xor(vmax, msk, 64); vmax = o > vmax ? o : vmax; }
    double q[2];
    const double pr[2] = {a.p_lo, a.p_hi};
    for (int k = 0; k < 2; ++k) {
      double h = (double)(n - 1) * pr[k];
      PPCX_OPAQUE(h);                            // rounded here: the product must not be fused into h - lo below
      int lo = (int)floor(h);
      if (lo > n - 1) lo = n - 1;
      if (lo < 0) lo = 0;
      int v0, v1;
      wave_select_pair(vals, n, lo, vmax, lane, &v0, &v1);
      q[k] = lo >= n - 1 ? (double)v0 : fma(h - (double)lo, (double)v1 - (double)v0, (double)v0);   // one rounding, as in the oracle
    }
    if (lane == 0) {
      double* o = a.ci + (long)cell * 4;
      o[0] = mean; o[1] = sd; o[2] = q[0]; o[3] = q[1];
    }
    __builtin_amdgcn_wave_barrier();             // the next cell reuses vals
  }
}

// -----------------------------------------------------------------------------------------------------
// The genes' dispersion tables (ppcx_disp.h), built once per model and whenever the exclusions change: one workgroup per
// gene. Phase 1: thread (panel, node) evaluates Fh and Dh at its node -- a loop over the gene's row in LDS, every thread on
// the same cell (so the y < 8 / y >= 8 branch of disp_cell is uniform). Phase 2: thread (panel, function) turns the panel's
// node values into its polynomial. 352 nodes x S cells per gene: a few milliseconds at 20 000 x 200.
// -----------------------------------------------------------------------------------------------------
constexpr int kDispThreads = 384;             // 352 nodes: one round
__global__ __launch_bounds__(kDispThreads) void ppcx_disp_build_kernel(const int* counts, int G, int S, const int* genes, int n_genes,
                                                                        DispFit fit, double* table) {
  extern __shared__ int s_row[];              // S counts, then (8-byte aligned) 2 x kDispPanels x kDispN node values
  const int gi = blockIdx.x;
  if (gi >= n_genes) return;
  const int g = genes ? genes[gi] : gi;
  const int tid = threadIdx.x;
  double* s_val = reinterpret_cast<double*>(s_row + ((S + 1) & ~1));
  for (int s = tid; s < S; s += kDispThreads) s_row[s] = counts[(long)g * S + s];
  __syncthreads();
  constexpr int NN = kDispPanels * kDispN;
  for (int t = tid; t < NN; t += kDispThreads) {
    const int p = t / kDispN, k = t - p * kDispN;
    double lo;
    const double sg = disp_node_sigma(fit, p, k, &lo);
    const DispPoint pt = disp_point(sg, lo);
    double F, D;
    disp_row(s_row, S, 0, 1, pt, &F, &D);
    s_val[(p * 2 + 0) * kDispN + k] = F; s_val[(p * 2 + 1) * kDispN + k] = D;
  }
  __syncthreads();
  for (int t = tid; t < 2 * kDispPanels; t += kDispThreads) {
    double out[kDispStride];
    disp_fit_panel(fit, s_val + t * kDispN, out);
    double* dst = table + (long)g * kDispGeneDoubles + (long)t * kDispStride;
#pragma unroll
    for (int m = 0; m < kDispStride; ++m) dst[m] = out[m];
  }
}

__global__ void ppcx_gather_kernel(const double* draws, long n_rows, int D, const int* cols, int n_cols, double* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * n_cols) return;
  const long r = i / n_cols; const int cidx = (int)(i % n_cols);
  out[i] = draws[r * D + cols[cidx]];
}

// a rank that leaves a gene-sharded fit early (a local failure) says so to every peer: their state machines stop waiting for it
__global__ void ppcx_xchg_abort_kernel(XchgArgs x) {
  const int k = threadIdx.x;
  if (k < x.nranks) sys_store(x.flags[k] + xchg_abort_index(x, x.rank), (unsigned long long)x.epoch);
  __threadfence_system();
}

__global__ void ppcx_fill_kernel(double* p, long n, double val) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = val;
}

// -----------------------------------------------------------------------------------------------------
// launch helpers (host)
// -----------------------------------------------------------------------------------------------------
static int loglik_generic_possible(const Dims& d) { return !d.x0_is_one ? 2 : ((d.C >= 2 && d.K > 0 && !d.x1_binary) ? 1 : 0); }
// LDS of a log-likelihood workgroup: the two tables, exp(exposure_s), and the design columns its genes read -- columns 1 .. C - 1
// where X[,1] == 1 (sweep_cells: S * C doubles in all; 8 960 samples fit at C = 2), the exposures and the whole of X for a design
// without the column of ones (generic_cells: S * (2 + C))
size_t loglik_lds_bytes(const Dims& d) {
  const size_t per_sample = loglik_generic_possible(d) == 2 ? (size_t)(2 + d.C) : (size_t)(d.C < 1 ? 1 : d.C);
  return sizeof(double) * (2 * kLogTabSize + 2 * kWinTabSize + (size_t)d.S * per_sample + 2 * kLdsPad);
}
// the instantiation a model runs: CM design columns (2, 4 or 8) and which route with an exp per cell its genes can need
// (lane_gene_sums: 0 none, 1 slopes on columns of any values, 2 no column of ones)
#define PPCX_BY_CM_GEN(KERNEL, CM, GEN, EXPR)                                                                     \
  do {                                                                                                            \
    if ((CM) <= 2) { if ((GEN) == 0) { auto k_ = KERNEL<2, 0>; EXPR; } else if ((GEN) == 1) { auto k_ = KERNEL<2, 1>; EXPR; } else { auto k_ = KERNEL<2, 2>; EXPR; } } \
    else if ((CM) <= 4) { if ((GEN) == 0) { auto k_ = KERNEL<4, 0>; EXPR; } else if ((GEN) == 1) { auto k_ = KERNEL<4, 1>; EXPR; } else { auto k_ = KERNEL<4, 2>; EXPR; } } \
    else if ((CM) <= 8) { if ((GEN) == 0) { auto k_ = KERNEL<8, 0>; EXPR; } else if ((GEN) == 1) { auto k_ = KERNEL<8, 1>; EXPR; } else { auto k_ = KERNEL<8, 2>; EXPR; } } \
    else { auto k_ = KERNEL<16, 0>; EXPR; }   /* 9 .. 16 columns: indicator designs only (ppcx_model_create refuses the others) */ \
  } while (0)
static const void* loglik_kernel_ptr(int CM, int gen) {
  const void* f = nullptr;
  PPCX_BY_CM_GEN(ppcx_loglik_kernel, CM, gen, f = (const void*)k_);
  return f;
}
int loglik_resident_workgroups_per_cu(int CM, const Dims& d) {
  int n = 0;
  const size_t lds_bytes = loglik_lds_bytes(d);
  const int gen = loglik_generic_possible(d);
  const void* f = loglik_kernel_ptr(CM, gen);
  if (lds_bytes > 64u * 1024u) {               // more than the default limit of dynamic LDS: ask for it once
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
  }
  hipError_t e = hipSuccess;
  PPCX_BY_CM_GEN(ppcx_loglik_kernel, CM, gen, e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_, 256, lds_bytes));
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
hipError_t launch_loglik_kernel(int CM, const LoglikArgs& a, hipStream_t st) {
  const size_t lds_bytes = loglik_lds_bytes(a.d);
  const dim3 grid((unsigned)((a.nbpc + 7) / 8 * 8) * (unsigned)a.nchains);
  LoglikArgs args = a;
  void* params[] = {&args};
  return hipLaunchKernel(loglik_kernel_ptr(CM, loglik_generic_possible(a.d)), grid, dim3(256), params, lds_bytes, st);
}
hipError_t launch_close_kernel(int CM, const CloseArgs& a, int nblocks, int nchains, hipStream_t st) {
  const dim3 grid(nblocks, nchains);
  if (CM <= 2) hipLaunchKernelGGL((ppcx_close_kernel<2>), grid, dim3(256), 0, st, a);
  else if (CM <= 4) hipLaunchKernelGGL((ppcx_close_kernel<4>), grid, dim3(256), 0, st, a);
  else if (CM <= 8) hipLaunchKernelGGL((ppcx_close_kernel<8>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((ppcx_close_kernel<16>), grid, dim3(256), 0, st, a);
  return hipGetLastError();
}
static const void* ls_kernel_ptr(int CM, int gen) {
  const void* f = nullptr;
  PPCX_BY_CM_GEN(ppcx_ls_kernel, CM, gen, f = (const void*)k_);
  return f;
}
static size_t ls_lds_bytes(const Dims& d) { const size_t a = loglik_lds_bytes(d); return a > sizeof(StepShared) ? a : sizeof(StepShared); }
int ls_resident_workgroups_per_cu(int CM, const Dims& d) {
  int n = 0;
  const size_t lds_bytes = ls_lds_bytes(d);
  const int gen = loglik_generic_possible(d);
  if (lds_bytes > 64u * 1024u) {
    if (hipFuncSetAttribute(ls_kernel_ptr(CM, gen), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
  }
  hipError_t e = hipSuccess;
  PPCX_BY_CM_GEN(ppcx_ls_kernel, CM, gen, e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_, 256, lds_bytes));
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
hipError_t launch_ls_kernel(int CM, const LoglikArgs& a, const StepArgs& sa, int n_srun, int n_chains_total, int spec, hipStream_t st,
                            hipEvent_t ev_start, hipEvent_t ev_stop) {
  const size_t lds_bytes = ls_lds_bytes(a.d);
  // runs: the first n_srun hold 7 range blocks (and a state machine) per chain, the others 8
  int runs = n_srun;
  if (a.nbpc > 7 * n_srun) runs += (a.nbpc - 7 * n_srun + 7) / 8;
  const dim3 grid((unsigned)runs * 8u * (unsigned)a.nchains);
  LoglikArgs args = a; StepArgs sargs = sa;
  void* params[] = {&args, &sargs, &n_srun, &n_chains_total, &spec};
  if (ev_start && ev_stop)
    return hipExtLaunchKernel(ls_kernel_ptr(CM, loglik_generic_possible(a.d)), grid, dim3(256), params, lds_bytes, st, ev_start, ev_stop, 0);
  return hipLaunchKernel(ls_kernel_ptr(CM, loglik_generic_possible(a.d)), grid, dim3(256), params, lds_bytes, st);
}
hipError_t launch_gene_kernel(int CM, const GeneArgs& a, int nblocks, int nchains, hipStream_t st) {
  const dim3 grid(nblocks, nchains);
  if (CM <= 2) hipLaunchKernelGGL((ppcx_gene_kernel<2>), grid, dim3(256), 0, st, a);
  else if (CM <= 4) hipLaunchKernelGGL((ppcx_gene_kernel<4>), grid, dim3(256), 0, st, a);
  else if (CM <= 8) hipLaunchKernelGGL((ppcx_gene_kernel<8>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((ppcx_gene_kernel<16>), grid, dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_step_kernel(const StepArgs& a, int nblocks, int nchains, hipStream_t st) {
  hipLaunchKernelGGL(ppcx_step_kernel, dim3(a.upd_vecs ? nblocks : 1, nchains), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_sum_shards_kernel(const ShardSumArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(ppcx_sum_shards_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_update_kernel(const UpdateArgs& a, int nblocks, int nchains, hipStream_t st) {
  hipLaunchKernelGGL(ppcx_update_kernel, dim3(nblocks, nchains), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_advi_kernel(const AdviArgs& a, int nblocks, hipStream_t st) {
  hipLaunchKernelGGL(ppcx_advi_kernel, dim3(nblocks), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_advi_elbo_kernel(const AdviElboArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(ppcx_advi_elbo_kernel, dim3(1), dim3(64), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_ppc_kernel(const PpcArgs& a, const double* T, int nblocks, hipStream_t st) {
  hipLaunchKernelGGL(ppcx_ppc_kernel, dim3(nblocks), dim3(kPpcThreads), a.scratch ? 0 : sizeof(int) * (size_t)a.n_gen, st, a, T);
  return hipGetLastError();
}
size_t ppc_wave_lds_bytes(int n_gen) { return sizeof(int) * 4 * (size_t)((n_gen + 1) & ~1); }
int ppc_wave_max_draws() { return kPpcWaveMaxDraws; }
hipError_t launch_ppc_table_kernel(const double* draws, long n_draws, const Dims& d, double tc, double* T, hipStream_t st) {
  const dim3 grid((unsigned)((n_draws + 31) / 32), (unsigned)((d.K + 31) / 32));
  hipLaunchKernelGGL(ppcx_ppc_table_kernel, grid, dim3(256), 0, st, draws, n_draws, d, tc, T);
  return hipGetLastError();
}
hipError_t launch_ppc_wave_kernel(const PpcArgs& a, const double* T, int nblocks, hipStream_t st) {
  const size_t lds = ppc_wave_lds_bytes(a.n_gen);
  if (lds > 64u * 1024u) {
    hipError_t e = hipFuncSetAttribute((const void*)ppcx_ppc_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(ppcx_ppc_wave_kernel, dim3(nblocks), dim3(256), lds, st, a, T);
  return hipGetLastError();
}
size_t disp_build_lds_bytes(int S) { return sizeof(int) * (size_t)((S + 1) & ~1) + sizeof(double) * 2 * kDispPanels * kDispN; }
hipError_t launch_disp_build_kernel(const int* counts, int G, int S, const int* genes, int n_genes, const DispFit& fit, double* table, hipStream_t st) {
  const size_t lds = disp_build_lds_bytes(S);
  if (lds > 64u * 1024u) {
    hipError_t e = hipFuncSetAttribute((const void*)ppcx_disp_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(ppcx_disp_build_kernel, dim3((unsigned)n_genes), dim3(kDispThreads), lds, st, counts, G, S, genes, n_genes, fit, table);
  return hipGetLastError();
}
hipError_t launch_gather_kernel(const double* draws, long n_rows, int D, const int* cols, int n_cols, double* out, hipStream_t st) {
  const long n = n_rows * n_cols;
  hipLaunchKernelGGL(ppcx_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, draws, n_rows, D, cols, n_cols, out);
  return hipGetLastError();
}
hipError_t launch_xchg_abort_kernel(const XchgArgs& x, hipStream_t st) {
  hipLaunchKernelGGL(ppcx_xchg_abort_kernel, dim3(1), dim3(64), 0, st, x);
  return hipGetLastError();
}
hipError_t launch_fill_kernel(double* p, long n, double val, hipStream_t st) {
  hipLaunchKernelGGL(ppcx_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, val);
  return hipGetLastError();
}

}  // namespace ppcx
