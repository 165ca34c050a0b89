// ppcx_kernels.hip -- gfx950 kernels of the NB hierarchical NUTS / posterior-predictive engine.
//
//   ppcx_gene_kernel<L,CM>  "kernel A": one leapfrog (or evaluation) for every chain, fused with the
//                           per-coordinate NUTS bookkeeping; streams the int32 count matrix once.
//                           Replaces lp_reduce + map_rect + X*alpha + the gene-level priors of
//                           inst/stan/negBinomial_MPI.stan:58-120,:205,:219-240 and Stan's leapfrog.
//   ppcx_chain_kernel       "kernel B": per-chain reduction of block partials, hyper-parameters,
//                           NUTS/adaptation state machine (ppcx_nuts.h).
//   ppcx_ppc_kernel         generated quantities (.stan:259-266) + credible-interval summary
//                           (R/utilities.R:685-703 / :733-784): NB draws straight into LDS, bitonic
//                           sort, type-7 quantiles, mean, sd.
//   ppcx_gather_kernel      column gather of the retained draws.
//
// Work decomposition of kernel A (DESIGN.md "lp/grad kernel"): a gene is owned by L lanes of one
// wavefront (L in {1,2,4,...,64}, chosen per problem so that ceil(S/L)*L wastes few lanes and the
// launch fills 1024 SIMDs evenly); lanes stride over that gene's samples, read the per-sample
// constants from LDS, and combine with an L-lane xor-shuffle butterfly. Per-block partial sums go to a
// slab that kernel B reduces in a fixed order, so results are bitwise reproducible for a fixed grid.
#include <hip/hip_runtime.h>
#include "ppcx_gene.h"
#include "ppcx_kernels.h"

namespace ppcx {

__device__ __forceinline__ double wave_xor_add(double v, int mask) { return v + __shfl_xor(v, mask, 64); }

template <int L, int CM>
__global__ __launch_bounds__(256, 2) void ppcx_gene_kernel(GeneArgs a) {
  constexpr int NCM = CM + 1;
  constexpr int GPW = 64 / L;                 // genes per wavefront
  extern __shared__ double lds[];
  const int chain = blockIdx.y;
  const Cmd c = a.cmds[chain];
  if (c.type == CMD_DONE) return;
  const Dims& d = a.d;
  const int S = d.S, C = d.C;
  double* wacc = lds;                          // [4][PT_COUNT]
  double* sE = lds + 4 * PT_COUNT;             // exp(exposure_s)
  double* sExpo = sE + S;
  double* sX = sExpo + S;                      // S x C column-major
  const int tid = threadIdx.x;
  for (int i = tid; i < 4 * PT_COUNT; i += 256) wacc[i] = 0.0;
  for (int i = tid; i < S; i += 256) { sE[i] = a.sampleE[i]; sExpo[i] = a.exposure[i]; }
  for (int i = tid; i < S * C; i += 256) sX[i] = a.X[i];
  __syncthreads();

  const VecRef v{a.vecs + (long)chain * V_COUNT * a.Dpad, a.Dpad};
  double* draws = a.draws ? a.draws + (long)chain * a.draws_chain_stride : nullptr;
  const int wave = tid >> 6, lane = tid & 63, sub = lane % L, gl = lane / L;
  const int ngroups = (d.G + GPW - 1) / GPW;
  const bool do_eval = c.type != CMD_FLUSH;
  const int m_merge = c.type == CMD_LEAF ? c.n_merge : 0;

  for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
    GeneCtx<CM> x;
    gene_begin<CM>(d, c, v, grp * GPW + gl, sub == 0, draws, x);
    if (!do_eval) continue;                   // CMD_FLUSH: bookkeeping only (uniform over the launch)

    CellAcc<CM> acc; acc.zero();
    gene_cells<CM>(d, x, a.counts + (long)x.gg * S, sE, sExpo, sX, sub, L, acc);
    // L-lane butterfly: every lane of the gene ends with the gene totals
#pragma unroll
    for (int msk = 1; msk < L; msk <<= 1) {
      acc.T1 = wave_xor_add(acc.T1, msk); acc.SP = wave_xor_add(acc.SP, msk); acc.T2u = wave_xor_add(acc.T2u, msk);
      acc.T3 = wave_xor_add(acc.T3, msk); acc.T4 = wave_xor_add(acc.T4, msk);
      if (!d.x0_is_one || (C >= 2 && d.K > 0)) {   // uniform: some gene of this launch may take the generic path
#pragma unroll
        for (int cc = 0; cc < CM; ++cc) if (cc < C) acc.T2x[cc] = wave_xor_add(acc.T2x[cc], msk);
      }
    }
    double pn[NCM], part[10];
    gene_end<CM>(d, c, v, x, acc, a.Sy, a.SyE, a.SyX, a.ncell, a.Lg1, part, pn);
#pragma unroll
    for (int msk = L; msk < 64; msk <<= 1) {
#pragma unroll
      for (int k = 0; k < 10; ++k) part[k] = wave_xor_add(part[k], msk);
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 10; ++k) wacc[wave * PT_COUNT + k] += part[k];
    }

    if (c.type == CMD_LEAF) {                  // iterative build_tree bookkeeping for this gene's coordinates
      NodeVals nv[NCM];
#pragma unroll
      for (int j = 0; j < NCM; ++j) nv[j] = NodeVals{pn[j], pn[j]};
      for (int lev = 0; lev < m_merge; ++lev) {
        double dots[6] = {0, 0, 0, 0, 0, 0};
        if (x.writer) {
#pragma unroll
          for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_merge_dots(v, x.idx[j], lev, pn[j], x.minv[j], &nv[j], dots);
        }
#pragma unroll
        for (int msk = L; msk < 64; msk <<= 1) {
#pragma unroll
          for (int k = 0; k < 6; ++k) dots[k] = wave_xor_add(dots[k], msk);
        }
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) wacc[wave * PT_COUNT + PT_DOTS + 6 * lev + k] += dots[k];
        }
      }
      if (!c.subtree_complete) {
        if (x.writer) {
#pragma unroll
          for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_store_slot(v, x.idx[j], m_merge, pn[j], nv[j]);
        }
      } else {
        double top[6] = {0, 0, 0, 0, 0, 0};
        if (x.writer) {
#pragma unroll
          for (int j = 0; j < NCM; ++j) if (j < x.ncoord) coord_top_dots(v, x.idx[j], c.dir, pn[j], x.minv[j], nv[j], top);
        }
#pragma unroll
        for (int msk = L; msk < 64; msk <<= 1) {
#pragma unroll
          for (int k = 0; k < 6; ++k) top[k] = wave_xor_add(top[k], msk);
        }
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) wacc[wave * PT_COUNT + PT_TOP + k] += top[k];
        }
      }
    }
  }
  __syncthreads();
  const int np = parts_used(c);
  double* slab = a.partials + ((long)chain * gridDim.x + blockIdx.x) * PT_COUNT;
  for (int k = tid; k < np; k += 256)
    slab[k] = ((wacc[k] + wacc[PT_COUNT + k]) + wacc[2 * PT_COUNT + k]) + wacc[3 * PT_COUNT + k];
}

// -----------------------------------------------------------------------------------------------------
// kernel B
// -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ppcx_chain_kernel(ChainArgs a) {
  __shared__ double sm[8][32];
  __shared__ double red[PT_COUNT];
  __shared__ double hv[V_COUNT * 8];           // the six hyper coordinates of every per-coordinate vector
  const int chain = blockIdx.x, tid = threadIdx.x;
  ChainState* stp = a.states + chain;
  if (stp->phase == PH_DONE) return;
  const Cmd ex = a.cmds[chain];
  double* hvg = a.hyper_vecs + (long)chain * V_COUNT * 8;
  for (int i = tid; i < V_COUNT * 8; i += 256) hv[i] = hvg[i];
  for (int i = tid; i < PT_COUNT; i += 256) red[i] = 0.0;
  const bool have_parts = stp->phase != PH_START;
  __syncthreads();
  if (have_parts) {
    const int np = parts_used(ex);
    const double* slab = a.partials + (long)chain * a.nblocks * PT_COUNT;
    for (int v0 = 0; v0 < np; v0 += 32) {
      const int vv = v0 + (tid & 31), ch = tid >> 5;
      double s = 0.0;
      if (vv < np) for (int b = ch; b < a.nblocks; b += 8) s += slab[(long)b * PT_COUNT + vv];
      sm[ch][tid & 31] = s;
      __syncthreads();
      if (tid < 32 && v0 + tid < np) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sm[k][tid];
        red[v0 + tid] = t;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  if (tid == 0) {
    ChainState st = *stp;
    ChainIO io;
    io.draws = a.draws ? a.draws + (long)chain * a.draws_chain_stride : nullptr;
    io.out.lp = a.out_lp ? a.out_lp + (long)chain * a.n_keep : nullptr;
    io.out.stepsize = a.out_stepsize ? a.out_stepsize + (long)chain * a.iter : nullptr;
    io.out.treedepth = a.out_treedepth ? a.out_treedepth + (long)chain * a.iter : nullptr;
    io.out.n_leapfrog = a.out_n_leapfrog ? a.out_n_leapfrog + (long)chain * a.iter : nullptr;
    io.out.divergent = a.out_divergent ? a.out_divergent + (long)chain * a.iter : nullptr;
    io.out.accept = a.out_accept ? a.out_accept + (long)chain * a.iter : nullptr;
    Cmd nc;
    chain_step(a.d, st, ex, red, have_parts, VecRef{hv, 8}, io, nc);
    *stp = st;
    a.cmds[chain] = nc;
    if (st.phase == PH_DONE) a.done[chain] = 1 + st.error;
  }
  __syncthreads();
  for (int i = tid; i < V_COUNT * 8; i += 256) hvg[i] = hv[i];
}

// -----------------------------------------------------------------------------------------------------
// posterior-predictive draws + credible intervals, one workgroup per (gene <= K, sample) cell
// -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ppcx_ppc_kernel(PpcArgs a) {
  extern __shared__ int ldsi[];
  __shared__ double sred[256];
  const Dims& d = a.d;
  const int cell = blockIdx.x;                 // g * S + s
  const int g = cell / d.S, s = cell % d.S;
  const int tid = threadIdx.x;
  int* vals = ldsi;
  const int n = a.n_gen, npad = a.n_pad;
  double sum = 0.0;
  for (int j = tid; j < npad; j += 256) {
    int val = 2147483647;                      // padding sorts to the end
    if (j < n) {
      long src = j;
      if (a.resample) {                        // R/utilities.R:760: sample(draws, n, replace = TRUE)
        const double u = coord_uniform((uint32_t)j, (uint32_t)cell, 5u, 0u, a.k0, 0x50504331u);
        src = (long)(u * (double)a.n_draws); if (src >= a.n_draws) src = a.n_draws - 1;
      }
      const double* u_ = a.draws + src * (long)d.D;
      double eta = a.exposure[s] + a.X[s] * u_[d.off_intercept + g];
      if (d.C >= 2) eta += a.X[(long)d.S + s] * u_[d.off_alpha1 + g];
      for (int cc = 2; cc < d.C; ++cc) eta += a.X[(long)cc * d.S + s] * u_[coef_index(d, cc, g)];
      const double phi = exp(-u_[d.off_sigma_raw + g]) * a.truncation_compensation;
      val = nb2_log_rng(eta, phi, a.k0, (uint32_t)cell, (uint32_t)j);
      sum += (double)val;
      if (a.counts_rng) a.counts_rng[(long)j * a.n_cells + cell] = val;
    }
    vals[j] = val;
  }
  // mean (fixed-order block reduction)
  sred[tid] = sum;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (tid < st) sred[tid] += sred[tid + st]; __syncthreads(); }
  const double mean = sred[0] / (double)n;
  __syncthreads();
  double ss = 0.0;
  for (int j = tid; j < n; j += 256) { const double t = (double)vals[j] - mean; ss += t * t; }
  sred[tid] = ss;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (tid < st) sred[tid] += sred[tid + st]; __syncthreads(); }
  const double sd = n > 1 ? sqrt(sred[0] / (double)(n - 1)) : NAN;
  // bitonic sort of the padded array in LDS
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < npad; i += 256) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const int x = vals[i], y = vals[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { vals[i] = y; vals[ixj] = x; }
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {                              // type-7 quantiles (R quantile default; rstan::summary)
    double q[2];
    const double pr[2] = {a.p_lo, a.p_hi};
    for (int k = 0; k < 2; ++k) {
      const double h = (double)(n - 1) * pr[k];
      const int lo = (int)floor(h);
      q[k] = lo >= n - 1 ? (double)vals[n - 1] : (double)vals[lo] + (h - (double)lo) * ((double)vals[lo + 1] - (double)vals[lo]);
    }
    double* o = a.ci + (long)cell * 4;
    o[0] = mean; o[1] = sd; o[2] = q[0]; o[3] = q[1];
  }
}

__global__ void ppcx_gather_kernel(const double* draws, long n_rows, int D, const int* cols, int n_cols, double* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * n_cols) return;
  const long r = i / n_cols; const int cidx = (int)(i % n_cols);
  out[i] = draws[r * D + cols[cidx]];
}

__global__ void ppcx_fill_kernel(double* p, long n, double val) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = val;
}

// -----------------------------------------------------------------------------------------------------
// launch helpers (host)
// -----------------------------------------------------------------------------------------------------
template <int L, int CM>
static hipError_t launch_gene_t(const GeneArgs& a, dim3 grid, size_t lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL((ppcx_gene_kernel<L, CM>), grid, dim3(256), lds_bytes, st, a);
  return hipGetLastError();
}
template <int CM>
static hipError_t launch_gene_l(int L, const GeneArgs& a, dim3 grid, size_t lds_bytes, hipStream_t st) {
  switch (L) {
    case 1: return launch_gene_t<1, CM>(a, grid, lds_bytes, st);
    case 2: return launch_gene_t<2, CM>(a, grid, lds_bytes, st);
    case 4: return launch_gene_t<4, CM>(a, grid, lds_bytes, st);
    case 8: return launch_gene_t<8, CM>(a, grid, lds_bytes, st);
    case 16: return launch_gene_t<16, CM>(a, grid, lds_bytes, st);
    case 32: return launch_gene_t<32, CM>(a, grid, lds_bytes, st);
    default: return launch_gene_t<64, CM>(a, grid, lds_bytes, st);
  }
}
hipError_t launch_gene_kernel(int L, int CM, const GeneArgs& a, int nblocks, int nchains, hipStream_t st) {
  const size_t lds_bytes = sizeof(double) * (4 * PT_COUNT + (size_t)a.d.S * (2 + a.d.C));
  const dim3 grid(nblocks, nchains);
  if (CM <= 2) return launch_gene_l<2>(L, a, grid, lds_bytes, st);
  if (CM <= 4) return launch_gene_l<4>(L, a, grid, lds_bytes, st);
  return launch_gene_l<8>(L, a, grid, lds_bytes, st);
}
hipError_t launch_chain_kernel(const ChainArgs& a, int nchains, hipStream_t st) {
  hipLaunchKernelGGL(ppcx_chain_kernel, dim3(nchains), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_ppc_kernel(const PpcArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(ppcx_ppc_kernel, dim3(a.n_cells), dim3(256), sizeof(int) * (size_t)a.n_pad, st, a);
  return hipGetLastError();
}
hipError_t launch_gather_kernel(const double* draws, long n_rows, int D, const int* cols, int n_cols, double* out, hipStream_t st) {
  const long n = n_rows * n_cols;
  hipLaunchKernelGGL(ppcx_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, draws, n_rows, D, cols, n_cols, out);
  return hipGetLastError();
}
hipError_t launch_fill_kernel(double* p, long n, double val, hipStream_t st) {
  hipLaunchKernelGGL(ppcx_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, val);
  return hipGetLastError();
}

}  // namespace ppcx
