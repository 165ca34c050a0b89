// ppcx_kernels.h -- argument blocks and launchers of the gfx950 kernels (ppcx_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "ppcx_gene.h"

namespace ppcx {

constexpr int kLdsPad = 256;     // entries the sweep may read past the per-sample arrays in LDS (4 x 64 lanes)
struct LoglikArgs {
  Dims d;
  CellData cd;                  // counts, dispersion tables, gene flags (ppcx_gene.h)
  const double* sampleE;        // exp(exposure_s)
  const double* exposure;       // S
  const double* X;              // S x C column-major
  double* vecs;                 // [chains][V_COUNT][Dpad] (read only here)
  long Dpad;
  const Cmd* cmds;              // [chains]
  double* sums;                 // [chains][3+CM][G] (GeneSumsV)
  const double* logtab;         // 2 x 256 doubles (device), ppcx_math.h table_log
  const double* wintab;         // 2 x 1024 doubles (device), ppcx_math.h window table
  const int* order;             // [G] launch position -> gene (host: gene_order)
  int lgL;                      // log2 of the lanes per gene
  int nchains;                  // chains of this launch
  const int* active;            // [nchains] their indices in cmds / vecs / sums (null: 0 .. nchains - 1)
  int nbpc;                     // workgroups per chain; wavefront j = 0 .. 4 nbpc - 1 of a chain takes the gene positions
  const int* bounds;            // [4 nbpc + 1]  bounds[j] .. bounds[j + 1] - 1 (host: plan_launch, balanced by cost)
#ifdef PPCX_TRACE               // development builds only (scripts/gpu_pass_trace.py): clock stamps of the passes' phases
  unsigned long long* trace;    // [kTraceBlocks][4 waves][kTracePasses][kTraceStamps] or null
#endif
};
#ifdef PPCX_TRACE
constexpr int kTraceBlocks = 64, kTracePasses = 8, kTraceStamps = 8;
#endif

struct CloseArgs {
  Dims d;
  const double* Sy;             // per-gene sufficient statistics over non-excluded cells
  const double* SyE;
  const double* SyX;            // [C][G]
  const double* SX;             // [C][G] sum of X_sc over the gene's non-excluded cells
  const double* ncell;          // [G] number of non-excluded cells
  const double* Lg1;            // per-gene sum of lgamma(y+1)
  const double* sums;
  double* vecs; long Dpad;
  const Cmd* cmds;
  double* partials;             // [chains][nblocks_close][PT_COUNT]
};

// the gene kernel of a pipelined round (ppcx_gene_kernel): close kernel + the per-coordinate work of the command + the
// constants of the anticipated next position
struct GeneArgs {
  CloseArgs c;
  double* draws; long draws_chain_stride;   // PRE_STORE_DRAW
  int spec;                     // anticipate the next leaf's position (models whose cell paths read the constants only)
};

// Direct exchange of a gene-sharded run (one shard per rank; the reference's map_rect over gene shards,
// inst/stan/negBinomial_MPI.stan:226-240): every rank's state machine adds the ranks' partial sums itself. Rank k's receive
// buffer is mapped into every rank (peer-mapped device memory: hipIpc handles between processes, plain pointers inside one);
// a state machine stores its PT_COUNT sums into every rank's buffer, then a sequence number, and waits until the sequence
// numbers of all ranks have arrived in its own buffer. No collective library call, no kernel boundary: the exchange happens
// inside the merged launch of a pipelined round, beside the log-likelihood workgroups.
constexpr int kMaxRanks = 16;
struct XchgArgs {
  int nranks = 1, rank = 0, max_chains = 0;
  int chain0 = 0;                              // first chain of the launch's chain group in the buffers
  unsigned epoch = 0;                          // of this fit: stale sequence numbers of earlier fits never match
  long long timeout_ticks = 0;                 // of the 100 MHz wall clock: a peer that does not arrive fails the chain
  double* recv[kMaxRanks];                     // [2 slots][nranks][max_chains][PT_COUNT]
  unsigned long long* flags[kMaxRanks];        // [2 slots][nranks][max_chains] sequence numbers, then [nranks] abort epochs
};
PPCX_HD long xchg_recv_index(const XchgArgs& x, int slot, int src, int chain) { return (((long)slot * x.nranks + src) * x.max_chains + x.chain0 + chain) * PT_COUNT; }
PPCX_HD long xchg_flag_index(const XchgArgs& x, int slot, int src, int chain) { return ((long)slot * x.nranks + src) * x.max_chains + x.chain0 + chain; }
PPCX_HD long xchg_abort_index(const XchgArgs& x, int src) { return 2L * x.nranks * x.max_chains + src; }
PPCX_HD size_t xchg_recv_doubles(int nranks, int max_chains) { return (size_t)2 * nranks * max_chains * PT_COUNT; }
PPCX_HD size_t xchg_flag_words(int nranks, int max_chains) { return (size_t)2 * nranks * max_chains + nranks; }

enum StepPhase : int { STEP_REDUCE = 1, STEP_ADVANCE = 2 };
struct StepArgs {
  Dims d;
  int phases;                   // STEP_REDUCE | STEP_ADVANCE (one launch), or the two halves around a shard exchange
  const ChainState* states_in; ChainState* states_out;   // double-buffered between rounds
  const Cmd* cmds_in; Cmd* cmds_out;
  const double* hyper_in; double* hyper_out;             // [chains][V_COUNT][8]
  const double* partials; int nblocks_close;             // [chains][slab_stride][PT_COUNT], rows 0 .. nblocks_close - 1 are summed
  int slab_stride;
  const double* t0; int nblocks_update;                  // [chains][nblocks_update]
  double* red;                                           // [chains][PT_COUNT]
  double* draws; long draws_chain_stride;
  int n_keep, iter;
  double* out_lp; double* out_stepsize; int* out_treedepth; int* out_n_leapfrog; int* out_divergent; double* out_accept;
  int* done;                    // [chains]
  // the per-coordinate work of the new command in the same launch (null upd_vecs: a separate ppcx_update_kernel does
  // it): grid.x workgroups per chain, each runs the step redundantly and updates its share of the coordinates
  double* upd_vecs; long upd_Dpad; double* upd_t0_out;
  XchgArgs x;                   // nranks > 1: the sums are exchanged with the other ranks' state machines (pipelined rounds)
};

constexpr int kMaxShards = 16;
struct ShardSumArgs { double* bufs[kMaxShards]; int n_shards; int n; };

struct UpdateArgs {
  Dims d;
  const Cmd* cmds;              // the commands the step kernel just wrote
  double* vecs; long Dpad;
  double* draws; long draws_chain_stride;
  double* t0_out;               // [chains][nblocks_update]
};

enum AdviOp : int { ADVI_DRAW = 0, ADVI_RESET = 1, ADVI_STEP = 2 };
struct AdviArgs {
  Dims d;
  double* vecs; long Dpad;      // slot c = evaluation slot c; slot 0 also stores mu/omega/history/initial point
  double* hyper;                // slot 0 hyper vectors [V_COUNT][8]
  Cmd* cmds;                    // [n_slots]
  const double* red;            // reduced sums of slot 0 (ADVI_STEP)
  int op, n_slots, first_iter;
  double eta_scaled;
  uint32_t k0, prev_draw, draw_base;
  double* out_draws; int out_row0;   // non-null: write the draws to [row][D] instead of the evaluation slots
  double* omega_part;           // [nblocks]
};
struct AdviElboArgs { Dims d; const Cmd* cmds; const double* red; int n_slots; double* acc; const double* omega_part; int n_omega_parts; };

struct PpcArgs {
  Dims d;
  const double* draws;          // [n_draws][D]
  long n_draws;
  const double* exposure; const double* X;
  double truncation_compensation, p_lo, p_hi;
  uint32_t k0;
  int n_gen, resample, n_cells;
  double* ci;                   // [K*S][4] mean, sd, lower, upper
  int* counts_rng;              // [n_gen][K*S] or null
  int* scratch;                 // null: a cell's draws in LDS (grid = n_cells); else [grid][n_gen] global scratch, cells in turn
};

hipError_t launch_loglik_kernel(int CM, const LoglikArgs& a, hipStream_t st);
int loglik_resident_workgroups_per_cu(int CM, const Dims& d);   // 0: the kernel cannot be launched with this much LDS
size_t loglik_lds_bytes(const Dims& d);
hipError_t launch_close_kernel(int CM, const CloseArgs& a, int nblocks, int nchains, hipStream_t st);
// pipelined rounds: the merged launch (the state machines take position 7 of the first n_srun runs of 8 x chains
// workgroups, log-likelihood range blocks everything else) and the gene kernel
// ev_start / ev_stop (both or neither): events attached to the dispatch itself (hipExtLaunchKernel) -- their elapsed time is the
// kernel's own duration, without the marker packets of a hipEventRecord on either side of the launch
hipError_t launch_ls_kernel(int CM, const LoglikArgs& a, const StepArgs& sa, int n_srun, int n_chains_total, int spec, hipStream_t st,
                            hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
int ls_resident_workgroups_per_cu(int CM, const Dims& d);
hipError_t launch_gene_kernel(int CM, const GeneArgs& a, int nblocks, int nchains, hipStream_t st);
hipError_t launch_step_kernel(const StepArgs& a, int nblocks, int nchains, hipStream_t st);
hipError_t launch_sum_shards_kernel(const ShardSumArgs& a, hipStream_t st);
hipError_t launch_update_kernel(const UpdateArgs& a, int nblocks, int nchains, hipStream_t st);
hipError_t launch_advi_kernel(const AdviArgs& a, int nblocks, hipStream_t st);
hipError_t launch_advi_elbo_kernel(const AdviElboArgs& a, hipStream_t st);
hipError_t launch_ppc_kernel(const PpcArgs& a, const double* T, int nblocks, hipStream_t st);   // one workgroup per cell
// one wavefront per cell (n_gen <= ppc_wave_max_draws()): the parameter table T[K][C + 1][n_draws], then the draws
size_t ppc_wave_lds_bytes(int n_gen);
int ppc_wave_max_draws();
hipError_t launch_ppc_table_kernel(const double* draws, long n_draws, const Dims& d, double tc, double* T, hipStream_t st);
hipError_t launch_ppc_wave_kernel(const PpcArgs& a, const double* T, int nblocks, hipStream_t st);
// the dispersion tables (ppcx_disp.h) of the genes in `genes` (null: genes 0 .. n_genes - 1), one workgroup per gene
hipError_t launch_disp_build_kernel(const int* counts, int G, int S, const int* genes, int n_genes, const DispFit& fit, double* table, hipStream_t st);
hipError_t launch_gather_kernel(const double* draws, long n_rows, int D, const int* cols, int n_cols, double* out, hipStream_t st);
hipError_t launch_fill_kernel(double* p, long n, double val, hipStream_t st);
hipError_t launch_xchg_abort_kernel(const XchgArgs& x, hipStream_t st);      // tells every peer that this rank has left the fit

}  // namespace ppcx
