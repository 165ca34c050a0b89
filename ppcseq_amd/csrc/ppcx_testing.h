/* ppcx_testing.h -- entry points that exist only in the TESTING build of the library (-DPPCX_TESTING:
 * tests/libppcx_testing.so, built by __graft_entry__.build() next to the product). They are test infrastructure
 * (fault injection, forcing a cell path, a stand-in provider of the nccl* entry points, kernel-level timing); the
 * shipped ppcseq_amd/libppcx.so neither exports them nor contains the code behind them. */
#ifndef PPCX_TESTING_H
#define PPCX_TESTING_H
#include "../../include/ppcx.h"
#ifdef __cplusplus
extern "C" {
#endif
/* keys: "fail_at_round" (a rank of a gene-sharded run reports a failure once it has issued that many rounds),
 * "fail_rank" (-1: every rank), "force_generic" (genes with slopes form eta per cell even in a factor design; applies to
 * models created afterwards),
 * "slope_cost_permille", "trim_slack_permille", "trim_extra_passes" (the launch plan's cost of a pass with slopes / slack of
 * a chain group's trimmed launch, per mille / passes per wavefront of such a launch beyond the fewest possible; -1 or 0:
 * built-in; plans made afterwards). */
PPCX_API int ppcx_testing_set(const char* key, long long value);
/* path of a shared object that provides ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce /
 * ncclGetErrorString instead of librccl (before the first communicator is created) */
PPCX_API int ppcx_testing_set_nccl_provider(const char* path);
enum { PPCX_BENCH_LOGLIK = 0, PPCX_BENCH_CLOSE = 1, PPCX_BENCH_LOGLIK_CLOSE = 2, PPCX_BENCH_STEP = 3, PPCX_BENCH_UPDATE = 4,
       PPCX_BENCH_STEP_REDUCE = 5, PPCX_BENCH_STEP_ADVANCE = 6, PPCX_BENCH_STEP_UPDATE = 7,
       PPCX_BENCH_GENE = 8,           /* the gene kernel of a pipelined round on a leaf command (apply + close + anticipate) */
       PPCX_BENCH_GENE_NO_PROP = 9,   /* the same without the proposal copy: what an index-based proposal store would save */
       PPCX_BENCH_GENE_NO_SPEC = 10,  /* ... without the anticipated constants */
       PPCX_BENCH_GENE_UPDATE_ONLY = 11,  /* ... the command's coordinate work only (no close) */
       PPCX_BENCH_GENE_NEW_TRANSITION = 12 };  /* the first leaf of a transition: fresh momenta for every coordinate */
PPCX_API int ppcx_testing_bench_kernel(ppcx_model* m, int which, int nchains, int warm_rounds, int reps, int n_merge,
                                       double* ms_per_launch, int* cmd_type);
/* mean microseconds per round of a chain's state machine by phase since the last call (ppcx_capi.hip) */
PPCX_API int ppcx_testing_sm_trace(double* out6);
#ifdef __cplusplus
}
#endif
#endif
