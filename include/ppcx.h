/* ppcx.h -- C ABI of the MI355X-native posterior-predictive-check engine for ppcseq's
 * negative-binomial hierarchical model.
 *
 * This is the drop-in boundary for the ONE hot path of stemangiola/ppcseq: what
 * `do_inference()` (R/utilities.R:1321-1547) obtains today from
 *     rstan::sampling(stanmodels$negBinomial_MPI, ...)            R/utilities.R:1497-1512
 *     rstan::summary(fit, "counts_rng", prob = c(p, 1-p))         R/utilities.R:685-703
 *     rstan::extract(fit, "lambda_log_param" / "sigma_raw")       R/utilities.R:738,743
 *     rstan::summary(fit, "alpha_sub_1") (slope)                  R/utilities.R:1531,1250-1263
 * whose model object is registered by src/RcppExports.cpp:15-25 / R/stanmodels.R:7-25.
 * The Stan data block (inst/stan/negBinomial_MPI.stan:142-173) carries CPU-threading packing
 * (counts_package, G_ind, symbol_end ...); this ABI takes the LOGICAL inputs instead and the
 * R shim in INTEGRATION.md undoes the packing.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types; every function returns a status
 * (0 = ok, < 0 = error class, message via ppcx_last_error()); the caller owns every buffer it
 * passes, the library never keeps a caller pointer after return; device memory is internal.
 * One call at a time per handle; different handles may be used from different threads.
 */
#ifndef PPCX_H
#define PPCX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define PPCX_API __attribute__((visibility("default")))
#else
#define PPCX_API
#endif

#define PPCX_OK 0
#define PPCX_ERR_ARG (-1)       /* invalid argument                                   */
#define PPCX_ERR_HIP (-2)       /* HIP runtime error (message has hipGetErrorString)  */
#define PPCX_ERR_INIT (-3)      /* no finite initial point after 100 attempts (Stan)  */
#define PPCX_ERR_STEPSIZE (-4)  /* step-size heuristic left (0, 1e7) (Stan)           */
#define PPCX_ERR_STALL (-5)     /* launch budget exhausted (internal guard)           */
#define PPCX_ERR_LIMIT (-6)     /* size limit of this build                           */
#define PPCX_ERR_CANCELLED (-7) /* the caller's progress callback ended the fit       */

typedef struct ppcx_model ppcx_model;
typedef struct ppcx_fit ppcx_fit;

/* ABI version: bumped whenever the layout or meaning of an argument changes. 100 = round 1; 200 = round 2 (dims[15] /
 * reals[6] / counts_rng + errbuf + errlen of ppcx_do_inference_C, ppcx_model_set_launch's second argument = workgroups);
 * 300 = this one: ppcx_do_inference_C takes the version its caller was written for as dims[0], so that a shim built for
 * another layout gets PPCX_ERR_ARG instead of reading past its arrays.                                                */
#define PPCX_VERSION 400
PPCX_API int ppcx_version(void);
PPCX_API int ppcx_device_count(void);
/* free and total memory of a device in bytes (what R/methods.R:178-195 asks the host about before keeping all draws) */
PPCX_API int ppcx_device_memory(int device, unsigned long long* free_bytes, unsigned long long* total_bytes);
PPCX_API const char* ppcx_last_error(void);

/* --- model = Stan data block (inst/stan/negBinomial_MPI.stan:142-173), logical form -------------
 * counts : G x S int32, gene-major (sample index fastest); genes 0..K-1 are the checked ones
 *          (how_many_to_check, R/utilities.R:1364-1368; G order R/utilities.R:949-952)
 * X      : S x C doubles, column-major (R model.matrix, R/utilities.R:887-900)
 * exposure_rate : S (R/utilities.R:1466-1473);  lambda_mu_mu = 5.612671 (R/methods.R:218)
 * excl   : n_excl cell ids g*S+s, 0-based (to_exclude, R/utilities.R:321-359 / .stan:105-115)      */
PPCX_API int ppcx_model_create(int device, int G, int S, int C, int K, const int32_t* counts, const double* X,
                      const double* exposure_rate, double lambda_mu_mu, int n_excl, const int32_t* excl,
                      ppcx_model** out);
PPCX_API int ppcx_model_set_exclusions(ppcx_model* m, int n_excl, const int32_t* excl);   /* pass 2 of R/methods.R:320-342 */
PPCX_API int ppcx_model_set_launch(ppcx_model* m, int lanes_per_gene, int workgroups); /* 0 = automatic; lanes: power of two <= 64 */
PPCX_API int ppcx_model_get_launch(const ppcx_model* m, int* lanes_per_gene, int* nblocks);
/* round structure of the NUTS fits of this model. pipelined: -1 = two launches per leapfrog wherever the model allows it
   (default), 0 = always the three-launch round, -2 = leave the setting as it is; stream_groups: 0 = by the number of chains
   (default), n = the chains run in n groups on their own streams, -1 = leave the setting as it is. A chain's draws do not
   depend on stream_groups. The initial values come from the environment variables PPCX_PIPELINE / PPCX_STREAM_GROUPS, read
   once when the model is created. */
PPCX_API int ppcx_model_set_rounds(ppcx_model* m, int pipelined, int stream_groups);
/* Progress of a running NUTS fit (rstan prints the chains' iterations; a fit of minutes need not be silent, and a chain that left
   the reference's 150 warm-up iterations with a very small step size -- every transition at the maximum tree depth, DESIGN.md
   section 4 -- shows up here as one chain still running long after the others): fn(user, first_chain_of_the_group,
   chains_in_the_group, chains_done, leapfrog_rounds_issued, seconds) is called from the thread that pumps the group's launches,
   at most every `every_seconds` and once when the group is done. NULL switches it off (default).
   The callback's return value is the caller's budget: nonzero ends the fit -- no further rounds are issued for the group, the
   other groups end at their next report, a gene-sharded run takes its peers along as after any failure -- and the fit call
   returns PPCX_ERR_CANCELLED with nothing to free; the model stays usable. */
typedef int (*ppcx_progress_fn)(void* user, int first_chain, int chains, int chains_done, long long rounds, double seconds);
PPCX_API int ppcx_model_set_progress(ppcx_model* m, ppcx_progress_fn fn, void* user, double every_seconds);
/* what a fit of `nchains` chains of this model runs: pipelined = 1 (two launches per leapfrog: any model with X[,1] = 1 whose
   slope columns are 0 / 1 indicators -- `~ 1`, `~ Label`, formulas of factors) or 0 (three launches: continuous covariates),
   and the number of chain groups */
PPCX_API int ppcx_model_get_rounds(const ppcx_model* m, int nchains, int* pipelined, int* stream_groups);
/* diagnostic: the log-likelihood launch planned for `nchains` chains -- lanes per gene, workgroups per chain and the
   gene-order positions bounds[0 .. 4 * workgroups_per_chain] delimiting the wavefronts' ranges (NULL to skip).
   nchains < 0: the launch of -nchains chains of one of SEVERAL chain groups, which leaves the workgroup slots it cannot use
   to the other groups' launches, at the lanes per gene in force (of the last fit, plan or ppcx_model_set_launch) */
PPCX_API int ppcx_model_get_plan(ppcx_model* m, int nchains, int* lanes_per_gene, int* workgroups_per_chain, int* bounds, int cap);
PPCX_API int ppcx_model_dim(const ppcx_model* m);          /* D = 2G + K*max(C-1,1) + 6 */
PPCX_API void ppcx_model_destroy(ppcx_model* m);

/* log_prob + gradient on the unconstrained scale, Stan parameter order (.stan:183-197), Jacobians
 * included, `~` constants dropped, neg_binomial_2_log_lpmf constants kept (what NUTS integrates).
 * u, grad: n_points x D ; lp: n_points.                                                             */
PPCX_API int ppcx_log_prob_grad(ppcx_model* m, int n_points, const double* u, double* lp, double* grad);

/* --- NUTS = rstan::sampling call of R/utilities.R:1497-1512 ------------------------------------- */
typedef struct {
  int chains, iter, warmup;          /* iter includes warmup (Stan convention)                     */
  unsigned long long seed;
  double adapt_delta;                /* 0.8  */
  int max_treedepth;                 /* 10   */
  double init_radius;                /* 2    (init = "random")                                     */
  double stepsize0;                  /* 1    */
  int init_buffer, term_buffer, window;   /* 75, 50, 25                                            */
  int chain_id_offset;               /* global id of chain 0 of this call (multi-GPU: rank*chains)  */
} ppcx_nuts_config;
PPCX_API void ppcx_nuts_config_default(ppcx_nuts_config* cfg);

PPCX_API int ppcx_fit_nuts(ppcx_model* m, const ppcx_nuts_config* cfg, ppcx_fit** out);
PPCX_API int ppcx_fit_info(const ppcx_fit* f, int* chains, int* n_keep, int* D, int* iter);
/* a fit holding draws produced elsewhere, [chains][n_keep][D] unconstrained (the host layer gathers the chains of other
 * ranks): ppcx_fit_ppc / ppcx_fit_get_columns then see the pooled posterior, as rstan::summary does over merged chains
 * (R/utilities.R:685-703, :1500-1501)                                                                                    */
PPCX_API int ppcx_fit_from_draws(ppcx_model* m, int chains, int n_keep, const double* draws, ppcx_fit** out);
/* kept draws, unconstrained, [chains][n_keep][D] */
PPCX_API int ppcx_fit_get_draws(ppcx_fit* f, double* out);
/* selected columns of the kept draws, [chains*n_keep][n_cols] */
PPCX_API int ppcx_fit_get_columns(ppcx_fit* f, int n_cols, const int32_t* cols, double* out);
/* lp: [chains][n_keep]; the rest [chains][iter] (warmup included); any pointer may be NULL */
PPCX_API int ppcx_fit_get_diagnostics(ppcx_fit* f, double* lp, double* stepsize, int32_t* treedepth,
                             int32_t* n_leapfrog, int32_t* divergent, double* accept);
/* [chains][D]: the diagonal of the inverse metric every chain ended its warm-up with, in the order of the unconstrained vector
 * (what rstan::get_adaptation_info(fit) prints as "Diagonal elements of inverse mass matrix"; the adapted step sizes are the
 * last column of ppcx_fit_get_diagnostics' stepsize). NUTS fits only.                                                        */
PPCX_API int ppcx_fit_get_inv_metric(ppcx_fit* f, double* inv_metric);
/* wall seconds of the sampling loop, gradient evaluations summed over chains, and the mean duration
 * (ms) / count of the HIP-event-timed log-likelihood-kernel launches with the chain-launches they covered    */
PPCX_API int ppcx_fit_get_timing(ppcx_fit* f, double* seconds, long long* grad_evals, double* gene_kernel_ms_mean,
                        long long* gene_kernel_samples, double* gene_kernel_chain_launches_mean);

/* mean HIP-event durations (ms) of the three kernels of a leapfrog over the timed launches, and the number of
 * (loglik, close, update) launch triples the run issued                                                   */
PPCX_API int ppcx_fit_get_kernel_times(ppcx_fit* f, double* loglik_ms, double* close_ms, double* update_ms,
                              long long* launch_triples);

/* generated quantities + credible intervals (.stan:259-266; R/utilities.R:685-703 full analysis,
 * :733-784 approximated analysis when resample != 0 and n_gen = how_many_posterior_draws).
 * ci: [K][S][4] = mean, sd, lower, upper.  counts_rng: NULL or [n_gen][K][S] int32.
 * n_gen = 0 means one predictive draw per kept posterior draw.                                      */
PPCX_API int ppcx_fit_ppc(ppcx_fit* f, double truncation_compensation, double p_lo, double p_hi,
                 unsigned long long seed, int n_gen, int resample, double* ci, int32_t* counts_rng);
/* duration (ms, HIP events) of the posterior-predictive kernel of the last ppcx_fit_ppc call and the negative-binomial
 * draws it generated (n_gen x K x S)                                                                                   */
PPCX_API int ppcx_fit_get_ppc_timing(ppcx_fit* f, double* kernel_ms, long long* nb_draws);
PPCX_API void ppcx_fit_free(ppcx_fit* f);

/* R .C() convention (all pointers, void return; character vectors arrive as char**): one do_inference() pass end to end --
 * what R/utilities.R:1482-1531 obtains from vb_iterative()/sampling(), summary(fit, "counts_rng"), extract() and
 * summary(fit, "alpha_sub_1").
 *   dims[33] = {PPCX_VERSION the caller was written for (anything else: status PPCX_ERR_ARG, nothing else is read),
 *               device, G, S, C, K, n_excl, chains, iter, warmup, n_gen, resample,
 *               approximate_posterior_inference (0 = NUTS, R/utilities.R:1497-1512; 1 = ADVI through the bounded
 *               vb_iterative retry, :1487-1494), save_generated_quantities (counts_rng is filled, :796),
 *               vb_output_samples, vb_iter (0 = 50000),
 *               n_devices (0: the one `device` above), devices[16] (the first n_devices are read)}
 *             With n_devices > 1 a NUTS pass deals its chains to those devices -- a host thread each, as rstan::sampling runs
 *             its chains on `cores` workers (R/utilities.R:1500-1501) -- and summarises the pooled chains on the first one;
 *             the result is that of one device running all the chains. A device may be named more than once.
 *   reals[6] = {lambda_mu_mu, truncation_compensation, p_lo, p_hi, seed, vb_tol_rel_obj (0 = 0.005, the value the
 *               reference hard-codes at :1492)}
 *   outputs  : ci [K*S*4] (mean, sd, .lower, .upper per checked cell), slope [K] (posterior mean of alpha_sub_1),
 *              counts_rng [n_draws*K*S] or NULL, status[1] (0 or a PPCX_ERR_* class), errbuf[0] (message, at most
 *              errlen[0] bytes including the terminator; may be NULL). No exception and no R condition crosses the ABI.   */
PPCX_API void ppcx_do_inference_C(const int* dims, const int* counts, const double* X, const double* exposure_rate,
                         const int* excl, const double* reals, double* ci, double* slope, int* counts_rng,
                         int* status, char** errbuf, const int* errlen);

/* --- ADVI = rstan::vb(model, output_samples, iter = 50000, tol_rel_obj = 0.005) through vb_iterative
 * (R/utilities.R:246-278, :1487-1494): mean-field Gaussian on the unconstrained scale, Stan defaults
 * grad_samples 1, elbo_samples 100, eval_elbo 100, adapt_iter 50. The result is a one-chain fit holding
 * output_samples draws of the approximation, so ppcx_fit_ppc / ppcx_fit_get_columns apply unchanged.          */
typedef struct {
  int output_samples, iter; double tol_rel_obj; int grad_samples, elbo_samples, eval_elbo, adapt_iter;
  unsigned long long seed; double init_radius;
} ppcx_advi_config;
PPCX_API void ppcx_advi_config_default(ppcx_advi_config* cfg);
PPCX_API int ppcx_fit_advi(ppcx_model* m, const ppcx_advi_config* cfg, ppcx_fit** out);
/* vb_iterative (R/utilities.R:246-278): retried with seed + attempt while ADVI fails to initialise or to find a step size */
PPCX_API int ppcx_fit_advi_iterative(ppcx_model* m, const ppcx_advi_config* cfg, int max_attempts, ppcx_fit** out);
PPCX_API int ppcx_fit_advi_info(const ppcx_fit* f, int* iterations, int* converged, double* elbo, double* eta);

/* --- gene shards = the reference's map_rect over gene shards (inst/stan/negBinomial_MPI.stan:226-240;
 * round-robin gene->shard assignment R/utilities.R:130-136; here contiguous gene ranges or the same round-robin deal,
 * ppcx_model_create_shard_strided). A shard model holds
 * genes [g0, g1) of a G_total-gene problem whose first K_total genes are the checked ones; the six
 * hyper-parameters are replicated and every leapfrog exchanges one vector of <= 76 partial sums (log density,
 * 6 hyper-gradient sums, kinetic energies, U-turn dot products).
 *   ppcx_fit_nuts_shards : all shards in this process on one device (testing, or splitting oversize problems)
 *   ppcx_fit_nuts_comm   : one shard per process / GPU, sums all-reduced with RCCL over xGMI              */
typedef struct ppcx_comm ppcx_comm;
PPCX_API int ppcx_model_create_shard(int device, int G_total, int S, int C, int K_total, int g0, int g1,
                            const int32_t* counts_shard, const double* X, const double* exposure_rate,
                            double lambda_mu_mu, int n_excl, const int32_t* excl_local, ppcx_model** out);
/* ... or every gene_stride-th gene from g0 on: genes g0, g0 + gene_stride, ..., n_genes of them (counts_shard holds their rows in
   that order). Shard r of N with g0 = r, gene_stride = N is the reference's round-robin deal (R/utilities.R:125-136): every shard
   gets its share of the K_total checked genes, which come first in the whole problem -- and with them an equal share of the work
   (a contiguous split gives the first shard every gene with a slope). Philox streams are addressed by the coordinate's index in
   the whole problem either way: a sharded run draws what the unsharded run draws. */
PPCX_API int ppcx_model_create_shard_strided(int device, int G_total, int S, int C, int K_total, int g0, int gene_stride, int n_genes,
                            const int32_t* counts_shard, const double* X, const double* exposure_rate,
                            double lambda_mu_mu, int n_excl, const int32_t* excl_local, ppcx_model** out);
PPCX_API int ppcx_fit_nuts_shards(ppcx_model** shards, int n_shards, const ppcx_nuts_config* cfg, ppcx_fit** fits);
PPCX_API int ppcx_comm_unique_id(char* out128);               /* rank 0 creates it, the host layer broadcasts it */
PPCX_API int ppcx_comm_create(int device, int nranks, int rank, const char* id128, ppcx_comm** out);
PPCX_API void ppcx_comm_destroy(ppcx_comm* c);
PPCX_API int ppcx_fit_nuts_comm(ppcx_model* shard, const ppcx_nuts_config* cfg, ppcx_comm* comm, ppcx_fit** out);

/* --- gene shards with a DIRECT exchange (the default between GPUs; exercised so far between processes that share one GPU and
 * between host threads -- no multi-GPU box was available to any round -- so the host layer, distributed.do_inference_shards, falls
 * back to ppcx_fit_nuts_comm on ALL ranks when its set-up fails on any): no collective library call per leapfrog. Every rank's
 * receive buffer (uncached device memory) is mapped into every rank -- hipIpc handles between processes -- and the chains'
 * state machines, which run inside the merged launch of a pipelined round beside the log-likelihood workgroups, store their
 * <= 76 partial sums into the peers' buffers over xGMI, then a sequence number, and wait for the peers' (rank-order sum:
 * identical bits, hence identical decisions, on every rank). A peer that leaves the fit or does not arrive within the timeout
 * fails the fit with PPCX_ERR_STALL on every rank. Needs a model that runs pipelined rounds (ppcx_model_get_rounds); the RCCL
 * path above serves the others.
 *   rank r: ppcx_xchg_create -> ppcx_xchg_handle (64 bytes) -> [host layer all-gathers the handles] -> ppcx_xchg_connect
 *           -> ppcx_fit_nuts_xchg (any number of fits, the same sequence on every rank) -> ppcx_xchg_destroy              */
typedef struct ppcx_xchg ppcx_xchg;
PPCX_API int ppcx_xchg_create(int device, int nranks, int rank, int max_chains, ppcx_xchg** out);
PPCX_API int ppcx_xchg_handle(ppcx_xchg* x, char* out64);
PPCX_API int ppcx_xchg_connect(ppcx_xchg* x, const char* handles /* nranks x 64 bytes, rank order */);
PPCX_API int ppcx_xchg_connect_local(ppcx_xchg** group, int n);   /* all ranks in this process (a host thread each) */
PPCX_API int ppcx_xchg_set_timeout(ppcx_xchg* x, double seconds); /* default 20 s */
PPCX_API void ppcx_xchg_destroy(ppcx_xchg* x);
PPCX_API int ppcx_fit_nuts_xchg(ppcx_model* shard, const ppcx_nuts_config* cfg, ppcx_xchg* x, ppcx_fit** out);
/* mean time (us) a chain's state machine waited for its peers per exchange, and the number of exchanges of the fit */
PPCX_API int ppcx_fit_get_xchg_timing(ppcx_fit* f, double* wait_us_per_exchange, long long* exchanges);

/* What the ranks of a gene-sharded run conclude at a poll from the max-reduced vector
 * [rounds, -rounds, chains done, -chains done, -(error status)] and their own status: PPCX_OK, the peer's / own error
 * class, or PPCX_ERR_STALL when the ranks disagree. Pure host logic, exported so that it can be tested without two GPUs. */
PPCX_API int ppcx_guard_decision(const double* reduced5, int local_status);

#ifdef __cplusplus
}
#endif
#endif
