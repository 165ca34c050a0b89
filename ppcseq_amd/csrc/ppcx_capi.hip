// ppcx_capi.hip -- host side of the C ABI declared in include/ppcx.h.
// Owns device memory, chooses the launch geometry, pumps the (gene kernel, chain kernel) launch pairs
// and copies results back. No torch types, no exceptions across the boundary.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string>
#include <thread>
#include <atomic>
#include <vector>
#include "../../include/ppcx.h"
#include "ppcx_kernels.h"

using namespace ppcx;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(PPCX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

// Test hooks exist only in the testing build (-DPPCX_TESTING: tests/libppcx_testing.so, built by __graft_entry__.build()
// for tests/ and scripts/); the shipped library has none of them and reads no environment variable after model creation.
#ifdef PPCX_TESTING
#include "ppcx_testing.h"
struct TestHooks {
  long long fail_at_round = 0; int fail_rank = -1;   // fault injection into the pump of a gene-sharded run
  int force_generic = 0;                             // every gene with slopes takes the per-cell-eta path
  int slope_cost_permille = 0;                       // plan: cost of a pass with slope genes relative to a plain one (0: built-in)
  int trim_slack_permille = -1;                      // plan: slack of a chain group's trimmed launch (-1: built-in)
  int trim_extra_passes = 0;                         // plan: passes per wavefront of a trimmed launch beyond the fewest possible
  std::string rccl_lib;                              // another provider of the nccl* entry points (tests/loopback)
};
static TestHooks g_test;
#endif

#ifdef PPCX_TRACE
static unsigned long long* g_trace_dev = nullptr;    // development builds: stamps of the log-likelihood passes (ppcx_kernels.h)
#endif
struct ppcx_model {
  int device;
  Dims d;
  int CM, L = 64;
  int L_override = 0, wgs_override = 0;        // ppcx_model_set_launch
  int n_cu = 256, wgs_per_cu = 4;              // resident workgroups of the log-likelihood kernel = n_cu * wgs_per_cu
  int ls_wgs_per_cu = 0;                       // the same for the merged launch of a pipelined round (0: cannot run)
  int nblocks_chosen = 0;                      // workgroups of the last planned launch (what ppcx_model_get_launch reports)
  // round structure of a NUTS fit (ppcx_model_set_rounds; initial values from PPCX_PIPELINE / PPCX_STREAM_GROUPS, read
  // once when the model is created): pipelined -1 = where it applies, 0 = never; stream_groups 0 = by the number of chains
  int opt_pipelined = -1, opt_stream_groups = 0;
  // progress reports of a running fit (ppcx_model_set_progress): the pump calls it at a poll, at most every progress_every s
  ppcx_progress_fn progress = nullptr; void* progress_user = nullptr; double progress_every = 1.0;
  // gene order of the log-likelihood launch (upload_counts): per position, whether the gene has slopes -- what a pass of a
  // wavefront costs (plan_launch)
  std::vector<char> pos_slope;
  struct Plan { int nbpc = 0; int* d_bounds = nullptr; };
  std::map<std::pair<int, int>, Plan> plans;   // (chains in the launch, resident workgroups it may use) -> ranges
  std::mutex plan_mutex;
  std::vector<int32_t> counts_host;            // original counts (exclusions are re-applied on a copy)
  std::vector<double> X_host, expo_host;
  int* d_counts = nullptr;
  double *d_E = nullptr, *d_expo = nullptr, *d_X = nullptr, *d_Sy = nullptr, *d_SyE = nullptr, *d_SyX = nullptr, *d_SX = nullptr, *d_ncell = nullptr, *d_Lg1 = nullptr;
  double* d_disp = nullptr;                    // [G][kDispGeneDoubles] the genes' dispersion tables (ppcx_disp.h)
  unsigned char* d_gflags = nullptr;           // [G] bit 0: the gene has excluded cells
  DispFit fit;                                 // nodes and transforms of the table build
  double *d_logtab = nullptr, *d_wintab = nullptr;
  double e_min = 1.0, e_max = 1.0;             // smallest and largest exp(exposure_s)
  int* d_order = nullptr;        // gene_order: position in the log-likelihood kernel's launch -> gene
  hipStream_t stream = nullptr;
  int live_fits = 0;             // fits that still point at this model: ppcx_model_destroy defers until the last one is freed
  bool destroy_requested = false;
};

struct ppcx_fit {
  ppcx_model* m;
  NutsConfig cfg;
  int chains, n_keep, iter;
  double* d_draws = nullptr;                   // [chains][n_keep][D]
  double *d_lp = nullptr, *d_stepsize = nullptr, *d_accept = nullptr;
  int *d_treedepth = nullptr, *d_nleap = nullptr, *d_div = nullptr;
  double seconds = 0; long long grad_evals = 0;
  double kA_ms_mean = 0; long long kA_samples = 0; double kA_chain_launches_mean = 0;
  double kC_ms_mean = 0, kU_ms_mean = 0; long long launch_triples = 0;
  double advi_elbo = 0, advi_eta = 0; int advi_converged = 0;
  double ppc_ms = 0; long long ppc_draws = 0;  // last ppcx_fit_ppc: kernel time (HIP events) and NB draws generated
  long long xchg_ticks = 0, xchg_count = 0;    // direct exchange: 100 MHz ticks the chains' state machines waited for peers, exchanges
  std::vector<double> inv_metric;              // [chains][D] diagonal of the adapted inverse metric (host; ppcx_fit_get_inv_metric)
};

extern "C" int ppcx_version(void) { return PPCX_VERSION; }
extern "C" int ppcx_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
extern "C" const char* ppcx_last_error(void) { return g_err.c_str(); }
extern "C" int ppcx_device_memory(int device, unsigned long long* free_bytes, unsigned long long* total_bytes) {
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(PPCX_ERR_ARG, "no such HIP device");
  HIPCHK(hipSetDevice(device));
  size_t f = 0, t = 0;
  HIPCHK(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (unsigned long long)f;
  if (total_bytes) *total_bytes = (unsigned long long)t;
  return PPCX_OK;
}

// Launch geometry of the log-likelihood kernel. The launch is resident: n_res = (workgroups the chip holds at once)
// divided among the chains of the launch, four wavefronts each; wavefront j of a chain walks the gene positions
// bounds[j] .. bounds[j + 1] of the gene order, 64 / L genes per pass.
//   pass cost (in row-sweep iterations of a lane): 5.8 for loading and closing the genes -- 129 vector instructions against
//   43.7 per iteration (SQ_INSTS_VALU at S = 40, 104, 200) are 3 iterations' worth of issue, but with the dependent loads at
//   the start of a pass they weigh like 5 to 6 in time (L = 8 against 16 at 4 chains, 4 against 8 at 3 chains: measured), ceil(S / L) for the row sweep
//   (a little more on the two-group path, 2.5 x on the generic path with an exp per cell), 0.75 per iteration of the
//   low-count loop, which lasts as long as the longest list of the pass needs -- from the kernel's instruction counts
//   (profiles/, SQ_INSTS_VALU: ~50 per row-sweep cell, ~100 per list cell, ~250 per pass).
//   L: minimise passes x pass cost for the full chain count (chosen once per fit: a gene's sums depend on L).
//   bounds: contiguous ranges of whole passes of (nearly) equal cost -- the boundary of wavefront j is where the running
//   cost crosses j / (wavefronts per chain) of the total; recomputed when chains finish and the others get their slots.
static double pass_cost(const ppcx_model* m, int L, int p, int n) {
  const int S = m->d.S;
  bool slope = false;
  for (int i = p; i < p + n; ++i) slope = slope || m->pos_slope[i];
  // a pass of genes with slopes: 1.15 x a plain one in a two-group design. With more indicator columns it costs 1.43 x (C = 3, all
  // genes with slopes against none: 83.9 vs 58.4 us per launch) -- but weighting it so makes the launch SLOWER (K = 1000 of 20 000
  // genes: weight 1.0 -> 63.4 us, 1.45 -> 67.5, 2.0 -> 72.8; scripts/gpu_factor_k.py): the four wavefronts of a SIMD come from four
  // workgroups and share its fp64 unit, so what counts is the SIMD's total, and a wavefront with fewer, dearer passes does not
  // relieve the three it shares the SIMD with, while the passes it gives up push the others over their share
  double slope_w = m->d.C > 2 ? 1.0 : 1.15;
#ifdef PPCX_TESTING
  if (g_test.slope_cost_permille > 0) slope_w = 1e-3 * g_test.slope_cost_permille;
#endif
  // ... and with an exp per cell (a continuous covariate: sweep_cells MODE 3; without the column of ones: generic_cells)
  double lin_w = 2.5;
#ifdef PPCX_TESTING
  if (g_test.slope_cost_permille > 0) lin_w = 1e-3 * g_test.slope_cost_permille;
#endif
  const double sweep = (double)((S + L - 1) / L) * (!m->d.x0_is_one ? 2.5 : (slope && !m->d.x1_binary ? lin_w : (slope ? slope_w : 1.0)));
  return 5.8 + sweep;
}
// reserve: workgroups of the same launch that are not log-likelihood workgroups (the state machines of a pipelined
// round's merged launch; they are dispatched first and resident for a part of the launch)
static int resident_workgroups(const ppcx_model* m, int reserve) {
  int n = m->wgs_override > 0 ? m->wgs_override : m->n_cu * (reserve > 0 ? m->ls_wgs_per_cu : m->wgs_per_cu);
  n -= reserve;
  return n < 1 ? 1 : n;
}
// workgroups per chain: all of the resident ones, a multiple of 8 (the kernel deals runs of 8 to the XCDs), not more
// than there are passes to hand out
constexpr double kTrimSlack = 0.0;
static int workgroups_per_chain(const ppcx_model* m, int L, int nch, int n_res, bool whole_runs = true, bool trim = false) {
  int nbpc = n_res / (nch < 1 ? 1 : nch);
  if (nbpc >= 8 && whole_runs) nbpc = nbpc / 8 * 8;
  if (nbpc < 1) nbpc = 1;
  const int gpw = 64 / L, npass = (m->d.G + gpw - 1) / gpw;
  if (nbpc > (npass + 3) / 4) nbpc = (npass + 3) / 4;
  // trim (a launch of one of several chain groups, whose launches share the chip): ... and not more than it takes to give every
  // wavefront the passes of the busiest one. A launch lasts as long as its wavefront with the most passes (plan_launch); with
  // 2500 passes for 1352 wavefronts (cfg3, a chain group of three) that is two, for a sixth of the wavefronts one -- scattered
  // by the cost balancing, so that nearly every workgroup keeps its slot for the whole launch with a wavefront less to run.
  // 1250 wavefronts with two passes each last as long and leave 8 % of the chip's slots (39 % in a launch of two chains) to the
  // launch of another chain group, which otherwise waits for them: cfg3, 8 chains in three groups, 2.75 -> 2.53 s per fit
  // (round 4). No slack beyond the whole run of 8 workgroups pays (kTrimSlack: 0 / 1.5 / 3 / 6 / 12 % -> 2.55 / 2.55 / 2.58 / 2.62 /
  // 2.66 s), nor a pass more per wavefront on fewer workgroups (2.82 s). Alone on the chip the trimmed launch is the slower one
  // (nobody takes the slots, and the balancing has less to work with: 3 chains 27.9 -> 30.3 us, 8 chains on one stream
  // 2.95 -> 3.31 s per fit), so a fit on one stream keeps every resident workgroup.
  // A gene's sums depend on the lanes per gene only: the chains do not change.
  if (trim) {
    double slack = kTrimSlack;
#ifdef PPCX_TESTING
    if (g_test.trim_slack_permille >= 0) slack = 1e-3 * g_test.trim_slack_permille;
#endif
    int maxp = (npass + 4 * nbpc - 1) / (4 * nbpc);
#ifdef PPCX_TESTING
    maxp += g_test.trim_extra_passes;
#endif
    int nb2 = (int)ceil((double)npass * (1.0 + slack) / (4.0 * maxp));
    if (nb2 >= 8 && whole_runs) nb2 = (nb2 + 7) / 8 * 8;
    if (nb2 < nbpc) nbpc = nb2;
  }
  return nbpc;
}
// chain groups on their own streams (ppcx_fit_nuts): the default for a fit of `nch` chains
static int default_stream_groups(int nch) { return nch >= 8 ? 3 : (nch >= 4 ? 2 : 1); }
// chains in a launch of a fit of `nch` chains: the fit runs them in groups on their own streams (ppcx_model_set_rounds, default
// default_stream_groups), every launch holds one group. A function of the fit's chain count and the model's setting only, so that a
// chain's lanes per gene -- hence its draws -- do not depend on anything else.
static int fit_launch_chains(const ppcx_model* m, int nch) {
  int g = default_stream_groups(nch);
  if (m->opt_stream_groups >= 1) g = m->opt_stream_groups < nch ? m->opt_stream_groups : nch;
  return (nch + g - 1) / g;
}
static void choose_launch(ppcx_model* m, int nchains) {
  const int G = m->d.G, S = m->d.S;
  // lanes per gene for launches of `nchains` chains: passes of the busiest wavefront x pass cost, smallest first. A pass costs
  // what its sweep costs plus 8 cell iterations' worth of loading and closing its genes (round 5: SQ_INSTS_VALU of a wavefront
  // = 22 per cell iteration + 165 per pass; rounds 2-4, with a cell of 41 instructions: 5.8).
  // Round 5 dropped two things. (a) The condition that a choice leave no wavefront slot of the chip without a pass (>= 1.2 passes
  // per slot), which kept few large passes from looking cheapest at small numbers of chains: with the cheaper cell the pass
  // overhead decides, and the measured order is the model's -- cfg3, kernel level, one chain: L = 8 9.4 us (2500 wavefronts of one
  // pass), 4 10.9, 16 11.4, 32 13.5; two chains: L = 4 13.5, 8 15.9, 16 17.6, 32 20.5; three: L = 4 15.8, 8 17.6; eight: L = 8
  // 38.6, 4 39.0. (b) Choosing L for all the chains of a fit when the fit runs them in groups (callers pass the chains of a
  // group's launch, fit_launch_chains): whole cfg3 fits by L = 4 / 8 / 16 (scripts/gpu_lanes_fits.py): 1 chain 0.920 / 0.918 /
  // 0.950 s, 2 chains 1.008 / 1.081 / 1.114, 3 chains 1.152 / 1.272 / 1.495, 4 chains (two groups) 1.256 / 1.281 / 1.358, 8 chains
  // (three groups) 1.634 / 1.756 / 1.978. (Round 3 had measured the same 9 % at 8 chains and not taken them: two of 61 fits with
  // L = 4 ended warm-up with a chain at tree depth 10, none of 233 with L = 8. Round 5 found where such chains come from --
  // log(-sigma_slope) wandering through its flat tail during the one adaptation window, DESIGN.md section 4 -- and lanes per gene
  // have no part in it beyond changing every chain's rounding, like a seed.)
  int bestL = 64; double best = 1e300;
  for (int L = 64; L >= 1; L >>= 1) {                      // (ties go to the larger L: shorter dependent chains per lane)
    const int gpw = 64 / L;
    // (L = 1, 2: a lane's four counts of a trip lie 4 L bytes apart, every request touches 64 / L times the lines: +20 %)
    const double npass = ceil((double)G / gpw), c_pass = (8.0 + (double)((S + L - 1) / L)) * (L <= 2 ? 1.2 : 1.0);
    const int wpc = 4 * workgroups_per_chain(m, L, nchains, resident_workgroups(m, 0));
    const double t = ceil(npass / wpc) * c_pass;           // passes of the busiest wavefront x pass cost
    if (t < best) { best = t; bestL = L; }
  }
  const int L = m->L_override > 0 ? m->L_override : bestL;
  if (L != m->L) {
    std::lock_guard<std::mutex> lk(m->plan_mutex);
    for (auto& kv : m->plans) (void)hipFree(kv.second.d_bounds);
    m->plans.clear();
    m->L = L;
  }
  m->nblocks_chosen = workgroups_per_chain(m, m->L, nchains, resident_workgroups(m, 0)) * nchains;
}
static void drop_plans(ppcx_model* m) {
  std::lock_guard<std::mutex> lk(m->plan_mutex);
  for (auto& kv : m->plans) (void)hipFree(kv.second.d_bounds);
  m->plans.clear();
}
// the ranges for a launch of `nch` chains beside `reserve` other workgroups
static int plan_launch(ppcx_model* m, int nch, int reserve, ppcx_model::Plan* out, bool trim = false) {
  std::lock_guard<std::mutex> lk(m->plan_mutex);
  const int n_res = resident_workgroups(m, reserve);
  const auto key = std::make_pair(nch, trim ? -n_res : n_res);   // (a trimmed plan under its own key)
  auto it = m->plans.find(key);
  if (it != m->plans.end()) { *out = it->second; return PPCX_OK; }
  const int G = m->d.G, L = m->L, gpw = 64 / L;
  // beside state machines (reserve > 0) the last run of the launch may be partial: their slots and the range blocks
  // together fill the chip
  const int nbpc = workgroups_per_chain(m, L, nch, n_res, reserve == 0, trim), wpc = 4 * nbpc, npass = (G + gpw - 1) / gpw;
  std::vector<double> cost(npass);
  double total = 0;
  for (int k = 0; k < npass; ++k) {
    const int p = k * gpw;
    cost[k] = pass_cost(m, L, p, G - p < gpw ? G - p : gpw);
    total += cost[k];
  }
  // boundary j where the running cost crosses j / wpc of the total, at the nearest whole pass
  std::vector<int> bounds(wpc + 1, G);
  bounds[0] = 0;
  {
    int k = 0; double cum = 0.0;
    for (int j = 1; j < wpc; ++j) {
      const double target = total * (double)j / (double)wpc;
      while (k < npass && cum + 0.5 * cost[k] < target) cum += cost[k++];
      bounds[j] = k * gpw < G ? k * gpw : G;
    }
  }
  // no wavefront gets more than its share of passes rounded up: the four wavefronts of a SIMD take turns on its fp64
  // unit, so the launch lasts as long as the SIMD with the most passes (measured: the 8 SIMDs whose wavefronts all had
  // one pass more than the rest finished 8 us after the others, profiles/r02_wave_trace.txt)
  {
    const long pmax = ((long)(npass + wpc - 1) / wpc) * gpw;
    for (int j = 0; j < wpc; ++j) if (bounds[j + 1] - bounds[j] > pmax) bounds[j + 1] = (int)(bounds[j] + pmax);
    bounds[wpc] = G;
    for (int j = wpc - 1; j >= 0; --j) if (bounds[j + 1] - bounds[j] > pmax) bounds[j] = (int)(bounds[j + 1] - pmax);
  }
  ppcx_model::Plan pl; pl.nbpc = nbpc;
  HIPCHK(hipMalloc(&pl.d_bounds, sizeof(int) * (size_t)(wpc + 1)));
  HIPCHK(hipMemcpy(pl.d_bounds, bounds.data(), sizeof(int) * (size_t)(wpc + 1), hipMemcpyHostToDevice));
  m->plans[key] = pl;
  *out = pl;
  return PPCX_OK;
}

static int upload_counts(ppcx_model* m, int n_excl, const int32_t* excl) {
  const int G = m->d.G, S = m->d.S, C = m->d.C;
  std::vector<int32_t> cnt(m->counts_host);
  for (int e = 0; e < n_excl; ++e) {
    if (excl[e] < 0 || excl[e] >= G * S) return fail(PPCX_ERR_ARG, "excluded cell id out of range");
    cnt[excl[e]] = -1;
  }
  std::vector<double> Sy(G, 0.0), SyE(G, 0.0), SyX((size_t)C * G, 0.0), SX((size_t)C * G, 0.0), ncell(G, 0.0), Lg1(G, 0.0);
  std::vector<unsigned char> gflags(G, 0);
  for (int g = 0; g < G; ++g) {
    double sy = 0, sye = 0, nc = 0, lg1 = 0;
    for (int s = 0; s < S; ++s) {
      const int y = cnt[(size_t)g * S + s];
      if (y < 0) { gflags[g] |= 1; continue; }
      sy += y; sye += (double)y * m->expo_host[s]; nc += 1.0; lg1 += lgamma((double)y + 1.0);
      for (int c = 0; c < C; ++c) { SyX[(size_t)c * G + g] += (double)y * m->X_host[(size_t)c * S + s]; SX[(size_t)c * G + g] += m->X_host[(size_t)c * S + s]; }
    }
    Sy[g] = sy; SyE[g] = sye; ncell[g] = nc; Lg1[g] = lg1;
  }
  HIPCHK(hipMemcpy(m->d_gflags, gflags.data(), (size_t)G, hipMemcpyHostToDevice));
  // gene_order: a wavefront holds several genes and runs the cell path its most demanding gene needs -- the slope columns if
  // one of them has slopes, the test for excluded cells if one of them has such cells. So neighbours in the launch should be
  // alike: genes with slopes first, among them and among the plain ones those with excluded cells first. Every count costs the
  // same since round 5 (ppcx_disp.h), so nothing else distinguishes two genes.
  {
    std::vector<int> ord(G);
    for (int g = 0; g < G; ++g) ord[g] = g;
    const int K = m->d.K;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
      const bool sa = a < K && C >= 2, sb = b < K && C >= 2;
      if (sa != sb) return sa;
      return gflags[a] > gflags[b];
    });
    // ... and the genes with slopes are dealt over the whole order in blocks of kOrderBlock positions (whole passes for every L
    // >= 4), evenly spaced among the plain genes' blocks: a wavefront's range is a contiguous piece of the order, a workgroup's
    // four wavefronts hold neighbouring ranges and a CU four workgroups in dispatch order, so with the slope genes at the front
    // a few CUs ran nothing but the dearer passes -- `~ group + age` at BASELINE size: 61 us per 8-chain launch whatever weight
    // the plan gave those passes, 46 us dealt out, against 40 us for the sum of the work; the order does not enter a gene's sums.
    // Only where those passes are much dearer (an exp per cell): with indicator columns they cost 1.15 x a plain one, and dealing
    // them out makes a chain group's launch of three chains 5 % slower (17.6 -> 18.6 us: shorter runs of alike passes).
    {
      int ns = 0;
      while (ns < G && ord[ns] < K && C >= 2) ++ns;         // genes with slopes: the first ns positions
      constexpr int kOrderBlock = 16;
      const int nbs = (ns + kOrderBlock - 1) / kOrderBlock, nb = (G + kOrderBlock - 1) / kOrderBlock;
      const bool dear_slopes = m->d.x0_is_one && !m->d.x1_binary;      // an exp per cell (sweep_cells MODE 3): 2.4 x a plain pass
      if (ns > 0 && ns < G && nbs < nb && dear_slopes) {
        std::vector<int> mixed; mixed.reserve(G);
        int is = 0, ip = ns, sb = 0;                       // next slope gene, next plain gene, slope blocks placed
        for (int b = 0; b < nb; ++b) {
          const bool slope_block = sb < nbs && b == (int)((long long)sb * nb / nbs);
          if (slope_block && is < ns) { for (int k = 0; k < kOrderBlock && is < ns; ++k) mixed.push_back(ord[is++]); ++sb; }
          else { for (int k = 0; k < kOrderBlock && ip < G; ++k) mixed.push_back(ord[ip++]); }
        }
        while (is < ns) mixed.push_back(ord[is++]);
        while (ip < G) mixed.push_back(ord[ip++]);
        ord.swap(mixed);
      }
    }
    HIPCHK(hipMemcpy(m->d_order, ord.data(), sizeof(int) * (size_t)G, hipMemcpyHostToDevice));
    m->pos_slope.resize(G);
    for (int p = 0; p < G; ++p) m->pos_slope[p] = ord[p] < K && C >= 2;
    drop_plans(m);
  }
  HIPCHK(hipMemcpy(m->d_counts, cnt.data(), sizeof(int32_t) * cnt.size(), hipMemcpyHostToDevice));
  // the dispersion tables of all genes from the counts now on the device (a few milliseconds; excluded cells are left out of them)
  {
    const hipError_t e = launch_disp_build_kernel(m->d_counts, G, S, nullptr, G, m->fit, m->d_disp, m->stream);
    if (e != hipSuccess) return fail(PPCX_ERR_HIP, std::string("dispersion-table kernel: ") + hipGetErrorString(e));
  }
  HIPCHK(hipMemcpy(m->d_Sy, Sy.data(), sizeof(double) * G, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(m->d_SyE, SyE.data(), sizeof(double) * G, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(m->d_SyX, SyX.data(), sizeof(double) * SyX.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(m->d_SX, SX.data(), sizeof(double) * SX.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(m->d_ncell, ncell.data(), sizeof(double) * G, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(m->d_Lg1, Lg1.data(), sizeof(double) * G, hipMemcpyHostToDevice));
  HIPCHK(hipStreamSynchronize(m->stream));
  return PPCX_OK;
}

extern "C" int ppcx_model_create(int device, int G, int S, int C, int K, const int32_t* counts, const double* X,
                                 const double* exposure, double lambda_mu_mu, int n_excl, const int32_t* excl,
                                 ppcx_model** out) {
  if (!out) return fail(PPCX_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (G < 1 || S < 1 || C < 1 || K < 0 || K > G) return fail(PPCX_ERR_ARG, "need G>=1, S>=1, C>=1, 0<=K<=G");
  if (C > kMaxC) return fail(PPCX_ERR_LIMIT, "C exceeds the 16 design columns this build supports");
  if ((long long)G * S > 2000000000LL) return fail(PPCX_ERR_LIMIT, "G*S exceeds int32 cell ids");
  if (!counts || !X || !exposure || (n_excl > 0 && !excl)) return fail(PPCX_ERR_ARG, "NULL input buffer");
  for (long long i = 0; i < (long long)G * S; ++i) if (counts[i] < 0) return fail(PPCX_ERR_ARG, "negative count");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(PPCX_ERR_ARG, "no such HIP device");
  HIPCHK(hipSetDevice(device));
  ppcx_model* m = new ppcx_model();
  m->device = device;
  m->d = make_dims(G, S, C, K, lambda_mu_mu);
  m->CM = C <= 2 ? 2 : (C <= 4 ? 4 : (C <= 8 ? 8 : 16));
  {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    m->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  m->counts_host.assign(counts, counts + (size_t)G * S);
  m->X_host.assign(X, X + (size_t)S * C);
  m->expo_host.assign(exposure, exposure + S);
  int x0 = 1;
  for (int s = 0; s < S; ++s) if (X[s] != 1.0) x0 = 0;
  m->d.x0_is_one = x0;
  int x1b = (C >= 2);                          // factor design: every slope column is a 0 / 1 indicator (C == 2: two groups)
  for (size_t i = (size_t)S; i < (size_t)S * C && x1b; ++i) if (X[i] != 0.0 && X[i] != 1.0) x1b = 0;
#ifdef PPCX_TESTING
  if (g_test.force_generic) x1b = 0;
#endif
  m->d.x1_binary = x1b;
  m->d.raw_consts = (!x0 || (C >= 2 && K > 0 && !x1b)) ? 1 : 0;
  // more than 8 design columns: the instantiation for factor designs only (a twelve-level factor, `~ a + b + c` of factors:
  // model.matrix columns that are all indicators, R/utilities.R:887-900); a continuous covariate among more than 8 columns,
  // or such a design without the column of ones, has no instantiation in this build
  if (C > 8 && (!x0 || (K > 0 && !x1b))) {
    delete m;
    return fail(PPCX_ERR_LIMIT, "more than 8 design columns are supported for designs of a column of ones and 0 / 1 indicator columns only");
  }
  // the two tuning knobs of a fit's round structure, read from the environment HERE and nowhere else
  if (const char* e = getenv("PPCX_PIPELINE")) if (atoi(e) == 0) m->opt_pipelined = 0;
  if (const char* e = getenv("PPCX_STREAM_GROUPS")) { const int v = atoi(e); if (v >= 1) m->opt_stream_groups = v; }
  // the log-likelihood kernel stages its tables (20 KB) and the per-sample constants in LDS: exp(exposure) and the design's slope
  // columns -- S * C doubles; S * (2 + C) without the column of ones
  if (loglik_lds_bytes(m->d) > 160u * 1024u) {
    delete m;
    return fail(PPCX_ERR_LIMIT, "the per-sample constants (S * C doubles; S * (2 + C) for a design without a column of ones) do not fit the 160 KB of LDS of a compute unit beside the 20 KB of tables");
  }
  m->wgs_per_cu = loglik_resident_workgroups_per_cu(m->CM, m->d);      // of the instantiation this model runs
  if (m->wgs_per_cu < 1) { delete m; return fail(PPCX_ERR_LIMIT, "the log-likelihood kernel cannot be resident with S * (2 + C) doubles of per-sample constants in LDS"); }
  m->ls_wgs_per_cu = ls_resident_workgroups_per_cu(m->CM, m->d);
  std::vector<double> E(S);
  for (int s = 0; s < S; ++s) E[s] = exp(exposure[s]);
#define MCHK(expr) do { int rc_ = (expr); if (rc_ != PPCX_OK) { ppcx_model_destroy(m); return rc_; } } while (0)
#define MHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ppcx_model_destroy(m); return fail(PPCX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
  MHIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  MHIP(hipMalloc(&m->d_counts, sizeof(int32_t) * ((size_t)G * S + 512)));   // + 512: the row sweep requests counts up to two trips of 4 x 64 lanes past the end
  MHIP(hipMemset(m->d_counts, 0, sizeof(int32_t) * ((size_t)G * S + 512)));
  MHIP(hipMalloc(&m->d_E, sizeof(double) * S));
  MHIP(hipMalloc(&m->d_expo, sizeof(double) * S));
  MHIP(hipMalloc(&m->d_X, sizeof(double) * (size_t)S * C));
  MHIP(hipMalloc(&m->d_Sy, sizeof(double) * G));
  MHIP(hipMalloc(&m->d_SyE, sizeof(double) * G));
  MHIP(hipMalloc(&m->d_SyX, sizeof(double) * (size_t)C * G));
  MHIP(hipMalloc(&m->d_SX, sizeof(double) * (size_t)C * G));
  MHIP(hipMalloc(&m->d_ncell, sizeof(double) * G));
  MHIP(hipMalloc(&m->d_disp, sizeof(double) * (size_t)G * kDispGeneDoubles));
  MHIP(hipMalloc(&m->d_gflags, (size_t)G));
  disp_fit_init(m->fit);
  MHIP(hipMalloc(&m->d_Lg1, sizeof(double) * G));
  MHIP(hipMalloc(&m->d_logtab, sizeof(double) * 2 * kLogTabSize));
  { double tab[2 * kLogTabSize]; fill_log_table(tab); MHIP(hipMemcpy(m->d_logtab, tab, sizeof(tab), hipMemcpyHostToDevice)); }
  MHIP(hipMalloc(&m->d_wintab, sizeof(double) * 2 * kWinTabSize));
  { std::vector<double> wt(2 * kWinTabSize); fill_window_log_table(wt.data()); MHIP(hipMemcpy(m->d_wintab, wt.data(), sizeof(double) * wt.size(), hipMemcpyHostToDevice)); }
  m->e_min = m->e_max = E[0];
  for (int s = 1; s < S; ++s) { if (E[s] < m->e_min) m->e_min = E[s]; if (E[s] > m->e_max) m->e_max = E[s]; }
  MHIP(hipMalloc(&m->d_order, sizeof(int) * (size_t)G));
  MHIP(hipMemcpy(m->d_E, E.data(), sizeof(double) * S, hipMemcpyHostToDevice));
  MHIP(hipMemcpy(m->d_expo, exposure, sizeof(double) * S, hipMemcpyHostToDevice));
  MHIP(hipMemcpy(m->d_X, X, sizeof(double) * (size_t)S * C, hipMemcpyHostToDevice));
  MCHK(upload_counts(m, n_excl, excl));
  choose_launch(m, 1);
  *out = m;
  return PPCX_OK;
}

extern "C" int ppcx_model_set_exclusions(ppcx_model* m, int n_excl, const int32_t* excl) {
  if (!m || (n_excl > 0 && !excl) || n_excl < 0) return fail(PPCX_ERR_ARG, "bad arguments");
  HIPCHK(hipSetDevice(m->device));
  return upload_counts(m, n_excl, excl);
}
extern "C" int ppcx_model_set_launch(ppcx_model* m, int lanes_per_gene, int workgroups) {
  if (!m) return fail(PPCX_ERR_ARG, "model is NULL");
  if (lanes_per_gene != 0 && (lanes_per_gene < 1 || lanes_per_gene > 64 || (lanes_per_gene & (lanes_per_gene - 1))))
    return fail(PPCX_ERR_ARG, "lanes_per_gene must be 0 or a power of two <= 64");
  if (workgroups < 0 || workgroups > (1 << 20)) return fail(PPCX_ERR_ARG, "workgroups must be 0 (automatic) or a positive count");
  HIPCHK(hipSetDevice(m->device));
  m->L_override = lanes_per_gene; m->wgs_override = workgroups;
  drop_plans(m);
  choose_launch(m, 1);
  return PPCX_OK;
}
extern "C" int ppcx_model_set_rounds(ppcx_model* m, int pipelined, int stream_groups) {
  if (!m) return fail(PPCX_ERR_ARG, "model is NULL");
  if (pipelined < -2 || pipelined > 1 || stream_groups < -1 || stream_groups > 64) return fail(PPCX_ERR_ARG, "pipelined must be -2 .. 1 and stream_groups -1 .. 64");
  if (pipelined != -2) m->opt_pipelined = pipelined;           // -2 / -1: leave that setting as it is
  if (stream_groups != -1) m->opt_stream_groups = stream_groups;
  return PPCX_OK;
}
static bool model_pipelines(const ppcx_model* m) {
  // the pipelined round needs cells that read the anticipated constants only: every design since round 5 (a per-cell linear
  // predictor reads the coefficients kept among the constants: Dims::raw_consts)
  return m->opt_pipelined != 0 && m->ls_wgs_per_cu >= 1;
}
extern "C" int ppcx_model_set_progress(ppcx_model* m, ppcx_progress_fn fn, void* user, double every_seconds) {
  if (!m || !(every_seconds >= 0)) return fail(PPCX_ERR_ARG, "bad arguments");
  m->progress = fn; m->progress_user = user; m->progress_every = every_seconds;
  return PPCX_OK;
}
extern "C" int ppcx_model_get_rounds(const ppcx_model* m, int nchains, int* pipelined, int* stream_groups) {
  if (!m || nchains < 1) return fail(PPCX_ERR_ARG, "bad arguments");
  if (pipelined) *pipelined = model_pipelines(m) ? 1 : 0;
  if (stream_groups) {
    int g = default_stream_groups(nchains);
    if (m->opt_stream_groups >= 1) g = m->opt_stream_groups < nchains ? m->opt_stream_groups : nchains;
    *stream_groups = g;
  }
  return PPCX_OK;
}
extern "C" int ppcx_model_get_launch(const ppcx_model* m, int* lanes_per_gene, int* nblocks) {
  if (!m) return fail(PPCX_ERR_ARG, "model is NULL");
  if (lanes_per_gene) *lanes_per_gene = m->L;
  if (nblocks) *nblocks = m->nblocks_chosen;
  return PPCX_OK;
}
// the plan of a log-likelihood launch of `nchains` chains: workgroups per chain, and the gene-order positions
// bounds[0 .. 4 * workgroups_per_chain] that delimit the wavefronts' ranges (bounds may be NULL; `cap` entries at most).
// A diagnostic: tests check its invariants, nothing in the product path reads it back.
extern "C" int ppcx_model_get_plan(ppcx_model* m, int nchains, int* lanes_per_gene, int* workgroups_per_chain, int* bounds, int cap) {
  if (!m || nchains == 0) return fail(PPCX_ERR_ARG, "bad arguments");
  HIPCHK(hipSetDevice(m->device));
  // nchains < 0: the launch of -nchains chains of ONE of several chain groups (trimmed: workgroups_per_chain), at the lanes
  // per gene in force -- those of the fit, chosen for all its chains
  const bool trim = nchains < 0;
  if (trim) nchains = -nchains; else choose_launch(m, nchains);
  ppcx_model::Plan pl;
  const int rc = plan_launch(m, nchains, 0, &pl, trim);
  if (rc != PPCX_OK) return rc;
  if (lanes_per_gene) *lanes_per_gene = m->L;
  if (workgroups_per_chain) *workgroups_per_chain = pl.nbpc;
  if (bounds) {
    const int n = 4 * pl.nbpc + 1;
    if (cap < n) return fail(PPCX_ERR_ARG, "bounds buffer too small");
    HIPCHK(hipMemcpy(bounds, pl.d_bounds, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
  }
  return PPCX_OK;
}
extern "C" int ppcx_model_dim(const ppcx_model* m) { return m ? m->d.D : PPCX_ERR_ARG; }
extern "C" void ppcx_model_destroy(ppcx_model* m) {
  if (!m) return;
  if (m->live_fits > 0) { m->destroy_requested = true; return; }   // freed by the last ppcx_fit_free
  (void)hipSetDevice(m->device);
  drop_plans(m);
  (void)hipFree(m->d_counts); (void)hipFree(m->d_E); (void)hipFree(m->d_expo); (void)hipFree(m->d_X);
  (void)hipFree(m->d_Sy); (void)hipFree(m->d_SyE); (void)hipFree(m->d_SyX); (void)hipFree(m->d_SX); (void)hipFree(m->d_ncell); (void)hipFree(m->d_disp); (void)hipFree(m->d_gflags); (void)hipFree(m->d_Lg1); (void)hipFree(m->d_logtab); (void)hipFree(m->d_wintab); (void)hipFree(m->d_order);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

extern "C" void ppcx_nuts_config_default(ppcx_nuts_config* c) {
  if (!c) return;
  c->chains = 3; c->iter = 300; c->warmup = 150; c->seed = 1; c->adapt_delta = 0.8; c->max_treedepth = 10;
  c->init_radius = 2.0; c->stepsize0 = 1.0; c->init_buffer = 75; c->term_buffer = 50; c->window = 25;
  c->chain_id_offset = 0;
}

// device scratch of one run of the launch pump. States, commands, hyper-coordinate vectors and the T0 slab
// are double-buffered: update launch k reads buffer k&1 and writes buffer (k+1)&1.
struct Work {
  double *vecs = nullptr, *hyper_vecs[2] = {nullptr, nullptr}, *partials = nullptr, *t0[2] = {nullptr, nullptr}, *sums = nullptr, *red = nullptr;
  Cmd* cmds[2] = {nullptr, nullptr}; ChainState* states[2] = {nullptr, nullptr}; int* done = nullptr;
  int* done_host = nullptr;
  long Dpad = 0; int nb_update = 1, nb_close = 1; long launches = 0;
  hipStream_t stream = nullptr; bool own_stream = false;
  bool pipelined = false;        // two launches per round (ppcx_ls_kernel + ppcx_gene_kernel) instead of three
  bool shared_chip = false;      // one of several chain groups of a fit: its launches leave the slots they cannot use (workgroups_per_chain)
  std::atomic<int>* stop = nullptr;   // shared by the chain groups of a fit: set when the progress callback ended one of them
  int *active = nullptr, *active_host = nullptr; int n_active = 0;   // chains still running (pump), 0 = all
  const XchgArgs* xchg = nullptr; int xchg_chain0 = 0;   // gene shards with the direct exchange: the group's first chain in the buffers
  ~Work() {
    (void)hipFree(vecs); (void)hipFree(partials); (void)hipFree(done); (void)hipFree(sums); (void)hipFree(red);
    for (int i = 0; i < 2; ++i) { (void)hipFree(hyper_vecs[i]); (void)hipFree(t0[i]); (void)hipFree(cmds[i]); (void)hipFree(states[i]); }
    if (done_host) (void)hipHostFree(done_host);
    (void)hipFree(active); if (active_host) (void)hipHostFree(active_host);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
};

static int work_alloc(Work& w, ppcx_model* m, int nchains) {
  const int D = m->d.D;
  if (!w.stream) w.stream = m->stream;
  w.Dpad = ((long)D + 31) / 32 * 32;
  // workgroups per chain of the step / update launches: every one of them repeats the step (reads the close kernel's
  // partial sums), so not too many, each with several coordinates per thread. The number does not depend on the chains
  // of the launch: it fixes the summation order of the kinetic energy, and a chain's results must not depend on its company.
  w.nb_update = (D + 255) / 256;
  if (w.nb_update > 80) w.nb_update = 80;
  if (w.nb_update < 1) w.nb_update = 1;
  HIPCHK(hipMalloc(&w.vecs, sizeof(double) * (size_t)nchains * V_COUNT * w.Dpad));
  w.nb_close = (m->d.G + 255) / 256;
  HIPCHK(hipMalloc(&w.partials, sizeof(double) * (size_t)nchains * w.nb_close * PT_COUNT));
  HIPCHK(hipMalloc(&w.sums, sizeof(double) * (size_t)nchains * (3 + m->CM) * m->d.G));
  HIPCHK(hipMemsetAsync(w.sums, 0, sizeof(double) * (size_t)nchains * (3 + m->CM) * m->d.G, w.stream));
  HIPCHK(hipMalloc(&w.done, sizeof(int) * nchains));
  HIPCHK(hipMalloc(&w.red, sizeof(double) * (size_t)nchains * PT_COUNT));
  HIPCHK(hipMemsetAsync(w.red, 0, sizeof(double) * (size_t)nchains * PT_COUNT, w.stream));
  HIPCHK(hipHostMalloc(&w.done_host, sizeof(int) * 2 * nchains));     // two polls in flight (pump)
  HIPCHK(hipHostMalloc(&w.active_host, sizeof(int) * nchains));
  HIPCHK(hipMalloc(&w.active, sizeof(int) * nchains));
  for (int i = 0; i < 2; ++i) {
    HIPCHK(hipMalloc(&w.hyper_vecs[i], sizeof(double) * (size_t)nchains * V_COUNT * 8));
    HIPCHK(hipMalloc(&w.t0[i], sizeof(double) * (size_t)nchains * w.nb_update));
    HIPCHK(hipMalloc(&w.cmds[i], sizeof(Cmd) * nchains));
    HIPCHK(hipMalloc(&w.states[i], sizeof(ChainState) * nchains));
    HIPCHK(hipMemsetAsync(w.hyper_vecs[i], 0, sizeof(double) * (size_t)nchains * V_COUNT * 8, w.stream));
    HIPCHK(hipMemsetAsync(w.t0[i], 0, sizeof(double) * (size_t)nchains * w.nb_update, w.stream));
    HIPCHK(hipMemsetAsync(w.cmds[i], 0, sizeof(Cmd) * nchains, w.stream));
    HIPCHK(hipMemsetAsync(w.states[i], 0, sizeof(ChainState) * nchains, w.stream));
  }
  HIPCHK(hipMemsetAsync(w.vecs, 0, sizeof(double) * (size_t)nchains * V_COUNT * w.Dpad, w.stream));
  HIPCHK(hipMemsetAsync(w.partials, 0, sizeof(double) * (size_t)nchains * w.nb_close * PT_COUNT, w.stream));
  HIPCHK(hipMemsetAsync(w.done, 0, sizeof(int) * nchains, w.stream));
  for (int c = 0; c < nchains; ++c) {          // inverse metric starts at identity
    HIPCHK(launch_fill_kernel(w.vecs + ((size_t)c * V_COUNT + V_MINV) * w.Dpad, w.Dpad, 1.0, w.stream));
    HIPCHK(launch_fill_kernel(w.hyper_vecs[0] + ((size_t)c * V_COUNT + V_MINV) * 8, 8, 1.0, w.stream));
  }
  w.launches = 0;
  // the callers upload the initial chain states / hyper vectors next, some of them with blocking copies on the NULL
  // stream, which does not order against this non-blocking stream: the zero fills above must have landed first
  HIPCHK(hipStreamSynchronize(w.stream));
  return PPCX_OK;
}

struct RunIO {                  // output buffers of a run (device pointers, may be null)
  double* draws = nullptr; long draws_stride = 0; int n_keep = 0, iter = 0;
  double *lp = nullptr, *stepsize = nullptr, *accept = nullptr; int *treedepth = nullptr, *nleap = nullptr, *div = nullptr;
};

// step kernel: reduce (+ optional) advance. After an ADVANCE launch the "current" buffers are the ones it wrote.
// The kinetic energy of freshly drawn momenta travels from the update of one round to the step of the next through the
// T0 slab, double-buffered like the states: a step launched at generation g (= w.launches) reads buffer g & 1, the
// update that belongs to the command it decides writes buffer (g + 1) & 1.
// with_update: the per-coordinate work of the new command in the same launch (ppcx_kernels.hip, ppcx_step_kernel).
static void step_args(ppcx_model* m, Work& w, const RunIO& io, int phases, bool with_update, StepArgs* o) {
  const int in = (int)(w.launches & 1), out = in ^ 1;
  StepArgs& sa = *o;
  sa.d = m->d; sa.phases = phases;
  sa.states_in = w.states[in]; sa.states_out = w.states[out];
  sa.cmds_in = w.cmds[in]; sa.cmds_out = w.cmds[out];
  sa.hyper_in = w.hyper_vecs[in]; sa.hyper_out = w.hyper_vecs[out];
  sa.partials = w.partials; sa.nblocks_close = w.nb_close; sa.slab_stride = w.nb_close; sa.t0 = w.t0[in]; sa.nblocks_update = w.nb_update; sa.red = w.red;
  sa.draws = io.draws; sa.draws_chain_stride = io.draws_stride; sa.n_keep = io.n_keep; sa.iter = io.iter;
  sa.out_lp = io.lp; sa.out_stepsize = io.stepsize; sa.out_treedepth = io.treedepth; sa.out_n_leapfrog = io.nleap;
  sa.out_divergent = io.div; sa.out_accept = io.accept; sa.done = w.done;
  sa.upd_vecs = nullptr; sa.upd_Dpad = 0; sa.upd_t0_out = nullptr;
  sa.x = XchgArgs();
  if (w.xchg) {                                  // this group's chains start at xchg_chain0 of the exchange buffers
    sa.x = *w.xchg;
    sa.x.chain0 = w.xchg_chain0;
  }
  if (with_update && (phases & STEP_ADVANCE)) { sa.upd_vecs = w.vecs; sa.upd_Dpad = w.Dpad; sa.upd_t0_out = w.t0[out]; }
}
static int launch_step(ppcx_model* m, Work& w, int nchains, const RunIO& io, int phases, bool with_update = false) {
  StepArgs sa;
  step_args(m, w, io, phases, with_update, &sa);
  hipError_t e = launch_step_kernel(sa, w.nb_update, nchains, w.stream);
  if (e != hipSuccess) return fail(PPCX_ERR_HIP, std::string("step kernel: ") + hipGetErrorString(e));
  if (phases & STEP_ADVANCE) w.launches++;
  return PPCX_OK;
}
// the per-coordinate work of the current command in a launch of its own (after a step without with_update)
static int launch_update(ppcx_model* m, Work& w, int nchains, const RunIO& io) {
  UpdateArgs ua;
  ua.d = m->d; ua.cmds = w.cmds[w.launches & 1]; ua.vecs = w.vecs; ua.Dpad = w.Dpad;
  ua.draws = io.draws; ua.draws_chain_stride = io.draws_stride; ua.t0_out = w.t0[w.launches & 1];
  hipError_t e = launch_update_kernel(ua, w.nb_update, nchains, w.stream);
  if (e != hipSuccess) return fail(PPCX_ERR_HIP, std::string("update kernel: ") + hipGetErrorString(e));
  return PPCX_OK;
}
// runs of a pipelined round's merged launch whose position 7 holds state machines: one per chain of the fit, `nact`
// (chains still running = columns of the launch) per run
static int step_runs(int nchains, int nact) { return (nchains + nact - 1) / nact; }
static int loglik_args(ppcx_model* m, Work& w, int nchains, int reserve, LoglikArgs* out) {
  const int nact = w.n_active > 0 ? w.n_active : nchains;
  ppcx_model::Plan pl;
  int rc = plan_launch(m, nact, reserve, &pl, w.shared_chip);
  if (rc != PPCX_OK) return rc;
  LoglikArgs& la = *out;
  la.d = m->d; la.cd.counts = m->d_counts; la.cd.disp = m->d_disp; la.cd.gflags = m->d_gflags; la.cd.Sy = m->d_Sy; la.cd.ncell = m->d_ncell; la.cd.e_min = m->e_min; la.cd.e_max = m->e_max; la.sampleE = m->d_E; la.exposure = m->d_expo; la.X = m->d_X;
  la.vecs = w.vecs; la.Dpad = w.Dpad; la.cmds = w.cmds[w.launches & 1]; la.sums = w.sums; la.logtab = m->d_logtab; la.wintab = m->d_wintab; la.order = m->d_order;
  la.lgL = 0; while ((1 << la.lgL) < m->L) ++la.lgL;
  la.nchains = nact; la.active = w.n_active > 0 ? w.active : nullptr; la.nbpc = pl.nbpc; la.bounds = pl.d_bounds;
#ifdef PPCX_TRACE
  la.trace = g_trace_dev;
#endif
  return PPCX_OK;
}
static int launch_loglik(ppcx_model* m, Work& w, int nchains) {
  LoglikArgs la;
  int rc = loglik_args(m, w, nchains, 0, &la);
  if (rc != PPCX_OK) return rc;
  hipError_t e = launch_loglik_kernel(m->CM, la, w.stream);
  if (e != hipSuccess) return fail(PPCX_ERR_HIP, std::string("loglik kernel: ") + hipGetErrorString(e));
  return PPCX_OK;
}
static void close_args(ppcx_model* m, Work& w, CloseArgs* o);
// pipelined round, first launch: the state machines that digest the previous gene kernel's sums beside the log-likelihood
// workgroups of this round (the command buffer the log-likelihood part reads is the one the state machines read, not
// the one they write)
static int launch_ls(ppcx_model* m, Work& w, int nchains, const RunIO& io, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
  const int nact = w.n_active > 0 ? w.n_active : nchains;
  const int n_srun = step_runs(nchains, nact);
  LoglikArgs la;
  int rc = loglik_args(m, w, nchains, n_srun * nact, &la);       // the state machines' slots are not log-likelihood workgroups
  if (rc != PPCX_OK) return rc;
  StepArgs sa;
  step_args(m, w, io, STEP_REDUCE | STEP_ADVANCE, false, &sa);
  hipError_t e = launch_ls_kernel(m->CM, la, sa, n_srun, nchains, 1, w.stream, ev_start, ev_stop);
  if (e != hipSuccess) return fail(PPCX_ERR_HIP, std::string("merged log-likelihood / step kernel: ") + hipGetErrorString(e));
  w.launches++;
  return PPCX_OK;
}
// pipelined round, second launch: the command the state machines just wrote, gene by gene
static int launch_gene_round(ppcx_model* m, Work& w, int nchains, const RunIO& io, int spec = 1) {
  GeneArgs ga;
  close_args(m, w, &ga.c);
  ga.draws = io.draws; ga.draws_chain_stride = io.draws_stride; ga.spec = spec;
  hipError_t e = launch_gene_kernel(m->CM, ga, w.nb_close, nchains, w.stream);
  if (e != hipSuccess) return fail(PPCX_ERR_HIP, std::string("gene kernel: ") + hipGetErrorString(e));
  return PPCX_OK;
}
static void close_args(ppcx_model* m, Work& w, CloseArgs* o) {
  CloseArgs& ca = *o;
  ca.d = m->d; ca.Sy = m->d_Sy; ca.SyE = m->d_SyE; ca.SyX = m->d_SyX; ca.SX = m->d_SX; ca.ncell = m->d_ncell; ca.Lg1 = m->d_Lg1;
  ca.sums = w.sums; ca.vecs = w.vecs; ca.Dpad = w.Dpad; ca.cmds = w.cmds[w.launches & 1]; ca.partials = w.partials;
}
static int launch_close(ppcx_model* m, Work& w, int nchains) {
  CloseArgs ca;
  ca.d = m->d; ca.Sy = m->d_Sy; ca.SyE = m->d_SyE; ca.SyX = m->d_SyX; ca.SX = m->d_SX; ca.ncell = m->d_ncell; ca.Lg1 = m->d_Lg1;
  ca.sums = w.sums; ca.vecs = w.vecs; ca.Dpad = w.Dpad; ca.cmds = w.cmds[w.launches & 1]; ca.partials = w.partials;
  hipError_t e = launch_close_kernel(m->CM, ca, w.nb_close, nchains, w.stream);
  if (e != hipSuccess) return fail(PPCX_ERR_HIP, std::string("close kernel: ") + hipGetErrorString(e));
  return PPCX_OK;
}
static int launch_gene(ppcx_model* m, Work& w, int nchains) {   // one gradient evaluation = loglik + close
  int rc = launch_loglik(m, w, nchains);
  return rc != PPCX_OK ? rc : launch_close(m, w, nchains);
}
static ChainState* current_states(Work& w) { return w.states[w.launches & 1]; }
static double* current_hyper(Work& w) { return w.hyper_vecs[w.launches & 1]; }

struct PumpStats { double kA_ms_sum = 0, kC_ms_sum = 0, kU_ms_sum = 0; long long kA_samples = 0; double chain_launches = 0; long long pairs = 0; };

// ---- RCCL, bound at run time (dlopen) so the library has no link-time dependency and shares the RCCL that
// the process may already have loaded (torch ships one)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId_t;
struct RcclApi {
  void* h = nullptr;
  int (*GetUniqueId)(ncclUniqueId_t*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId_t, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
  if (g_rccl.h) return PPCX_OK;
#ifdef PPCX_TESTING
  // another provider of the five nccl* entry points below (tests/loopback: ranks of one host over shared memory, so that
  // the RCCL path runs with two ranks on a one-GPU box, where RCCL itself refuses two ranks on a device)
  if (!g_test.rccl_lib.empty()) {
    g_rccl.h = dlopen(g_test.rccl_lib.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl.h) return fail(PPCX_ERR_HIP, std::string("cannot load the nccl provider ") + g_test.rccl_lib + ": " + dlerror());
  }
#endif
  if (!g_rccl.h) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.h) break; }
  }
  if (!g_rccl.h) return fail(PPCX_ERR_HIP, "cannot load librccl.so");
  g_rccl.GetUniqueId = (int (*)(ncclUniqueId_t*))dlsym(g_rccl.h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId_t, int))dlsym(g_rccl.h, "ncclCommInitRank");
  g_rccl.CommDestroy = (int (*)(ncclComm_t))dlsym(g_rccl.h, "ncclCommDestroy");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))dlsym(g_rccl.h, "ncclAllReduce");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce) return fail(PPCX_ERR_HIP, "librccl.so lacks the expected symbols");
  return PPCX_OK;
}
struct ppcx_comm { ncclComm_t comm = nullptr; int nranks = 1, rank = 0, device = 0; double* d_guard = nullptr; double* h_guard = nullptr; };
// Every rank of a gene-sharded run replicates the chains' state machines and must issue the same launches. At every
// poll the ranks compare (rounds issued, chains done, local error) with ONE max-reduction of [x, -x] pairs: if the
// counts differ anywhere, or any rank failed, every rank leaves the pump with the same status instead of waiting for
// a collective its peers will never issue.
static int guard_decision(const double* g, int local_rc);
static int comm_guard(ppcx_comm* c, hipStream_t st, long long pairs, int n_done, int local_rc, int* all_rc) {
  // the vector travels through pinned host memory (the device reads it in place): a failing upload cannot keep this rank
  // out of the collective its peers are about to enter
  double* v = c->h_guard;
  v[0] = (double)pairs; v[1] = -(double)pairs; v[2] = (double)n_done; v[3] = -(double)n_done; v[4] = local_rc != PPCX_OK ? (double)(-local_rc) : 0.0;
  const std::string local_msg = g_err;
  hipError_t he = hipMemcpyAsync(c->d_guard, v, sizeof(double) * 5, hipMemcpyHostToDevice, st);
  const int e = g_rccl.AllReduce(c->d_guard, c->d_guard, 5, /*ncclDouble*/ 8, /*ncclMax*/ 2, c->comm, st);
  if (e != 0) return fail(PPCX_ERR_HIP, std::string("ncclAllReduce (guard): ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"));
  if (he == hipSuccess) he = hipMemcpyAsync(c->h_guard + 8, c->d_guard, sizeof(double) * 5, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  if (he != hipSuccess) return fail(PPCX_ERR_HIP, std::string("guard exchange: ") + hipGetErrorString(he));
  const int d = guard_decision(c->h_guard + 8, local_rc);
  *all_rc = d;
  if (d != PPCX_OK) {
    if (local_rc != PPCX_OK) g_err = local_msg;
    else if (d == PPCX_ERR_STALL) g_err = "the ranks of the gene-sharded run disagree on the rounds issued or the chains finished";
    else g_err = "another rank of the gene-sharded run reported an error";
  }
  return PPCX_OK;
}

// One shard of a run: its model (all genes, or a contiguous gene range) and its device scratch.
struct Shard { ppcx_model* m; Work* w; RunIO io; };

// What the ranks of a gene-sharded run conclude from the max-reduced guard vector [rounds, -rounds, done, -done, error]
// (a pure function: tests/test_abi.py drives it through ppcx_guard_decision without a GPU).
static int guard_decision(const double* g, int local_rc) {
  if (g[4] != 0.0) return local_rc != PPCX_OK ? local_rc : -(int)g[4];
  if (g[0] != -g[1] || g[2] != -g[3]) return PPCX_ERR_STALL;
  return PPCX_OK;
}
extern "C" int ppcx_guard_decision(const double* reduced5, int local_rc) { return reduced5 ? guard_decision(reduced5, local_rc) : PPCX_ERR_ARG; }

// Launch rounds until every chain reports done. A round is (loglik, close, step + update) -- three launches -- or, pipelined
// (Work::pipelined), (merged log-likelihood / step launch, gene kernel) -- two. With several shards in one process they share
// shard 0's stream and their partial sums are added by ppcx_sum_shards_kernel; with a communicator the sums are all-reduced
// over the ranks (RCCL, xGMI) between reduce and advance.
static int pump(std::vector<Shard>& sh, int nchains, ppcx_comm* comm, long long max_pairs, bool time_kernels,
                PumpStats* stats) {
  const int ns = (int)sh.size();
  hipStream_t st = sh[0].w->stream;
  int rc = PPCX_OK;
  // Several ranks (one gene shard per process): a rank that fails must not leave its peers waiting in a collective. It
  // stops launching kernels but keeps issuing the per-round all-reduces until the next poll, where comm_guard lets every
  // rank see the failure (or a disagreement on the rounds issued) and leave together. Every local failure inside the
  // loop -- a launch, an event, a copy -- becomes local_rc; only a failing collective returns at once (its peers are
  // then in an undefined state anyway).
  const bool guarded = comm && comm->comm && comm->nranks > 1;
  const bool piped = ns == 1 && !(comm && comm->comm) && sh[0].w->pipelined;
  int local_rc = PPCX_OK;
#define PUMP_TRY(expr) do { if (local_rc == PPCX_OK) { const int r_ = (expr); if (r_ != PPCX_OK) { if (!guarded) return r_; local_rc = r_; } } } while (0)
#define PUMP_HIP(expr) do { if (local_rc == PPCX_OK) { const hipError_t e_ = (expr); if (e_ != hipSuccess) { const int r_ = fail(PPCX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); if (!guarded) return r_; local_rc = r_; } } } while (0)
  if (!piped) for (int k = 0; k < ns; ++k) {                      // PH_START: first command, then its coordinate work
    PUMP_TRY(launch_step(sh[k].m, *sh[k].w, nchains, sh[k].io, STEP_REDUCE | STEP_ADVANCE));
    PUMP_TRY(launch_update(sh[k].m, *sh[k].w, nchains, sh[k].io));
  }
  const int batch = 32, sample_every = 16;
  struct Events {                // destroyed on every exit path
    hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
  } evs;
  if (time_kernels) for (int i = 0; i < 4; ++i) if (hipEventCreate(&evs.e[i]) != hipSuccess) { evs.e[i] = nullptr; time_kernels = false; }
  hipEvent_t &ev0 = evs.e[0], &ev1 = evs.e[1], &ev2 = evs.e[2], &ev3 = evs.e[3];
  // The poll. Plain: after every batch of rounds the done flags are copied back and the stream is waited for -- the GPU then
  // idles until the host has woken up and launched again. A pipelined single-process fit polls ONE BATCH BEHIND instead: the
  // flags of batch b are looked at while batch b + 1 is already queued, so the queue never runs dry (single stream: 2 % of a
  // fit were such bubbles). The chains notice one batch later that they are all done (32 rounds of kernels that return at
  // once), and the list of active chains is still rewritten on an idle stream, a few times per fit.
  bool lookahead = piped && !guarded;
  if (lookahead) for (int i = 4; i < 6; ++i) if (hipEventCreateWithFlags(&evs.e[i], hipEventDisableTiming) != hipSuccess) { evs.e[i] = nullptr; lookahead = false; }
  long long pairs = 0; int n_done = 0, n_done_applied = 0;
  const auto t_start = std::chrono::steady_clock::now(); auto t_report = t_start;
  Work& w0 = *sh[0].w;
  int cur = 0; bool have_prev = false, sampled_prev = false;
  while (true) {
    bool sampled = false;
    for (int i = 0; i < batch; ++i, ++pairs) {
      const bool smp = time_kernels && !sampled && (pairs / batch) % sample_every == 0 && i == batch / 2 && local_rc == PPCX_OK;
      if (smp && !piped) PUMP_HIP(hipEventRecord(ev0, st));
      if (piped) {
        // a sampled merged launch carries its own start / stop events (hipExtLaunchKernel): the kernel's duration as the
        // profiler's kernel trace sees it; a hipEventRecord on either side adds its marker packets (3-4 us on a 38 us launch)
        PUMP_TRY(launch_ls(sh[0].m, w0, nchains, sh[0].io, smp ? ev0 : nullptr, smp ? ev1 : nullptr));
        if (smp) sampled = true;
        PUMP_TRY(launch_gene_round(sh[0].m, w0, nchains, sh[0].io));
        if (smp) { PUMP_HIP(hipEventRecord(ev2, st)); PUMP_HIP(hipEventRecord(ev3, st)); }
        continue;
      }
      for (int k = 0; k < ns; ++k) PUMP_TRY(launch_loglik(sh[k].m, *sh[k].w, nchains));
      if (smp) { PUMP_HIP(hipEventRecord(ev1, st)); sampled = true; }
      const bool exchange = ns > 1 || (comm && comm->comm);
      for (int k = 0; k < ns; ++k) {
        PUMP_TRY(launch_close(sh[k].m, *sh[k].w, nchains));
        if (smp && k == ns - 1) PUMP_HIP(hipEventRecord(ev2, st));
        PUMP_TRY(launch_step(sh[k].m, *sh[k].w, nchains, sh[k].io, exchange ? STEP_REDUCE : (STEP_REDUCE | STEP_ADVANCE), !exchange));
      }
      if (ns > 1) {
        ShardSumArgs sa; sa.n_shards = ns; sa.n = nchains * PT_COUNT;
        for (int k = 0; k < ns; ++k) sa.bufs[k] = sh[k].w->red;
        PUMP_HIP(launch_sum_shards_kernel(sa, st));
      }
      if (comm && comm->nranks >= 1 && comm->comm) {                 // issued by every rank every round, failed or not
        const int e = g_rccl.AllReduce(w0.red, w0.red, (size_t)nchains * PT_COUNT, /*ncclDouble*/ 8, /*ncclSum*/ 0, comm->comm, st);
        if (e != 0) return fail(PPCX_ERR_HIP, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"));
      }
      for (int k = 0; k < ns; ++k) {
        if (exchange) PUMP_TRY(launch_step(sh[k].m, *sh[k].w, nchains, sh[k].io, STEP_ADVANCE, true));   // step + coordinate update in one launch
      }
      if (smp) PUMP_HIP(hipEventRecord(ev3, st));
    }
#ifdef PPCX_TESTING
    if (g_test.fail_at_round > 0 && pairs >= g_test.fail_at_round && local_rc == PPCX_OK &&
        (g_test.fail_rank < 0 || (!comm && !w0.xchg) || g_test.fail_rank == (comm ? comm->rank : w0.xchg->rank)))   // fault injection
      local_rc = fail(PPCX_ERR_HIP, "injected failure (ppcx_testing_set fail_at_round)");
#endif
    int* flags = w0.done_host + (lookahead ? cur * nchains : 0);
    PUMP_HIP(hipMemcpyAsync(flags, w0.done, sizeof(int) * nchains, hipMemcpyDeviceToHost, st));
    bool sampled_chk = sampled;
    if (lookahead) {
      PUMP_HIP(hipEventRecord(evs.e[4 + cur], st));
      if (!have_prev) { have_prev = true; sampled_prev = sampled; cur ^= 1; continue; }   // the first batch is looked at after the second is queued
      flags = w0.done_host + (cur ^ 1) * nchains;
      PUMP_HIP(hipEventSynchronize(evs.e[4 + (cur ^ 1)]));
      sampled_chk = sampled_prev; sampled_prev = sampled;
    } else {
      PUMP_HIP(hipStreamSynchronize(st));
    }
    if (sampled_chk && n_done == 0 && local_rc == PPCX_OK) {   // only launches in which every chain was still active
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) {
        stats->kA_ms_sum += ms; stats->kA_samples++; stats->chain_launches += nchains;
        if (hipEventElapsedTime(&ms, ev1, ev2) == hipSuccess) stats->kC_ms_sum += ms;
        if (hipEventElapsedTime(&ms, ev2, ev3) == hipSuccess) stats->kU_ms_sum += ms;
      }
    }
    n_done = 0;
    rc = local_rc;
    if (local_rc == PPCX_OK) for (int c = 0; c < nchains; ++c) {
      if (flags[c]) ++n_done;
      if (flags[c] == 2) rc = fail(PPCX_ERR_INIT, "no finite initial point after 100 attempts");
      if (flags[c] == 3) rc = fail(PPCX_ERR_STEPSIZE, "step-size heuristic diverged");
      if (flags[c] == 5) rc = fail(PPCX_ERR_STALL, "gene-shard exchange: a peer rank left the fit or did not arrive within the timeout");
    }
    if (pairs > max_pairs && n_done < nchains && rc == PPCX_OK) rc = fail(PPCX_ERR_STALL, "launch budget exhausted before the chains finished");
    if (guarded) {
      int all_rc = PPCX_OK;
      const int grc = comm_guard(comm, st, pairs, n_done, rc, &all_rc);
      if (grc != PPCX_OK) return grc;
      if (all_rc != PPCX_OK) { rc = all_rc; break; }
    } else if (rc != PPCX_OK) break;
    if (sh[0].m->progress && local_rc == PPCX_OK) {   // a blocking call of minutes need not be silent: rounds issued, chains done
      const auto now = std::chrono::steady_clock::now();
      if (std::chrono::duration<double>(now - t_report).count() >= sh[0].m->progress_every || n_done == nchains) {
        t_report = now;
        const int stop = sh[0].m->progress(sh[0].m->progress_user, sh[0].w->xchg_chain0, nchains, n_done, pairs, std::chrono::duration<double>(now - t_start).count());
        if (stop != 0 && n_done < nchains) {     // the caller's budget is spent: a local failure like any other
          local_rc = fail(PPCX_ERR_CANCELLED, "the progress callback ended the fit");
          if (w0.stop) w0.stop->store(1);
          if (!guarded) { rc = local_rc; break; }
        }
      }
    }
    if (w0.stop && w0.stop->load() && local_rc == PPCX_OK && n_done < nchains) {   // another chain group of this fit was ended
      local_rc = fail(PPCX_ERR_CANCELLED, "the progress callback ended the fit");
      if (!guarded) { rc = local_rc; break; }
    }
    if (n_done == nchains) break;
    // fewer chains in the launch: the others get their wavefronts (the list is rewritten on an idle stream: with the poll one
    // batch behind the queued batch is waited for first, and its newer flags are the ones applied)
    if (n_done > n_done_applied) {
      if (lookahead) {
        PUMP_HIP(hipStreamSynchronize(st));
        flags = w0.done_host + cur * nchains;
        have_prev = false;                       // both batches are finished and looked at: start over
      }
      int na = 0;
      for (int c = 0; c < nchains; ++c) if (!flags[c]) w0.active_host[na++] = c;
      n_done_applied = nchains - na;
      if (na == 0) { n_done = nchains; break; }
      for (int k = 0; k < ns; ++k) {
        Work& wk = *sh[k].w;
        PUMP_HIP(hipMemcpyAsync(wk.active, w0.active_host, sizeof(int) * na, hipMemcpyHostToDevice, st));
        wk.n_active = na;
      }
      PUMP_HIP(hipStreamSynchronize(st));
    }
    if (lookahead) cur ^= 1;
  }
  if (lookahead) (void)hipStreamSynchronize(st);   // a queued batch may still be running: nothing is freed under it
#undef PUMP_TRY
#undef PUMP_HIP
  stats->pairs = pairs;
  return rc;
}
static int pump(ppcx_model* m, Work& w, int nchains, const RunIO& io, long long max_pairs, bool time_kernels,
                PumpStats* stats, ppcx_comm* comm = nullptr) {
  std::vector<Shard> sh(1);
  sh[0].m = m; sh[0].w = &w; sh[0].io = io;
  return pump(sh, nchains, comm, max_pairs, time_kernels, stats);
}

extern "C" int ppcx_log_prob_grad(ppcx_model* m, int n_points, const double* u, double* lp, double* grad) {
  if (!m || n_points < 1 || !u || !lp) return fail(PPCX_ERR_ARG, "bad arguments");
  HIPCHK(hipSetDevice(m->device));
  const int D = m->d.D;
  const int maxb = 256;                         // points per batch (grid.y)
  for (int p0 = 0; p0 < n_points; p0 += maxb) {
    const int nb = n_points - p0 < maxb ? n_points - p0 : maxb;
    choose_launch(m, nb);
    Work w;
    int rc = work_alloc(w, m, nb);
    if (rc != PPCX_OK) return rc;
    std::vector<ChainState> states(nb);
    NutsConfig cfg; memset(&cfg, 0, sizeof cfg);
    cfg.chains = nb; cfg.iter = 0; cfg.warmup = 0; cfg.seed = 0; cfg.adapt_delta = 0.8; cfg.max_treedepth = 10;
    cfg.init_radius = 2; cfg.stepsize0 = 1; cfg.init_buffer = 75; cfg.term_buffer = 50; cfg.window = 25; cfg.chain_id_offset = 0;
    for (int c = 0; c < nb; ++c) state_init(states[c], cfg, c, 1);
    HIPCHK(hipMemcpyAsync(w.states[0], states.data(), sizeof(ChainState) * nb, hipMemcpyHostToDevice, m->stream));
    std::vector<double> hq((size_t)nb * V_COUNT * 8, 0.0);
    for (int c = 0; c < nb; ++c) {
      const double* uc = u + (size_t)(p0 + c) * D;
      HIPCHK(hipMemcpyAsync(w.vecs + ((size_t)c * V_COUNT + V_Q1) * w.Dpad, uc, sizeof(double) * D, hipMemcpyHostToDevice, m->stream));
      for (int k = 0; k < 6; ++k) hq[((size_t)c * V_COUNT + V_Q1) * 8 + k] = uc[hyper_index(m->d, k)];
      for (int k = 0; k < 8; ++k) hq[((size_t)c * V_COUNT + V_MINV) * 8 + k] = 1.0;
    }
    HIPCHK(hipMemcpyAsync(w.hyper_vecs[0], hq.data(), sizeof(double) * hq.size(), hipMemcpyHostToDevice, m->stream));
    RunIO io;
    PumpStats ps;
    rc = pump(m, w, nb, io, 64, false, &ps);
    if (rc != PPCX_OK) return rc;
    HIPCHK(hipMemcpy(states.data(), current_states(w), sizeof(ChainState) * nb, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hq.data(), current_hyper(w), sizeof(double) * hq.size(), hipMemcpyDeviceToHost));
    for (int c = 0; c < nb; ++c) {
      lp[p0 + c] = states[c].sc.lp_eval;
      if (grad) {
        double* gc = grad + (size_t)(p0 + c) * D;
        HIPCHK(hipMemcpy(gc, w.vecs + ((size_t)c * V_COUNT + V_G1) * w.Dpad, sizeof(double) * D, hipMemcpyDeviceToHost));
        for (int k = 0; k < 6; ++k) gc[hyper_index(m->d, k)] = hq[((size_t)c * V_COUNT + V_G1) * 8 + k];
      }
    }
  }
  return PPCX_OK;
}

#ifdef PPCX_TESTING
// ---- testing build only (ppcx_testing.h) ---------------------------------------------------------------------------
extern "C" int ppcx_testing_set(const char* key, long long value) {
  if (!key) return fail(PPCX_ERR_ARG, "key is NULL");
  const std::string k(key);
  if (k == "fail_at_round") g_test.fail_at_round = value;
  else if (k == "fail_rank") g_test.fail_rank = (int)value;
  else if (k == "force_generic") g_test.force_generic = (int)value;
  else if (k == "slope_cost_permille") g_test.slope_cost_permille = (int)value;
  else if (k == "trim_slack_permille") g_test.trim_slack_permille = (int)value;
  else if (k == "trim_extra_passes") g_test.trim_extra_passes = (int)value;
  else return fail(PPCX_ERR_ARG, "unknown test hook " + k);
  return PPCX_OK;
}
extern "C" int ppcx_testing_set_nccl_provider(const char* path) {
  if (g_rccl.h) return fail(PPCX_ERR_ARG, "the nccl entry points are bound already");
  g_test.rccl_lib = path ? path : "";
  return PPCX_OK;
}
// Kernel-level timing: mean duration (ms) of `reps` back-to-back launches of one kernel (`which`, ppcx_testing.h) of the
// three-launch round on the command the chains hold after `warm_rounds` rounds of a real run. n_merge >= 0 overrides the
// tree position of that command (number of subtree merges the leaf closes), so every variant is timed on the same work.
static std::mutex g_sm_mutex; static long long g_sm_ticks[6] = {0, 0, 0, 0, 0, 0}; static long long g_sm_rounds = 0;
// mean microseconds per round a chain's state machine (the workgroup beside the log-likelihood workgroups of a pipelined round)
// spent in its phases, over the fits since the last call: [0] until state, command, hyper vectors and slab have arrived,
// [1] folding the slab and staging in LDS, [2] the exchange between ranks, [3] chain_step, [4] after it; out[5] = rounds counted
extern "C" int ppcx_testing_sm_trace(double* out6) {
  std::lock_guard<std::mutex> lk(g_sm_mutex);
  for (int k = 0; k < 5; ++k) out6[k] = g_sm_rounds ? 1e-2 * (double)g_sm_ticks[k] / (double)g_sm_rounds : 0.0;
  out6[5] = (double)g_sm_rounds;
  for (int k = 0; k < 6; ++k) g_sm_ticks[k] = 0;
  g_sm_rounds = 0;
  return PPCX_OK;
}
extern "C" int ppcx_testing_bench_kernel(ppcx_model* m, int which, int nchains, int warm_rounds, int reps, int n_merge,
                                         double* ms_per_launch, int* cmd_type) {
  if (!m || nchains < 1 || reps < 1 || !ms_per_launch || which < 0 || which > PPCX_BENCH_GENE_NEW_TRANSITION) return fail(PPCX_ERR_ARG, "bad arguments");
  HIPCHK(hipSetDevice(m->device));
  choose_launch(m, nchains);
  Work w;
  int rc = work_alloc(w, m, nchains);
  if (rc != PPCX_OK) return rc;
  NutsConfig nc; memset(&nc, 0, sizeof nc);
  nc.chains = nchains; nc.iter = 1000000; nc.warmup = 1000000; nc.seed = 1; nc.adapt_delta = 0.8; nc.max_treedepth = 10;
  nc.init_radius = 2; nc.stepsize0 = 1; nc.init_buffer = 75; nc.term_buffer = 50; nc.window = 25;
  std::vector<ChainState> states(nchains);
  for (int c = 0; c < nchains; ++c) state_init(states[c], nc, c, 0);
  HIPCHK(hipMemcpyAsync(w.states[0], states.data(), sizeof(ChainState) * nchains, hipMemcpyHostToDevice, m->stream));
  RunIO io; io.iter = nc.iter;
  hipStream_t st = m->stream;
  if ((rc = launch_step(m, w, nchains, io, STEP_REDUCE | STEP_ADVANCE)) != PPCX_OK) return rc;
  if ((rc = launch_update(m, w, nchains, io)) != PPCX_OK) return rc;
  for (int i = 0; i < warm_rounds; ++i) {
    if ((rc = launch_gene(m, w, nchains)) != PPCX_OK) return rc;
    if ((rc = launch_step(m, w, nchains, io, STEP_REDUCE | STEP_ADVANCE)) != PPCX_OK) return rc;
    if ((rc = launch_update(m, w, nchains, io)) != PPCX_OK) return rc;
  }
  HIPCHK(hipStreamSynchronize(st));
  Cmd* dcmds = w.cmds[w.launches & 1];
  std::vector<Cmd> cmds(nchains);
  HIPCHK(hipMemcpy(cmds.data(), dcmds, sizeof(Cmd) * nchains, hipMemcpyDeviceToHost));
  if (cmd_type) *cmd_type = cmds[0].type;
  if (n_merge >= 0) for (int c = 0; c < nchains; ++c) {
    // a chain still searching its step size after the warm rounds is timed on a leaf as well (the gene kernel's variants:
    // with step-size trials among the chains the launch takes as long as their fresh momenta, whatever the others do)
    if (which >= PPCX_BENCH_GENE && cmds[c].type == CMD_EPS_TRY) { cmds[c].type = CMD_LEAF; cmds[c].pre_dir = cmds[c].dir; cmds[c].next_dir = cmds[c].dir; cmds[c].leaf_n = 1; }
    if (cmds[c].type != CMD_LEAF) continue;
    cmds[c].n_merge = n_merge; cmds[c].subtree_complete = 0;
    cmds[c].eps *= 1e-3;                         // keep the repeated second half kicks on a bounded trajectory
  }
  if (which >= PPCX_BENCH_GENE) for (int c = 0; c < nchains; ++c) {   // the gene kernel of a pipelined round: apply + close + anticipate
    cmds[c].evaluated = 1; cmds[c].updated = 0;
    // a plain leaf inside a subtree: the proposal copy is its only pre-operation (the command left by the warm rounds may be a
    // transition's first leaf, whose fresh momenta -- Philox, Box-Muller -- are a thirtieth of a fit's rounds, not the typical one)
    if (cmds[c].type == CMD_LEAF) { cmds[c].pre_flags = PRE_PROP; cmds[c].prop_slot = n_merge >= 0 ? n_merge : 0; cmds[c].prop_src = -1; }
    if (which == PPCX_BENCH_GENE_NEW_TRANSITION && cmds[c].type == CMD_LEAF) { cmds[c].pre_flags = PRE_NEW_TRANSITION | PRE_SAVE_NEAR; cmds[c].rng_c1 = 7; }
    if (which == PPCX_BENCH_GENE_NO_PROP) cmds[c].pre_flags &= ~PRE_PROP;   // what the kernel would cost without the proposal copies
    if (which == PPCX_BENCH_GENE_UPDATE_ONLY) cmds[c].evaluated = 0;        // apply the command only (no close, its own constants)
  }
  HIPCHK(hipMemcpy(dcmds, cmds.data(), sizeof(Cmd) * nchains, hipMemcpyHostToDevice));
  auto one = [&]() -> int {
    switch (which) {
      case PPCX_BENCH_CLOSE: return launch_close(m, w, nchains);
      case PPCX_BENCH_LOGLIK_CLOSE: return launch_gene(m, w, nchains);
      case PPCX_BENCH_STEP: return launch_step(m, w, nchains, io, STEP_REDUCE | STEP_ADVANCE);
      case PPCX_BENCH_UPDATE: return launch_update(m, w, nchains, io);
      case PPCX_BENCH_STEP_REDUCE: return launch_step(m, w, nchains, io, STEP_REDUCE);
      case PPCX_BENCH_STEP_ADVANCE: return launch_step(m, w, nchains, io, STEP_ADVANCE);
      case PPCX_BENCH_STEP_UPDATE: return launch_step(m, w, nchains, io, STEP_REDUCE | STEP_ADVANCE, true);
      case PPCX_BENCH_GENE: case PPCX_BENCH_GENE_NO_PROP: case PPCX_BENCH_GENE_UPDATE_ONLY: case PPCX_BENCH_GENE_NEW_TRANSITION:
        return launch_gene_round(m, w, nchains, io);
      case PPCX_BENCH_GENE_NO_SPEC: return launch_gene_round(m, w, nchains, io, 0);
      default: return launch_loglik(m, w, nchains);
    }
  };
  struct Ev { hipEvent_t e0 = nullptr, e1 = nullptr; ~Ev() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); } } ev;   // destroyed on every return
  HIPCHK(hipEventCreate(&ev.e0)); HIPCHK(hipEventCreate(&ev.e1));
  for (int i = 0; i < 3; ++i) if ((rc = launch_gene(m, w, nchains)) != PPCX_OK) return rc;
  HIPCHK(hipEventRecord(ev.e0, st));
  for (int i = 0; i < reps; ++i) if ((rc = one()) != PPCX_OK) return rc;
  HIPCHK(hipEventRecord(ev.e1, st));
  HIPCHK(hipStreamSynchronize(st));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  *ms_per_launch = (double)ms / reps;
#ifdef PPCX_TRACE
  if (const char* path = getenv("PPCX_TRACE_FILE")) {       // one more launch, stamped; the stamps go to the file as raw uint64
    const size_t n = (size_t)kTraceBlocks * 4 * kTracePasses * kTraceStamps;
    HIPCHK(hipMalloc(&g_trace_dev, sizeof(unsigned long long) * n));
    HIPCHK(hipMemset(g_trace_dev, 0, sizeof(unsigned long long) * n));
    rc = one();
    HIPCHK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(n);
    HIPCHK(hipMemcpy(h.data(), g_trace_dev, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    (void)hipFree(g_trace_dev); g_trace_dev = nullptr;
    if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), sizeof(unsigned long long), n, f); fclose(f); }
    if (rc != PPCX_OK) return rc;
  }
#endif
  return PPCX_OK;
}
#endif

static void fit_attach(ppcx_fit* f, ppcx_model* m) { f->m = m; m->live_fits++; }
extern "C" void ppcx_fit_free(ppcx_fit* f) {
  if (!f) return;
  ppcx_model* m = f->m;
  (void)hipSetDevice(m->device);
  (void)hipFree(f->d_draws); (void)hipFree(f->d_lp); (void)hipFree(f->d_stepsize); (void)hipFree(f->d_accept);
  (void)hipFree(f->d_treedepth); (void)hipFree(f->d_nleap); (void)hipFree(f->d_div);
  delete f;
  if (--m->live_fits == 0 && m->destroy_requested) ppcx_model_destroy(m);
}

// ---- direct exchange between the ranks of a gene-sharded run (ppcx_kernels.h XchgArgs) ---------------------------------
struct ppcx_xchg {
  int device = 0, nranks = 1, rank = 0, max_chains = 0;
  void* local = nullptr; size_t bytes = 0;       // this rank's receive buffer: sums, then sequence numbers and abort words
  void* peer[kMaxRanks] = {};                    // every rank's buffer as this process sees it (peer[rank] = local)
  bool opened[kMaxRanks] = {};                   // mapped through an IPC handle (to be closed)
  bool connected = false;
  unsigned epoch = 0;                            // fits run over this group
  double timeout_s = 20.0;
  std::mutex* mu = nullptr;
};
static size_t xchg_bytes(int nranks, int max_chains) { return sizeof(double) * xchg_recv_doubles(nranks, max_chains) + sizeof(unsigned long long) * xchg_flag_words(nranks, max_chains); }
extern "C" int ppcx_xchg_create(int device, int nranks, int rank, int max_chains, ppcx_xchg** out) {
  if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || max_chains < 1 || max_chains > 1024)
    return fail(PPCX_ERR_ARG, "need 1 <= nranks <= 16, 0 <= rank < nranks, 1 <= max_chains <= 1024");
  *out = nullptr;
  HIPCHK(hipSetDevice(device));
  ppcx_xchg* x = new ppcx_xchg();
  x->device = device; x->nranks = nranks; x->rank = rank; x->max_chains = max_chains; x->bytes = xchg_bytes(nranks, max_chains);
  // uncached device memory: a peer's stores must be seen by loads of a kernel that is already running
  hipError_t e = hipExtMallocWithFlags(&x->local, x->bytes, hipDeviceMallocUncached);
  if (e == hipSuccess) e = hipMemset(x->local, 0, x->bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(x->local); delete x; return fail(PPCX_ERR_HIP, std::string("exchange buffer: ") + hipGetErrorString(e)); }
  x->peer[rank] = x->local;
  if (nranks == 1) x->connected = true;
  *out = x;
  return PPCX_OK;
}
extern "C" int ppcx_xchg_handle(ppcx_xchg* x, char* out64) {
  if (!x || !out64) return fail(PPCX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(x->device));
  hipIpcMemHandle_t h;
  HIPCHK(hipIpcGetMemHandle(&h, x->local));
  static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
  memcpy(out64, &h, 64);
  return PPCX_OK;
}
extern "C" int ppcx_xchg_connect(ppcx_xchg* x, const char* handles) {
  if (!x || !handles) return fail(PPCX_ERR_ARG, "NULL argument");
  if (x->connected) return fail(PPCX_ERR_ARG, "the exchange group is connected already");
  HIPCHK(hipSetDevice(x->device));
  for (int k = 0; k < x->nranks; ++k) {
    if (k == x->rank) continue;
    hipIpcMemHandle_t h; memcpy(&h, handles + (size_t)k * 64, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(PPCX_ERR_HIP, "hipIpcOpenMemHandle (rank " + std::to_string(k) + "): " + hipGetErrorString(e));
    x->peer[k] = p; x->opened[k] = true;
  }
  x->connected = true;
  return PPCX_OK;
}
// ranks that live in ONE process (host threads, a shard model each -- on one device or several): plain device pointers
extern "C" int ppcx_xchg_connect_local(ppcx_xchg** group, int n) {
  if (!group || n < 1 || n > kMaxRanks) return fail(PPCX_ERR_ARG, "bad group");
  for (int k = 0; k < n; ++k) if (!group[k] || group[k]->nranks != n || group[k]->rank != k || group[k]->connected || group[k]->max_chains != group[0]->max_chains)
    return fail(PPCX_ERR_ARG, "group[k] must be the unconnected rank k of n, all with the same max_chains");
  for (int k = 0; k < n; ++k) {
    for (int j = 0; j < n; ++j) {
      if (group[k]->device != group[j]->device) {
        (void)hipSetDevice(group[k]->device);
        hipError_t e = hipDeviceEnablePeerAccess(group[j]->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(PPCX_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
      }
      group[k]->peer[j] = group[j]->local;
    }
    group[k]->connected = true;
  }
  return PPCX_OK;
}
extern "C" int ppcx_xchg_set_timeout(ppcx_xchg* x, double seconds) {
  if (!x || !(seconds > 0)) return fail(PPCX_ERR_ARG, "bad arguments");
  x->timeout_s = seconds;
  return PPCX_OK;
}
extern "C" void ppcx_xchg_destroy(ppcx_xchg* x) {
  if (!x) return;
  (void)hipSetDevice(x->device);
  for (int k = 0; k < x->nranks; ++k) if (x->opened[k] && x->peer[k]) (void)hipIpcCloseMemHandle(x->peer[k]);
  (void)hipFree(x->local);
  delete x;
}
static void xchg_fill(const ppcx_xchg* x, XchgArgs* a) {
  a->nranks = x->nranks; a->rank = x->rank; a->max_chains = x->max_chains; a->chain0 = 0; a->epoch = x->epoch;
  a->timeout_ticks = (long long)(x->timeout_s * 1e8);
  const size_t nd = xchg_recv_doubles(x->nranks, x->max_chains);
  for (int k = 0; k < kMaxRanks; ++k) {
    a->recv[k] = k < x->nranks ? (double*)x->peer[k] : nullptr;
    a->flags[k] = k < x->nranks ? (unsigned long long*)((double*)x->peer[k] + nd) : nullptr;
  }
}

static int fit_nuts_impl(ppcx_model* m, const ppcx_nuts_config* cfg, ppcx_xchg* xg, ppcx_fit** out);
extern "C" int ppcx_fit_nuts(ppcx_model* m, const ppcx_nuts_config* cfg, ppcx_fit** out) { return fit_nuts_impl(m, cfg, nullptr, out); }
// One gene shard per rank, the ranks' sums added by the state machines themselves (direct exchange): the pipelined round of
// ppcx_fit_nuts with one more step inside the merged launch. Every rank calls it with the same configuration.
extern "C" int ppcx_fit_nuts_xchg(ppcx_model* shard, const ppcx_nuts_config* cfg, ppcx_xchg* xg, ppcx_fit** out) {
  if (!xg) return fail(PPCX_ERR_ARG, "exchange group is NULL");
  return fit_nuts_impl(shard, cfg, xg, out);
}
static int fit_nuts_impl(ppcx_model* m, const ppcx_nuts_config* cfg, ppcx_xchg* xg, ppcx_fit** out) {
  if (!m || !cfg || !out) return fail(PPCX_ERR_ARG, "NULL argument");
  *out = nullptr;
  if (cfg->chains < 1 || cfg->chains > 1024 || cfg->iter < 1 || cfg->warmup < 0 || cfg->warmup > cfg->iter)
    return fail(PPCX_ERR_ARG, "need 1<=chains<=1024, iter>=1, 0<=warmup<=iter");
  if (cfg->max_treedepth < 1 || cfg->max_treedepth > kMaxDepth) return fail(PPCX_ERR_LIMIT, "max_treedepth must be in 1..10");
  HIPCHK(hipSetDevice(m->device));
  const int nch = cfg->chains, D = m->d.D, iter = cfg->iter, n_keep = cfg->iter - cfg->warmup;
  choose_launch(m, (xg && xg->nranks > 1) ? nch : fit_launch_chains(m, nch));   // (between ranks: one group, below)
  // Round structure. Pipelined (default where it applies): two launches per leapfrog, the state machine beside the
  // log-likelihood workgroups (ppcx_kernels.hip, "Pipelined rounds"). It needs a model whose cells read the anticipated
  // constants only (no per-cell linear predictor). The choice must not depend on the number of chains: the two round
  // structures sum the kinetic energy of fresh momenta in different orders, and a chain's draws may not depend on its
  // company. (With more chains than the chip holds workgroups the state machines simply run ahead of the log-likelihood
  // workgroups instead of beside them.) ppcx_model_set_rounds(m, 0, ...) selects the three-launch round.
  const bool piped = model_pipelines(m);
  // The exchange group is looked at BEFORE anything is allocated (a refused call leaves nothing behind), and the group's fit
  // counter -- the epoch in every sequence number -- moves before anything that can fail on one rank alone: a rank whose
  // allocation fails has then counted this fit like its peers, and it tells them that it has left (leave() below) instead of
  // letting them wait for the timeout, now and in every later fit of the group.
  XchgArgs xa;
  bool xa_live = false;
  if (xg) {
    if (!xg->connected) return fail(PPCX_ERR_ARG, "the exchange group is not connected");
    if (xg->device != m->device) return fail(PPCX_ERR_ARG, "the exchange group lives on another device than the shard");
    if (cfg->chains > xg->max_chains) return fail(PPCX_ERR_ARG, "more chains than the exchange group was created for");
    if (!piped) return fail(PPCX_ERR_LIMIT, "the direct exchange runs inside pipelined rounds, which ppcx_model_set_rounds (or a model too large for "
                                             "the merged launch's LDS) rules out: use ppcx_fit_nuts_comm");
    xg->epoch += 1;                             // every rank counts the fits of the group: sequence numbers of earlier fits never match
    xchg_fill(xg, &xa);
    xa_live = xg->nranks > 1;
  }
  ppcx_fit* f = nullptr;
  auto leave = [&](int rc) {                    // every failure from here on: the peers' state machines are told, the fit is freed
    const std::string msg = g_err;
    if (xa_live) { (void)launch_xchg_abort_kernel(xa, m->stream); (void)hipStreamSynchronize(m->stream); }
    if (f) ppcx_fit_free(f);
    g_err = msg;
    return rc;
  };
  f = new ppcx_fit();
  fit_attach(f, m); f->chains = nch; f->n_keep = n_keep; f->iter = iter;
  f->inv_metric.assign((size_t)nch * D, 1.0);
  NutsConfig nc;
  nc.chains = nch; nc.iter = iter; nc.warmup = cfg->warmup; nc.seed = cfg->seed; nc.adapt_delta = cfg->adapt_delta;
  nc.max_treedepth = cfg->max_treedepth; nc.init_radius = cfg->init_radius; nc.stepsize0 = cfg->stepsize0;
  nc.init_buffer = cfg->init_buffer; nc.term_buffer = cfg->term_buffer; nc.window = cfg->window;
  nc.chain_id_offset = cfg->chain_id_offset;
  f->cfg = nc;
#define FHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return leave(fail(PPCX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_))); } while (0)
  if (n_keep > 0) FHIP(hipMalloc(&f->d_draws, sizeof(double) * (size_t)nch * n_keep * D));
  if (n_keep > 0) FHIP(hipMalloc(&f->d_lp, sizeof(double) * (size_t)nch * n_keep));
  FHIP(hipMalloc(&f->d_stepsize, sizeof(double) * (size_t)nch * iter));
  FHIP(hipMalloc(&f->d_accept, sizeof(double) * (size_t)nch * iter));
  FHIP(hipMalloc(&f->d_treedepth, sizeof(int) * (size_t)nch * iter));
  FHIP(hipMalloc(&f->d_nleap, sizeof(int) * (size_t)nch * iter));
  FHIP(hipMalloc(&f->d_div, sizeof(int) * (size_t)nch * iter));
  if (n_keep > 0) FHIP(hipMemsetAsync(f->d_draws, 0, sizeof(double) * (size_t)nch * n_keep * D, m->stream));
  if (n_keep > 0) FHIP(hipMemsetAsync(f->d_lp, 0, sizeof(double) * (size_t)nch * n_keep, m->stream));
  FHIP(hipMemsetAsync(f->d_stepsize, 0, sizeof(double) * (size_t)nch * iter, m->stream));
  FHIP(hipMemsetAsync(f->d_accept, 0, sizeof(double) * (size_t)nch * iter, m->stream));
  FHIP(hipMemsetAsync(f->d_treedepth, 0, sizeof(int) * (size_t)nch * iter, m->stream));
  FHIP(hipMemsetAsync(f->d_nleap, 0, sizeof(int) * (size_t)nch * iter, m->stream));
  FHIP(hipMemsetAsync(f->d_div, 0, sizeof(int) * (size_t)nch * iter, m->stream));
  FHIP(hipStreamSynchronize(m->stream));
  // Chains can also be split into groups that run on their own streams from their own host threads
  // (ppcx_model_set_rounds): while one group sits in its memory-bound gene kernel another group's log-likelihood
  // workgroups have the CUs: measured at cfg3 / 8 chains, pipelined rounds (final kernels of round 3, mean of two fits):
  // 3.13 s per fit on one stream, 2.97 s with two groups, 2.93 s with three. Default: three groups from eight chains on, two
  // from four (whole fits at cfg3 size, one group -> two: 4 chains 2.00 -> 1.82 s, 5 chains 2.40 -> 2.07, 6 chains
  // 2.86 -> 2.37, 7 chains 3.21 -> 2.64; three chains are faster on one stream; four groups are slower everywhere). A chain's draws do not depend on the grouping (tests/test_gpu_configs.py); the per-kernel event timings of a
  // fit are only meaningful with one group (bench.py takes its roofline sample from a fit on one stream).
  int ngrp = default_stream_groups(nch);
  if (m->opt_stream_groups >= 1) ngrp = m->opt_stream_groups < nch ? m->opt_stream_groups : nch;
  // Between ranks (direct exchange) the chains run as ONE group on one stream: a chain's state machine spins inside its merged
  // launch until every peer's copy of that chain has published its sums, so the peers' launches of the SAME group must be running
  // at the same time. With several groups a rank's launch of group A can sit in front of its launch of group B while the peer
  // has them the other way round -- each state machine then waits for a launch that is queued behind the one it is waiting in,
  // until the exchange's timeout ends the fit.
  if (xa_live) ngrp = 1;
  struct Group { int c0 = 0, n = 0; Work w; RunIO io; PumpStats ps; int rc = PPCX_OK; std::string err; long long leap = 0, xticks = 0, xcount = 0; };
  std::vector<Group> grp(ngrp);
  std::atomic<int> stop{0};
  const long long max_pairs = ((long long)iter * ((1LL << cfg->max_treedepth) + 8) + 100000) * (piped ? 2 : 1);
  for (int g = 0; g < ngrp; ++g) {
    Group& G = grp[g];
    G.w.pipelined = piped;
    G.c0 = (int)((long long)nch * g / ngrp); G.n = (int)((long long)nch * (g + 1) / ngrp) - G.c0;
    G.w.xchg_chain0 = G.c0;                    // also what a progress report names the group by
    G.w.stop = &stop;
    G.w.shared_chip = ngrp > 1;
    if (xg && xg->nranks > 1) G.w.xchg = &xa;
    if (g > 0) { FHIP(hipStreamCreateWithFlags(&G.w.stream, hipStreamNonBlocking)); G.w.own_stream = true; }
    int rc = work_alloc(G.w, m, G.n);
    if (rc != PPCX_OK) return leave(rc);
    std::vector<ChainState> states(G.n);
    NutsConfig ncg = nc; ncg.chain_id_offset = nc.chain_id_offset + G.c0;
    for (int c = 0; c < G.n; ++c) state_init(states[c], ncg, c, 0);
    FHIP(hipMemcpyAsync(G.w.states[0], states.data(), sizeof(ChainState) * G.n, hipMemcpyHostToDevice, G.w.stream));
    FHIP(hipStreamSynchronize(G.w.stream));     // `states` is a host temporary
    const size_t c0 = (size_t)G.c0;
    G.io.draws = f->d_draws ? f->d_draws + c0 * n_keep * D : nullptr; G.io.draws_stride = (long)n_keep * D;
    G.io.n_keep = n_keep; G.io.iter = iter;
    G.io.lp = f->d_lp ? f->d_lp + c0 * n_keep : nullptr;
    G.io.stepsize = f->d_stepsize + c0 * iter; G.io.accept = f->d_accept + c0 * iter;
    G.io.treedepth = f->d_treedepth + c0 * iter; G.io.nleap = f->d_nleap + c0 * iter; G.io.div = f->d_div + c0 * iter;
    FHIP(hipStreamSynchronize(G.w.stream));
  }
  const auto t0 = std::chrono::steady_clock::now();
  auto run_group = [&](Group* G) {
    (void)hipSetDevice(m->device);
    G->rc = pump(m, G->w, G->n, G->io, max_pairs, true, &G->ps);
    if (G->rc != PPCX_OK) { G->err = g_err; return; }
    std::vector<ChainState> states(G->n);
    if (hipMemcpy(states.data(), current_states(G->w), sizeof(ChainState) * G->n, hipMemcpyDeviceToHost) != hipSuccess) {
      G->rc = PPCX_ERR_HIP; G->err = "reading back the chain states failed"; return;
    }
    for (int c = 0; c < G->n; ++c) { G->leap += states[c].sc.total_leapfrogs; G->xticks += states[c].xc.ticks; G->xcount += states[c].xc.count; }
#ifdef PPCX_TESTING
    { std::lock_guard<std::mutex> lk(g_sm_mutex); for (int c = 0; c < G->n; ++c) { for (int k = 0; k < 6; ++k) g_sm_ticks[k] += states[c].tr.t[k]; g_sm_rounds += states[c].tr.n; } }
#endif
    // the adapted inverse metric (what rstan::get_adaptation_info prints): the genes' coordinates, then the six hyper-parameters
    std::vector<double> hq((size_t)G->n * V_COUNT * 8);
    bool ok = hipMemcpy(hq.data(), current_hyper(G->w), sizeof(double) * hq.size(), hipMemcpyDeviceToHost) == hipSuccess;
    for (int c = 0; c < G->n && ok; ++c) {
      double* dst = f->inv_metric.data() + (size_t)(G->c0 + c) * D;
      ok = hipMemcpy(dst, G->w.vecs + ((size_t)c * V_COUNT + V_MINV) * G->w.Dpad, sizeof(double) * D, hipMemcpyDeviceToHost) == hipSuccess;
      for (int k = 0; k < 6; ++k) dst[hyper_index(m->d, k)] = hq[((size_t)c * V_COUNT + V_MINV) * 8 + k];
    }
    if (!ok) { G->rc = PPCX_ERR_HIP; G->err = "reading back the inverse metric failed"; }
  };
  {
    std::vector<std::thread> th;
    for (int g = 1; g < ngrp; ++g) th.emplace_back(run_group, &grp[g]);
    run_group(&grp[0]);
    for (auto& t : th) t.join();
  }
  const auto t1 = std::chrono::steady_clock::now();
  for (int g = 0; g < ngrp; ++g) if (grp[g].rc != PPCX_OK) {
    const int rc = grp[g].rc; const std::string e = grp[g].err;
    return leave(fail(rc, e));                   // (the peers' state machines wait for this rank: leave() tells them it has left)
  }
  f->seconds = std::chrono::duration<double>(t1 - t0).count();
  f->grad_evals = 0;
  PumpStats ps;
  for (int g = 0; g < ngrp; ++g) {
    f->grad_evals += grp[g].leap; f->xchg_ticks += grp[g].xticks; f->xchg_count += grp[g].xcount;
    ps.kA_ms_sum += grp[g].ps.kA_ms_sum; ps.kC_ms_sum += grp[g].ps.kC_ms_sum; ps.kU_ms_sum += grp[g].ps.kU_ms_sum;
    ps.kA_samples += grp[g].ps.kA_samples; ps.chain_launches += grp[g].ps.chain_launches; ps.pairs += grp[g].ps.pairs;
  }
  f->kA_samples = ps.kA_samples;
  f->kA_ms_mean = ps.kA_samples ? ps.kA_ms_sum / (double)ps.kA_samples : 0.0;
  f->kA_chain_launches_mean = ps.kA_samples ? ps.chain_launches / (double)ps.kA_samples : 0.0;
  f->kC_ms_mean = ps.kA_samples ? ps.kC_ms_sum / (double)ps.kA_samples : 0.0;
  f->kU_ms_mean = ps.kA_samples ? ps.kU_ms_sum / (double)ps.kA_samples : 0.0;
  f->launch_triples = ps.pairs;
  *out = f;
  return PPCX_OK;
}

// ---- ADVI: mean-field variational inference, the reference's default path (rstan::vb through vb_iterative,
// R/utilities.R:246-278,1487-1494; Stan's advi.hpp algorithm restated: adapt_eta over {100,10,1,0.1,0.01},
// stochastic gradient ascent with the running-squared-gradient step, ELBO every eval_elbo iterations from
// elbo_samples draws, convergence when the mean or median of the relative ELBO changes drops below tol_rel_obj) ----
struct AdviRun {
  ppcx_model* m; Work* w; int nslot, nb_advi; uint32_t k0; uint32_t draw_id = 1; double* d_acc = nullptr; double* d_omega = nullptr;
  double lp_const = 0, ent_const = 0; int elbo_samples = 100;
};
static int advi_launch(AdviRun& r, int op, int n_slots, double eta_scaled, int first_iter, uint32_t prev_draw, uint32_t draw_base,
                       double* out_draws, int out_row0) {
  AdviArgs a;
  a.d = r.m->d; a.vecs = r.w->vecs; a.Dpad = r.w->Dpad; a.hyper = r.w->hyper_vecs[0]; a.cmds = r.w->cmds[0]; a.red = r.w->red;
  a.op = op; a.n_slots = n_slots; a.first_iter = first_iter; a.eta_scaled = eta_scaled; a.k0 = r.k0; a.prev_draw = prev_draw;
  a.draw_base = draw_base; a.out_draws = out_draws; a.out_row0 = out_row0; a.omega_part = r.d_omega;
  hipError_t e = launch_advi_kernel(a, r.nb_advi, r.w->stream);
  if (e != hipSuccess) return fail(PPCX_ERR_HIP, std::string("advi kernel: ") + hipGetErrorString(e));
  return PPCX_OK;
}
static int advi_eval(AdviRun& r, int n_slots) {      // gradient evaluation of the first n_slots slots
  int rc = launch_loglik(r.m, *r.w, n_slots);
  if (rc == PPCX_OK) rc = launch_close(r.m, *r.w, n_slots);
  if (rc == PPCX_OK) { RunIO io; rc = launch_step(r.m, *r.w, n_slots, io, STEP_REDUCE); }
  return rc;
}
static int advi_elbo(AdviRun& r, double* elbo) {     // Stan advi::calc_ELBO
  HIPCHK(hipMemsetAsync(r.d_acc, 0, sizeof(double) * 4, r.w->stream));
  int left = r.elbo_samples, rc;
  while (left > 0) {
    const int nb = left < r.nslot ? left : r.nslot;
    if ((rc = advi_launch(r, ADVI_DRAW, nb, 0.0, 0, 0, r.draw_id, nullptr, 0)) != PPCX_OK) return rc;
    r.draw_id += nb;
    if ((rc = advi_eval(r, nb)) != PPCX_OK) return rc;
    AdviElboArgs ea; ea.d = r.m->d; ea.cmds = r.w->cmds[0]; ea.red = r.w->red; ea.n_slots = nb; ea.acc = r.d_acc;
    ea.omega_part = r.d_omega; ea.n_omega_parts = r.nb_advi;
    hipError_t e = launch_advi_elbo_kernel(ea, r.w->stream);
    if (e != hipSuccess) return fail(PPCX_ERR_HIP, std::string("advi elbo kernel: ") + hipGetErrorString(e));
    left -= nb;
  }
  double acc[4];
  HIPCHK(hipMemcpyAsync(acc, r.d_acc, sizeof(acc), hipMemcpyDeviceToHost, r.w->stream));
  HIPCHK(hipStreamSynchronize(r.w->stream));
  if (acc[1] < 1.0) return fail(PPCX_ERR_INIT, "ADVI: every ELBO evaluation was non-finite");
  *elbo = acc[0] / (double)r.elbo_samples + r.lp_const * (acc[1] / (double)r.elbo_samples) + r.ent_const + acc[3];
  return PPCX_OK;
}
// draw the next gradient sample into slot 0 and evaluate it
static int advi_fresh_grad(AdviRun& r, uint32_t* id) {
  *id = r.draw_id++;
  int rc = advi_launch(r, ADVI_DRAW, 1, 0.0, 0, 0, *id, nullptr, 0);
  return rc != PPCX_OK ? rc : advi_eval(r, 1);
}
// one stochastic-gradient step (uses the gradient at draw *id), then draw + evaluate the next sample
static int advi_step(AdviRun& r, double eta, int iter_counter, uint32_t* id) {
  const uint32_t next = r.draw_id++;
  int rc = advi_launch(r, ADVI_STEP, 1, eta / sqrt((double)iter_counter), iter_counter == 1, *id, next, nullptr, 0);
  *id = next;
  return rc != PPCX_OK ? rc : advi_eval(r, 1);
}

extern "C" void ppcx_advi_config_default(ppcx_advi_config* c) {
  if (!c) return;
  c->output_samples = 1000; c->iter = 50000; c->tol_rel_obj = 0.005; c->grad_samples = 1; c->elbo_samples = 100;
  c->eval_elbo = 100; c->adapt_iter = 50; c->seed = 1; c->init_radius = 2.0;
}

extern "C" int ppcx_fit_advi(ppcx_model* m, const ppcx_advi_config* cfg, ppcx_fit** out) {
  if (!m || !cfg || !out) return fail(PPCX_ERR_ARG, "NULL argument");
  *out = nullptr;
  if (cfg->output_samples < 1 || cfg->iter < 1 || cfg->elbo_samples < 1 || cfg->eval_elbo < 1 || cfg->adapt_iter < 1 || !(cfg->tol_rel_obj > 0))
    return fail(PPCX_ERR_ARG, "bad ADVI configuration");
  if (cfg->grad_samples != 1) return fail(PPCX_ERR_LIMIT, "grad_samples must be 1 (the reference's value)");
  HIPCHK(hipSetDevice(m->device));
  const Dims& d = m->d;
  const int D = d.D;
  AdviRun r; r.m = m;
  r.nslot = cfg->elbo_samples < 32 ? cfg->elbo_samples : 32;
  choose_launch(m, r.nslot);
  Work w; r.w = &w;
  int rc = work_alloc(w, m, r.nslot);
  if (rc != PPCX_OK) return rc;
  r.nb_advi = (D + 255) / 256; if (r.nb_advi > 1024) r.nb_advi = 1024;
  r.k0 = seed32(cfg->seed); r.elbo_samples = cfg->elbo_samples;
  const double HL2PI = 0.91893853320467274178;
  const int n2 = d.C > 2 ? d.C - 2 : 0;
  r.lp_const = -(6.0 + 2.0 * d.G + (double)n2 * d.K) * HL2PI - 5.0 * log(2.0) - (d.C >= 2 ? d.K * log(2.0) : 0.0) - (double)n2 * d.K * log(2.5);
  r.ent_const = 0.5 * (double)D * (1.0 + 2.0 * HL2PI);
  struct Guard { double* a = nullptr; double* b = nullptr; ~Guard() { (void)hipFree(a); (void)hipFree(b); } } guard;
  HIPCHK(hipMalloc(&guard.a, sizeof(double) * 4)); r.d_acc = guard.a;
  HIPCHK(hipMalloc(&guard.b, sizeof(double) * r.nb_advi)); r.d_omega = guard.b;
  hipStream_t st = w.stream;
  // ---- initial point: init = "random" U(-R, R), retried until the density and gradient are finite
  std::vector<double> q0(D), red(PT_COUNT);
  bool ok = false;
  for (int attempt = 0; attempt < 100 && !ok; ++attempt) {
    for (int i = 0; i < D; ++i) q0[i] = (2.0 * coord_uniform((uint32_t)i, (uint32_t)attempt, 0u, 0u, r.k0, 0x41445649u) - 1.0) * cfg->init_radius;
    Cmd c; cmd_clear(c); c.type = CMD_EVAL; c.dir = 1;
    for (int k = 0; k < 6; ++k) c.hyp_q[k] = q0[hyper_index(d, k)];
    c.hy = make_hyper(c.hyp_q, d.lambda_mu_mu);
    HIPCHK(hipMemcpyAsync(w.vecs + (size_t)V_Q1 * w.Dpad, q0.data(), sizeof(double) * D, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(w.cmds[0], &c, sizeof(Cmd), hipMemcpyHostToDevice, st));
    { RunIO io0; if ((rc = launch_update(m, w, 1, io0)) != PPCX_OK) return rc; }   // a command without a step: only the constants of the uploaded point
    if ((rc = advi_eval(r, 1)) != PPCX_OK) return rc;
    HIPCHK(hipMemcpyAsync(red.data(), w.red, sizeof(double) * PT_COUNT, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double g6[6];
    const double lp = hyper_close(d, c.hy, c.hyp_q, red[PT_LP], red.data() + PT_H0, g6);
    ok = isfinite(lp) && red[PT_NONFINITE] == 0.0;
    for (int k = 0; k < 6; ++k) ok = ok && isfinite(g6[k]);
  }
  if (!ok) return fail(PPCX_ERR_INIT, "ADVI: no finite initial point after 100 attempts");
  HIPCHK(hipMemcpyAsync(w.vecs + (size_t)V_Q0 * w.Dpad, q0.data(), sizeof(double) * D, hipMemcpyHostToDevice, st));
  {
    std::vector<double> hv((size_t)V_COUNT * 8, 0.0);
    for (int k = 0; k < 8; ++k) hv[V_MINV * 8 + k] = 1.0;
    for (int k = 0; k < 6; ++k) hv[V_Q0 * 8 + k] = q0[hyper_index(d, k)];
    HIPCHK(hipMemcpyAsync(w.hyper_vecs[0], hv.data(), sizeof(double) * hv.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  auto reset = [&]() { return advi_launch(r, ADVI_RESET, 0, 0.0, 0, 0, 0, nullptr, 0); };
  if ((rc = reset()) != PPCX_OK) return rc;
  // ---- adapt_eta
  double elbo_init = 0, elbo_best = -INFINITY, eta_best = 0;
  if ((rc = advi_elbo(r, &elbo_init)) != PPCX_OK) return rc;
  const double eta_seq[5] = {100, 10, 1, 0.1, 0.01};
  bool tuned = false;
  for (int e = 0; e < 5 && !tuned; ++e) {
    uint32_t id;
    if ((rc = advi_fresh_grad(r, &id)) != PPCX_OK) return rc;
    for (int it = 1; it <= cfg->adapt_iter; ++it) if ((rc = advi_step(r, eta_seq[e], it, &id)) != PPCX_OK) return rc;
    double elbo = -INFINITY;
    if (advi_elbo(r, &elbo) != PPCX_OK || !isfinite(elbo)) elbo = -INFINITY;
    if (elbo < elbo_best && elbo_best > elbo_init) tuned = true;
    else if (e < 4) { elbo_best = elbo; eta_best = eta_seq[e]; }
    else { if (elbo > elbo_init) { eta_best = eta_seq[e]; tuned = true; } else return fail(PPCX_ERR_STEPSIZE, "ADVI: all proposed step-sizes failed"); }
    if ((rc = reset()) != PPCX_OK) return rc;
  }
  // ---- stochastic gradient ascent
  int cb_size = (int)fmax(0.1 * cfg->iter / cfg->eval_elbo, 2.0);
  std::vector<double> cb;
  double elbo = 0, elbo_prev = -INFINITY;
  uint32_t id;
  if ((rc = advi_fresh_grad(r, &id)) != PPCX_OK) return rc;
  int iters_done = 0; bool converged = false;
  for (int it = 1; it <= cfg->iter && !converged; ++it) {
    if ((rc = advi_step(r, eta_best, it, &id)) != PPCX_OK) return rc;
    iters_done = it;
    if (it % cfg->eval_elbo == 0) {
      elbo_prev = elbo;
      if ((rc = advi_elbo(r, &elbo)) != PPCX_OK) return rc;
      const double delta = fabs((elbo - elbo_prev) / elbo);
      cb.push_back(delta); if ((int)cb.size() > cb_size) cb.erase(cb.begin());
      double mean = 0; for (double x : cb) mean += x; mean /= cb.size();
      std::vector<double> srt(cb); std::sort(srt.begin(), srt.end());
      const double med = srt.size() % 2 ? srt[srt.size() / 2] : 0.5 * (srt[srt.size() / 2 - 1] + srt[srt.size() / 2]);
      if (mean < cfg->tol_rel_obj || med < cfg->tol_rel_obj) converged = true;
      if (!converged) { if ((rc = advi_fresh_grad(r, &id)) != PPCX_OK) return rc; }   // the ELBO draws used slot 0
    }
  }
  // ---- output_samples draws from the fitted approximation (kept as a one-chain fit)
  ppcx_fit* f = new ppcx_fit();
  fit_attach(f, m); f->chains = 1; f->n_keep = cfg->output_samples; f->iter = iters_done;
  memset(&f->cfg, 0, sizeof f->cfg);
#define AHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ppcx_fit_free(f); return fail(PPCX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
  AHIP(hipMalloc(&f->d_draws, sizeof(double) * (size_t)cfg->output_samples * D));
  AHIP(hipMalloc(&f->d_lp, sizeof(double) * (size_t)cfg->output_samples));
  AHIP(hipMemset(f->d_lp, 0, sizeof(double) * (size_t)cfg->output_samples));
  AHIP(hipMalloc(&f->d_stepsize, sizeof(double) * (size_t)(iters_done > 0 ? iters_done : 1)));
  AHIP(hipMemset(f->d_stepsize, 0, sizeof(double) * (size_t)(iters_done > 0 ? iters_done : 1)));
  for (int row = 0; row < cfg->output_samples; row += 64) {
    const int nb = cfg->output_samples - row < 64 ? cfg->output_samples - row : 64;
    if ((rc = advi_launch(r, ADVI_DRAW, nb, 0.0, 0, 0, r.draw_id, f->d_draws, row)) != PPCX_OK) { ppcx_fit_free(f); return rc; }
    r.draw_id += nb;
  }
  AHIP(hipStreamSynchronize(st));
  f->grad_evals = (long long)r.draw_id; f->seconds = 0; f->advi_elbo = elbo; f->advi_eta = eta_best; f->advi_converged = converged ? 1 : 0;
  *out = f;
  return PPCX_OK;
}
extern "C" int ppcx_fit_advi_info(const ppcx_fit* f, int* iterations, int* converged, double* elbo, double* eta) {
  if (!f) return fail(PPCX_ERR_ARG, "fit is NULL");
  if (iterations) *iterations = f->iter;
  if (converged) *converged = f->advi_converged;
  if (elbo) *elbo = f->advi_elbo;
  if (eta) *eta = f->advi_eta;
  return PPCX_OK;
}

// ---- gene shards (second mode; the reference's map_rect over gene shards, .stan:226-240) ---------
extern "C" int ppcx_model_create_shard_strided(int device, int G_total, int S, int C, int K_total, int g0, int gene_stride, int n_genes,
                                               const int32_t* counts_shard, const double* X, const double* exposure,
                                               double lambda_mu_mu, int n_excl, const int32_t* excl_local, ppcx_model** out) {
  if (g0 < 0 || gene_stride < 1 || n_genes < 1 || K_total < 0 || K_total > G_total ||
      (long long)g0 + (long long)gene_stride * (n_genes - 1) >= G_total) return fail(PPCX_ERR_ARG, "bad gene shard");
  // the shard's checked genes: its genes among the first K_total of the whole problem (they come first in the shard too)
  int kl = 0;
  if (g0 < K_total) kl = (K_total - g0 + gene_stride - 1) / gene_stride;
  if (kl > n_genes) kl = n_genes;
  int rc = ppcx_model_create(device, n_genes, S, C, kl, counts_shard, X, exposure, lambda_mu_mu, n_excl, excl_local, out);
  if (rc != PPCX_OK) return rc;
  (*out)->d.Gt = G_total; (*out)->d.Kt = K_total; (*out)->d.g0 = g0; (*out)->d.k0 = g0 < K_total ? g0 : K_total; (*out)->d.gstride = gene_stride;
  return PPCX_OK;
}
extern "C" int ppcx_model_create_shard(int device, int G_total, int S, int C, int K_total, int g0, int g1,
                                       const int32_t* counts_shard, const double* X, const double* exposure,
                                       double lambda_mu_mu, int n_excl, const int32_t* excl_local, ppcx_model** out) {
  if (g0 < 0 || g1 <= g0 || g1 > G_total) return fail(PPCX_ERR_ARG, "bad gene range");
  return ppcx_model_create_shard_strided(device, G_total, S, C, K_total, g0, 1, g1 - g0, counts_shard, X, exposure, lambda_mu_mu, n_excl, excl_local, out);
}

static int fit_sharded(ppcx_model** models, int ns, const ppcx_nuts_config* cfg, ppcx_comm* comm, ppcx_fit** fits) {
  if (!models || !cfg || !fits || ns < 1 || ns > kMaxShards) return fail(PPCX_ERR_ARG, "bad shard arguments");
  for (int k = 0; k < ns; ++k) { fits[k] = nullptr; if (!models[k]) return fail(PPCX_ERR_ARG, "NULL shard model"); }
  if (cfg->chains < 1 || cfg->chains > 1024 || cfg->iter < 1 || cfg->warmup < 0 || cfg->warmup > cfg->iter)
    return fail(PPCX_ERR_ARG, "need 1<=chains<=1024, iter>=1, 0<=warmup<=iter");
  if (cfg->max_treedepth < 1 || cfg->max_treedepth > kMaxDepth) return fail(PPCX_ERR_LIMIT, "max_treedepth must be in 1..10");
  const int dev = models[0]->device;
  for (int k = 0; k < ns; ++k) if (models[k]->device != dev) return fail(PPCX_ERR_ARG, "in-process shards must share a device");
  HIPCHK(hipSetDevice(dev));
  const int nch = cfg->chains, iter = cfg->iter, n_keep = cfg->iter - cfg->warmup;
  NutsConfig nc;
  nc.chains = nch; nc.iter = iter; nc.warmup = cfg->warmup; nc.seed = cfg->seed; nc.adapt_delta = cfg->adapt_delta;
  nc.max_treedepth = cfg->max_treedepth; nc.init_radius = cfg->init_radius; nc.stepsize0 = cfg->stepsize0;
  nc.init_buffer = cfg->init_buffer; nc.term_buffer = cfg->term_buffer; nc.window = cfg->window;
  nc.chain_id_offset = cfg->chain_id_offset;
  std::vector<Work> works(ns);
  std::vector<Shard> sh(ns);
  int rc = PPCX_OK;
  auto cleanup = [&]() { for (int k = 0; k < ns; ++k) { ppcx_fit_free(fits[k]); fits[k] = nullptr; } };
#define SHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(PPCX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
  for (int k = 0; k < ns; ++k) {
    ppcx_model* m = models[k];
    const int D = m->d.D;
    choose_launch(m, nch);
    ppcx_fit* f = new ppcx_fit();
    fits[k] = f;
    fit_attach(f, m); f->chains = nch; f->n_keep = n_keep; f->iter = iter; f->cfg = nc;
    if (n_keep > 0) SHIP(hipMalloc(&f->d_draws, sizeof(double) * (size_t)nch * n_keep * D));
    if (n_keep > 0) SHIP(hipMalloc(&f->d_lp, sizeof(double) * (size_t)nch * n_keep));
    SHIP(hipMalloc(&f->d_stepsize, sizeof(double) * (size_t)nch * iter));
    SHIP(hipMalloc(&f->d_accept, sizeof(double) * (size_t)nch * iter));
    SHIP(hipMalloc(&f->d_treedepth, sizeof(int) * (size_t)nch * iter));
    SHIP(hipMalloc(&f->d_nleap, sizeof(int) * (size_t)nch * iter));
    SHIP(hipMalloc(&f->d_div, sizeof(int) * (size_t)nch * iter));
    if (n_keep > 0) SHIP(hipMemset(f->d_draws, 0, sizeof(double) * (size_t)nch * n_keep * D));
    if (n_keep > 0) SHIP(hipMemset(f->d_lp, 0, sizeof(double) * (size_t)nch * n_keep));
    SHIP(hipMemset(f->d_stepsize, 0, sizeof(double) * (size_t)nch * iter));
    SHIP(hipMemset(f->d_accept, 0, sizeof(double) * (size_t)nch * iter));
    SHIP(hipMemset(f->d_treedepth, 0, sizeof(int) * (size_t)nch * iter));
    SHIP(hipMemset(f->d_nleap, 0, sizeof(int) * (size_t)nch * iter));
    SHIP(hipMemset(f->d_div, 0, sizeof(int) * (size_t)nch * iter));
    works[k].stream = models[0]->stream;        // all in-process shards are ordered on one stream
    if ((rc = work_alloc(works[k], m, nch)) != PPCX_OK) { cleanup(); return rc; }
    std::vector<ChainState> states(nch);
    for (int c = 0; c < nch; ++c) state_init(states[c], nc, c, 0);   // every shard replicates the same chains
    SHIP(hipMemcpyAsync(works[k].states[0], states.data(), sizeof(ChainState) * nch, hipMemcpyHostToDevice, works[k].stream));
    SHIP(hipStreamSynchronize(works[k].stream));   // `states` is a host temporary
    sh[k].m = m; sh[k].w = &works[k];
    RunIO& io = sh[k].io;
    io.draws = f->d_draws; io.draws_stride = (long)n_keep * D; io.n_keep = n_keep; io.iter = iter;
    io.lp = f->d_lp; io.stepsize = f->d_stepsize; io.accept = f->d_accept; io.treedepth = f->d_treedepth;
    io.nleap = f->d_nleap; io.div = f->d_div;
  }
  SHIP(hipStreamSynchronize(models[0]->stream));
  const long long max_pairs = (long long)iter * ((1LL << cfg->max_treedepth) + 8) + 100000;
  PumpStats ps;
  const auto t0 = std::chrono::steady_clock::now();
  rc = pump(sh, nch, comm, max_pairs, true, &ps);
  const auto t1 = std::chrono::steady_clock::now();
  if (rc != PPCX_OK) { cleanup(); return rc; }
  std::vector<ChainState> states(nch);
  SHIP(hipMemcpy(states.data(), current_states(works[0]), sizeof(ChainState) * nch, hipMemcpyDeviceToHost));
  long long leap = 0;
  for (int c = 0; c < nch; ++c) leap += states[c].sc.total_leapfrogs;
  for (int k = 0; k < ns; ++k) {
    ppcx_fit* f = fits[k];
    f->seconds = std::chrono::duration<double>(t1 - t0).count();
    f->grad_evals = leap;
    f->kA_samples = ps.kA_samples;
    f->kA_ms_mean = ps.kA_samples ? ps.kA_ms_sum / (double)ps.kA_samples : 0.0;
    f->kA_chain_launches_mean = ps.kA_samples ? ps.chain_launches / (double)ps.kA_samples : 0.0;
    f->kC_ms_mean = ps.kA_samples ? ps.kC_ms_sum / (double)ps.kA_samples : 0.0;
    f->kU_ms_mean = ps.kA_samples ? ps.kU_ms_sum / (double)ps.kA_samples : 0.0;
    f->launch_triples = ps.pairs;
  }
  return PPCX_OK;
}
extern "C" int ppcx_fit_nuts_shards(ppcx_model** models, int n_shards, const ppcx_nuts_config* cfg, ppcx_fit** fits) {
  return fit_sharded(models, n_shards, cfg, nullptr, fits);
}

// ---- one gene shard per process, sums all-reduced over RCCL -------------------------------------------
extern "C" int ppcx_comm_unique_id(char* out128) {
  if (!out128) return fail(PPCX_ERR_ARG, "NULL buffer");
  int rc = rccl_load();
  if (rc != PPCX_OK) return rc;
  ncclUniqueId_t id;
  const int e = g_rccl.GetUniqueId(&id);
  if (e != 0) return fail(PPCX_ERR_HIP, "ncclGetUniqueId failed");
  memcpy(out128, id.internal, 128);
  return PPCX_OK;
}
extern "C" int ppcx_comm_create(int device, int nranks, int rank, const char* id128, ppcx_comm** out) {
  if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(PPCX_ERR_ARG, "bad communicator arguments");
  *out = nullptr;
  int rc = rccl_load();
  if (rc != PPCX_OK) return rc;
  HIPCHK(hipSetDevice(device));
  ncclUniqueId_t id; memcpy(id.internal, id128, 128);
  ppcx_comm* c = new ppcx_comm();
  c->nranks = nranks; c->rank = rank; c->device = device;
  const int e = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
  if (e != 0) { delete c; return fail(PPCX_ERR_HIP, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error")); }
  if (hipMalloc(&c->d_guard, sizeof(double) * 8) != hipSuccess || hipHostMalloc(&c->h_guard, sizeof(double) * 16) != hipSuccess) {
    ppcx_comm_destroy(c); return fail(PPCX_ERR_HIP, "allocating the communicator's guard buffers failed");
  }
  *out = c;
  return PPCX_OK;
}
extern "C" void ppcx_comm_destroy(ppcx_comm* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  (void)hipFree(c->d_guard); if (c->h_guard) (void)hipHostFree(c->h_guard);
  delete c;
}
extern "C" int ppcx_fit_nuts_comm(ppcx_model* shard, const ppcx_nuts_config* cfg, ppcx_comm* comm, ppcx_fit** out) {
  if (!comm) return fail(PPCX_ERR_ARG, "communicator is NULL");
  return fit_sharded(&shard, 1, cfg, comm, out);
}

// A fit that holds draws produced elsewhere (other ranks' chains gathered by the host layer): ppcx_fit_ppc and
// ppcx_fit_get_columns then work on the pooled posterior, as rstan::summary does over merged chains (R/utilities.R:685-703).
extern "C" int ppcx_fit_from_draws(ppcx_model* m, int chains, int n_keep, const double* draws, ppcx_fit** out) {
  if (!m || !draws || !out || chains < 1 || n_keep < 1) return fail(PPCX_ERR_ARG, "bad arguments");
  *out = nullptr;
  HIPCHK(hipSetDevice(m->device));
  ppcx_fit* f = new ppcx_fit();
  fit_attach(f, m); f->chains = chains; f->n_keep = n_keep; f->iter = n_keep;
  memset(&f->cfg, 0, sizeof f->cfg);
  const size_t n = (size_t)chains * n_keep * m->d.D;
  hipError_t e = hipMalloc(&f->d_draws, sizeof(double) * n);
  if (e == hipSuccess) e = hipMemcpy(f->d_draws, draws, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e != hipSuccess) { ppcx_fit_free(f); return fail(PPCX_ERR_HIP, hipGetErrorString(e)); }
  *out = f;
  return PPCX_OK;
}

extern "C" int ppcx_fit_info(const ppcx_fit* f, int* chains, int* n_keep, int* D, int* iter) {
  if (!f) return fail(PPCX_ERR_ARG, "fit is NULL");
  if (chains) *chains = f->chains;
  if (n_keep) *n_keep = f->n_keep;
  if (D) *D = f->m->d.D;
  if (iter) *iter = f->iter;
  return PPCX_OK;
}
extern "C" int ppcx_fit_get_draws(ppcx_fit* f, double* out) {
  if (!f || !out) return fail(PPCX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(f->m->device));
  if (f->n_keep > 0) HIPCHK(hipMemcpy(out, f->d_draws, sizeof(double) * (size_t)f->chains * f->n_keep * f->m->d.D, hipMemcpyDeviceToHost));
  return PPCX_OK;
}
extern "C" int ppcx_fit_get_columns(ppcx_fit* f, int n_cols, const int32_t* cols, double* out) {
  if (!f || !cols || !out || n_cols < 1) return fail(PPCX_ERR_ARG, "bad arguments");
  const int D = f->m->d.D;
  for (int i = 0; i < n_cols; ++i) if (cols[i] < 0 || cols[i] >= D) return fail(PPCX_ERR_ARG, "column out of range");
  HIPCHK(hipSetDevice(f->m->device));
  const long rows = (long)f->chains * f->n_keep;
  if (rows == 0) return PPCX_OK;
  int* d_cols = nullptr; double* d_out = nullptr;
  HIPCHK(hipMalloc(&d_cols, sizeof(int) * n_cols));
  hipError_t e = hipMalloc(&d_out, sizeof(double) * (size_t)rows * n_cols);
  if (e != hipSuccess) { (void)hipFree(d_cols); return fail(PPCX_ERR_HIP, hipGetErrorString(e)); }
  int rc = PPCX_OK;
  do {
    if ((e = hipMemcpy(d_cols, cols, sizeof(int) * n_cols, hipMemcpyHostToDevice)) != hipSuccess) break;
    if ((e = launch_gather_kernel(f->d_draws, rows, D, d_cols, n_cols, d_out, f->m->stream)) != hipSuccess) break;
    if ((e = hipStreamSynchronize(f->m->stream)) != hipSuccess) break;
    e = hipMemcpy(out, d_out, sizeof(double) * (size_t)rows * n_cols, hipMemcpyDeviceToHost);
  } while (0);
  if (e != hipSuccess) rc = fail(PPCX_ERR_HIP, hipGetErrorString(e));
  (void)hipFree(d_cols); (void)hipFree(d_out);
  return rc;
}
extern "C" int ppcx_fit_get_diagnostics(ppcx_fit* f, double* lp, double* stepsize, int32_t* treedepth,
                                        int32_t* n_leapfrog, int32_t* divergent, double* accept) {
  if (!f) return fail(PPCX_ERR_ARG, "fit is NULL");
  HIPCHK(hipSetDevice(f->m->device));
  const size_t ni = (size_t)f->chains * f->iter, nk = (size_t)f->chains * f->n_keep;
  if (lp && nk && f->d_lp) HIPCHK(hipMemcpy(lp, f->d_lp, sizeof(double) * nk, hipMemcpyDeviceToHost));
  if (stepsize && f->d_stepsize) HIPCHK(hipMemcpy(stepsize, f->d_stepsize, sizeof(double) * ni, hipMemcpyDeviceToHost));
  if (treedepth && f->d_treedepth) HIPCHK(hipMemcpy(treedepth, f->d_treedepth, sizeof(int) * ni, hipMemcpyDeviceToHost));
  if (n_leapfrog && f->d_nleap) HIPCHK(hipMemcpy(n_leapfrog, f->d_nleap, sizeof(int) * ni, hipMemcpyDeviceToHost));
  if (divergent && f->d_div) HIPCHK(hipMemcpy(divergent, f->d_div, sizeof(int) * ni, hipMemcpyDeviceToHost));
  if (accept && f->d_accept) HIPCHK(hipMemcpy(accept, f->d_accept, sizeof(double) * ni, hipMemcpyDeviceToHost));
  return PPCX_OK;
}
extern "C" int ppcx_fit_get_inv_metric(ppcx_fit* f, double* out) {
  if (!f || !out) return fail(PPCX_ERR_ARG, "bad arguments");
  if (f->inv_metric.empty()) return fail(PPCX_ERR_ARG, "this fit has no adapted metric (not a NUTS fit)");
  memcpy(out, f->inv_metric.data(), sizeof(double) * f->inv_metric.size());
  return PPCX_OK;
}
extern "C" int ppcx_fit_get_kernel_times(ppcx_fit* f, double* loglik_ms, double* close_ms, double* update_ms,
                                         long long* launch_triples) {
  if (!f) return fail(PPCX_ERR_ARG, "fit is NULL");
  if (loglik_ms) *loglik_ms = f->kA_ms_mean;
  if (close_ms) *close_ms = f->kC_ms_mean;
  if (update_ms) *update_ms = f->kU_ms_mean;
  if (launch_triples) *launch_triples = f->launch_triples;
  return PPCX_OK;
}
extern "C" int ppcx_fit_get_xchg_timing(ppcx_fit* f, double* wait_us_per_exchange, long long* exchanges) {
  if (!f) return fail(PPCX_ERR_ARG, "fit is NULL");
  if (wait_us_per_exchange) *wait_us_per_exchange = f->xchg_count > 0 ? (double)f->xchg_ticks / 100.0 / (double)f->xchg_count : 0.0;
  if (exchanges) *exchanges = f->xchg_count;
  return PPCX_OK;
}
extern "C" int ppcx_fit_get_ppc_timing(ppcx_fit* f, double* kernel_ms, long long* nb_draws) {
  if (!f) return fail(PPCX_ERR_ARG, "fit is NULL");
  if (kernel_ms) *kernel_ms = f->ppc_ms;
  if (nb_draws) *nb_draws = f->ppc_draws;
  return PPCX_OK;
}
extern "C" int ppcx_fit_get_timing(ppcx_fit* f, double* seconds, long long* grad_evals, double* gene_kernel_ms_mean,
                                   long long* gene_kernel_samples, double* gene_kernel_chain_launches_mean) {
  if (!f) return fail(PPCX_ERR_ARG, "fit is NULL");
  if (seconds) *seconds = f->seconds;
  if (grad_evals) *grad_evals = f->grad_evals;
  if (gene_kernel_ms_mean) *gene_kernel_ms_mean = f->kA_ms_mean;
  if (gene_kernel_samples) *gene_kernel_samples = f->kA_samples;
  if (gene_kernel_chain_launches_mean) *gene_kernel_chain_launches_mean = f->kA_chain_launches_mean;
  return PPCX_OK;
}

extern "C" int ppcx_fit_ppc(ppcx_fit* f, double truncation_compensation, double p_lo, double p_hi,
                            unsigned long long seed, int n_gen, int resample, double* ci, int32_t* counts_rng) {
  if (!f || !ci) return fail(PPCX_ERR_ARG, "NULL argument");
  ppcx_model* m = f->m;
  const long n_draws = (long)f->chains * f->n_keep;
  if (n_draws < 1) return fail(PPCX_ERR_ARG, "fit holds no kept draws");
  if (m->d.K < 1) return PPCX_OK;
  if (n_gen <= 0) n_gen = (int)n_draws;
  if (!resample && n_gen > n_draws) return fail(PPCX_ERR_ARG, "n_gen exceeds the kept draws (use resample)");
  if (!(p_lo >= 0.0 && p_hi <= 1.0 && p_lo <= p_hi)) return fail(PPCX_ERR_ARG, "need 0 <= p_lo <= p_hi <= 1");
  HIPCHK(hipSetDevice(m->device));
  const int n_cells = m->d.K * m->d.S;
  double* d_ci = nullptr; int* d_rng = nullptr; int* d_scratch = nullptr;
  HIPCHK(hipMalloc(&d_ci, sizeof(double) * (size_t)n_cells * 4));
  // a cell's draws live in LDS (160 KB per CU; 4 KB of it is the kernel's static scratch) when they fit; beyond that
  // 1024 workgroups share the cells and keep the current cell's draws in their slice of a global scratch buffer
  const int kLdsDraws = 39680;
  int nblocks = n_cells;
  const bool wave_kernel = n_gen <= ppc_wave_max_draws();      // one wavefront per cell (else one workgroup per cell)
  double* d_T = nullptr;                                       // the checked genes' parameters, transposed: [K][C + 1][draws]
  {
    hipError_t e = hipMalloc(&d_T, sizeof(double) * (size_t)m->d.K * (m->d.C + 1) * (size_t)n_draws);
    if (e != hipSuccess) { (void)hipFree(d_ci); return fail(PPCX_ERR_HIP, hipGetErrorString(e)); }
  }
  if (wave_kernel) {
    nblocks = (n_cells + 3) / 4; if (nblocks > 4096) nblocks = 4096;
  } else if (n_gen > kLdsDraws) {
    nblocks = n_cells < 1024 ? n_cells : 1024;
    hipError_t e = hipMalloc(&d_scratch, sizeof(int) * (size_t)nblocks * n_gen);
    if (e != hipSuccess) { (void)hipFree(d_ci); (void)hipFree(d_T); return fail(PPCX_ERR_HIP, hipGetErrorString(e)); }
  }
  if (counts_rng) {
    hipError_t e = hipMalloc(&d_rng, sizeof(int) * (size_t)n_gen * n_cells);
    if (e != hipSuccess) { (void)hipFree(d_ci); (void)hipFree(d_scratch); (void)hipFree(d_T); return fail(PPCX_ERR_HIP, hipGetErrorString(e)); }
  }
  PpcArgs pa;
  pa.d = m->d; pa.draws = f->d_draws; pa.n_draws = n_draws; pa.exposure = m->d_expo; pa.X = m->d_X;
  pa.truncation_compensation = truncation_compensation; pa.p_lo = p_lo; pa.p_hi = p_hi; pa.k0 = seed32(seed);
  pa.n_gen = n_gen; pa.resample = resample ? 1 : 0; pa.n_cells = n_cells; pa.ci = d_ci; pa.counts_rng = d_rng; pa.scratch = d_scratch;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  (void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
  if (ev0) (void)hipEventRecord(ev0, m->stream);
  hipError_t e = hipSuccess;
  e = launch_ppc_table_kernel(f->d_draws, n_draws, m->d, truncation_compensation, d_T, m->stream);
  if (e == hipSuccess) e = wave_kernel ? launch_ppc_wave_kernel(pa, d_T, nblocks, m->stream) : launch_ppc_kernel(pa, d_T, nblocks, m->stream);
  if (ev1) (void)hipEventRecord(ev1, m->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
  if (e == hipSuccess && ev0 && ev1) { float ms = 0; if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) { f->ppc_ms = ms; f->ppc_draws = (long long)n_gen * n_cells; } }
  if (ev0) (void)hipEventDestroy(ev0);
  if (ev1) (void)hipEventDestroy(ev1);
  if (e == hipSuccess) e = hipMemcpy(ci, d_ci, sizeof(double) * (size_t)n_cells * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess && counts_rng) e = hipMemcpy(counts_rng, d_rng, sizeof(int) * (size_t)n_gen * n_cells, hipMemcpyDeviceToHost);
  (void)hipFree(d_ci); (void)hipFree(d_rng); (void)hipFree(d_scratch); (void)hipFree(d_T);
  if (e != hipSuccess) return fail(PPCX_ERR_HIP, hipGetErrorString(e));
  return PPCX_OK;
}

// vb_iterative (R/utilities.R:246-278): rstan::vb is retried until it returns; the reference passes no seed, so every
// attempt is a fresh random start. Here attempt k runs with seed + k and the retries are bounded.
static int fit_advi_iterative(ppcx_model* m, ppcx_advi_config cfg, int max_attempts, ppcx_fit** out) {
  int rc = PPCX_ERR_ARG;
  for (int k = 0; k < max_attempts; ++k) {
    rc = ppcx_fit_advi(m, &cfg, out);
    if (rc == PPCX_OK || (rc != PPCX_ERR_INIT && rc != PPCX_ERR_STEPSIZE)) return rc;
    cfg.seed += 1;
  }
  return rc;
}
extern "C" int ppcx_fit_advi_iterative(ppcx_model* m, const ppcx_advi_config* cfg, int max_attempts, ppcx_fit** out) {
  if (!m || !cfg || !out || max_attempts < 1) return fail(PPCX_ERR_ARG, "NULL argument");
  return fit_advi_iterative(m, *cfg, max_attempts, out);
}

// One NUTS pass with the CHAINS dealt to several devices of this process -- what rstan::sampling(chains, cores) does with its
// worker processes (R/utilities.R:1497-1512, :1377-1386) -- behind the .C() entry: one host thread per device creates the model
// there, runs its share of the chains (global chain ids: the chains' Philox streams do not depend on the device they run on) and
// hands back the checked genes' columns of its kept draws; the first device then holds a model of the K checked genes and computes
// the generated quantities from the pooled chains, as rstan::summary does over merged chains (:685-703). Lanes per gene are those a
// single device would choose for ALL the chains, so the result does not depend on the number of devices.
static int nuts_over_devices(const int* devs, int ndev, int G, int S, int C, int K, const int* counts, const double* X, const double* exposure,
                             double lmm, int n_excl, const int* excl, const ppcx_nuts_config& cfg0, double tc, double p_lo, double p_hi,
                             unsigned long long seed, int n_gen, int resample, double* ci, double* slope, int* counts_rng) {
  const int chains = cfg0.chains, n_keep = cfg0.iter - cfg0.warmup;
  if (n_keep < 1) return fail(PPCX_ERR_ARG, "no kept draws");
  if (ndev > chains) ndev = chains;
  const int per = (chains + ndev - 1) / ndev;
  const int n2 = C > 2 ? C - 2 : 0, nsl = C - 1 > 1 ? C - 1 : 1;
  const int Dk = 2 * K + K * nsl + 6;
  std::vector<double> pooled((size_t)chains * n_keep * Dk, 0.0);
  std::vector<int> rcs(ndev, PPCX_OK); std::vector<std::string> errs(ndev);
  auto work = [&](int r) {
    const int c0 = r * per, n = chains - c0 < per ? chains - c0 : per;
    if (n <= 0) return;
    ppcx_model* m = nullptr; ppcx_fit* f = nullptr;
    int rc = ppcx_model_create(devs[r], G, S, C, K, counts, X, exposure, lmm, n_excl, excl, &m);
    if (rc == PPCX_OK) {
      choose_launch(m, fit_launch_chains(m, chains));   // the geometry of ONE fit of all the chains
      m->L_override = m->L;
      ppcx_nuts_config cfg = cfg0; cfg.chains = n; cfg.chain_id_offset = cfg0.chain_id_offset + c0;
      rc = ppcx_fit_nuts(m, &cfg, &f);
    }
    if (rc == PPCX_OK) {
      const Dims& d = m->d;
      std::vector<int32_t> cols;
      for (int k = 0; k < 3; ++k) cols.push_back(k);
      for (int k = 0; k < K; ++k) cols.push_back(d.off_intercept + k);
      for (int k = 0; k < K; ++k) cols.push_back(d.off_alpha1 + k);
      for (int k = 0; k < n2 * K; ++k) cols.push_back(d.off_alpha2 + k);
      for (int k = 0; k < K; ++k) cols.push_back(d.off_sigma_raw + k);
      for (int k = 0; k < 3; ++k) cols.push_back(d.off_tail + k);
      rc = (int)cols.size() == Dk ? ppcx_fit_get_columns(f, Dk, cols.data(), pooled.data() + (size_t)c0 * n_keep * Dk)
                                  : fail(PPCX_ERR_ARG, "checked columns do not match the K-gene model");
    }
    if (rc != PPCX_OK) errs[r] = g_err;          // (g_err is per thread)
    rcs[r] = rc;
    ppcx_fit_free(f); ppcx_model_destroy(m);
  };
  {
    std::vector<std::thread> th;
    for (int r = 1; r < ndev; ++r) th.emplace_back(work, r);
    work(0);
    for (auto& t : th) t.join();
  }
  for (int r = 0; r < ndev; ++r) if (rcs[r] != PPCX_OK) return fail(rcs[r], "device " + std::to_string(devs[r]) + ": " + errs[r]);
  // the K checked genes on the first device: cell ids g * S + s and draw indices are those of the full model
  std::vector<int> ex_k;
  for (int e = 0; e < n_excl; ++e) if (excl[e] / S < K) ex_k.push_back(excl[e]);
  ppcx_model* mk = nullptr; ppcx_fit* fk = nullptr;
  int rc = ppcx_model_create(devs[0], K, S, C, K, counts, X, exposure, lmm, (int)ex_k.size(), ex_k.data(), &mk);
  if (rc == PPCX_OK) rc = ppcx_fit_from_draws(mk, chains, n_keep, pooled.data(), &fk);
  if (rc == PPCX_OK) rc = ppcx_fit_ppc(fk, tc, p_lo, p_hi, seed, n_gen, resample, ci, counts_rng);
  if (rc == PPCX_OK && slope) {
    for (int k = 0; k < K; ++k) {
      double s = 0;
      for (long r = 0; r < (long)chains * n_keep; ++r) s += pooled[(size_t)r * Dk + 3 + K + k];
      slope[k] = s / ((double)chains * n_keep);
    }
  }
  ppcx_fit_free(fk); ppcx_model_destroy(mk);
  return rc;
}

extern "C" void ppcx_do_inference_C(const int* dims, const int* counts, const double* X, const double* exposure,
                                    const int* excl, const double* reals, double* ci, double* slope, int* counts_rng,
                                    int* status, char** errbuf, const int* errlen) {
  if (!status) return;
  auto finish = [&](int rc) {
    *status = rc;
    if (errbuf && errbuf[0] && errlen && errlen[0] > 0) {
      const char* msg = rc == PPCX_OK ? "" : g_err.c_str();
      strncpy(errbuf[0], msg, (size_t)errlen[0] - 1);
      errbuf[0][errlen[0] - 1] = 0;
    }
  };
  if (!dims || !reals || !ci) { finish(fail(PPCX_ERR_ARG, "dims, reals and ci must not be NULL")); return; }
  if (dims[0] != PPCX_VERSION) {               // a shim written for another argument layout: nothing else is read
    finish(fail(PPCX_ERR_ARG, "ppcx_do_inference_C: dims[0] must be the ABI version the caller was written for (" + std::to_string(PPCX_VERSION) + "), got " + std::to_string(dims[0])));
    return;
  }
  dims += 1;                                   // the fields below are numbered as in include/ppcx.h, after the version
  const int device = dims[0], G = dims[1], S = dims[2], C = dims[3], K = dims[4], n_excl = dims[5];
  const int vb = dims[11], save_rng = dims[12];
  if (save_rng && !counts_rng) { finish(fail(PPCX_ERR_ARG, "save_generated_quantities without a counts_rng buffer")); return; }
  const int n_devices = dims[15];
  if (n_devices < 0 || n_devices > 16) { finish(fail(PPCX_ERR_ARG, "n_devices must be 0 .. 16")); return; }
  if (!vb && n_devices > 1 && K > 0) {          // the chains over several devices (ADVI is one chain: the first device)
    ppcx_nuts_config cfg; ppcx_nuts_config_default(&cfg);
    cfg.chains = dims[6]; cfg.iter = dims[7]; cfg.warmup = dims[8]; cfg.seed = (unsigned long long)reals[4];
    finish(nuts_over_devices(dims + 16, n_devices, G, S, C, K, counts, X, exposure, reals[0], n_excl, excl, cfg, reals[1], reals[2], reals[3],
                             (unsigned long long)reals[4], dims[9], dims[10], ci, slope, save_rng ? counts_rng : nullptr));
    return;
  }
  ppcx_model* m = nullptr; ppcx_fit* f = nullptr;
  int rc = ppcx_model_create(n_devices >= 1 ? dims[16] : device, G, S, C, K, counts, X, exposure, reals[0], n_excl, excl, &m);
  if (rc == PPCX_OK) {
    if (vb) {
      ppcx_advi_config ac; ppcx_advi_config_default(&ac);
      ac.output_samples = dims[13]; ac.iter = dims[14] > 0 ? dims[14] : 50000; ac.seed = (unsigned long long)reals[4];
      if (reals[5] > 0) ac.tol_rel_obj = reals[5];
      rc = fit_advi_iterative(m, ac, 5, &f);
    } else {
      ppcx_nuts_config cfg; ppcx_nuts_config_default(&cfg);
      cfg.chains = dims[6]; cfg.iter = dims[7]; cfg.warmup = dims[8]; cfg.seed = (unsigned long long)reals[4];
      rc = ppcx_fit_nuts(m, &cfg, &f);
    }
  }
  if (rc == PPCX_OK) rc = ppcx_fit_ppc(f, reals[1], reals[2], reals[3], (unsigned long long)reals[4], dims[9], dims[10], ci,
                                       save_rng ? counts_rng : nullptr);
  if (rc == PPCX_OK && slope && K > 0) {
    std::vector<int32_t> cols(K);
    for (int k = 0; k < K; ++k) cols[k] = m->d.off_alpha1 + k;
    int chains, n_keep; ppcx_fit_info(f, &chains, &n_keep, nullptr, nullptr);
    std::vector<double> a((size_t)chains * n_keep * K);
    rc = ppcx_fit_get_columns(f, K, cols.data(), a.data());
    if (rc == PPCX_OK) for (int k = 0; k < K; ++k) {
      double s = 0; for (long r = 0; r < (long)chains * n_keep; ++r) s += a[(size_t)r * K + k];
      slope[k] = s / ((double)chains * n_keep);
    }
  }
  ppcx_fit_free(f); ppcx_model_destroy(m);
  finish(rc);
}
