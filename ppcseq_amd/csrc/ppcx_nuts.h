// ppcx_nuts.h -- device-resident NUTS: the command protocol between the kernels of a leapfrog
// and the per-chain scalar state machine.
//
// Reference behaviour replaced: rstan::sampling(...) at R/utilities.R:1497-1512 (chains, iter, warmup=150,
// init="random", seed, no control= list => Stan defaults). rstan/Stan are third-party and not vendored
// (DESCRIPTION:32,59-60); the algorithm below is a restatement of the *published* Stan sampler
// (multinomial NUTS with the generalised U-turn criterion and its two cross-subtree checks, diagonal
// Euclidean metric, dual-averaging step size, windowed variance adaptation).
//
// MI355X-first structure: one leapfrog round = three launches on one stream.
//   * log-likelihood kernel : the gradient evaluation of one leapfrog for every chain -- the count matrix streamed once,
//     a handful of sums per gene (ppcx_gene.h lane_gene_sums);
//   * close kernel          : per gene: priors, gradient, second half kick, tree bookkeeping of the gene's coordinates
//     (U-turn dot products, subtree slots), per-workgroup partial sums into a slab;
//   * step kernel           : reduces the slab in a fixed order and runs the scalar NUTS / adaptation state machine
//     (`chain_step` / `chain_advance`), emitting the next command, and applies that command to every gene-owned coordinate
//     (proposal / sample copies, draw storage, Welford / metric updates, momentum refresh, first half kick + drift, the
//     constants of the new position): every workgroup of its grid repeats the step on the same inputs and then takes its
//     share of the coordinates (a separate update kernel does the coordinate work at initialisation and for ADVI).
//   State and commands are double-buffered between rounds. The host only pumps the launches and polls a done flag: no
//   per-leapfrog host round trip. Stan's recursive build_tree is evaluated iteratively: a completed left subtree of
//   level d parks its (rho, p_begin, p_end, proposal, log weight) in slot d until its right sibling completes.
#pragma once
#include "ppcx_model.h"

namespace ppcx {

constexpr int kMaxDepth = 10;
constexpr int kLev = kMaxDepth;

// per-coordinate vectors of one chain (each of length D, stride Dpad)
enum Vec : int {
  V_Q0 = 0, V_P0, V_G0,          // backward end of the trajectory: position, momentum, grad log p
  V_Q1, V_P1, V_G1,              // forward end
  V_RHO, V_PNEAR, V_MINV, V_WM, V_WM2, V_SQ, V_SG,
  V_C0, V_C1, V_C2, V_C3,        // constants derived from the position being evaluated (ppcx_gene.h coord_consts)
  V_LRHO,                        // slots 0..kLev-1
  V_LPBEG = V_LRHO + kLev,
  V_LPEND = V_LPBEG + kLev,
  V_LPQ = V_LPEND + kLev,
  V_LPG = V_LPQ + kLev,
  V_COUNT = V_LPG + kLev
};

enum CmdType : int { CMD_DONE = 0, CMD_EVAL = 1, CMD_EPS_TRY = 2, CMD_LEAF = 3, CMD_FLUSH = 4 };
enum PreFlag : int {
  PRE_PROP = 1, PRE_SAMPLE = 2, PRE_STORE_DRAW = 4, PRE_WELFORD = 8, PRE_METRIC = 16, PRE_INIT = 32,
  PRE_NEW_TRANSITION = 64, PRE_EPS_TRY = 128, PRE_SAVE_NEAR = 256
};

struct Cmd {
  int type;
  int dir;                       // which end this leapfrog advances (0 backward, 1 forward)
  double eps;                    // signed step
  int pre_flags;
  int pre_dir;                   // end advanced by the previous leaf (source of "leaf" proposals)
  int prop_slot, prop_src;       // PRE_PROP   : proposal[prop_slot] <- (prop_src < 0 ? end[pre_dir] : proposal[prop_src])
  int sample_src;                // PRE_SAMPLE : sample <- (sample_src < 0 ? end[pre_dir] : proposal[sample_src])
  int draw_index;                // PRE_STORE_DRAW
  int welford_n, metric_n;       // counts for PRE_WELFORD / PRE_METRIC
  unsigned rng_c1, rng_c3;       // Philox counter words for momentum / init draws
  double init_radius;
  int leaf_n, n_merge, subtree_complete;   // tree position of this leaf inside the current subtree
  int next_dir;                  // the direction the NEXT leaf advances if the tree goes on: dir inside a subtree, the next doubling's
                                 // direction at a leaf that completes one (drawn from its own Philox stream, so it is known ahead)
  double hyp_q[6];               // hyper-parameter values (unconstrained) to evaluate the genes at
  uint32_t k0, k1;               // Philox key of this chain
  // pipelined rounds (ppcx_ls_kernel / ppcx_gene_kernel): has the log-likelihood kernel evaluated this command's position
  // (the per-gene sums belong to it), and has the per-coordinate work of the command (pre-operations, first half kick,
  // drift) been applied. The classic three-launch round ignores both.
  int evaluated, updated;
  Hyper hy;                      // make_hyper(hyp_q): computed once by kernel B so kernel A holds it in SGPRs
};

struct VecRef {
  double* base; long stride;
  PPCX_HD double& at(int v, int i) const { return base[(long)v * stride + i]; }
};
struct CoordVals { double q, p, g, minv; };
// first half kick and drift of one coordinate: ph = p + eps/2 g, qn = q + eps minv ph. One definition with explicit
// fused operations, because the gene kernel evaluates it twice for the same leaf -- once ahead of the state machine's
// decision, from registers, to hand the log-likelihood kernel the constants of the next position, and once for real --
// and both must give the same bits.
PPCX_HD void kick_drift(double q, double p, double g, double eps, double minv, double* ph, double* qn) {
  const double h = fma(0.5 * eps, g, p);
  *ph = h;
  *qn = fma(eps * minv, h, q);
}
// both ends and the metric of one coordinate, fetched ahead of the state machine so that the memory latency
// overlaps it (kernel B); coord_pre loads from memory when no cache is given
struct CoordCache {
  double q[2], p[2], g[2], minv;
  // constant indices only, so that the cache stays in registers (an end picked by a run-time index would put it on the stack)
  PPCX_HD double qe(int e) const { return e ? q[1] : q[0]; }
  PPCX_HD double pe(int e) const { return e ? p[1] : p[0]; }
  PPCX_HD double ge(int e) const { return e ? g[1] : g[0]; }
};
PPCX_HD CoordCache coord_prefetch(const VecRef& v, int i) {
  CoordCache c;
  c.q[0] = v.at(V_Q0, i); c.p[0] = v.at(V_P0, i); c.g[0] = v.at(V_G0, i);
  c.q[1] = v.at(V_Q1, i); c.p[1] = v.at(V_P1, i); c.g[1] = v.at(V_G1, i);
  c.minv = v.at(V_MINV, i);
  return c;
}

// the same, restricted to what command c reads of coordinate i: the end it advances, and the other end only when a
// proposal / sample copy takes it from there (a leaf after the tree turned round)
PPCX_HD CoordCache coord_prefetch_for(const Cmd& c, const VecRef& v, int i) {
  CoordCache cc;
  const int e = c.dir, o = 1 - c.dir;
  const double qe = v.at(V_Q0 + 3 * e, i), pe = v.at(V_P0 + 3 * e, i), ge = v.at(V_G0 + 3 * e, i);
  double qo = 0.0, go = 0.0;
  if (c.pre_dir == o && (c.pre_flags & (PRE_PROP | PRE_SAMPLE))) { qo = v.at(V_Q0 + 3 * o, i); go = v.at(V_G0 + 3 * o, i); }
  cc.q[0] = e ? qo : qe; cc.q[1] = e ? qe : qo;
  cc.p[0] = e ? 0.0 : pe; cc.p[1] = e ? pe : 0.0;
  cc.g[0] = e ? go : ge; cc.g[1] = e ? ge : go;
  cc.minv = v.at(V_MINV, i);
  return cc;
}

// ---------------------------------------------------------------------------------------------------
// per-coordinate pre-operations (lazy bookkeeping decided by the previous chain_advance) followed by
// loading the coordinate's current end state. Every lane of a gene computes the values; only the
// `writer` lane stores, and nothing stored here is re-read in the same launch.
// ---------------------------------------------------------------------------------------------------
// `i` indexes the vectors, `flat` is the coordinate's column in this (shard's) unconstrained vector = draws column,
// `rid` its index in the whole problem's vector = Philox stream id. i != flat only for kernel B's LDS copy of
// the hyper coordinates; rid != flat only for gene shards.
// MASK: the pre-operations this instantiation knows (the others are compiled out); fmask: those still to do (0 after
// gene_rare_pre has done all of them in a pass of its own) -- see gene_coord_update.
constexpr int kPreCommon = PRE_PROP | PRE_SAMPLE | PRE_SAVE_NEAR;      // what a leaf inside a transition can carry
template <int MASK = ~0>
PPCX_HD CoordVals coord_pre(const Cmd& c, const VecRef& v, int i, int flat, int rid, bool writer, double* draws, int D,
                            uint32_t k0, uint32_t k1, double* T0, const CoordCache* cc = nullptr, int fmask = ~0) {
  const int f = c.pre_flags & MASK & fmask;
  if (f & PRE_PROP) {
    double q_, g_;
    if (c.prop_src < 0) { q_ = cc ? cc->qe(c.pre_dir) : v.at(V_Q0 + 3 * c.pre_dir, i); g_ = cc ? cc->ge(c.pre_dir) : v.at(V_G0 + 3 * c.pre_dir, i); }
    else { q_ = v.at(V_LPQ + c.prop_src, i); g_ = v.at(V_LPG + c.prop_src, i); }
    if (writer) { v.at(V_LPQ + c.prop_slot, i) = q_; v.at(V_LPG + c.prop_slot, i) = g_; }
  }
  double sq = 0.0, sg = 0.0;
  bool have_s = false;
  if (f & PRE_SAMPLE) {
    if (c.sample_src < 0) { sq = cc ? cc->qe(c.pre_dir) : v.at(V_Q0 + 3 * c.pre_dir, i); sg = cc ? cc->ge(c.pre_dir) : v.at(V_G0 + 3 * c.pre_dir, i); }
    else { sq = v.at(V_LPQ + c.sample_src, i); sg = v.at(V_LPG + c.sample_src, i); }
    have_s = true;
    if (writer) { v.at(V_SQ, i) = sq; v.at(V_SG, i) = sg; }
  }
  if (!have_s && (f & (PRE_STORE_DRAW | PRE_WELFORD | PRE_NEW_TRANSITION | PRE_EPS_TRY))) {
    sq = v.at(V_SQ, i); sg = v.at(V_SG, i);
  }
  if ((f & PRE_STORE_DRAW) && writer && draws) draws[(long)c.draw_index * D + flat] = sq;
  double minv = cc ? cc->minv : v.at(V_MINV, i);
  if (f & (PRE_WELFORD | PRE_METRIC)) {
    double m = v.at(V_WM, i), m2 = v.at(V_WM2, i);
    if (f & PRE_WELFORD) {                     // Welford update with the new sample (Stan welford_var_estimator)
      const double dlt = sq - m;
      m += dlt / (double)c.welford_n;
      m2 += (sq - m) * dlt;
    }
    if (f & PRE_METRIC) {                      // regularised variance -> inverse metric; restart the estimator
      const double n = (double)c.metric_n;
      const double var = m2 / (n - 1.0);
      minv = (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0));
      m = 0.0; m2 = 0.0;
      if (writer) v.at(V_MINV, i) = minv;
    }
    if (writer) { v.at(V_WM, i) = m; v.at(V_WM2, i) = m2; }
  }
  CoordVals r;
  r.minv = minv;
  if (f & PRE_INIT) {                          // init = "random": U(-R, R) on the unconstrained scale
    r.q = (2.0 * coord_uniform((uint32_t)rid, c.rng_c1, 0u, 0u, k0, k1) - 1.0) * c.init_radius;
    r.p = 0.0; r.g = 0.0;
    if (writer) { v.at(V_Q1, i) = r.q; v.at(V_P1, i) = 0.0; v.at(V_G1, i) = 0.0; }
  } else if (f & PRE_NEW_TRANSITION) {         // both ends restart at the current sample with fresh momentum
    r.q = sq; r.g = sg;
    r.p = coord_normal((uint32_t)rid, c.rng_c1, 1u, 0u, k0, k1) / sqrt(minv);
    if (writer) {
      v.at(V_Q0, i) = sq; v.at(V_Q1, i) = sq; v.at(V_G0, i) = sg; v.at(V_G1, i) = sg;
      v.at(V_P0, i) = r.p; v.at(V_P1, i) = r.p; v.at(V_RHO, i) = r.p;
      *T0 += r.p * r.p * minv;
    }
  } else if (f & PRE_EPS_TRY) {                // init_stepsize trial: forward end <- sample, fresh momentum
    r.q = sq; r.g = sg;
    r.p = coord_normal((uint32_t)rid, c.rng_c1, 3u, c.rng_c3, k0, k1) / sqrt(minv);
    if (writer) { v.at(V_Q1, i) = sq; v.at(V_G1, i) = sg; v.at(V_P1, i) = r.p; *T0 += r.p * r.p * minv; }
  } else {
    if (cc) { r.q = cc->qe(c.dir); r.p = cc->pe(c.dir); r.g = cc->ge(c.dir); }
    else { r.q = v.at(V_Q0 + 3 * c.dir, i); r.p = v.at(V_P0 + 3 * c.dir, i); r.g = v.at(V_G0 + 3 * c.dir, i); }
  }
  if ((f & PRE_SAVE_NEAR) && writer) v.at(V_PNEAR, i) = r.p;
  return r;
}

// running (rho, p_begin) of the node that the current leaf closes, per coordinate
struct NodeVals { double nrho, npbeg; };

// contribution of coordinate i to the six U-turn dot products of the merge at level d (slot d holds the
// completed left sibling). Mirrors Stan's three compute_criterion calls inside build_tree.
// Lr, Lb, Le: the parked left subtree of level d (rho, p_begin, p_end) of this coordinate
PPCX_HD void coord_merge_dots_vals(double Lr, double Lb, double Le, double p_end, double minv, NodeVals* nv, double* dots);
// the parked left subtree of level d of coordinate i: (rho, p_begin, p_end). A level-0 subtree is one leaf, whose three
// values are the same number: only V_LRHO holds it (two stores and two loads less per coordinate and leaf pair).
PPCX_HD void coord_load_slot(const VecRef& v, int i, int d, double* Lr, double* Lb, double* Le) {
  *Lr = v.at(V_LRHO + d, i);
  if (d == 0) { *Lb = *Lr; *Le = *Lr; }
  else { *Lb = v.at(V_LPBEG + d, i); *Le = v.at(V_LPEND + d, i); }
}
PPCX_HD void coord_merge_dots(const VecRef& v, int i, int d, double p_end, double minv, NodeVals* nv, double* dots) {
  double Lr, Lb, Le;
  coord_load_slot(v, i, d, &Lr, &Lb, &Le);
  coord_merge_dots_vals(Lr, Lb, Le, p_end, minv, nv, dots);
}
PPCX_HD void coord_merge_dots_vals(double Lr, double Lb, double Le, double p_end, double minv, NodeVals* nv, double* dots) {
  const double pes = minv * p_end;
  const double rs = Lr + nv->nrho;              // rho_subtree = rho_init + rho_final
  dots[0] += (minv * Lb) * rs;                  // p_sharp_beg . rho_subtree
  dots[1] += pes * rs;                          // p_sharp_end . rho_subtree
  const double e1 = Lr + nv->npbeg;             // rho_init + p_final_beg
  dots[2] += (minv * Lb) * e1;                  // p_sharp_beg
  dots[3] += (minv * nv->npbeg) * e1;           // p_sharp_final_beg
  const double e2 = nv->nrho + Le;              // rho_final + p_init_end
  dots[4] += (minv * Le) * e2;                  // p_sharp_init_end
  dots[5] += pes * e2;                          // p_sharp_end
  nv->nrho = rs; nv->npbeg = Lb;
}
// park the node closed by this leaf in slot m (it is a left child at level m)
PPCX_HD void coord_store_slot(const VecRef& v, int i, int m, double p_end, const NodeVals& nv) {
  v.at(V_LRHO + m, i) = nv.nrho;
  if (m > 0) { v.at(V_LPBEG + m, i) = nv.npbeg; v.at(V_LPEND + m, i) = p_end; }     // level 0: all three are p_end (coord_load_slot)
}
// the subtree is complete: the three top-level criteria of base_nuts::transition, and rho += rho_subtree
PPCX_HD void coord_top_dots(const VecRef& v, int i, int dir, double p_end, double minv, const NodeVals& nv, double* top) {
  const double rho_old = v.at(V_RHO, i), far = v.at(V_P0 + 3 * (1 - dir), i), near = v.at(V_PNEAR, i);
  const double pes = minv * p_end;
  const double rt = rho_old + nv.nrho;
  top[0] += (minv * far) * rt;
  top[1] += pes * rt;
  const double e1 = rho_old + nv.npbeg;         // old tree + first leaf of the new subtree
  top[2] += (minv * far) * e1;
  top[3] += (minv * nv.npbeg) * e1;
  const double e2 = nv.nrho + near;             // new subtree + adjacent end of the old tree
  top[4] += (minv * near) * e2;
  top[5] += pes * e2;
  v.at(V_RHO, i) = rt;
}

// ---------------------------------------------------------------------------------------------------
// partial sums exchanged from kernel A to kernel B (per block, then reduced in a fixed order)
// ---------------------------------------------------------------------------------------------------
enum Part : int {
  PT_LP = 0, PT_H0 = 1,          // 1..6 hyper-gradient sums
  PT_T0 = 7, PT_T1 = 8, PT_NONFINITE = 9,
  PT_DOTS = 10,                  // 6 per level
  PT_TOP = PT_DOTS + 6 * kLev,
  PT_COUNT = PT_TOP + 6
};
PPCX_HD int parts_used(const Cmd& c) {
  if (c.type == CMD_LEAF) return c.subtree_complete ? PT_COUNT : PT_DOTS + 6 * c.n_merge;
  return PT_DOTS;
}

struct NutsConfig {
  int chains, iter, warmup;
  unsigned long long seed;
  double adapt_delta; int max_treedepth; double init_radius, stepsize0;
  int init_buffer, term_buffer, window;
  int chain_id_offset;           // global id of local chain 0 (multi-GPU: rank * chains)
};

enum Phase : int { PH_START = 0, PH_EVAL_ONLY, PH_INIT, PH_EPS, PH_TREE, PH_FLUSH, PH_DONE };

// scalar part of a chain's state: lives in registers while the state machine runs
struct ChainScalars {
  int phase, error;
  int eval_only;
  uint32_t k0, k1;
  int iter, warmup, max_depth;
  double adapt_delta, init_radius;
  int init_buffer, term_buffer, window, adapt_windows;   // adapt_windows: warmup >= 20
  int it, init_attempt;
  int eps_dir, eps_attempt, eps_call;
  double eps;
  int depth, leaf_n, dir, n_leapfrog, divergent;
  uint32_t rng_j;
  double H0, lsw_tree, sum_metro;
  double V_sample;
  double T0h;                    // kinetic energy of the fresh hyper momenta
  double T0g; int T0g_held;      // pipelined rounds: the gene coordinates' share of it, taken when the command is carried (the
                                 // round after the one that drew the momenta), so that it does not depend on the slab's
                                 // layout at the round that closes the command
  double mu, s_bar, x_bar; int da_counter;
  int win_next, win_size, win_counter, wn;
  double lp_eval;                // CMD_EVAL result
  long long total_leapfrogs;
};
// gene shards with the direct exchange (ppcx_kernels.hip xchg_sums): exchanges this chain has taken part in -- the sequence
// number its ranks stamp their contributions with -- and the wall-clock ticks (100 MHz) its state machine has waited for peers.
// Kept apart from ChainScalars: those live in registers while a state machine runs, and ppcx_step_kernel has none to spare.
struct XchgCount { unsigned count; unsigned pad_; long long ticks; };

// per-level log weights / potentials of the parked left subtrees (indexed at run time: kept in LDS)
struct TreeArrays { double Llsw[kLev + 1], LV[kLev + 1]; };
#ifdef PPCX_TESTING
// testing build: 100 MHz ticks the chain's state machine (step_role_pipelined) has spent in its phases, and the rounds counted
struct SmTrace { long long t[6]; long long n; };
struct ChainState { ChainScalars sc; TreeArrays ta; XchgCount xc; SmTrace tr; };
#else
struct ChainState { ChainScalars sc; TreeArrays ta; XchgCount xc; };
#endif

struct Reduced {
  double lp_genes, hsum[6], T0, T1, nonfinite;
  double dots[kLev][6], top[6];
};

struct ChainOut {                // per-chain diagnostic arrays (device pointers; may be null)
  double* lp;                    // [n_keep]
  double* stepsize; int* treedepth; int* n_leapfrog; int* divergent; double* accept;   // [iter]
};

PPCX_HD double tree_uniform(ChainScalars& st) {
  return coord_uniform(st.rng_j++, (uint32_t)st.it, 2u, 0u, st.k0, st.k1);
}
PPCX_HD bool all_positive(const double* d6) {
  return d6[0] > 0 && d6[1] > 0 && d6[2] > 0 && d6[3] > 0 && d6[4] > 0 && d6[5] > 0;
}

PPCX_HD void state_init(ChainState& cs, const NutsConfig& cfg, int local_chain, int eval_only) {
  ChainScalars& st = cs.sc;
  st.phase = PH_START; st.error = 0; st.eval_only = eval_only;
  st.k0 = seed32(cfg.seed); st.k1 = (uint32_t)(cfg.chain_id_offset + local_chain);
  st.iter = cfg.iter; st.warmup = cfg.warmup; st.max_depth = cfg.max_treedepth > kMaxDepth ? kMaxDepth : cfg.max_treedepth;
  st.adapt_delta = cfg.adapt_delta; st.init_radius = cfg.init_radius;
  st.init_buffer = cfg.init_buffer; st.term_buffer = cfg.term_buffer; st.window = cfg.window;
  st.adapt_windows = cfg.warmup >= 20;
  if (!st.adapt_windows) { st.init_buffer = st.term_buffer = st.window = 0; }
  else if (st.init_buffer + st.window + st.term_buffer > cfg.warmup) {   // Stan windowed_adaptation ctor
    st.init_buffer = (int)(0.15 * cfg.warmup); st.term_buffer = (int)(0.1 * cfg.warmup);
    st.window = cfg.warmup - (st.init_buffer + st.term_buffer);
  }
  st.it = 0; st.init_attempt = 0; st.eps_dir = 0; st.eps_attempt = 0; st.eps_call = 0; st.eps = cfg.stepsize0;
  st.depth = 0; st.leaf_n = 0; st.dir = 1; st.n_leapfrog = 0; st.divergent = 0; st.rng_j = 0;
  st.H0 = 0; st.lsw_tree = 0; st.sum_metro = 0; st.V_sample = 0; st.T0h = 0; st.T0g = 0; st.T0g_held = 0;
  st.mu = 0; st.s_bar = 0; st.x_bar = 0; st.da_counter = 0;
  st.win_next = st.init_buffer + st.window - 1; st.win_size = st.window; st.win_counter = 0; st.wn = 0;
  st.lp_eval = 0; st.total_leapfrogs = 0;
  cs.xc.count = 0; cs.xc.pad_ = 0; cs.xc.ticks = 0;
  for (int d = 0; d <= kLev; ++d) { cs.ta.Llsw[d] = 0; cs.ta.LV[d] = 0; }
}

PPCX_HD void cmd_clear(Cmd& c) {
  c.type = CMD_DONE; c.dir = 1; c.eps = 0.0; c.pre_flags = 0; c.pre_dir = 1; c.prop_slot = 0; c.prop_src = -1;
  c.sample_src = -1; c.draw_index = 0; c.welford_n = 0; c.metric_n = 0; c.rng_c1 = 0; c.rng_c3 = 0;
  c.init_radius = 0; c.leaf_n = 0; c.n_merge = 0; c.subtree_complete = 0; c.next_dir = 1;
  for (int k = 0; k < 6; ++k) c.hyp_q[k] = 0.0;
  c.k0 = 0; c.k1 = 0; c.evaluated = 0; c.updated = 0;
}
// Pipelined rounds: after closing leaf `ex` the gene kernel assumes that the tree goes on -- the next leaf of the same
// subtree, or the first leaf of the next doubling in the direction ex.next_dir -- and writes the constants of that position
// for the log-likelihood launch that runs beside the state machine. True unless the transition ends here.
PPCX_HD bool spec_continues(const Cmd& ex, const Cmd& nc) {
  return ex.type == CMD_LEAF && nc.type == CMD_LEAF && nc.dir == ex.next_dir &&
         nc.eps == (ex.next_dir == ex.dir ? ex.eps : -ex.eps) &&
         (nc.pre_flags & (PRE_NEW_TRANSITION | PRE_INIT | PRE_EPS_TRY | PRE_METRIC)) == 0;
}
// does a command wait for a gradient evaluation (as opposed to CMD_FLUSH / CMD_DONE, which only move data)
PPCX_HD bool cmd_evaluates(const Cmd& c) { return c.type == CMD_EVAL || c.type == CMD_EPS_TRY || c.type == CMD_LEAF; }

// ----- helpers that fill in the next command ---------------------------------------------------------
PPCX_HD void issue_eps_try(ChainScalars& st, Cmd& nc) {
  nc.type = CMD_EPS_TRY; nc.pre_flags |= PRE_EPS_TRY; nc.dir = 1; nc.eps = st.eps;
  // (Round 3 had an opaque asm barrier on st.eps_attempt here: hipcc ROCm 7.2 emitted 0 for rng_c3 on the halving / doubling
  // path of PH_EPS in ppcx_step_kernel. Round 4 traced it: the optimised LLVM IR is correct, the AMDGPU backend drops the
  // copy on one of the two predecessor paths after AMDGPUCodeGenPrepare has broken the SLP-made <2 x i32> phi of
  // (rng_c1, rng_c3) into scalars; the library is built without SLP vectorisation -- ppcseq_amd/build.py, DESIGN.md
  // section 3, profiles/r04_miscompile/. Round 5 keeps the barrier as well: it costs nothing measurable, and a build of these
  // sources with other flags -- a packager's Makefile, an R package build -- must not get the wrong Philox streams silently.)
  PPCX_OPAQUE(st.eps_attempt);
  nc.rng_c1 = (unsigned)st.eps_call; nc.rng_c3 = (unsigned)st.eps_attempt;
  st.eps_attempt++;
  st.phase = PH_EPS;
}
PPCX_HD void start_eps_heuristic(ChainScalars& st, Cmd& nc) {   // Stan base_hmc::init_stepsize
  st.eps_dir = 0; st.eps_attempt = 0;
  if (st.eps == 0 || st.eps > 1e7 || isnan(st.eps)) { st.error = 2; nc.type = CMD_DONE; st.phase = PH_DONE; return; }
  issue_eps_try(st, nc);
}
// direction of doubling number `depth` (0-based) of the current transition: forward if the uniform of Philox counter
// (depth, iteration, 7, 0) exceeds 1/2. A stream of its own -- not the transition's sequence of scalar uniforms, whose
// consumption depends on the data -- so that the direction of the NEXT doubling is known while the current one is built
// (the gene kernel anticipates the first leaf of the next doubling, see Cmd::next_dir).
PPCX_HD int doubling_dir(const ChainScalars& st, int depth) {
  return coord_uniform((uint32_t)depth, (uint32_t)st.it, 7u, 0u, st.k0, st.k1) > 0.5 ? 1 : 0;
}
PPCX_HD void set_leaf(ChainScalars& st, Cmd& nc, int leaf_n) {
  st.leaf_n = leaf_n;
  nc.type = CMD_LEAF; nc.dir = st.dir; nc.eps = st.dir ? st.eps : -st.eps;
  nc.leaf_n = leaf_n;
  int m = 0;
  while (m < st.depth && ((leaf_n >> m) & 1) == 0) ++m;          // trailing zeros, capped at the subtree depth
  nc.n_merge = m;
  nc.subtree_complete = (leaf_n == (1 << st.depth));
  nc.next_dir = nc.subtree_complete ? doubling_dir(st, st.depth + 1) : st.dir;
  st.phase = PH_TREE;
}
PPCX_HD void start_doubling(ChainScalars& st, Cmd& nc) {
  st.dir = doubling_dir(st, st.depth);
  nc.pre_flags |= PRE_SAVE_NEAR;
  set_leaf(st, nc, 1);
}
PPCX_HD void start_transition(ChainScalars& st, Cmd& nc) {
  nc.pre_flags |= PRE_NEW_TRANSITION;
  nc.rng_c1 = (unsigned)st.it;
  st.depth = 0; st.rng_j = 0; st.lsw_tree = 0.0; st.sum_metro = 0.0; st.n_leapfrog = 0; st.divergent = 0;
  start_doubling(st, nc);
}

// A transition has ended: diagnostics, adaptation (Stan adapt_diag_e_nuts::transition), next command.
PPCX_HD void end_transition(ChainScalars& st, Cmd& nc, const ChainOut& out) {
  const int it = st.it;
  const double accept = st.sum_metro / (double)st.n_leapfrog;
  if (out.stepsize) out.stepsize[it] = st.eps;
  if (out.treedepth) out.treedepth[it] = st.depth;
  if (out.n_leapfrog) out.n_leapfrog[it] = st.n_leapfrog;
  if (out.divergent) out.divergent[it] = st.divergent;
  if (out.accept) out.accept[it] = accept;
  bool metric_updated = false;
  if (it < st.warmup) {
    // dual averaging (stepsize_adaptation::learn_stepsize; gamma 0.05, t0 10, kappa 0.75)
    ++st.da_counter;
    const double as = accept > 1.0 ? 1.0 : accept;
    const double eta = 1.0 / ((double)st.da_counter + 10.0);
    st.s_bar = (1.0 - eta) * st.s_bar + eta * (st.adapt_delta - as);
    const double x = st.mu - st.s_bar * sqrt((double)st.da_counter) / 0.05;
    const double x_eta = pow((double)st.da_counter, -0.75);
    st.x_bar = (1.0 - x_eta) * st.x_bar + x_eta * x;
    st.eps = exp(x);
    if (st.adapt_windows) {                    // windowed_adaptation / var_adaptation::learn_variance
      const int W = st.warmup, cw = st.win_counter;
      const bool in_window = (cw >= st.init_buffer) && (cw < W - st.term_buffer) && (cw != W);
      if (in_window) { ++st.wn; nc.pre_flags |= PRE_WELFORD; nc.welford_n = st.wn; }
      const bool end_window = (cw == st.win_next) && (cw != W);
      if (end_window) {
        if (st.win_next != W - st.term_buffer - 1) {            // compute_next_window
          st.win_size *= 2;
          st.win_next = cw + st.win_size;
          if (st.win_next != W - st.term_buffer - 1) {
            const int boundary = st.win_next + 2 * st.win_size;
            if (boundary >= W - st.term_buffer) st.win_next = W - st.term_buffer - 1;
          }
        }
        nc.pre_flags |= PRE_METRIC; nc.metric_n = st.wn; st.wn = 0;
        metric_updated = true;
      }
      ++st.win_counter;
    }
    if (it == st.warmup - 1 && !metric_updated) st.eps = exp(st.x_bar);   // complete_adaptation
  } else {
    const int k = it - st.warmup;
    nc.pre_flags |= PRE_STORE_DRAW; nc.draw_index = k;
    if (out.lp) out.lp[k] = -st.V_sample;
  }
  st.it = it + 1;
  if (st.it >= st.iter) { nc.type = CMD_FLUSH; st.phase = PH_FLUSH; return; }
  if (metric_updated) { st.eps_call++; start_eps_heuristic(st, nc); }
  else start_transition(st, nc);
}

// ---------------------------------------------------------------------------------------------------
// The scalar state machine. `ex` is the command kernel A just executed, `rd` its reduced partial sums
// (hyper-coordinate contributions already added), `lp` the complete log density at the evaluated point.
// Fills `nc` (except hyp_q, which the caller sets after the hyper pre-ops/half step).
// ---------------------------------------------------------------------------------------------------
PPCX_HD void chain_advance(ChainScalars& st, TreeArrays& ta, const Cmd& ex, const Reduced& rd, double lp,
                           bool grads_finite, const ChainOut& out, Cmd& nc) {
  cmd_clear(nc);
  switch (st.phase) {
    case PH_START: {
      nc.type = CMD_EVAL; nc.dir = 1; nc.eps = 0.0;
      if (st.eval_only) { st.phase = PH_EVAL_ONLY; }
      else { nc.pre_flags = PRE_INIT; nc.rng_c1 = (unsigned)st.init_attempt; nc.init_radius = st.init_radius; st.phase = PH_INIT; }
      return;
    }
    case PH_EVAL_ONLY: { st.lp_eval = lp; nc.type = CMD_DONE; st.phase = PH_DONE; return; }
    case PH_INIT: {
      if (!(isfinite(lp) && grads_finite)) {
        if (++st.init_attempt >= 100) { st.error = 1; nc.type = CMD_DONE; st.phase = PH_DONE; return; }
        nc.type = CMD_EVAL; nc.dir = 1; nc.eps = 0.0; nc.pre_flags = PRE_INIT;
        nc.rng_c1 = (unsigned)st.init_attempt; nc.init_radius = st.init_radius;
        return;
      }
      st.V_sample = -lp;
      nc.pre_flags |= PRE_SAMPLE; nc.sample_src = -1; nc.pre_dir = 1;       // sample <- the evaluated init point
      if (st.iter <= 0) { nc.type = CMD_FLUSH; st.phase = PH_FLUSH; return; }
      start_eps_heuristic(st, nc);
      return;
    }
    case PH_EPS: {
      const double H0 = st.V_sample + 0.5 * rd.T0;
      double h = -lp + 0.5 * rd.T1; if (isnan(h)) h = INFINITY;
      const double dH = H0 - h;
#if defined(PPCX_DEBUG_EPS)           // development aid: the energies of every step-size trial
#if defined(__HIP_DEVICE_COMPILE__)
      if (threadIdx.x == 0 && blockIdx.x == 0)
#endif
      printf("eps trial: chain key %u eps %.6g V %.10g T0 %.10g lp %.10g T1 %.10g dH %.10g | attempt %d call %d ex.rng_c1 %u ex.rng_c3 %u ex.flags %d\n", st.k1, st.eps, st.V_sample, rd.T0, lp, rd.T1, dH, st.eps_attempt, st.eps_call, ex.rng_c1, ex.rng_c3, ex.pre_flags);
#endif
      const double thr = -0.22314355131420976;   // log(0.8)
      bool finished = false;
      if (st.eps_dir == 0) st.eps_dir = dH > thr ? 1 : -1;
      else if (st.eps_dir == 1 && !(dH > thr)) finished = true;
      else if (st.eps_dir == -1 && !(dH < thr)) finished = true;
      else {
        st.eps = st.eps_dir == 1 ? 2.0 * st.eps : 0.5 * st.eps;
        if (st.eps > 1e7 || st.eps == 0.0) finished = true;
      }
      if (!finished) { issue_eps_try(st, nc); return; }
      st.mu = log(10.0 * st.eps); st.s_bar = 0.0; st.x_bar = 0.0; st.da_counter = 0;   // set_mu + restart
      start_transition(st, nc);
      return;
    }
    case PH_TREE: {
      if (ex.pre_flags & PRE_NEW_TRANSITION) st.H0 = st.V_sample + 0.5 * rd.T0;
      const double Vn = -lp;
      double h = Vn + 0.5 * rd.T1; if (isnan(h)) h = INFINITY;
      if ((h - st.H0) > 1000.0) st.divergent = 1;
      ++st.n_leapfrog; ++st.total_leapfrogs;
      const double dlt = st.H0 - h;
      st.sum_metro += dlt > 0.0 ? 1.0 : (dlt < -700.0 ? 0.0 : fast_exp(dlt));
      double n_lsw = dlt, n_V = Vn; int n_src = -1;            // the node closed so far: this leaf
      bool valid = !st.divergent;
      if (valid) for (int d = 0; d < ex.n_merge; ++d) {         // merges in post-order, as the recursion unwinds
        const double lsw_sub = log_sum_exp(ta.Llsw[d], n_lsw);
        bool take_final = n_lsw > lsw_sub;
        if (!take_final) take_final = tree_uniform(st) < fast_exp(n_lsw - lsw_sub);
        if (!take_final) { n_src = d; n_V = ta.LV[d]; }
        n_lsw = lsw_sub;
        if (!all_positive(rd.dots[d])) { valid = false; break; }
      }
      nc.pre_dir = ex.dir;
      if (!valid) { end_transition(st, nc, out); return; }
      if (!ex.subtree_complete) {
        const int m = ex.n_merge;
        ta.Llsw[m] = n_lsw; ta.LV[m] = n_V;
        nc.pre_flags |= PRE_PROP; nc.prop_slot = m; nc.prop_src = n_src;
        set_leaf(st, nc, ex.leaf_n + 1);
        return;
      }
      // the new subtree is valid and complete (base_nuts::transition after build_tree)
      ++st.depth;
      bool accept = n_lsw > st.lsw_tree;
      if (!accept) accept = tree_uniform(st) < fast_exp(n_lsw - st.lsw_tree);
      if (accept) { nc.pre_flags |= PRE_SAMPLE; nc.sample_src = n_src; st.V_sample = n_V; }
      st.lsw_tree = log_sum_exp(st.lsw_tree, n_lsw);
      const bool persist = all_positive(rd.top);
      if (!persist || st.depth >= st.max_depth) { end_transition(st, nc, out); return; }
      start_doubling(st, nc);
      return;
    }
    case PH_FLUSH: { nc.type = CMD_DONE; st.phase = PH_DONE; return; }
    default: { nc.type = CMD_DONE; st.phase = PH_DONE; return; }
  }
}

}  // namespace ppcx
