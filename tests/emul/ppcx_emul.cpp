// CPU emulation harness (TEST INFRASTRUCTURE): drives the SAME __host__ __device__ source that the
// gfx950 kernels are built from (ppcseq_amd/csrc/ppcx_{math,model,nuts,gene}.h) with plain loops in
// place of wavefront lanes and workgroups, so the host logic -- command protocol, iterative NUTS tree, adaptation --
// can be checked against the oracle without a GPU. It is never loaded by the product package.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../ppcseq_amd/csrc/ppcx_gene.h"

using namespace ppcx;

static const double* window_table() { static double t[2 * kWinTabSize]; static bool init = false; if (!init) { fill_window_log_table(t); init = true; } return t; }
static const double* log_table() { static double t[2 * kLogTabSize]; static bool init = false; if (!init) { fill_log_table(t); init = true; } return t; }

struct EmulModel {
  Dims d; int CM;
  std::vector<int> counts; std::vector<double> E, expo, X, Sy, SyE, SyX, SX, ncell, Lg1;
  std::vector<double> disp; std::vector<unsigned char> gflags; double e_min = 1.0, e_max = 1.0;     // dispersion tables (ppcx_disp.h), bit 0: excluded cells
};

static EmulModel make_model(int G, int S, int C, int K, const int32_t* counts, const double* X, const double* expo,
                            double lmm, int n_excl, const int32_t* excl) {
  EmulModel m; m.d = make_dims(G, S, C, K, lmm); m.CM = C <= 2 ? 2 : (C <= 4 ? 4 : (C <= 8 ? 8 : 16));
  m.counts.assign(counts, counts + (size_t)G * S);
  m.counts.resize((size_t)G * S + 64, 0);      // the row sweep requests counts up to two trips of 4 lanes past the end
  for (int e = 0; e < n_excl; ++e) m.counts[excl[e]] = -1;
  m.X.assign(X, X + (size_t)S * C); m.X.resize((size_t)S * C + 64, 0.0); m.expo.assign(expo, expo + S); m.E.assign(S + 64, 0.0);
  int x0 = 1;
  for (int s = 0; s < S; ++s) { m.E[s] = exp(expo[s]); if (X[s] != 1.0) x0 = 0; }
  m.e_min = m.e_max = m.E[0];
  for (int s = 1; s < S; ++s) { if (m.E[s] < m.e_min) m.e_min = m.E[s]; if (m.E[s] > m.e_max) m.e_max = m.E[s]; }
  m.d.x0_is_one = x0;
  int x1b = (C >= 2);
  for (size_t i = (size_t)S; i < (size_t)S * C && x1b; ++i) if (X[i] != 0.0 && X[i] != 1.0) x1b = 0;
  m.d.x1_binary = x1b; m.d.raw_consts = (!x0 || (C >= 2 && K > 0 && !x1b)) ? 1 : 0;
  m.Sy.assign(G, 0); m.SyE.assign(G, 0); m.SyX.assign((size_t)C * G, 0); m.SX.assign((size_t)C * G, 0); m.ncell.assign(G, 0); m.Lg1.assign(G, 0);
  m.gflags.assign(G, 0); m.disp.assign((size_t)G * kDispGeneDoubles, 0.0);
  DispFit fit; disp_fit_init(fit);
  for (int g = 0; g < G; ++g) { for (int s = 0; s < S; ++s) {
    int y = m.counts[(size_t)g * S + s]; if (y < 0) { m.gflags[g] |= 1; continue; }
    m.Sy[g] += y; m.SyE[g] += (double)y * expo[s]; m.ncell[g] += 1; m.Lg1[g] += lgamma((double)y + 1.0);
    for (int c = 0; c < C; ++c) { m.SyX[(size_t)c * G + g] += (double)y * X[(size_t)c * S + s]; m.SX[(size_t)c * G + g] += X[(size_t)c * S + s]; }
  } disp_build_gene_host(fit, m.counts.data() + (size_t)g * S, S, m.disp.data() + (size_t)g * kDispGeneDoubles); }
  return m;
}

template <int CM>
static void gene_pass(const EmulModel& m, const Cmd& c, const VecRef& v, double* red) {
  constexpr int NCM = CM + 1;
  const Dims& d = m.d;
  for (int k = 0; k < PT_COUNT; ++k) red[k] = 0.0;
  if (c.type == CMD_DONE || c.type == CMD_FLUSH) return;
  for (int g = 0; g < d.G; ++g) {
    // log-likelihood kernel (one lane per gene here)
    CellData cd; cd.counts = m.counts.data(); cd.disp = m.disp.data(); cd.gflags = m.gflags.data(); cd.Sy = m.Sy.data(); cd.ncell = m.ncell.data(); cd.e_min = m.e_min; cd.e_max = m.e_max;
    GeneSumsV<CM> o;
    lane_gene_sums<CM, 1>(d, c, v, cd, g, gene_pre_load(d, v, cd, g), 0, m.E.data(), m.expo.data(), m.X.data(), log_table(), window_table(), o);
    // close kernel
    GeneCtx<CM> x2;
    gene_load<CM>(d, c, v, g, x2);
    double pn[NCM], minv[NCM], part[10];
    gene_finish<CM>(d, c, v, x2, o, m.Sy.data(), m.SyE.data(), m.SyX.data(), m.SX.data(), m.ncell.data(), m.Lg1.data(), part, pn, minv);
    for (int k = 0; k < 10; ++k) red[k] += part[k];
    if (c.type == CMD_LEAF) {
      NodeVals nv[NCM];
      for (int j = 0; j < NCM; ++j) nv[j] = NodeVals{pn[j], pn[j]};
      for (int lev = 0; lev < c.n_merge; ++lev)
        for (int j = 0; j < x2.ncoord; ++j) coord_merge_dots(v, x2.idx[j], lev, pn[j], minv[j], &nv[j], red + PT_DOTS + 6 * lev);
      if (!c.subtree_complete) { for (int j = 0; j < x2.ncoord; ++j) coord_store_slot(v, x2.idx[j], c.n_merge, pn[j], nv[j]); }
      else for (int j = 0; j < x2.ncoord; ++j) coord_top_dots(v, x2.idx[j], c.dir, pn[j], minv[j], nv[j], red + PT_TOP);
    }
  }
}
static void gene_pass_dispatch(const EmulModel& m, const Cmd& c, const VecRef& v, double* red) {
  if (m.CM == 2) gene_pass<2>(m, c, v, red);
  else if (m.CM == 4) gene_pass<4>(m, c, v, red);
  else if (m.CM == 8) gene_pass<8>(m, c, v, red);
  else gene_pass<16>(m, c, v, red);
}
// kernel B: state machine, then the per-coordinate operations of the new command; returns T0 of the gene coordinates
static double update_pass(const EmulModel& m, ChainState& st, const Cmd& ex, const double* red, double T0_prev,
                          bool have_parts, const VecRef& v, const VecRef& h, const ChainIO& io, Cmd& nc) {
  Reduced rd;
  std::vector<double> r2(red, red + PT_COUNT);
  r2[PT_T0] = T0_prev;
  chain_step(SerialLanes{}, m.d, st.sc, st.ta, ex, r2.data(), have_parts, h, io, rd, nc);
  double T0 = 0.0;
  if (nc.type != CMD_DONE)
    for (int i = 3; i < m.d.off_tail; ++i) coord_update(m.d, nc, v, i, io.draws, &T0);
  return T0;
}

struct EmulCfg { int chains, iter, warmup; unsigned long long seed; double adapt_delta; int max_treedepth;
                 double init_radius, stepsize0; int init_buffer, term_buffer, window, chain_id_offset; };

extern "C" __attribute__((visibility("default")))
int emul_log_prob_grad(int G, int S, int C, int K, const int32_t* counts, const double* X, const double* expo,
                       double lmm, int n_excl, const int32_t* excl, const double* u, double* lp, double* grad) {
  EmulModel m = make_model(G, S, C, K, counts, X, expo, lmm, n_excl, excl);
  const int D = m.d.D;
  std::vector<double> vecs((size_t)V_COUNT * D, 0.0), hv((size_t)V_COUNT * 8, 0.0), red(PT_COUNT, 0.0);
  for (int i = 0; i < D; ++i) { vecs[(size_t)V_Q1 * D + i] = u[i]; vecs[(size_t)V_MINV * D + i] = 1.0; }
  for (int k = 0; k < 6; ++k) { hv[V_Q1 * 8 + k] = u[hyper_index(m.d, k)]; hv[V_MINV * 8 + k] = 1.0; }
  NutsConfig nc; memset(&nc, 0, sizeof nc); nc.chains = 1; nc.max_treedepth = 10; nc.adapt_delta = 0.8; nc.init_radius = 2; nc.stepsize0 = 1;
  ChainState st; state_init(st, nc, 0, 1);
  Cmd c, n; cmd_clear(c);
  ChainIO io; memset(&io, 0, sizeof io);
  VecRef v{vecs.data(), D}, h{hv.data(), 8};
  double T0 = update_pass(m, st, c, red.data(), 0.0, false, v, h, io, n); c = n;
  while (c.type != CMD_DONE) {
    gene_pass_dispatch(m, c, v, red.data());
    T0 = update_pass(m, st, c, red.data(), T0, true, v, h, io, n); c = n;
  }
  *lp = st.sc.lp_eval;
  for (int i = 0; i < D; ++i) grad[i] = vecs[(size_t)V_G1 * D + i];
  for (int k = 0; k < 6; ++k) grad[hyper_index(m.d, k)] = hv[V_G1 * 8 + k];
  return 0;
}

// draws [chains][n_keep][D]; lp [chains][n_keep]; diagnostics [chains][iter]
extern "C" __attribute__((visibility("default")))
int emul_fit_nuts(int G, int S, int C, int K, const int32_t* counts, const double* X, const double* expo, double lmm,
                  int n_excl, const int32_t* excl, const EmulCfg* cfg, double* draws, double* lp, double* stepsize,
                  int* treedepth, int* n_leapfrog, int* divergent, double* accept) {
  EmulModel m = make_model(G, S, C, K, counts, X, expo, lmm, n_excl, excl);
  const int D = m.d.D, nk = cfg->iter - cfg->warmup;
  NutsConfig nc; nc.chains = cfg->chains; nc.iter = cfg->iter; nc.warmup = cfg->warmup; nc.seed = cfg->seed;
  nc.adapt_delta = cfg->adapt_delta; nc.max_treedepth = cfg->max_treedepth; nc.init_radius = cfg->init_radius;
  nc.stepsize0 = cfg->stepsize0; nc.init_buffer = cfg->init_buffer; nc.term_buffer = cfg->term_buffer;
  nc.window = cfg->window; nc.chain_id_offset = cfg->chain_id_offset;
  int rc = 0;
  for (int ch = 0; ch < cfg->chains; ++ch) {
    std::vector<double> vecs((size_t)V_COUNT * D, 0.0), hv((size_t)V_COUNT * 8, 0.0), red(PT_COUNT, 0.0);
    for (int i = 0; i < D; ++i) vecs[(size_t)V_MINV * D + i] = 1.0;
    for (int k = 0; k < 8; ++k) hv[V_MINV * 8 + k] = 1.0;
    ChainState st; state_init(st, nc, ch, 0);
    Cmd c, n; cmd_clear(c);
    ChainIO io;
    io.draws = draws + (size_t)ch * nk * D;
    io.out.lp = lp + (size_t)ch * nk; io.out.stepsize = stepsize + (size_t)ch * cfg->iter;
    io.out.treedepth = treedepth + (size_t)ch * cfg->iter; io.out.n_leapfrog = n_leapfrog + (size_t)ch * cfg->iter;
    io.out.divergent = divergent + (size_t)ch * cfg->iter; io.out.accept = accept + (size_t)ch * cfg->iter;
    VecRef v{vecs.data(), D}, h{hv.data(), 8};
    double T0 = update_pass(m, st, c, red.data(), 0.0, false, v, h, io, n); c = n;
    long guard = 0;
    while (c.type != CMD_DONE) {
      gene_pass_dispatch(m, c, v, red.data());
      T0 = update_pass(m, st, c, red.data(), T0, true, v, h, io, n); c = n;
      if (++guard > 50000000L) { rc = -5; break; }
    }
    if (st.sc.error) rc = -3;
  }
  return rc;
}

// ---- pipelined rounds (ppcx_ls_kernel + ppcx_gene_kernel on the device): the same protocol with plain loops.
// ls_first_s: run the state machine before the log-likelihood part of the merged launch (on the device they run side by
// side: neither may depend on the other, so both orders must give the same chain).
template <int CM>
static void pipelined_loglik(const EmulModel& m, const Cmd& x, const VecRef& v, std::vector<GeneSumsV<CM>>& sums) {
  if (x.type == CMD_DONE || x.type == CMD_FLUSH) return;
  if (x.evaluated && x.type != CMD_LEAF) return;               // closed, and nothing was anticipated after it
  CellData cd; cd.counts = m.counts.data(); cd.disp = m.disp.data(); cd.gflags = m.gflags.data(); cd.Sy = m.Sy.data(); cd.ncell = m.ncell.data(); cd.e_min = m.e_min; cd.e_max = m.e_max;
  for (int g = 0; g < m.d.G; ++g)
    lane_gene_sums<CM, 1>(m.d, x, v, cd, g, gene_pre_load(m.d, v, cd, g), 0, m.E.data(), m.expo.data(), m.X.data(), log_table(), window_table(), sums[g]);
}
template <int CM>
static void pipelined_gene(const EmulModel& m, const Cmd& y, const VecRef& v, double* draws, bool spec,
                           std::vector<GeneSumsV<CM>>& sums, double* red) {
  constexpr int NCM = CM + 1;
  const Dims& d = m.d;
  if (y.type == CMD_DONE) return;
  const bool do_update = !y.updated, do_close = y.evaluated && y.type != CMD_FLUSH;
  if (do_close) for (int k = 0; k < PT_COUNT; ++k) if (k != PT_T0) red[k] = 0.0;
  double T0 = 0.0;
  for (int g = 0; g < d.G; ++g) {
    GeneCtx<CM> x;
    gene_index<CM>(d, g, x);
    if (do_update) gene_coord_update<CM>(d, y, v, x, draws, &T0, !do_close);
    if (!do_close) continue;
    if (do_update) gene_params<CM>(d, v, x); else gene_load<CM>(d, y, v, g, x);
    double pn[NCM], minv[NCM], gn[NCM], part[10];
    GeneSumsV<CM> o = sums[g];
    gene_finish<CM>(d, y, v, x, o, m.Sy.data(), m.SyE.data(), m.SyX.data(), m.SX.data(), m.ncell.data(), m.Lg1.data(), part, pn, minv, gn);
    for (int k = 0; k < 10; ++k) if (k != PT_T0) red[k] += part[k];
    if (y.type == CMD_LEAF) {
      NodeVals nv[NCM];
      for (int j = 0; j < NCM; ++j) nv[j] = NodeVals{pn[j], pn[j]};
      for (int lev = 0; lev < y.n_merge; ++lev)
        for (int j = 0; j < x.ncoord; ++j) coord_merge_dots(v, x.idx[j], lev, pn[j], minv[j], &nv[j], red + PT_DOTS + 6 * lev);
      if (!y.subtree_complete) { for (int j = 0; j < x.ncoord; ++j) coord_store_slot(v, x.idx[j], y.n_merge, pn[j], nv[j]); }
      else for (int j = 0; j < x.ncoord; ++j) coord_top_dots(v, x.idx[j], y.dir, pn[j], minv[j], nv[j], red + PT_TOP);
      if (spec) gene_spec_consts<CM>(d, y, v, x, pn, gn, minv);
    }
  }
  if (do_update) red[PT_T0] = T0;
}
template <int CM>
static int pipelined_chain(const EmulModel& m, const NutsConfig& nc, int ch, const ChainIO& io, bool spec, bool ls_first_s,
                           long* rounds, long* carried) {
  const int D = m.d.D;
  std::vector<double> vecs((size_t)V_COUNT * D, 0.0), hv((size_t)V_COUNT * 8, 0.0), red(PT_COUNT, 0.0);
  for (int i = 0; i < D; ++i) vecs[(size_t)V_MINV * D + i] = 1.0;
  for (int k = 0; k < 8; ++k) hv[V_MINV * 8 + k] = 1.0;
  std::vector<GeneSumsV<CM>> sums(m.d.G);
  ChainState st; state_init(st, nc, ch, 0);
  Cmd x, y; cmd_clear(x);
  VecRef v{vecs.data(), D}, h{hv.data(), 8};
  Reduced rd;
  for (long guard = 0; guard < 100000000L; ++guard) {
    if (!ls_first_s) pipelined_loglik<CM>(m, x, v, sums);
    const Cmd x_seen = x;                                         // what the log-likelihood part of the launch reads
    const bool stepped = chain_step_pipelined(SerialLanes{}, m.d, st.sc, st.ta, x, red.data(), h, io, rd, y, spec);
    if (ls_first_s) pipelined_loglik<CM>(m, x_seen, v, sums);
    ++*rounds; if (!stepped) ++*carried;
    x = y;
    if (x.type == CMD_DONE) return st.sc.error ? -3 : 0;
    pipelined_gene<CM>(m, x, v, io.draws, spec, sums, red.data());
  }
  return -5;
}
// rounds[chains], carried[chains]: launches per chain and how many of them only carried a command (mis-anticipations)
extern "C" __attribute__((visibility("default")))
int emul_fit_nuts_pipelined(int G, int S, int C, int K, const int32_t* counts, const double* X, const double* expo, double lmm,
                            int n_excl, const int32_t* excl, const EmulCfg* cfg, int spec, int ls_first_s, double* draws, double* lp,
                            double* stepsize, int* treedepth, int* n_leapfrog, int* divergent, double* accept, long* rounds,
                            long* carried) {
  EmulModel m = make_model(G, S, C, K, counts, X, expo, lmm, n_excl, excl);
  const int D = m.d.D, nk = cfg->iter - cfg->warmup;
  NutsConfig nc; nc.chains = cfg->chains; nc.iter = cfg->iter; nc.warmup = cfg->warmup; nc.seed = cfg->seed;
  nc.adapt_delta = cfg->adapt_delta; nc.max_treedepth = cfg->max_treedepth; nc.init_radius = cfg->init_radius;
  nc.stepsize0 = cfg->stepsize0; nc.init_buffer = cfg->init_buffer; nc.term_buffer = cfg->term_buffer;
  nc.window = cfg->window; nc.chain_id_offset = cfg->chain_id_offset;
  int rc = 0;
  for (int ch = 0; ch < cfg->chains; ++ch) {
    ChainIO io;
    io.draws = draws + (size_t)ch * nk * D;
    io.out.lp = lp + (size_t)ch * nk; io.out.stepsize = stepsize + (size_t)ch * cfg->iter;
    io.out.treedepth = treedepth + (size_t)ch * cfg->iter; io.out.n_leapfrog = n_leapfrog + (size_t)ch * cfg->iter;
    io.out.divergent = divergent + (size_t)ch * cfg->iter; io.out.accept = accept + (size_t)ch * cfg->iter;
    rounds[ch] = 0; carried[ch] = 0;
    int r;
    if (m.CM == 2) r = pipelined_chain<2>(m, nc, ch, io, spec != 0, ls_first_s != 0, rounds + ch, carried + ch);
    else if (m.CM == 4) r = pipelined_chain<4>(m, nc, ch, io, spec != 0, ls_first_s != 0, rounds + ch, carried + ch);
    else if (m.CM == 8) r = pipelined_chain<8>(m, nc, ch, io, spec != 0, ls_first_s != 0, rounds + ch, carried + ch);
    else r = pipelined_chain<16>(m, nc, ch, io, spec != 0, ls_first_s != 0, rounds + ch, carried + ch);
    if (r != 0) rc = r;
  }
  return rc;
}

// the dispersion table of one row (ppcx_disp.h): Fh and Dh at the given sigma_raw -- by table lookup (in[i] = 1) or, outside
// the tabulated range, by the direct evaluation that also builds the table; direct[] always holds the direct evaluation
extern "C" __attribute__((visibility("default")))
int emul_disp_table(const int32_t* row, int S, int n, const double* sigma, double* F, double* D, double* F_direct, double* D_direct, int* in) {
  DispFit fit; disp_fit_init(fit);
  std::vector<double> t(kDispGeneDoubles);
  disp_build_gene_host(fit, row, S, t.data());
  for (int i = 0; i < n; ++i) {
    const DispRef r = disp_ref(sigma[i]);
    const DispPoint pt = disp_point(sigma[i]);
    disp_row(row, S, 0, 1, pt, &F_direct[i], &D_direct[i]);
    in[i] = r.in ? 1 : 0;
    if (r.in) {
      const double* pc = t.data() + (long)r.panel * 2 * kDispStride;
      F[i] = disp_horner(pc, r.x); D[i] = disp_horner(pc + kDispStride, r.x);
    } else { F[i] = F_direct[i]; D[i] = D_direct[i]; }
  }
  return 0;
}

extern "C" __attribute__((visibility("default")))
int emul_nb2_log_rng(double eta, double phi, unsigned long long seed, unsigned cell, unsigned draw) {
  return nb2_log_rng(eta, phi, seed32(seed), cell, draw);
}
