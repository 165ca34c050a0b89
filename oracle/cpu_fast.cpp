// cpu_fast.cpp -- TEST / BENCH INFRASTRUCTURE: an OPTIMISED CPU comparator for bench.py's `cpu_baseline` leg.
//
// oracle/ppc_oracle.c is a literal restatement of the Stan program (libm lgamma, series digamma and a log1p_exp per
// cell: ~500 ns per cell and thread), which makes "GPU vs CPU" ratios meaningless as a bar. This file evaluates the
// same log density and gradient (inst/stan/negBinomial_MPI.stan:58-120,:180-240) the way the PRODUCT formulates it
// -- per-gene sufficient statistics, one table logarithm and one reciprocal per cell, per-gene dispersion tables:
// the __host__ __device__ headers of ppcseq_amd/csrc compiled for the host -- with OpenMP threads over genes
// (mirrors map_rect / STAN_NUM_THREADS, R/utilities.R:1383-1386,1479), built -O3 -march=native.
// Only bench.py (cpu_baseline) and tests/ load it; the product package never does.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../ppcseq_amd/csrc/ppcx_gene.h"

using namespace ppcx;

struct FastModel {
  Dims d; int CM;
  std::vector<int> counts; std::vector<double> E, expo, X, Sy, SyE, SyX, SX, ncell, Lg1, tab, wtab;
  std::vector<double> disp; std::vector<unsigned char> gflags; double e_min = 1.0, e_max = 1.0;     // dispersion tables (ppcx_disp.h), bit 0: excluded cells
  std::vector<double> vecs, hv;
  int threads = 1;                 // of ppcf_lp_callback
};

extern "C" __attribute__((visibility("default")))
void* ppcf_model_create(int G, int S, int C, int K, const int32_t* counts, const double* X, const double* expo, double lmm) {
  FastModel* m = new FastModel();
  m->d = make_dims(G, S, C, K, lmm); m->CM = C <= 2 ? 2 : (C <= 4 ? 4 : (C <= 8 ? 8 : 16));
  m->counts.assign(counts, counts + (size_t)G * S); m->counts.resize((size_t)G * S + 64, 0);
  m->X.assign(X, X + (size_t)S * C); m->X.resize((size_t)S * C + 64, 0.0); m->expo.assign(expo, expo + S); m->E.assign(S + 64, 0.0);
  int x0 = 1;
  for (int s = 0; s < S; ++s) { m->E[s] = exp(expo[s]); if (X[s] != 1.0) x0 = 0; }
  m->e_min = m->e_max = m->E[0];
  for (int s = 1; s < S; ++s) { if (m->E[s] < m->e_min) m->e_min = m->E[s]; if (m->E[s] > m->e_max) m->e_max = m->E[s]; }
  m->d.x0_is_one = x0;
  int x1b = (C >= 2);
  for (size_t i = (size_t)S; i < (size_t)S * C && x1b; ++i) if (X[i] != 0.0 && X[i] != 1.0) x1b = 0;
  m->d.x1_binary = x1b; m->d.raw_consts = (!x0 || (C >= 2 && K > 0 && !x1b)) ? 1 : 0;
  m->Sy.assign(G, 0); m->SyE.assign(G, 0); m->SyX.assign((size_t)C * G, 0); m->SX.assign((size_t)C * G, 0); m->ncell.assign(G, 0); m->Lg1.assign(G, 0);
  m->gflags.assign(G, 0); m->disp.assign((size_t)G * kDispGeneDoubles, 0.0);
  DispFit fit; disp_fit_init(fit);
#pragma omp parallel for schedule(dynamic, 16)
  for (int g = 0; g < G; ++g) {
    for (int s = 0; s < S; ++s) {
      const int y = m->counts[(size_t)g * S + s];
      m->Sy[g] += y; m->SyE[g] += (double)y * expo[s]; m->ncell[g] += 1; m->Lg1[g] += lgamma((double)y + 1.0);
      for (int c = 0; c < C; ++c) { m->SyX[(size_t)c * G + g] += (double)y * X[(size_t)c * S + s]; m->SX[(size_t)c * G + g] += X[(size_t)c * S + s]; }
    }
    disp_build_gene_host(fit, m->counts.data() + (size_t)g * S, S, m->disp.data() + (size_t)g * kDispGeneDoubles);
  }
  m->tab.resize(2 * kLogTabSize); fill_log_table(m->tab.data());
  m->wtab.resize(2 * kWinTabSize); fill_window_log_table(m->wtab.data());
  m->vecs.assign((size_t)V_COUNT * m->d.D, 0.0); m->hv.assign((size_t)V_COUNT * 8, 0.0);
  return m;
}
extern "C" __attribute__((visibility("default"))) void ppcf_model_destroy(void* h) { delete (FastModel*)h; }
extern "C" __attribute__((visibility("default"))) int ppcf_dim(void* h) { return ((FastModel*)h)->d.D; }

template <int CM>
static double eval(FastModel& m, const double* u, double* grad, int threads) {
  constexpr int NCM = CM + 1;
  const Dims& d = m.d; const int D = d.D;
  VecRef v{m.vecs.data(), D};
  Cmd c; cmd_clear(c); c.type = CMD_EVAL; c.dir = 1; c.eps = 0.0;
  for (int k = 0; k < 6; ++k) c.hyp_q[k] = u[hyper_index(d, k)];
  c.hy = make_hyper(c.hyp_q, d.lambda_mu_mu);
  const double* tab = m.tab.data();
  CellData cd; cd.counts = m.counts.data(); cd.disp = m.disp.data(); cd.gflags = m.gflags.data(); cd.Sy = m.Sy.data(); cd.ncell = m.ncell.data(); cd.e_min = m.e_min; cd.e_max = m.e_max;
  double lp = 0.0, h[6] = {0, 0, 0, 0, 0, 0};
#pragma omp parallel for schedule(static) num_threads(threads) reduction(+ : lp, h[:6])
  for (int g = 0; g < d.G; ++g) {
    GeneCtx<CM> x;
    gene_index<CM>(d, g, x);
    for (int j = 0; j < x.ncoord; ++j) {                       // what the update kernel leaves next to the coordinates
      const int i = x.idx[j];
      v.at(V_Q1, i) = u[i]; v.at(V_P1, i) = 0.0; v.at(V_MINV, i) = 1.0;
      coord_consts(d, v, i, u[i]);
    }
    GeneSumsV<CM> o;
    lane_gene_sums<CM, 1>(d, c, v, cd, g, gene_pre_load(d, v, cd, g), 0, m.E.data(), m.expo.data(), m.X.data(), tab, m.wtab.data(), o);
    gene_load<CM>(d, c, v, g, x);
    double pn[NCM], minv[NCM], gn[NCM], part[10];
    gene_finish<CM>(d, c, v, x, o, m.Sy.data(), m.SyE.data(), m.SyX.data(), m.SX.data(), m.ncell.data(), m.Lg1.data(), part, pn, minv, gn);
    lp += part[PT_LP];
    for (int k = 0; k < 6; ++k) h[k] += part[PT_H0 + k];
    if (grad) for (int j = 0; j < x.ncoord; ++j) grad[x.idx[j]] = gn[j];
  }
  double g6[6];
  const double total = hyper_close(d, c.hy, c.hyp_q, lp, h, g6);
  if (grad) for (int k = 0; k < 6; ++k) grad[hyper_index(d, k)] = g6[k];
  return total;
}
// log density and gradient at u (unconstrained, Stan order); grad may be NULL
extern "C" __attribute__((visibility("default")))
double ppcf_log_prob_grad(void* h, const double* u, double* grad, int threads) {
  FastModel& m = *(FastModel*)h;
  if (threads < 1) threads = 1;
  if (m.CM == 2) return eval<2>(m, u, grad, threads);
  if (m.CM == 4) return eval<4>(m, u, grad, threads);
  if (m.CM == 8) return eval<8>(m, u, grad, threads);
  return eval<16>(m, u, grad, threads);
}

// The same evaluation as a callback fn(ctx, u, grad) for the oracle's NUTS driver (ppco_nuts_chain_fn): ctx is the model,
// whose scratch vectors make it single-caller -- one model per chain -- and ppcf_set_threads fixes its OpenMP threads.
extern "C" __attribute__((visibility("default"))) void ppcf_set_threads(void* h, int threads) { ((FastModel*)h)->threads = threads < 1 ? 1 : threads; }
extern "C" __attribute__((visibility("default")))
double ppcf_lp_callback(const void* ctx, const double* u, double* grad) {
  FastModel* m = (FastModel*)ctx;
  return ppcf_log_prob_grad(m, u, grad, m->threads);
}
