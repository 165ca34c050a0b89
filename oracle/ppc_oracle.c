/* TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.
 *
 * CPU restatement (plain C, fp64) of the ppcseq hot path:
 *   - the negative-binomial hierarchical log density of inst/stan/negBinomial_MPI.stan
 *     (parameters :180-199, transforms :200-206, model :208-258, lp_reduce :58-120,
 *      merge_coefficients :122-139) and its analytic gradient,
 *   - the Stan-default NUTS that rstan::sampling() runs for R/utilities.R:1497-1512
 *     (multinomial NUTS, diag_e metric, dual averaging, windowed variance adaptation;
 *      third-party rstan/StanHeaders, NOT in /root/reference -- version floors only,
 *      DESCRIPTION:32,59-60 -- so the published algorithm is restated),
 *   - generated quantities neg_binomial_2_log_rng (:259-266),
 *   - the credible-interval summary of R/utilities.R:685-703 (type-7 quantiles, mean, sd).
 *
 * PARITY PIN STATUS: the reference holds no numeric golden vectors for this path
 * (SURVEY.md 8c); density/gradient are pinned against independent scipy/mpmath/torch
 * evaluations (tests/test_oracle_density.py), the sampler against analytic targets
 * (tests/test_oracle_nuts.py) and the end-to-end outlier calls against the reference's
 * only known answers (tests/testthat/test-ppcSeq.R:26-30: tot_deleterious = 0,1,0).
 * NUTS-draw parity versus rstan itself remains "parity unpinned" (no R/Stan here).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "philox_spec.h"

#define PPCO_EXPORT __attribute__((visibility("default")))

/* ----------------------------------------------------------------------------------- */
/* special functions                                                                   */
/* ----------------------------------------------------------------------------------- */

/* digamma: upward recurrence to x >= 10, then the asymptotic series (Abramowitz&Stegun 6.3.18) */
PPCO_EXPORT double ppco_digamma(double x) {
  double r = 0.0;
  while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
  double f = 1.0 / (x * x);
  double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 +
             f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + log(x) - 0.5 / x + t;
}

static inline double log_sum_exp2(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  double m = a > b ? a : b;
  return m + log1p(exp(-fabs(a - b)));
}

/* neg_binomial_2_log_lpmf(y | eta, phi), all terms kept (target += form; .stan:97-103).
 * Returns the value and the partials w.r.t. eta and phi. */
static inline double nb2log(double y, double eta, double phi, double* d_eta, double* d_phi) {
  double logphi = log(phi);
  double lse = log_sum_exp2(eta, logphi);          /* log(exp(eta)+phi) */
  double lp = lgamma(y + phi) - lgamma(phi) - lgamma(y + 1.0) + y * eta + phi * logphi - (y + phi) * lse;
  if (d_eta) {
    double mu_over = exp(eta - lse);               /* mu/(mu+phi) */
    *d_eta = y - (y + phi) * mu_over;
    *d_phi = ppco_digamma(y + phi) - ppco_digamma(phi) + logphi + 1.0 - lse - (y + phi) * exp(-lse);
  }
  return lp;
}

/* ----------------------------------------------------------------------------------- */
/* model                                                                               */
/* ----------------------------------------------------------------------------------- */
typedef struct {
  int G, S, C, K;
  const int32_t* counts;     /* G x S, gene-major (sample index fastest) */
  const double* X;           /* S x C, column-major (R native)            */
  const double* exposure;    /* S                                          */
  double lambda_mu_mu;
  int n_excl;
  const int32_t* excl;       /* 0-based cell ids g*S+s (to_exclude, R/utilities.R:321-359) */
  int n_threads;
} ppco_model;

PPCO_EXPORT int ppco_dim(int G, int C, int K) {
  int kc = C - 1 > 1 ? C - 1 : 1;
  return 2 * G + K * kc + 6;
}

/* offsets into the unconstrained vector, Stan declaration order (.stan:183-197) */
typedef struct { int lambda_mu, lambda_sigma, lambda_skew, intercept, alpha1, alpha2, sigma_raw,
                 sigma_slope, sigma_intercept, sigma_sigma, D; } ppco_off;
static ppco_off offsets(int G, int C, int K) {
  ppco_off o;
  o.lambda_mu = 0; o.lambda_sigma = 1; o.lambda_skew = 2;
  o.intercept = 3;
  o.alpha1 = 3 + G;
  o.alpha2 = o.alpha1 + K;
  int n2 = (C - 2 > 0 ? C - 2 : 0) * K;
  o.sigma_raw = o.alpha2 + n2;
  o.sigma_slope = o.sigma_raw + G;
  o.sigma_intercept = o.sigma_slope + 1;
  o.sigma_sigma = o.sigma_slope + 2;
  o.D = o.sigma_slope + 3;
  return o;
}

/* log density on the unconstrained scale (+ Jacobians), and gradient if grad != NULL. */
PPCO_EXPORT double ppco_log_prob_grad(const ppco_model* m, const double* u, double* grad) {
  const int G = m->G, S = m->S, C = m->C, K = m->K;
  const ppco_off o = offsets(G, C, K);
  if (grad) memset(grad, 0, sizeof(double) * (size_t)o.D);

  /* transforms (.stan:183-197): offset / lower=0 / upper=0 */
  const double lambda_mu = u[o.lambda_mu] + m->lambda_mu_mu;
  const double lambda_sigma = exp(u[o.lambda_sigma]);
  const double lambda_skew = u[o.lambda_skew];
  const double sigma_slope = -exp(u[o.sigma_slope]);
  const double sigma_intercept = u[o.sigma_intercept];
  const double sigma_sigma = exp(u[o.sigma_sigma]);
  const double* intercept = u + o.intercept;
  const double* alpha1 = u + o.alpha1;
  const double* alpha2 = u + o.alpha2;
  const double* sigma_raw = u + o.sigma_raw;

  double lp = 0.0;
  /* Jacobians of the constraining transforms */
  lp += u[o.lambda_sigma] + u[o.sigma_slope] + u[o.sigma_sigma];
  /* hyper priors (.stan:210-216); `~` drops constants */
  lp += -0.5 * (lambda_mu - m->lambda_mu_mu) * (lambda_mu - m->lambda_mu_mu) / 4.0;
  lp += -0.5 * lambda_sigma * lambda_sigma / 4.0;
  lp += -0.5 * lambda_skew * lambda_skew;
  lp += -0.5 * sigma_intercept * sigma_intercept / 4.0;
  lp += -0.5 * sigma_slope * sigma_slope / 4.0;
  lp += -0.5 * sigma_sigma * sigma_sigma / 4.0;

  double g_lmu = 0, g_lsig = 0, g_lskew = 0, g_sslope = 0, g_sint = 0, g_ssig = 0; /* constrained-scale */
  if (grad) {
    g_lmu = -(lambda_mu - m->lambda_mu_mu) / 4.0;
    g_lsig = -lambda_sigma / 4.0;
    g_lskew = -lambda_skew;
    g_sint = -sigma_intercept / 4.0;
    g_sslope = -sigma_slope / 4.0;
    g_ssig = -sigma_sigma / 4.0;
  }

  /* gene-level priors (.stan:219-223) */
  const double xi = lambda_mu + m->lambda_mu_mu;     /* location: offset added twice, App. D.1 */
  const double om = lambda_sigma, a = lambda_skew;
  const double SQRT2 = 1.4142135623730951, SQRT_2_OVER_PI = 0.7978845608028654;
  for (int g = 0; g < G; ++g) {
    double z = (intercept[g] - xi) / om;
    double x = -a * z / SQRT2;
    double ec = erfc(x);
    lp += -log(om) - 0.5 * z * z + log(ec);
    double r = sigma_raw[g] - (sigma_slope * intercept[g] + sigma_intercept);
    lp += -log(sigma_sigma) - 0.5 * r * r / (sigma_sigma * sigma_sigma);
    if (grad) {
      double ratio = SQRT_2_OVER_PI * exp(-x * x) / ec;    /* d/dt log erfc(-t/sqrt2) at t=a z */
      double dz = -z + a * ratio;                          /* d lp / d z */
      grad[o.intercept + g] += dz / om;
      g_lmu += -dz / om;
      g_lsig += -1.0 / om - dz * z / om;
      g_lskew += z * ratio;
      double rs = r / (sigma_sigma * sigma_sigma);
      grad[o.sigma_raw + g] += -rs;
      grad[o.intercept + g] += sigma_slope * rs;
      g_sslope += intercept[g] * rs;
      g_sint += rs;
      g_ssig += -1.0 / sigma_sigma + r * r / (sigma_sigma * sigma_sigma * sigma_sigma);
    }
  }
  if (C >= 2) for (int k = 0; k < K; ++k) {               /* double_exponential(0,1) */
    lp += -fabs(alpha1[k]);
    if (grad) grad[o.alpha1 + k] += (alpha1[k] > 0) ? -1.0 : (alpha1[k] < 0 ? 1.0 : 0.0);
  }
  if (C >= 3) for (int i = 0; i < (C - 2) * K; ++i) {     /* normal(0,2.5) */
    lp += -0.5 * alpha2[i] * alpha2[i] / 6.25;
    if (grad) grad[o.alpha2 + i] += -alpha2[i] / 6.25;
  }

  /* likelihood: sum over all cells (.stan:97-103) ... */
  /* per-gene sums are added afterwards in gene order: an OpenMP reduction combines the threads' partial sums in the
     order they arrive, and the last bits of lp then differ from run to run (NUTS trajectories separate after ~50
     iterations; the distributional tests saw two different "oracle" posteriors on two boxes) */
  double lik = 0.0;
  double* lik_g = (double*)malloc(sizeof(double) * (size_t)(G > 0 ? G : 1));
  int nt = m->n_threads > 0 ? m->n_threads : 1;
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static)
  for (int g = 0; g < G; ++g) {
    const double phi = exp(-sigma_raw[g]);              /* sigma = 1/exp(sigma_raw) (.stan:203) */
    double acc = 0.0, d_int = 0.0, d_phi_sum = 0.0;
    double d_alpha[16];
    for (int c = 0; c < C && c < 16; ++c) d_alpha[c] = 0.0;
    for (int s = 0; s < S; ++s) {
      double eta = m->exposure[s] + m->X[s] * intercept[g];  /* X[,1] is the intercept column */
      if (g < K) {
        if (C >= 2) eta += m->X[(size_t)S + s] * alpha1[g];
        for (int c = 2; c < C; ++c) eta += m->X[(size_t)c * S + s] * alpha2[(c - 2) + (C - 2) * g];
      }
      double de = 0.0, dp = 0.0;
      acc += nb2log((double)m->counts[(size_t)g * S + s], eta, phi, grad ? &de : NULL, &dp);
      if (grad) {
        d_phi_sum += dp;
        for (int c = 0; c < C; ++c) d_alpha[c] += m->X[(size_t)c * S + s] * de;
      }
    }
    lik_g[g] = acc;
    if (grad) {
      d_int = d_alpha[0];
      grad[o.intercept + g] += d_int;
      grad[o.sigma_raw + g] += -phi * d_phi_sum;
      if (g < K) {
        if (C >= 2) grad[o.alpha1 + g] += d_alpha[1];
        for (int c = 2; c < C; ++c) grad[o.alpha2 + (c - 2) + (C - 2) * g] += d_alpha[c];
      }
    }
  }
  for (int g = 0; g < G; ++g) lik += lik_g[g];
  free(lik_g);
  /* ... minus the same over the excluded cells (.stan:105-115) */
  for (int e = 0; e < m->n_excl; ++e) {
    int cell = m->excl[e], g = cell / S, s = cell % S;
    const double phi = exp(-sigma_raw[g]);
    double eta = m->exposure[s] + m->X[s] * intercept[g];
    if (g < K) {
      if (C >= 2) eta += m->X[(size_t)S + s] * alpha1[g];
      for (int c = 2; c < C; ++c) eta += m->X[(size_t)c * S + s] * alpha2[(c - 2) + (C - 2) * g];
    }
    double de = 0.0, dp = 0.0;
    lik -= nb2log((double)m->counts[cell], eta, phi, grad ? &de : NULL, &dp);
    if (grad) {
      grad[o.intercept + g] -= m->X[s] * de;
      grad[o.sigma_raw + g] -= -phi * dp;
      if (g < K) {
        if (C >= 2) grad[o.alpha1 + g] -= m->X[(size_t)S + s] * de;
        for (int c = 2; c < C; ++c) grad[o.alpha2 + (c - 2) + (C - 2) * g] -= m->X[(size_t)c * S + s] * de;
      }
    }
  }
  lp += lik;

  if (grad) {                                           /* chain rule to the unconstrained scale */
    grad[o.lambda_mu] = g_lmu;
    grad[o.lambda_sigma] = g_lsig * lambda_sigma + 1.0;
    grad[o.lambda_skew] = g_lskew;
    grad[o.sigma_slope] = g_sslope * sigma_slope + 1.0;
    grad[o.sigma_intercept] = g_sint;
    grad[o.sigma_sigma] = g_ssig * sigma_sigma + 1.0;
  }
  return lp;
}

/* ----------------------------------------------------------------------------------- */
/* generic log-density callback so the sampler can be validated on analytic targets    */
/* ----------------------------------------------------------------------------------- */
typedef double (*ppco_lp_fn)(const void* ctx, const double* u, double* grad);

static double model_lp(const void* ctx, const double* u, double* grad) {
  return ppco_log_prob_grad((const ppco_model*)ctx, u, grad);
}
/* independent N(mean_i, sd_i^2) target: ctx = {D, mean[D], sd[D]} packed as doubles */
static double gauss_lp(const void* ctx, const double* u, double* grad) {
  const double* c = (const double*)ctx; int D = (int)c[0];
  double lp = 0;
  for (int i = 0; i < D; ++i) {
    double z = (u[i] - c[1 + i]) / c[1 + D + i];
    lp += -0.5 * z * z; if (grad) grad[i] = -z / c[1 + D + i];
  }
  return lp;
}

/* ----------------------------------------------------------------------------------- */
/* NUTS (Stan defaults).                                             */
/* RNG addressing (shared SPECIFICATION with the HIP sampler, see DESIGN.md "RNG"):     */
/*   key = (seed32, chain_id)                                                           */
/*   init values      : counter (i, attempt, 0, 0)  -> 1 uniform per coordinate         */
/*   momentum         : counter (i>>1, iter, 1, 0)   -> Box-Muller pair, cos=even i      */
/*   tree scalars     : counter (j, iter, 2, 0)      -> j-th uniform of the transition   */
/*   doubling direction: counter (depth, iter, 7, 0) -> forward iff the uniform > 1/2    */
/*   init_stepsize p  : counter (i>>1, call, 3, attempt)                                 */
/* ----------------------------------------------------------------------------------- */
typedef struct {
  int chains, iter, warmup;         /* per chain: iter total, of which warmup */
  uint64_t seed;
  double adapt_delta;               /* 0.8  */
  int max_treedepth;                /* 10   */
  double init_radius;               /* 2    */
  double stepsize0;                 /* 1    */
  int init_buffer, term_buffer, window; /* 75, 50, 25 */
  int max_leapfrogs_total;          /* >0: stop the chain early after this many gradient evals (bounded CPU baseline) */
} ppco_nuts_cfg;

typedef struct {
  int D; ppco_lp_fn fn; const void* ctx;
  uint32_t k0, k1;
  double eps;
  double* minv;                     /* diag inverse metric */
  /* current point */
  double *q, *p, *g; double V;
  long n_grad;
} chain_t;

static inline uint32_t seed32(uint64_t s) { return (uint32_t)s ^ (uint32_t)((s >> 32) * 0x9E3779B9u); }

static inline double coord_normal(const chain_t* c, int i, uint32_t c1, uint32_t c2, uint32_t c3) {
  ppco_u4 r = ppco_philox4x32_10((uint32_t)(i >> 1), c1, c2, c3, c->k0, c->k1);
  double u1 = ppco_u01(r.v[0], r.v[1]), u2 = ppco_u01(r.v[2], r.v[3]);
  double rad = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925 * u2;
  return (i & 1) ? rad * sin(t) : rad * cos(t);
}
static inline double scalar_uniform(const chain_t* c, uint32_t j, uint32_t iter) {
  ppco_u4 r = ppco_philox4x32_10(j, iter, 2u, 0u, c->k0, c->k1);
  return ppco_u01(r.v[0], r.v[1]);
}

/* the direction of doubling number `depth` of a transition has a stream of its own (the product anticipates the next
 * doubling's direction while the current subtree is being built: ppcx_nuts.h doubling_dir) */
static inline double dir_uniform(const chain_t* c, uint32_t depth, uint32_t iter) {
  ppco_u4 r = ppco_philox4x32_10(depth, iter, 7u, 0u, c->k0, c->k1);
  return ppco_u01(r.v[0], r.v[1]);
}

static void eval(chain_t* c) {              /* V = -lp, g = dV/dq */
  double lp = c->fn(c->ctx, c->q, c->g);
  for (int i = 0; i < c->D; ++i) c->g[i] = -c->g[i];
  c->V = -lp; c->n_grad++;
}
static double kinetic(const chain_t* c) {
  double t = 0; for (int i = 0; i < c->D; ++i) t += c->p[i] * c->p[i] * c->minv[i];
  return 0.5 * t;
}
static void leapfrog(chain_t* c, double eps) {
  for (int i = 0; i < c->D; ++i) { c->p[i] -= 0.5 * eps * c->g[i]; c->q[i] += eps * c->minv[i] * c->p[i]; }
  eval(c);
  for (int i = 0; i < c->D; ++i) c->p[i] -= 0.5 * eps * c->g[i];
}
static inline double Hnan(double h) { return isnan(h) ? INFINITY : h; }

typedef struct { double *q, *p, *g; double V; } pspoint;
static void ps_alloc(pspoint* z, int D) { z->q = malloc(sizeof(double) * D); z->p = malloc(sizeof(double) * D); z->g = malloc(sizeof(double) * D); }
static void ps_free(pspoint* z) { free(z->q); free(z->p); free(z->g); }
static void ps_save(pspoint* z, const chain_t* c) { memcpy(z->q, c->q, sizeof(double) * c->D); memcpy(z->p, c->p, sizeof(double) * c->D); memcpy(z->g, c->g, sizeof(double) * c->D); z->V = c->V; }
static void ps_load(const pspoint* z, chain_t* c) { memcpy(c->q, z->q, sizeof(double) * c->D); memcpy(c->p, z->p, sizeof(double) * c->D); memcpy(c->g, z->g, sizeof(double) * c->D); c->V = z->V; }
static void ps_copy(pspoint* a, const pspoint* b, int D) { memcpy(a->q, b->q, sizeof(double) * D); memcpy(a->p, b->p, sizeof(double) * D); memcpy(a->g, b->g, sizeof(double) * D); a->V = b->V; }

/* init_stepsize heuristic (Stan base_hmc::init_stepsize) */
static void init_stepsize(chain_t* c, uint32_t call) {
  if (c->eps == 0 || c->eps > 1e7 || isnan(c->eps)) return;
  pspoint z0; ps_alloc(&z0, c->D); ps_save(&z0, c);
  uint32_t attempt = 0;
  for (int i = 0; i < c->D; ++i) c->p[i] = coord_normal(c, i, call, 3u, attempt) / sqrt(c->minv[i]);
  ++attempt;
  double H0 = c->V + kinetic(c);
  leapfrog(c, c->eps);
  double h = Hnan(c->V + kinetic(c));
  double dH = H0 - h;
  int direction = dH > log(0.8) ? 1 : -1;
  for (;;) {
    ps_load(&z0, c);
    for (int i = 0; i < c->D; ++i) c->p[i] = coord_normal(c, i, call, 3u, attempt) / sqrt(c->minv[i]);
    ++attempt;
    H0 = c->V + kinetic(c);
    leapfrog(c, c->eps);
    h = Hnan(c->V + kinetic(c));
    dH = H0 - h;
    if (direction == 1 && !(dH > log(0.8))) break;
    else if (direction == -1 && !(dH < log(0.8))) break;
    else c->eps = direction == 1 ? 2 * c->eps : 0.5 * c->eps;
    if (c->eps > 1e7 || c->eps == 0) break;
  }
  ps_load(&z0, c);
  ps_free(&z0);
}

typedef struct {
  chain_t* c; double eps; double H0; int max_depth;
  uint32_t iter, rng_j;
  int n_leapfrog; double sum_metro; int divergent;
  pspoint* scratch; /* one proposal buffer per depth level */
} tree_t;

static inline int criterion(const double* psm, const double* psp, const double* rho, int D) {
  double a = 0, b = 0;
  for (int i = 0; i < D; ++i) { a += psp[i] * rho[i]; b += psm[i] * rho[i]; }
  return a > 0 && b > 0;
}

/* recursive build_tree, Stan base_nuts (multinomial, generalised U-turn with the
 * extra cross-subtree checks of Stan >= 2.21) */
static int build_tree(tree_t* t, int depth, pspoint* z_propose,
                      double* psharp_beg, double* psharp_end, double* rho,
                      double* p_beg, double* p_end, int sign, double* log_sum_weight) {
  chain_t* c = t->c; const int D = c->D;
  if (depth == 0) {
    leapfrog(c, sign * t->eps);
    ++t->n_leapfrog;
    double h = Hnan(c->V + kinetic(c));
    if ((h - t->H0) > 1000.0) t->divergent = 1;
    *log_sum_weight = log_sum_exp2(*log_sum_weight, t->H0 - h);
    t->sum_metro += (t->H0 - h > 0) ? 1.0 : exp(t->H0 - h);
    ps_save(z_propose, c);
    for (int i = 0; i < D; ++i) { psharp_beg[i] = c->minv[i] * c->p[i]; psharp_end[i] = psharp_beg[i];
      rho[i] += c->p[i]; p_beg[i] = c->p[i]; p_end[i] = c->p[i]; }
    return !t->divergent;
  }
  double* buf = malloc(sizeof(double) * D * 6);
  double *p_init_end = buf, *psharp_init_end = buf + D, *rho_init = buf + 2 * D,
         *p_final_beg = buf + 3 * D, *psharp_final_beg = buf + 4 * D, *rho_final = buf + 5 * D;
  memset(rho_init, 0, sizeof(double) * D); memset(rho_final, 0, sizeof(double) * D);
  double lsw_init = -INFINITY;
  int ok = build_tree(t, depth - 1, z_propose, psharp_beg, psharp_init_end, rho_init, p_beg, p_init_end, sign, &lsw_init);
  if (!ok) { free(buf); return 0; }
  pspoint* z_final = &t->scratch[depth];
  ps_save(z_final, c);
  double lsw_final = -INFINITY;
  ok = build_tree(t, depth - 1, z_final, psharp_final_beg, psharp_end, rho_final, p_final_beg, p_end, sign, &lsw_final);
  if (!ok) { free(buf); return 0; }
  double lsw_sub = log_sum_exp2(lsw_init, lsw_final);
  *log_sum_weight = log_sum_exp2(*log_sum_weight, lsw_sub);
  if (lsw_final > lsw_sub) ps_copy(z_propose, z_final, D);
  else {
    double ap = exp(lsw_final - lsw_sub);
    if (scalar_uniform(c, t->rng_j++, t->iter) < ap) ps_copy(z_propose, z_final, D);
  }
  int persist;
  {
    double* rho_sub = malloc(sizeof(double) * D * 2); double* rho_ext = rho_sub + D;
    for (int i = 0; i < D; ++i) { rho_sub[i] = rho_init[i] + rho_final[i]; rho[i] += rho_sub[i]; }
    persist = criterion(psharp_beg, psharp_end, rho_sub, D);
    for (int i = 0; i < D; ++i) rho_ext[i] = rho_init[i] + p_final_beg[i];
    persist &= criterion(psharp_beg, psharp_final_beg, rho_ext, D);
    for (int i = 0; i < D; ++i) rho_ext[i] = rho_final[i] + p_init_end[i];
    persist &= criterion(psharp_init_end, psharp_end, rho_ext, D);
    free(rho_sub);
  }
  free(buf);
  return persist;
}

typedef struct { double accept_stat; int n_leapfrog, depth, divergent; double energy; } trans_info;

static void transition(chain_t* c, uint32_t iter, int max_depth, trans_info* info) {
  const int D = c->D;
  for (int i = 0; i < D; ++i) c->p[i] = coord_normal(c, i, iter, 1u, 0u) / sqrt(c->minv[i]);
  pspoint z_fwd, z_bck, z_sample, z_propose;
  ps_alloc(&z_fwd, D); ps_alloc(&z_bck, D); ps_alloc(&z_sample, D); ps_alloc(&z_propose, D);
  ps_save(&z_fwd, c); ps_save(&z_bck, c); ps_save(&z_sample, c); ps_save(&z_propose, c);
  double* buf = malloc(sizeof(double) * D * 12);
  double *p_fwd_fwd = buf, *psh_fwd_fwd = buf + D, *p_fwd_bck = buf + 2 * D, *psh_fwd_bck = buf + 3 * D,
         *p_bck_fwd = buf + 4 * D, *psh_bck_fwd = buf + 5 * D, *p_bck_bck = buf + 6 * D, *psh_bck_bck = buf + 7 * D,
         *rho = buf + 8 * D, *rho_fwd = buf + 9 * D, *rho_bck = buf + 10 * D, *rho_ext = buf + 11 * D;
  for (int i = 0; i < D; ++i) {
    double ps = c->minv[i] * c->p[i];
    p_fwd_fwd[i] = p_fwd_bck[i] = p_bck_fwd[i] = p_bck_bck[i] = c->p[i];
    psh_fwd_fwd[i] = psh_fwd_bck[i] = psh_bck_fwd[i] = psh_bck_bck[i] = ps;
    rho[i] = c->p[i];
  }
  tree_t t; t.c = c; t.eps = c->eps; t.max_depth = max_depth; t.iter = iter; t.rng_j = 0;
  t.n_leapfrog = 0; t.sum_metro = 0; t.divergent = 0;
  t.scratch = malloc(sizeof(pspoint) * (max_depth + 1));
  for (int d = 0; d <= max_depth; ++d) ps_alloc(&t.scratch[d], D);
  double log_sum_weight = 0.0;
  t.H0 = c->V + kinetic(c);
  int depth = 0;
  while (depth < max_depth) {
    memset(rho_fwd, 0, sizeof(double) * D); memset(rho_bck, 0, sizeof(double) * D);
    int valid; double lsw_sub = -INFINITY;
    if (dir_uniform(c, (uint32_t)depth, iter) > 0.5) {
      ps_load(&z_fwd, c);
      memcpy(rho_bck, rho, sizeof(double) * D); memcpy(p_bck_fwd, p_fwd_fwd, sizeof(double) * D); memcpy(psh_bck_fwd, psh_fwd_fwd, sizeof(double) * D);
      valid = build_tree(&t, depth, &z_propose, psh_fwd_bck, psh_fwd_fwd, rho_fwd, p_fwd_bck, p_fwd_fwd, 1, &lsw_sub);
      ps_save(&z_fwd, c);
    } else {
      ps_load(&z_bck, c);
      memcpy(rho_fwd, rho, sizeof(double) * D); memcpy(p_fwd_bck, p_bck_bck, sizeof(double) * D); memcpy(psh_fwd_bck, psh_bck_bck, sizeof(double) * D);
      valid = build_tree(&t, depth, &z_propose, psh_bck_fwd, psh_bck_bck, rho_bck, p_bck_fwd, p_bck_bck, -1, &lsw_sub);
      ps_save(&z_bck, c);
    }
    if (!valid) break;
    ++depth;
    if (lsw_sub > log_sum_weight) ps_copy(&z_sample, &z_propose, D);
    else {
      double ap = exp(lsw_sub - log_sum_weight);
      if (scalar_uniform(c, t.rng_j++, iter) < ap) ps_copy(&z_sample, &z_propose, D);
    }
    log_sum_weight = log_sum_exp2(log_sum_weight, lsw_sub);
    for (int i = 0; i < D; ++i) rho[i] = rho_bck[i] + rho_fwd[i];
    int persist = criterion(psh_bck_bck, psh_fwd_fwd, rho, D);
    for (int i = 0; i < D; ++i) rho_ext[i] = rho_bck[i] + p_fwd_bck[i];
    persist &= criterion(psh_bck_bck, psh_fwd_bck, rho_ext, D);
    for (int i = 0; i < D; ++i) rho_ext[i] = rho_fwd[i] + p_bck_fwd[i];
    persist &= criterion(psh_bck_fwd, psh_fwd_fwd, rho_ext, D);
    if (!persist) break;
  }
  info->n_leapfrog = t.n_leapfrog; info->depth = depth; info->divergent = t.divergent;
  info->accept_stat = t.sum_metro / (double)t.n_leapfrog;
  ps_load(&z_sample, c);
  info->energy = c->V + kinetic(c);
  for (int d = 0; d <= max_depth; ++d) ps_free(&t.scratch[d]);
  free(t.scratch); free(buf);
  ps_free(&z_fwd); ps_free(&z_bck); ps_free(&z_sample); ps_free(&z_propose);
}

/* One chain. Outputs (all optional): draws[n_keep*D] (row per kept draw, unconstrained),
 * lp[n_keep], and per-iteration diagnostics for ALL iterations (warmup included):
 * stepsize[iter], treedepth[iter], n_leapfrog[iter], divergent[iter], accept[iter].
 * Returns number of iterations completed (== cfg->iter unless bounded), <0 on init failure. */
static int run_chain(ppco_lp_fn fn, const void* ctx, int D, const ppco_nuts_cfg* cfg, int chain_id,
                     double* draws, double* lp_out, double* stepsize, int* treedepth, int* n_leapfrog,
                     int* divergent, double* accept, double* metric_out) {
  chain_t c; memset(&c, 0, sizeof c);
  c.D = D; c.fn = fn; c.ctx = ctx; c.k0 = seed32(cfg->seed); c.k1 = (uint32_t)chain_id;
  c.q = malloc(sizeof(double) * D); c.p = calloc(D, sizeof(double)); c.g = malloc(sizeof(double) * D);
  c.minv = malloc(sizeof(double) * D);
  for (int i = 0; i < D; ++i) c.minv[i] = 1.0;
  c.eps = cfg->stepsize0;
  /* init = "random": U(-R, R) on the unconstrained scale, retried up to 100 times */
  int ok = 0;
  for (uint32_t attempt = 0; attempt < 100 && !ok; ++attempt) {
    for (int i = 0; i < D; ++i) {
      ppco_u4 r = ppco_philox4x32_10((uint32_t)i, attempt, 0u, 0u, c.k0, c.k1);
      c.q[i] = (2.0 * ppco_u01(r.v[0], r.v[1]) - 1.0) * cfg->init_radius;
    }
    eval(&c);
    ok = isfinite(c.V);
    for (int i = 0; i < D && ok; ++i) ok = isfinite(c.g[i]);
  }
  if (!ok) { free(c.q); free(c.p); free(c.g); free(c.minv); return -1; }

  /* adaptation state */
  const int W = cfg->warmup;
  int init_buffer = cfg->init_buffer, term_buffer = cfg->term_buffer, window = cfg->window;
  int adapt_on = W > 0;
  if (W < 20) { init_buffer = term_buffer = window = 0; /* Stan: no windowed adaptation */ }
  else if (init_buffer + window + term_buffer > W) {
    init_buffer = (int)(0.15 * W); term_buffer = (int)(0.1 * W); window = W - (init_buffer + term_buffer);
  }
  int next_window = init_buffer + window - 1, window_size = window, counter_w = 0;
  double mu = 0, s_bar = 0, x_bar = 0; int da_counter = 0;
  const double gamma = 0.05, kappa = 0.75, t0 = 10;
  double* wm = calloc(D, sizeof(double)); double* wm2 = calloc(D, sizeof(double)); int wn = 0;
  uint32_t ss_call = 0;

  init_stepsize(&c, ss_call++);
  mu = log(10 * c.eps);

  int done = 0;
  for (int it = 0; it < cfg->iter; ++it) {
    trans_info info;
    transition(&c, (uint32_t)it, cfg->max_treedepth, &info);
    if (stepsize) stepsize[it] = c.eps;
    if (treedepth) treedepth[it] = info.depth;
    if (n_leapfrog) n_leapfrog[it] = info.n_leapfrog;
    if (divergent) divergent[it] = info.divergent;
    if (accept) accept[it] = info.accept_stat;
    if (it < W && adapt_on) {
      /* dual averaging (Stan stepsize_adaptation::learn_stepsize) */
      ++da_counter;
      double as = info.accept_stat > 1 ? 1 : info.accept_stat;
      double eta = 1.0 / (da_counter + t0);
      s_bar = (1 - eta) * s_bar + eta * (cfg->adapt_delta - as);
      double x = mu - s_bar * sqrt((double)da_counter) / gamma;
      double x_eta = pow((double)da_counter, -kappa);
      x_bar = (1 - x_eta) * x_bar + x_eta * x;
      c.eps = exp(x);
      /* windowed variance adaptation (Stan var_adaptation::learn_variance) */
      if (W >= 20) {
        int in_window = (counter_w >= init_buffer) && (counter_w < W - term_buffer) && (counter_w != W);
        if (in_window) { /* Welford */
          ++wn;
          for (int i = 0; i < D; ++i) { double d = c.q[i] - wm[i]; wm[i] += d / wn; wm2[i] += (c.q[i] - wm[i]) * d; }
        }
        int end_window = (counter_w == next_window) && (counter_w != W);
        if (end_window) {
          /* compute_next_window */
          if (next_window != W - term_buffer - 1) {
            window_size *= 2;
            next_window = counter_w + window_size;
            if (next_window != W - term_buffer - 1) {
              int boundary = next_window + 2 * window_size;
              if (boundary >= W - term_buffer) next_window = W - term_buffer - 1;
            }
          }
          double n = (double)wn;
          for (int i = 0; i < D; ++i) {
            double var = wm2[i] / (n - 1.0);
            c.minv[i] = (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0));
          }
          memset(wm, 0, sizeof(double) * D); memset(wm2, 0, sizeof(double) * D); wn = 0;
          ++counter_w;
          init_stepsize(&c, ss_call++);
          mu = log(10 * c.eps); da_counter = 0; s_bar = 0; x_bar = 0;
        } else ++counter_w;
      }
      if (it == W - 1) c.eps = exp(x_bar);          /* complete_adaptation */
    }
    if (it >= W) {
      int k = it - W;
      if (draws) memcpy(draws + (size_t)k * D, c.q, sizeof(double) * D);
      if (lp_out) lp_out[k] = -c.V;
    }
    done = it + 1;
    if (cfg->max_leapfrogs_total > 0 && c.n_grad >= cfg->max_leapfrogs_total) break;
  }
  if (metric_out) memcpy(metric_out, c.minv, sizeof(double) * D);
  free(wm); free(wm2); free(c.q); free(c.p); free(c.g); free(c.minv);
  return done;
}

/* draws: [chains][n_keep][D]; diagnostics: [chains][iter]; returns 0 ok. iters_done[chains]. */
PPCO_EXPORT int ppco_nuts_model(const ppco_model* m, const ppco_nuts_cfg* cfg, double* draws, double* lp,
                                double* stepsize, int* treedepth, int* n_leapfrog, int* divergent,
                                double* accept, double* metric, int* iters_done) {
  int D = ppco_dim(m->G, m->C, m->K); int nk = cfg->iter - cfg->warmup; int rc = 0;
  for (int ch = 0; ch < cfg->chains; ++ch) {
    int r = run_chain(model_lp, m, D, cfg, ch,
                      draws ? draws + (size_t)ch * nk * D : NULL, lp ? lp + (size_t)ch * nk : NULL,
                      stepsize ? stepsize + (size_t)ch * cfg->iter : NULL, treedepth ? treedepth + (size_t)ch * cfg->iter : NULL,
                      n_leapfrog ? n_leapfrog + (size_t)ch * cfg->iter : NULL, divergent ? divergent + (size_t)ch * cfg->iter : NULL,
                      accept ? accept + (size_t)ch * cfg->iter : NULL, metric ? metric + (size_t)ch * D : NULL);
    if (iters_done) iters_done[ch] = r;
    if (r < 0) rc = -1;
  }
  return rc;
}
/* One chain of the same sampler on a caller-supplied log density (fn(ctx, u, grad) returns lp and fills grad): bench.py's
 * `--cpu-full-cfg2` leg drives the optimised CPU comparator (oracle/cpu_fast.cpp) through it, one chain per host thread, so
 * that a whole CPU fit -- same seeds, chains, warm-up and iterations as the GPU fit -- is timed rather than extrapolated.
 * chain_id is the chain's global id (its Philox stream). Returns the iterations completed, < 0 on init failure. */
PPCO_EXPORT int ppco_nuts_chain_fn(ppco_lp_fn fn, const void* ctx, int D, const ppco_nuts_cfg* cfg, int chain_id,
                                   double* draws, double* lp, double* stepsize, int* treedepth, int* n_leapfrog,
                                   int* divergent, double* accept) {
  return run_chain(fn, ctx, D, cfg, chain_id, draws, lp, stepsize, treedepth, n_leapfrog, divergent, accept, NULL);
}
PPCO_EXPORT int ppco_nuts_gauss(int D, const double* mean, const double* sd, const ppco_nuts_cfg* cfg,
                                double* draws, double* lp, double* stepsize, int* treedepth,
                                int* n_leapfrog, int* divergent, double* accept) {
  double* ctx = malloc(sizeof(double) * (1 + 2 * D));
  ctx[0] = D; memcpy(ctx + 1, mean, sizeof(double) * D); memcpy(ctx + 1 + D, sd, sizeof(double) * D);
  int nk = cfg->iter - cfg->warmup; int rc = 0;
  for (int ch = 0; ch < cfg->chains; ++ch) {
    int r = run_chain(gauss_lp, ctx, D, cfg, ch, draws + (size_t)ch * nk * D, lp ? lp + (size_t)ch * nk : NULL,
                      stepsize ? stepsize + (size_t)ch * cfg->iter : NULL, treedepth ? treedepth + (size_t)ch * cfg->iter : NULL,
                      n_leapfrog ? n_leapfrog + (size_t)ch * cfg->iter : NULL, divergent ? divergent + (size_t)ch * cfg->iter : NULL,
                      accept ? accept + (size_t)ch * cfg->iter : NULL, NULL);
    if (r < 0) rc = -1;
  }
  free(ctx);
  return rc;
}

/* ----------------------------------------------------------------------------------- */
/* generated quantities: neg_binomial_2_log_rng (.stan:259-266), gamma-Poisson mixture  */
/* Stream addressing: key = (seed32, 0x50504331), counter = (blk, cell, draw, 4) for the  */
/* gamma variate and (blk, cell, draw, 8) for the Poisson variate                        */
/* ----------------------------------------------------------------------------------- */
static double gamma_rng(double a, ppco_stream* st) {     /* Marsaglia-Tsang 2000, unit scale */
  double boost = 1.0;
  if (a < 1.0) { boost = pow(ppco_stream_uniform(st), 1.0 / a); a += 1.0; }
  double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
  for (;;) {
    double x = ppco_stream_normal(st);
    double v = 1.0 + c * x;
    if (v <= 0) continue;
    v = v * v * v;
    double u = ppco_stream_uniform(st);
    double x2 = x * x;
    if (u < 1.0 - 0.0331 * x2 * x2) return d * v * boost;
    if (log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) return d * v * boost;
  }
}
static int64_t poisson_rng(double lam, ppco_stream* st) {
  if (lam < 10.0) {                                     /* Knuth multiplication */
    double L = exp(-lam), p = ppco_stream_uniform(st); int64_t k = 0;
    while (p > L) { ++k; p *= ppco_stream_uniform(st); }
    return k;
  }
  /* PTRS, Hoermann 1993 */
  double slam = sqrt(lam), loglam = log(lam);
  double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
  double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
  for (;;) {
    double U = ppco_stream_uniform(st) - 0.5, V = ppco_stream_uniform(st);
    double us = 0.5 - fabs(U);
    double kf = floor((2.0 * a / us + b) * U + lam + 0.43);
    if (us >= 0.07 && V <= vr) return (int64_t)kf;
    if (kf < 0 || (us < 0.013 && V > us)) continue;
    if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + kf * loglam - lgamma(kf + 1.0)) return (int64_t)kf;
  }
}
PPCO_EXPORT int32_t ppco_nb2_log_rng(double eta, double phi, uint64_t seed, uint32_t cell, uint32_t draw) {
  if (!(phi > 0.0) || !isfinite(phi) || !isfinite(eta)) return 2147483647;
  ppco_stream st; ppco_stream_init(&st, seed32(seed), 0x50504331u, cell, draw, 4u);
  double lam = gamma_rng(phi, &st) * (exp(eta) / phi);
  if (!(lam < 1073741824.0)) return 1073741823;          /* Stan errors above 2^30; we saturate */
  ppco_stream sp; ppco_stream_init(&sp, seed32(seed), 0x50504331u, cell, draw, 8u);
  int64_t k = poisson_rng(lam, &sp);
  return k > 2147483647LL ? 2147483647 : (int32_t)k;
}

/* counts_rng[d][g][s] for g<K from unconstrained draws [n_draws][D] */
PPCO_EXPORT void ppco_generated_quantities(const ppco_model* m, const double* draws, int n_draws,
                                           double truncation_compensation, uint64_t seed, int32_t* out) {
  const int G = m->G, S = m->S, C = m->C, K = m->K; const ppco_off o = offsets(G, C, K);
#pragma omp parallel for schedule(static) num_threads(m->n_threads > 0 ? m->n_threads : 1)
  for (int d = 0; d < n_draws; ++d) {
    const double* u = draws + (size_t)d * o.D;
    for (int g = 0; g < K; ++g) {
      double phi = exp(-u[o.sigma_raw + g]) * truncation_compensation;
      for (int s = 0; s < S; ++s) {
        double eta = m->exposure[s] + m->X[s] * u[o.intercept + g];
        if (C >= 2) eta += m->X[(size_t)S + s] * u[o.alpha1 + g];
        for (int c = 2; c < C; ++c) eta += m->X[(size_t)c * S + s] * u[o.alpha2 + (c - 2) + (C - 2) * g];
        out[((size_t)d * K + g) * S + s] = ppco_nb2_log_rng(eta, phi, seed, (uint32_t)(g * S + s), (uint32_t)d);
      }
    }
  }
}

/* approximated analysis (fit_to_counts_rng_approximated, R/utilities.R:733-784): per checked cell (s, g <= K) resample the
 * posterior with replacement -- i_supersampled = sample(seq_len(n_draws), n_gen, replace = TRUE) (:760) -- and draw
 * rnbinom(mu = exp(lambda_log_param[i, s, g] + exposure_rate[s]), size = exp(-sigma_raw[i, g]) * truncation_compensation)
 * (:762-766). R's global RNG is replaced by the project's Philox specification: the index of predictive draw j of cell
 * c = g*S + s is floor(u * n_draws) with u the first uniform of block (j, c, 5, 0) under key (seed32, 'PPC1'); the
 * count is neg_binomial_2_log_rng on stream (c, j) as in the full analysis. out: [n_gen][K][S]. */
PPCO_EXPORT void ppco_generated_quantities_approx(const ppco_model* m, const double* draws, int n_draws, int n_gen,
                                                  double truncation_compensation, uint64_t seed, int32_t* out) {
  const int G = m->G, S = m->S, C = m->C, K = m->K; const ppco_off o = offsets(G, C, K);
  const uint32_t k0 = seed32(seed);
#pragma omp parallel for schedule(static) num_threads(m->n_threads > 0 ? m->n_threads : 1)
  for (int j = 0; j < n_gen; ++j) {
    for (int g = 0; g < K; ++g) for (int s = 0; s < S; ++s) {
      const uint32_t cell = (uint32_t)(g * S + s);
      const ppco_u4 r = ppco_philox4x32_10((uint32_t)j, cell, 5u, 0u, k0, 0x50504331u);
      long src = (long)(ppco_u01(r.v[0], r.v[1]) * (double)n_draws);
      if (src >= n_draws) src = n_draws - 1;
      const double* u = draws + (size_t)src * o.D;
      double eta = m->exposure[s] + m->X[s] * u[o.intercept + g];
      if (C >= 2) eta += m->X[(size_t)S + s] * u[o.alpha1 + g];
      for (int c = 2; c < C; ++c) eta += m->X[(size_t)c * S + s] * u[o.alpha2 + (c - 2) + (C - 2) * g];
      const double phi = exp(-u[o.sigma_raw + g]) * truncation_compensation;
      out[((size_t)j * K + g) * S + s] = ppco_nb2_log_rng(eta, phi, seed, cell, (uint32_t)j);
    }
  }
}

/* ----------------------------------------------------------------------------------- */
/* credible-interval summary (R/utilities.R:685-703): mean, sd, type-7 quantiles         */
/* x: [n_draws][n_cells] int32; out: [n_cells][4] = mean, sd, lower, upper               */
/* ----------------------------------------------------------------------------------- */
static int cmp_i32(const void* a, const void* b) { int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y); }
static double quantile7(const int32_t* sorted, int n, double p) {
  double h = (n - 1) * p; int lo = (int)floor(h); if (lo >= n - 1) return sorted[n - 1];
  return fma(h - lo, (double)sorted[lo + 1] - (double)sorted[lo], (double)sorted[lo]);   /* one rounding, whatever the compiler contracts */
}
PPCO_EXPORT void ppco_summarise(const int32_t* x, int n_draws, int n_cells, double p_lo, double p_hi, double* out) {
#pragma omp parallel
  {
    int32_t* col = malloc(sizeof(int32_t) * n_draws);
#pragma omp for schedule(static)
    for (int c = 0; c < n_cells; ++c) {
      double mean = 0;
      for (int d = 0; d < n_draws; ++d) { col[d] = x[(size_t)d * n_cells + c]; mean += col[d]; }
      mean /= n_draws;
      double ss = 0; for (int d = 0; d < n_draws; ++d) ss += (col[d] - mean) * (col[d] - mean);
      qsort(col, n_draws, sizeof(int32_t), cmp_i32);
      out[4 * c + 0] = mean; out[4 * c + 1] = n_draws > 1 ? sqrt(ss / (n_draws - 1)) : NAN;
      out[4 * c + 2] = quantile7(col, n_draws, p_lo); out[4 * c + 3] = quantile7(col, n_draws, p_hi);
    }
    free(col);
  }
}

/* ----------------------------------------------------------------------------------------------- */
/* ADVI, mean-field (rstan::vb through vb_iterative, R/utilities.R:246-278,1487-1494): restatement */
/* of Stan's published advi.hpp algorithm (third party, not in the container). Monte-Carlo draws   */
/* use the project's Philox specification: key (seed32, 'ADVI'), counter (i>>1, draw_id, 6, 0),     */
/* draw ids consumed in the order documented in DESIGN.md ("ADVI").                                 */
/* ----------------------------------------------------------------------------------------------- */
typedef struct { int output_samples, iter; double tol_rel_obj; int grad_samples, elbo_samples, eval_elbo, adapt_iter;
                 uint64_t seed; double init_radius; } ppco_advi_cfg;

static double advi_eta(uint32_t i, uint32_t draw, uint32_t k0) {
  ppco_u4 r = ppco_philox4x32_10(i >> 1, draw, 6u, 0u, k0, 0x41445649u);
  double u1 = ppco_u01(r.v[0], r.v[1]), u2 = ppco_u01(r.v[2], r.v[3]);
  double rad = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925 * u2;
  return (i & 1) ? rad * sin(t) : rad * cos(t);
}
typedef struct { const ppco_model* m; int D; uint32_t k0; uint32_t draw_id; double *mu, *om, *hm, *ho, *zeta, *g; double lp_const; int elbo_samples; } advi_t;

static double advi_calc_elbo(advi_t* a) {
  double acc = 0; int ok = 0;
  for (int s = 0; s < a->elbo_samples; ++s) {
    uint32_t id = a->draw_id++;
    for (int i = 0; i < a->D; ++i) a->zeta[i] = a->mu[i] + exp(a->om[i]) * advi_eta((uint32_t)i, id, a->k0);
    double lp = ppco_log_prob_grad(a->m, a->zeta, NULL);
    if (isfinite(lp)) { acc += lp + a->lp_const; ++ok; }
  }
  if (!ok) return -INFINITY;
  double ent = 0.5 * a->D * (1.0 + 1.8378770664093454836);
  for (int i = 0; i < a->D; ++i) ent += a->om[i];
  return acc / a->elbo_samples + ent;
}
static void advi_grad_at(advi_t* a, uint32_t id) {
  for (int i = 0; i < a->D; ++i) a->zeta[i] = a->mu[i] + exp(a->om[i]) * advi_eta((uint32_t)i, id, a->k0);
  ppco_log_prob_grad(a->m, a->zeta, a->g);
}
static void advi_step(advi_t* a, double eta, int it, uint32_t* id) {
  for (int i = 0; i < a->D; ++i) {
    double e = advi_eta((uint32_t)i, *id, a->k0);
    double gm = a->g[i], go = a->g[i] * e * exp(a->om[i]) + 1.0;
    if (!(isfinite(gm) && isfinite(go))) continue;
    a->hm[i] = it == 1 ? gm * gm : 0.1 * gm * gm + 0.9 * a->hm[i];
    a->ho[i] = it == 1 ? go * go : 0.1 * go * go + 0.9 * a->ho[i];
    double es = eta / sqrt((double)it);
    a->mu[i] += es * gm / (1.0 + sqrt(a->hm[i]));
    a->om[i] += es * go / (1.0 + sqrt(a->ho[i]));
  }
  *id = a->draw_id++;
  advi_grad_at(a, *id);
}
static int cmp_d(const void* x, const void* y) { double a = *(const double*)x, b = *(const double*)y; return (a > b) - (a < b); }

/* out_draws: [output_samples][D]; mu_out, omega_out: [D]; info: {iterations, converged, elbo, eta} */
PPCO_EXPORT int ppco_advi(const ppco_model* m, const ppco_advi_cfg* cfg, double* out_draws, double* mu_out, double* omega_out, double* info) {
  const int D = ppco_dim(m->G, m->C, m->K);
  advi_t a; a.m = m; a.D = D; a.k0 = seed32(cfg->seed); a.draw_id = 1; a.elbo_samples = cfg->elbo_samples;
  double* buf = calloc((size_t)D * 7, sizeof(double));
  a.mu = buf; a.om = buf + D; a.hm = buf + 2 * D; a.ho = buf + 3 * D; a.zeta = buf + 4 * D; a.g = buf + 5 * D;
  double* q0 = buf + 6 * D;
  const double HL2PI = 0.91893853320467274178; const int n2 = m->C > 2 ? m->C - 2 : 0;
  a.lp_const = -(6.0 + 2.0 * m->G + (double)n2 * m->K) * HL2PI - 5.0 * log(2.0) - (m->C >= 2 ? m->K * log(2.0) : 0.0) - (double)n2 * m->K * log(2.5);
  int ok = 0;
  for (uint32_t attempt = 0; attempt < 100 && !ok; ++attempt) {
    for (int i = 0; i < D; ++i) { ppco_u4 r = ppco_philox4x32_10((uint32_t)i, attempt, 0u, 0u, a.k0, 0x41445649u); q0[i] = (2.0 * ppco_u01(r.v[0], r.v[1]) - 1.0) * cfg->init_radius; }
    double lp = ppco_log_prob_grad(m, q0, a.g);
    ok = isfinite(lp);
    for (int i = 0; i < D && ok; ++i) ok = isfinite(a.g[i]);
  }
  if (!ok) { free(buf); return -3; }
#define ADVI_RESET() do { memcpy(a.mu, q0, sizeof(double) * D); memset(a.om, 0, sizeof(double) * D); memset(a.hm, 0, sizeof(double) * D); memset(a.ho, 0, sizeof(double) * D); } while (0)
  ADVI_RESET();
  double elbo_init = advi_calc_elbo(&a), elbo_best = -INFINITY, eta_best = 0;
  const double eta_seq[5] = {100, 10, 1, 0.1, 0.01};
  int tuned = 0;
  for (int e = 0; e < 5 && !tuned; ++e) {
    uint32_t id = a.draw_id++; advi_grad_at(&a, id);
    for (int it = 1; it <= cfg->adapt_iter; ++it) advi_step(&a, eta_seq[e], it, &id);
    double elbo = advi_calc_elbo(&a); if (!isfinite(elbo)) elbo = -INFINITY;
    if (elbo < elbo_best && elbo_best > elbo_init) tuned = 1;
    else if (e < 4) { elbo_best = elbo; eta_best = eta_seq[e]; }
    else { if (elbo > elbo_init) { eta_best = eta_seq[e]; tuned = 1; } else { free(buf); return -4; } }
    ADVI_RESET();
  }
  int cb_size = (int)fmax(0.1 * cfg->iter / cfg->eval_elbo, 2.0), ncb = 0;
  double* cb = calloc(cb_size, sizeof(double)); double* srt = calloc(cb_size, sizeof(double));
  double elbo = 0, elbo_prev = -INFINITY; int converged = 0, iters = 0;
  uint32_t id = a.draw_id++; advi_grad_at(&a, id);
  for (int it = 1; it <= cfg->iter && !converged; ++it) {
    advi_step(&a, eta_best, it, &id); iters = it;
    if (it % cfg->eval_elbo == 0) {
      elbo_prev = elbo; elbo = advi_calc_elbo(&a);
      double delta = fabs((elbo - elbo_prev) / elbo);
      if (ncb < cb_size) cb[ncb++] = delta; else { memmove(cb, cb + 1, sizeof(double) * (cb_size - 1)); cb[cb_size - 1] = delta; }
      double mean = 0; for (int k = 0; k < ncb; ++k) mean += cb[k]; mean /= ncb;
      memcpy(srt, cb, sizeof(double) * ncb); qsort(srt, ncb, sizeof(double), cmp_d);
      double med = ncb % 2 ? srt[ncb / 2] : 0.5 * (srt[ncb / 2 - 1] + srt[ncb / 2]);
      if (mean < cfg->tol_rel_obj || med < cfg->tol_rel_obj) converged = 1;
      if (!converged) { id = a.draw_id++; advi_grad_at(&a, id); }
    }
  }
  for (int r = 0; r < cfg->output_samples; ++r) { uint32_t d = a.draw_id++; for (int i = 0; i < D; ++i) out_draws[(size_t)r * D + i] = a.mu[i] + exp(a.om[i]) * advi_eta((uint32_t)i, d, a.k0); }
  if (mu_out) memcpy(mu_out, a.mu, sizeof(double) * D);
  if (omega_out) memcpy(omega_out, a.om, sizeof(double) * D);
  if (info) { info[0] = iters; info[1] = converged; info[2] = elbo; info[3] = eta_best; }
  free(cb); free(srt); free(buf);
  return 0;
}
