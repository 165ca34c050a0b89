/* dot_c_host.c -- a plain-C caller of the drop-in boundary, with exactly the argument shapes R's .C() passes.
 *
 * Test infrastructure (tests/test_gpu_chost.py runs it as a child process on the GPU box). No Python, no C++, no HIP
 * headers: the library is bound with dlopen() and `ppcx_do_inference_C` is called the way R calls a routine registered for
 * .C() -- every argument a pointer (int*, double*, char** for a character vector), void return -- which is what the shim
 * r/ppcx_do_inference.R does (it replaces R/utilities.R:1482-1531; the reference registers its native code through
 * src/RcppExports.cpp:15-25 and NAMESPACE:143 useDynLib).
 *
 * It runs the reference's testthat case (tests/testthat/test-ppcSeq.R:7-32: bundled `counts`, three checked genes + 50
 * negative controls, ~ Label, percent_false_positive_genes = 1, the defaults = ADVI + approximated analysis) as
 * identify_outliers does (R/methods.R:155-167, :268-342): the discovery pass, the cells it flags as deleterious excluded,
 * the test pass with truncation_compensation = 0.7352941; then the flag rules of R/utilities.R:651-663 and :493-513, and
 * prints tot_deleterious_outliers of the checked genes -- the reference's only assertion is that they are `0 1 0`.
 *
 * usage: dot_c_host <libppcx.so> <fixture.txt> [seed] [nuts]
 * build: gcc -O2 -o dot_c_host dot_c_host.c -ldl -lm
 */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef void (*do_inference_fn)(const int* dims, const int* counts, const double* X, const double* exposure,
                                const int* excl, const double* reals, double* ci, double* slope, int* counts_rng,
                                int* status, char** errbuf, const int* errlen);

static int g_n_devices = 0;     /* argv[5]: the chains of a NUTS pass dealt to that many devices (all of them device 0 on a one-GPU box) */

/* find_optimal_number_of_chains, R/utilities.R:291-303 */
static int optimal_chains(double draws) {
  int best = 2; double best_tot = 1e300;
  for (int c = 2; c <= 100; ++c) { const double t = draws / c + 150.0 * c; if (t < best_tot) { best_tot = t; best = c; } }
  return best;
}

typedef struct { int G, S, C, K; int* counts; double* X; double* expo; } Data;

/* one do_inference() pass (R/utilities.R:1321-1547) through the .C() entry; flags[g*S+s] = deleterious_outliers */
static int pass(do_inference_fn f, const Data* d, int vb, int approx_analysis, double thr, double draws, int n_excl,
                const int* excl, double trunc, double seed, int* flags, double* upper_out) {
  const int G = d->G, S = d->S, C = d->C, K = d->K;
  const int draws_practical = approx_analysis ? 1000 : (int)draws;               /* R/utilities.R:1372 */
  int chains = optimal_chains(draws_practical); if (chains < 3) chains = 3; if (chains > 4) chains = 4;  /* cores = 4 */
  const int iter = (int)ceil((double)draws_practical / chains) + 150;             /* :1502 */
  const int n_gen = approx_analysis ? (int)draws : 0;
  int dims[33] = {400, 0, G, S, C, K, n_excl, chains, iter, 150, n_gen, approx_analysis, vb, 0, draws_practical, 50000,
                  g_n_devices, 0};                                                /* n_devices, devices[16] (all device 0 here) */
  double reals[6] = {5.612671, trunc, thr, 1.0 - thr, seed, 0.005};
  double* ci = (double*)calloc((size_t)K * S * 4, sizeof(double));
  double* slope = (double*)calloc((size_t)K, sizeof(double));
  int rng_dummy[1] = {0}, status[1] = {-99}, errlen[1] = {256};
  char msg[256]; memset(msg, ' ', 255); msg[255] = 0;
  char* errbuf[1] = {msg};
  int excl_dummy[1] = {0};
  f(dims, d->counts, d->X, d->expo, n_excl > 0 ? excl : excl_dummy, reals, ci, slope, rng_dummy, status, errbuf, errlen);
  if (status[0] != 0) { fprintf(stderr, "ppcx error %d: %s\n", status[0], msg); free(ci); free(slope); return status[0]; }
  /* mean of X[,2] (R/utilities.R:499) */
  double xm = 0; for (int s = 0; s < S; ++s) xm += d->X[(size_t)S + s]; xm /= S;
  for (int g = 0; g < K; ++g) for (int s = 0; s < S; ++s) {
    const double* c4 = ci + ((size_t)g * S + s) * 4;
    const double y = (double)d->counts[(size_t)g * S + s];
    const int ppc = y >= c4[2] && y <= c4[3];                                     /* between(), inclusive: :657 */
    const int higher = !ppc && y > c4[0];                                         /* :658 */
    const int x_high = d->X[(size_t)S + s] > xm;
    const int group_high = (slope[g] > 0 && x_high) || (slope[g] < 0 && !x_high); /* :500-506 */
    flags[(size_t)g * S + s] = !ppc && (higher == group_high);                    /* :510 */
    if (upper_out) upper_out[(size_t)g * S + s] = c4[3];
  }
  free(ci); free(slope);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <libppcx.so> <fixture.txt> [seed] [nuts] [n_devices]\n", argv[0]); return 2; }
  if (argc > 5) g_n_devices = atoi(argv[5]);
  const double seed = argc > 3 ? atof(argv[3]) : 1.0;
  const int vb = !(argc > 4 && strcmp(argv[4], "nuts") == 0);
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  do_inference_fn f = (do_inference_fn)dlsym(h, "ppcx_do_inference_C");
  if (!f) { fprintf(stderr, "ppcx_do_inference_C not exported\n"); return 2; }
  FILE* fp = fopen(argv[2], "r");
  if (!fp) { perror(argv[2]); return 2; }
  Data d;
  if (fscanf(fp, "%d %d %d %d", &d.G, &d.S, &d.C, &d.K) != 4) return 2;
  d.counts = (int*)malloc(sizeof(int) * (size_t)d.G * d.S);
  d.X = (double*)malloc(sizeof(double) * (size_t)d.S * d.C);
  d.expo = (double*)malloc(sizeof(double) * (size_t)d.S);
  for (long i = 0; i < (long)d.G * d.S; ++i) if (fscanf(fp, "%d", &d.counts[i]) != 1) return 2;
  for (long i = 0; i < (long)d.S * d.C; ++i) if (fscanf(fp, "%lf", &d.X[i]) != 1) return 2;
  for (int s = 0; s < d.S; ++s) if (fscanf(fp, "%lf", &d.expo[s]) != 1) return 2;
  fclose(fp);
  const int S = d.S, K = d.K;
  /* thresholds of identify_outliers, pfp = 1, detrimental only (R/methods.R:156-167) */
  const double thr2 = 1.0 / 100.0 / S * 2.0, thr1 = fmax(0.05, 2.0 * thr2);
  const double draws1 = fmax(10.0 / thr1, 1000.0), draws2 = fmax(10.0 / thr2, 1000.0);
  int* flags = (int*)calloc((size_t)K * S, sizeof(int));
  double* upper = (double*)calloc((size_t)K * S, sizeof(double));
  /* pass 1: discovery -- always the full posterior analysis (R/methods.R:273) */
  int rc = pass(f, &d, vb, 0, thr1, draws1, 0, NULL, 1.0, seed, flags, NULL);
  if (rc) return 1;
  int n_excl = 0; int* excl = (int*)malloc(sizeof(int) * (size_t)K * S);
  for (int i = 0; i < K * S; ++i) if (flags[i]) excl[n_excl++] = i;               /* to_exclude, :292-300 */
  /* pass 2: test (R/methods.R:320-342) */
  rc = pass(f, &d, vb, 1, thr2, draws2, n_excl, excl, 0.7352941, seed, flags, upper);
  if (rc) return 1;
  for (int g = 0; g < K; ++g) {
    int tot = 0; for (int s = 0; s < S; ++s) tot += flags[(size_t)g * S + s];
    printf(g ? " %d" : "%d", tot);
  }
  printf("\n");
  for (int g = 0; g < K; ++g) for (int s = 0; s < S; ++s) if (flags[(size_t)g * S + s])
    printf("outlier gene %d sample %d count %d upper %.1f\n", g + 1, s + 1, d.counts[(size_t)g * S + s], upper[(size_t)g * S + s]);
  return 0;
}
