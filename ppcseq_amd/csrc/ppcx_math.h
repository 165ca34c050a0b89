// ppcx_math.h -- scalar building blocks of the MI355X NB hierarchical engine.
//
// Everything here is `__host__ __device__` so the *same* arithmetic is compiled into the gfx950
// kernels (ppcx_kernels.hip) and into the CPU emulation harness that tests/ uses to check the host
// logic without a GPU (tests/emul). Nothing here comes from the reference: Stan Math (third party,
// not vendored) is where the reference's arithmetic lives; the formulas below are written from the
// published definitions cited at each function.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PPCX_HD __host__ __device__ __forceinline__
#else
#define PPCX_HD inline
#endif

namespace ppcx {

// ---------------------------------------------------------------------------------------------
// Lean fp64 reciprocal and logarithm for the per-cell loop. ocml's log()/division are correctly
// rounded via double-double arithmetic (~65 and ~22 VALU instructions); the cell loop is bound by
// fp64 issue, so it uses:
//   fast_rcp : v_rcp_f64 seed (accurate to 4.6e-8 only) + one Newton step, two FMAs: relative error 1.2e-16 on average,
//              2.2e-15 at most (scripts/micro/rcp_accuracy.hip, profiles/r03_valu_rates_micro.txt)
//   fast_log : the classic argument reduction x = 2^k m, m in [sqrt(1/2), sqrt 2), s = f/(2+f),
//              log m = f - (f^2/2 - s (f^2/2 + R(s^2))) with the 7-term minimax R of Sun's fdlibm
//              e_log.c (public domain algorithm; max error < 1 ulp), ~30 VALU instructions.
// Valid for finite x > 0 (x = 1 + exp(t) >= 1 or x = y + phi > 0 in this code); inf/NaN propagate.
// ---------------------------------------------------------------------------------------------
PPCX_HD double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = __builtin_amdgcn_rcp(x);
  return fma(r, fma(-x, r, 1.0), r);
#else
  return 1.0 / x;
#endif
}

PPCX_HD double fast_log(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double m = __builtin_amdgcn_frexp_mant(x);      // [0.5, 1)
  int k = __builtin_amdgcn_frexp_exp(x);
#else
  int k;
  double m = frexp(x, &k);
#endif
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;                               // [sqrt(1/2), sqrt 2)
  k = lo ? k - 1 : k;
  const double f = m - 1.0;
  const double s = f * fast_rcp(2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  // k ln2_hi - ((hfsq - (s (hfsq + R) + k ln2_lo)) - f)
  return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// ---------------------------------------------------------------------------------------------
// Table-driven fp64 logarithm for the cell loop (no division, no frexp, no selects): for x = 2^e m,
// m in [1,2), the top 8 mantissa bits j pick c_j = 1 + (j + 1/2)/256; with r = m/c_j - 1 (one FMA with the
// tabulated 1/c_j, |r| < 2^-9)  log x = e ln2 + log c_j + log1p(r), log1p by its degree-5 Taylor polynomial
// (truncation < r^6/6 ~ 9e-18). The table is two arrays of 256 doubles, tab[j] = 1/c_j and tab[256 + j] =
// log c_j (4 KB), in LDS on the device: two 8-byte reads, whose 64-bank mapping spreads the lanes' random j
// over 32 bank pairs (one 16-byte {1/c, log c} read maps them to 8 bank quads and serialises 4-5 deep).
// Valid for finite normal x > 0; inf/NaN inputs give finite garbage, which the callers catch through the
// non-finite gradient that accompanies them (u = inf makes u/w NaN).
// ---------------------------------------------------------------------------------------------
constexpr int kLogTabSize = 256;          // entries; the table holds 2 * kLogTabSize doubles
constexpr int kLogTabBits = 8;

PPCX_HD unsigned long long dbl_bits(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (unsigned long long)__double_as_longlong(x);
#else
  unsigned long long b; __builtin_memcpy(&b, &x, 8); return b;
#endif
}
PPCX_HD double bits_dbl(unsigned long long b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double((long long)b);
#else
  double x; __builtin_memcpy(&x, &b, 8); return x;
#endif
}
inline void fill_log_table(double* t /* 2 * kLogTabSize */) {          // host: exact-to-rounding entries
  // entries for the mantissa m' = frexp(x) in [1/2, 1):  t[j] = 2/c_j (so r = m' t[j] - 1), t[256 + j] = log(1/t[j]) = log(c_j/2)
  for (int j = 0; j < kLogTabSize; ++j) {
    const long double c = 1.0L + ((long double)j + 0.5L) / (long double)kLogTabSize;
    t[j] = (double)(2.0L / c);
    t[kLogTabSize + j] = (double)logl(1.0L / (long double)t[j]);   // log of the reciprocal actually stored
  }
}
PPCX_HD double table_log(double x, const double* tab) {
  const int j = (int)(dbl_bits(x) >> (52 - kLogTabBits)) & (kLogTabSize - 1);
#if defined(__HIP_DEVICE_COMPILE__)
  const double m = __builtin_amdgcn_frexp_mant(x);      // v_frexp_mant_f64 / v_frexp_exp_i32_f64: [1/2, 1) x 2^e
  const int e = __builtin_amdgcn_frexp_exp(x);
#else
  int e;
  const double m = frexp(x, &e);
#endif
  const double cinv = tab[j], logc = tab[kLogTabSize + j];
  const double r = fma(m, cinv, -1.0);
  double p = fma(r, 0.2, -0.25);
  p = fma(r, p, 1.0 / 3.0);
  p = fma(r, p, -0.5);
  p = fma(r, p, 1.0);
  return fma((double)e, 6.93147180559945286227e-01, fma(r, p, logc));
}

// The WINDOWED table (round 5) for arguments known to lie in [1, 16) -- the cells of one gene after scaling by a power of two
// (ppcx_model.h GeneWindow): the index is the two low bits of the exponent field and the top 8 mantissa bits, the entries
// are 1/c and log c of the bin's centre c = 2^e (1 + (j + 1/2)/256), so that r = x/c - 1 comes from ONE multiply-add with x
// itself and log x = log c + log1p(r): no v_frexp_mant / v_frexp_exp / conversion / exponent multiply-add per cell.
// 1024 pairs of doubles (16 KB), wt[2 j] = 1/c_j and wt[2 j + 1] = log c_j: ONE 16-byte LDS read per cell. (A random gather of
// the lanes' entries costs 11 LDS cycles per wave-instruction as one ds_read_b128 or as two ds_read_b64, and 19.5 as the
// ds_read2st64_b64 that hipcc makes of two arrays: scripts/micro/lds_gather.hip.)
constexpr int kWinTabBits = 10, kWinTabSize = 1 << kWinTabBits;
constexpr int kWinBinades = 4;            // [1, 2), [2, 4), [4, 8), [8, 16)
inline void fill_window_log_table(double* t /* 2 * kWinTabSize */) {    // host
  for (int j = 0; j < kWinTabSize; ++j) {
    const int e2 = j >> 8, m8 = j & 255;            // e2: exponent field & 3; field = 1023 + e  =>  e = (e2 + 1) & 3
    const int e = (e2 + 1) & 3;
    const long double c = ldexpl(1.0L + ((long double)m8 + 0.5L) / 256.0L, e);
    t[2 * j] = (double)(1.0L / c);
    t[2 * j + 1] = (double)logl(1.0L / (long double)t[2 * j]);        // log of the reciprocal actually stored
  }
}
struct WinEntry { double cinv, logc; };
PPCX_HD WinEntry window_entry(const double* wt, int j) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  const v2d_t e = *reinterpret_cast<const v2d_t*>(wt + 2 * j);         // 16-byte aligned: one ds_read_b128
  return WinEntry{e.x, e.y};
#else
  return WinEntry{wt[2 * j], wt[2 * j + 1]};
#endif
}
// log x for x in [1, 16) (the host's statement of the windowed cell's logarithm, ppcx_model.h cell_back_win)
PPCX_HD double window_log(double x, const double* wt) {
  const int j = (int)(dbl_bits(x) >> (52 - 8)) & (kWinTabSize - 1);
  const WinEntry e = window_entry(wt, j);
  const double r = fma(x, e.cinv, -1.0);
  double p = fma(r, 0.2, -0.25);
  p = fma(r, p, 1.0 / 3.0);
  p = fma(r, p, -0.5);
  p = fma(r, p, 1.0);
  return fma(r, p, e.logc);
}

// exp(x) for |x| < 700 by x = k ln2 + r, |r| <= ln2/2 and the rational form of Sun's fdlibm e_exp.c
// (public domain algorithm; error < 1 ulp): c = r - r^2 P(r^2), exp(r) = 1 + r + r c / (2 - c).
PPCX_HD double fast_exp(double x) {
  const double kf = rint(x * 1.44269504088896338700e+00);
  const double hi = fma(kf, -6.93147180369123816490e-01, x);
  const double lo = kf * 1.90821492927058770002e-10;
  const double r = hi - lo;
  const double t = r * r;
  const double c = r - t * (1.66666666666666019037e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05 +
                   t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
  const double y = 1.0 - ((lo - (r * c) * fast_rcp(2.0 - c)) - hi);
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ldexp(y, (int)kf);
#else
  return ldexp(y, (int)kf);
#endif
}

// ---------------------------------------------------------------------------------------------
// Stirling tails for x >= 8, r = 1/x (A&S 6.1.40, 6.3.18 give the asymptotic series):
//   lgamma(x)  = (x - 1/2) ln x - x + ln(2 pi)/2 + lg_tail(r),      lg_tail(r) = r F(r^2)
//   digamma(x) = ln x - dg_tail(r),                                  dg_tail(r) = r/2 + r^2 G(r^2)
// F and G (both 1/12 - ... at 0) are smooth on r^2 in [0, 1/64]; instead of the asymptotic series (7 terms for 2e-15
// at x = 8) they are evaluated by the degree-4 polynomials that interpolate them at the Chebyshev nodes of that
// interval (scripts/fit/stirling_tails.py, mpmath at 60 digits): maximum absolute error of the double-precision
// evaluation 3.7e-16 (lg_tail) and 4.9e-16 (dg_tail) over x in [8, 1e6] -- one regime for every x >= 8.
// ---------------------------------------------------------------------------------------------
// Shorter polynomials of the same F and G for large arguments (same script, other intervals): degree 2 on x >= 32
// (maximum absolute error 5.4e-16 / 1.2e-16, the size of the degree-4 polynomials' own error), degree 1 on x >= 256
// (9.0e-17 / 1.8e-18). The log-likelihood kernel uses them for whole passes of genes whose smallest row-sweep count is
// that large (ppcx_gene.h lane_gene_sums; the host orders the genes by that tier).
constexpr int kTailX2 = 32, kTailX1 = 256;
constexpr double kStirlingF2[3] = {8.33333333333160509e-02, -2.77777745910256493e-03, 7.92780214438562267e-04};
constexpr double kStirlingG2[3] = {8.33333333332123838e-02, -8.33333110387448305e-03, 3.96216261085104680e-03};
constexpr double kStirlingF1[2] = {8.33333333333102361e-02, -2.77776566774899170e-03};
constexpr double kStirlingG1[2] = {8.33333333332178378e-02, -8.33327278343192618e-03};
// tier of a gene from its smallest count among the row-sweep cells (counts >= 8): 2, 1 or 0
PPCX_HD int tail_tier(int min_sweep_count) { return min_sweep_count >= kTailX1 ? 2 : (min_sweep_count >= kTailX2 ? 1 : 0); }
// ... granted only to a gene ALL of whose S cells are row-sweep cells (no count below 8, none excluded): a pass of tier 1 or 2
// then evaluates every cell without looking at its count (no compare, no branch per cell)
PPCX_HD int gene_tier(int min_sweep_count, int n_sweep, int S) { return n_sweep == S ? tail_tier(min_sweep_count) : 0; }

PPCX_HD void stirling_tails(double rx, double* lgt, double* dgt) {
  const double r2 = rx * rx;
  double t = fma(r2, 7.72651446721163817e-04, -5.94317590856362882e-04);
  t = fma(r2, t, 7.93645716111539040e-04);
  t = fma(r2, t, -2.77777776791245188e-03);
  t = fma(r2, t, 8.33333333333302478e-02);
  *lgt = rx * t;
  double d = fma(r2, 6.82627523986508236e-03, -4.15672846375406482e-03);
  d = fma(r2, d, 3.96819926156938719e-03);
  d = fma(r2, d, -8.33333322714054948e-03);
  d = fma(r2, d, 8.33333333333001886e-02);
  *dgt = fma(r2, d, 0.5 * rx);
}
// The "Stirling excess" of the dispersion phi, the per-gene constants of the cell loop (ppcx_model.h):
//   dlt = lgamma(phi) - [(phi - 1/2) ln phi - phi + ln(2 pi)/2]          dps = ln phi - digamma(phi)
// For phi >= 8 they ARE the Stirling tails of 1/phi (no cancellation); below, phi is shifted by 8 and the leading
// terms are subtracted analytically. lnphi is passed in because the caller knows it exactly (ln phi = -sigma_raw).
// `any_small` (wave-uniform on the device) says whether some lane needs the shifted form.
PPCX_HD void stirling_excess(double phi, double lnphi, const double* tab, bool any_small, double* dlt, double* dps) {
  const bool small = phi < 8.0;
  const double xs = small ? phi + 8.0 : phi;
  const double rs = fast_rcp(xs);
  double lgt, dgt;
  stirling_tails(rs, &lgt, &dgt);
  *dlt = lgt; *dps = dgt;
  if (any_small) {
    double P = phi, dP = 1.0;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const double f = phi + (double)k;
      dP = fma(dP, f, P);
      P = P * f;
    }
    const double lxs = table_log(xs, tab), lP = table_log(P, tab);
    // lgamma(phi) = (xs - 1/2) ln xs - xs + c + lgt - ln P ;  minus (phi - 1/2) ln phi - phi + c
    const double d1 = (xs - 0.5) * lxs - 8.0 - lP - (phi - 0.5) * lnphi + lgt;
    const double d2 = (lnphi - lxs) + dgt + dP * fast_rcp(P);
    *dlt = small ? d1 : lgt;
    *dps = small ? d2 : dgt;
  }
}

// ---------------------------------------------------------------------------------------------
// log erfc(x) and  R(x) = exp(-x^2)/erfc(x)  (the inverse Mills-type ratio the skew-normal
// gradient needs). Written as Stan Math's skew_normal_lpdf evaluates it -- log(erfc(.)) directly --
// so that where erfc underflows (x > ~26.5) the density is log(0) = -inf exactly as in the reference:
// such points are rejected as initial values and count as divergent proposals, the same as in Stan.
// ---------------------------------------------------------------------------------------------
PPCX_HD void log_erfc_and_ratio(double x, double* log_erfc, double* ratio) {
  const double e = erfc(x);
  *log_erfc = e > 0.0 ? fast_log(e) : -INFINITY;            // log(0) = -inf, as libm's log gives it
  const double x2 = x * x;
  *ratio = (x2 < 700.0 ? fast_exp(-x2) : 0.0) / e;          // e = 0 only for x > 26.5: 0/0 = NaN, as exp(-x^2)/erfc(x) there
}

PPCX_HD double log_sum_exp(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const double m = a > b ? a : b;
  return m + fast_log(1.0 + fast_exp(-fabs(a - b)));
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).
// Counter-based, so every coordinate / cell / draw addresses its own stream without state.
// Stream addressing is documented in DESIGN.md ("RNG").
// ---------------------------------------------------------------------------------------------
struct u4 { uint32_t x, y, z, w; };

PPCX_HD u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u4{c0, c1, c2, c3};
}
// 53 random bits -> (0,1), never 0 or 1
PPCX_HD double u01(uint32_t a, uint32_t b) {
  const uint64_t v = (((uint64_t)a << 32) | b) >> 11;
  return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}
PPCX_HD uint32_t seed32(uint64_t s) { return (uint32_t)s ^ (uint32_t)((s >> 32) * 0x9E3779B9u); }

// standard normal for coordinate i: Box-Muller on block (i>>1); cosine branch for even i
PPCX_HD double coord_normal(uint32_t i, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const u4 r = philox4x32_10(i >> 1, c1, c2, c3, k0, k1);
  const double u1 = u01(r.x, r.y), u2 = u01(r.z, r.w);
  const double rad = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925 * u2;
  return (i & 1u) ? rad * sin(t) : rad * cos(t);
}
PPCX_HD double coord_uniform(uint32_t i, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const u4 r = philox4x32_10(i, c1, c2, c3, k0, k1);
  return u01(r.x, r.y);
}

// sequential stream over a fixed (key, c1, c2, c3): c0 is the block index. 2 uniforms per block.
struct Stream {
  uint32_t k0, k1, c1, c2, c3, blk;
  int have; double b0, b1;
  int have_n; double spare;
  PPCX_HD void init(uint32_t k0_, uint32_t k1_, uint32_t c1_, uint32_t c2_, uint32_t c3_) {
    k0 = k0_; k1 = k1_; c1 = c1_; c2 = c2_; c3 = c3_; blk = 0; have = 0; have_n = 0; b0 = b1 = spare = 0.0;
  }
  PPCX_HD double uniform() {
    if (have == 0) {
      const u4 r = philox4x32_10(blk++, c1, c2, c3, k0, k1);
      b1 = u01(r.x, r.y); b0 = u01(r.z, r.w); have = 2;
    }
    --have;
    return have == 1 ? b1 : b0;
  }
  PPCX_HD double normal() {
    if (have_n) { have_n = 0; return spare; }
    const double u1 = uniform(), u2 = uniform();
    const double rad = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925 * u2;
    spare = rad * sin(t); have_n = 1;
    return rad * cos(t);
  }
};

// The transcendental functions of the posterior-predictive draws. On the device the lean versions above (fast_log, fast_exp:
// < 1 ulp) and a sine / cosine of 2 pi u reduced on u itself; on the host libm. The integers drawn agree with the oracle's
// (libm) unless an accept / reject comparison or a floor() is decided by the last bits: ~1e-13 per draw.
#if defined(__HIP_DEVICE_COMPILE__)
PPCX_HD double rng_log(double x) { return fast_log(x); }
PPCX_HD double rng_exp(double x) { return (x > -700.0 && x < 700.0) ? fast_exp(x) : exp(x); }
PPCX_HD double rng_div(double a, double b) { return a * fast_rcp(b); }      // v_rcp_f64 + Newton instead of the ~30-instruction division
// sin(2 pi u), cos(2 pi u), u in (0, 1): the quadrant and the reduced argument are exact in u; kernel polynomials of
// Sun's fdlibm k_sin.c / k_cos.c (public domain algorithm) on [0, pi/4]
PPCX_HD void sincos_2pi(double u, double* sn, double* cs) {
  const double q4 = floor(u * 4.0);
  double r = u - q4 * 0.25;                              // [0, 1/4), exact
  const bool flip = r > 0.125;
  r = flip ? 0.25 - r : r;                               // [0, 1/8], exact
  const double t = r * 6.283185307179586476925, z = t * t;
  const double ps = z * (-1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
  const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double s0 = fma(t, ps, t), c0 = fma(z, pc, fma(-0.5, z, 1.0));
  const double sq = flip ? c0 : s0, cq = flip ? s0 : c0;  // of the angle inside the quadrant
  const int q = (int)q4 & 3;
  *sn = q == 0 ? sq : (q == 1 ? cq : (q == 2 ? -sq : -cq));
  *cs = q == 0 ? cq : (q == 1 ? -sq : (q == 2 ? -cq : sq));
}
// lgamma(k + 1) for an integer k >= 0 held in a double: a table of ln k! below 8, Stirling with the tail polynomial above
PPCX_HD double lgamma_int1(double kf) {
  if (kf < 8.0) {
    const int k = (int)kf;
    return k < 2 ? 0.0 : (k == 2 ? 6.93147180559945286e-01 : (k == 3 ? 1.79175946922805496e+00 : (k == 4 ? 3.17805383034794575e+00 :
           (k == 5 ? 4.78749174278204581e+00 : (k == 6 ? 6.57925121201010121e+00 : 8.52516136106541467e+00)))));
  }
  const double x = kf + 1.0, rx = fast_rcp(x);
  double lgt, dgt;
  stirling_tails(rx, &lgt, &dgt);
  return (x - 0.5) * fast_log(x) - x + 9.18938533204672742e-01 + lgt;
}
#else
PPCX_HD double rng_log(double x) { return log(x); }
PPCX_HD double rng_exp(double x) { return exp(x); }
PPCX_HD double rng_div(double a, double b) { return a / b; }
PPCX_HD void sincos_2pi(double u, double* sn, double* cs) { const double t = 6.283185307179586476925 * u; *sn = sin(t); *cs = cos(t); }
PPCX_HD double lgamma_int1(double kf) { return lgamma(kf + 1.0); }
#endif

// sequential stream of the posterior-predictive draws: as Stream, with the lean functions
struct RngStream {
  uint32_t k0, k1, c1, c2, c3, blk;
  int have; double b0, b1;
  int have_n; double spare;
  PPCX_HD void init(uint32_t k0_, uint32_t k1_, uint32_t c1_, uint32_t c2_, uint32_t c3_) {
    k0 = k0_; k1 = k1_; c1 = c1_; c2 = c2_; c3 = c3_; blk = 0; have = 0; have_n = 0; b0 = b1 = spare = 0.0;
  }
  PPCX_HD double uniform() {
    if (have == 0) {
      const u4 r = philox4x32_10(blk++, c1, c2, c3, k0, k1);
      b1 = u01(r.x, r.y); b0 = u01(r.z, r.w); have = 2;
    }
    --have;
    return have == 1 ? b1 : b0;
  }
  PPCX_HD double normal() {                      // Box-Muller; the sine branch is the spare
    if (have_n) { have_n = 0; return spare; }
    const double u1 = uniform(), u2 = uniform();
    const double rad = sqrt(-2.0 * rng_log(u1));
    double sn, cs;
    sincos_2pi(u2, &sn, &cs);
    spare = rad * sn; have_n = 1;
    return rad * cs;
  }
};

// Gamma(shape a, scale 1): Marsaglia & Tsang, "A simple method for generating gamma variables" (2000), as begin + attempts:
// the posterior-predictive kernel runs the attempts of a wavefront's lanes in ONE loop in which a lane that has accepted
// goes on to its next draw at once, instead of every draw waiting for the slowest lane's rejections.
// Stream: key (seed32, 'PPC1'), counter (block, cell, draw, 4).
struct GammaDraw { RngStream st; double d, c, boost; };
PPCX_HD void gamma_begin(GammaDraw& g, double a, uint32_t k0, uint32_t cell, uint32_t draw) {
  g.st.init(k0, 0x50504331u, cell, draw, 4u);
  g.boost = 1.0;
  if (a < 1.0) { g.boost = pow(g.st.uniform(), 1.0 / a); a += 1.0; }
  g.d = a - 1.0 / 3.0; g.c = rng_div(1.0, sqrt(9.0 * g.d));
}
PPCX_HD bool gamma_attempt(GammaDraw& g, double* out) {
  const double x = g.st.normal();
  double v = 1.0 + g.c * x;
  if (v <= 0.0) return false;
  v = v * v * v;
  const double u = g.st.uniform();
  const double x2 = x * x;
  if (u < 1.0 - 0.0331 * x2 * x2 || rng_log(u) < 0.5 * x2 + g.d * (1.0 - v + rng_log(v))) { *out = g.d * v * g.boost; return true; }
  return false;
}
// Poisson(lam): Knuth multiplication below 10, Hoermann's PTRS (1993) above; begin + attempts likewise.
// Stream: key (seed32, 'PPC1'), counter (block, cell, draw, 8) -- its own stream (round 4), so that the gamma part's stream
// state need not travel with the draw from the gamma loop to the Poisson loop.
struct PoissonDraw {
  RngStream st; double lam;
  double L, p; long long k;                     // Knuth
  double loglam, b, a, invalpha, vr;            // PTRS
  int n_att;
};
PPCX_HD void poisson_begin(PoissonDraw& q, double lam, uint32_t k0, uint32_t cell, uint32_t draw) {
  q.st.init(k0, 0x50504331u, cell, draw, 8u);
  q.lam = lam; q.n_att = 0; q.k = 0; q.L = 0.0; q.p = 0.0; q.loglam = 0.0; q.b = 0.0; q.a = 0.0; q.invalpha = 0.0; q.vr = 0.0;
  if (lam < 10.0) { q.L = rng_exp(-lam); q.p = q.st.uniform(); }
  else {
    const double slam = sqrt(lam);
    q.loglam = rng_log(lam);
    q.b = 0.931 + 2.53 * slam; q.a = -0.059 + 0.02483 * q.b;
    q.invalpha = 1.1239 + rng_div(1.1328, q.b - 3.4); q.vr = 0.9277 - rng_div(3.6224, q.b - 2.0);
  }
}
PPCX_HD bool poisson_attempt(PoissonDraw& q, long long* out) {
  if (q.lam < 10.0) {
    if (!(q.p > q.L) || q.k >= 4096) { *out = q.k; return true; }
    ++q.k; q.p *= q.st.uniform();
    return false;
  }
  if (++q.n_att > 4096) { *out = (long long)q.lam; return true; }     // unreachable for finite lam (acceptance > 85 %)
  const double U = q.st.uniform() - 0.5, V = q.st.uniform();
  const double us = 0.5 - fabs(U);
  const double kf = floor((rng_div(2.0 * q.a, us) + q.b) * U + q.lam + 0.43);
  if (us >= 0.07 && V <= q.vr) { *out = (long long)kf; return true; }
  if (kf < 0 || (us < 0.013 && V > us)) return false;
  if (rng_log(V) + rng_log(q.invalpha) - rng_log(rng_div(q.a, us * us) + q.b) <= -q.lam + kf * q.loglam - lgamma_int1(kf)) { *out = (long long)kf; return true; }
  return false;
}
// neg_binomial_2_log_rng(eta, phi) as a gamma-Poisson mixture (inst/stan/negBinomial_MPI.stan:264); Stan raises above 2^30,
// we saturate. lam_of_gamma: the Poisson mean from the gamma variate.
PPCX_HD bool nb2_invalid(double eta, double phi) { return !(phi > 0.0) || !isfinite(phi) || !isfinite(eta); }
PPCX_HD int32_t nb2_log_rng(double eta, double phi, uint32_t k0, uint32_t cell, uint32_t draw) {
  if (nb2_invalid(eta, phi)) return 2147483647;                       // invalid draw: sorts last
  GammaDraw g; gamma_begin(g, phi, k0, cell, draw);
  double gam = 0.0;
  int it = 0;
  while (!gamma_attempt(g, &gam)) if (++it >= 4096) { gam = g.d * g.boost; break; }   // bounded: every wave must be able to leave the loop
  const double lam = gam * rng_div(rng_exp(eta), phi);
  if (!(lam < 1073741824.0)) return 1073741823;
  PoissonDraw q; poisson_begin(q, lam, k0, cell, draw);
  long long k = 0;
  while (!poisson_attempt(q, &k)) {}
  return k > 2147483647LL ? 2147483647 : (int32_t)k;
}

}  // namespace ppcx
