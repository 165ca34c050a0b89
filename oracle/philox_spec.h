/* TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.
 *
 * Philox4x32-10 counter-based generator (Salmon et al., SC'11), restated from the
 * published algorithm. The oracle and the HIP product each carry their OWN copy of
 * this specification (this file is only ever included from oracle/); sharing the
 * *specification* (not the code) is what lets the GPU sampler and the CPU oracle be
 * compared draw-for-draw. The reference (rstan) uses boost::ecuyer1988, which is not
 * in the container; bit-identical draws vs Stan are not a goal.
 *
 * Stream addressing used everywhere in this project:
 *   key     = (seed_lo, seed_hi ^ stream_tag)
 *   counter = (c0, c1, c2, c3) chosen by the caller (documented at each call site)
 */
#ifndef PPCO_PHILOX_SPEC_H
#define PPCO_PHILOX_SPEC_H
#include <stdint.h>
#include <math.h>

typedef struct { uint32_t v[4]; } ppco_u4;

static inline ppco_u4 ppco_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                         uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  ppco_u4 o = {{c0, c1, c2, c3}};
  return o;
}

/* two 32-bit words -> double in (0,1): 53 random bits, never 0 or 1 */
static inline double ppco_u01(uint32_t a, uint32_t b) {
  uint64_t x = (((uint64_t)a << 32) | b) >> 11; /* 53 bits */
  return ((double)x + 0.5) * (1.0 / 9007199254740992.0);
}

/* A sequential stream over a fixed (key, c1, c2, c3) with c0 as the running block index. */
typedef struct {
  uint32_t k0, k1, c1, c2, c3, blk;
  int have;          /* doubles still buffered (0..2) */
  double buf[2];
  int have_n; double spare_n; /* Box-Muller spare */
} ppco_stream;

static inline void ppco_stream_init(ppco_stream* s, uint32_t k0, uint32_t k1,
                                    uint32_t c1, uint32_t c2, uint32_t c3) {
  s->k0 = k0; s->k1 = k1; s->c1 = c1; s->c2 = c2; s->c3 = c3; s->blk = 0;
  s->have = 0; s->have_n = 0; s->spare_n = 0.0;
}
static inline double ppco_stream_uniform(ppco_stream* s) {
  if (s->have == 0) {
    ppco_u4 r = ppco_philox4x32_10(s->blk++, s->c1, s->c2, s->c3, s->k0, s->k1);
    s->buf[1] = ppco_u01(r.v[0], r.v[1]);   /* handed out first  */
    s->buf[0] = ppco_u01(r.v[2], r.v[3]);   /* handed out second */
    s->have = 2;
  }
  return s->buf[--s->have];
}
/* N(0,1) by Box-Muller; the sine branch is kept as a spare for the next call. */
static inline double ppco_stream_normal(ppco_stream* s) {
  if (s->have_n) { s->have_n = 0; return s->spare_n; }
  double u1 = ppco_stream_uniform(s), u2 = ppco_stream_uniform(s);
  double r = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925 * u2;
  s->spare_n = r * sin(t); s->have_n = 1;
  return r * cos(t);
}
#endif
