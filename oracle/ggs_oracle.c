/*
 * ggs_oracle.c -- CPU ORACLE (test infrastructure; see ggs_oracle.h header).
 *
 * Plain C restatement of the reference's Grouped-Gibbs arithmetic.  Build with
 *   gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp
 * (-ffp-contract=off matters: every product and sum below must round exactly
 * once, as the JVM's strict double arithmetic does).
 *
 * Reference paths are relative to src/main/java/cc/mallet/ in the reference
 * checkout.  "GGS" = topics/LDAGroupedGibbsSampler.java, "UPLDA" =
 * topics/UncollapsedParallelLDA.java, "MSLDA" = topics/ModifiedSimpleLDA.java.
 *
 * Third-party arithmetic that is NOT in the reference tree and is restated from
 * its published algorithm:
 *   - java.util.Random (JDK 8 spec): 48-bit LCG, nextInt(bound), nextDouble,
 *     nextGaussian (polar method using StrictMath.log/sqrt).
 *   - StrictMath.log / StrictMath.pow = fdlibm 5.3 e_log.c / e_pow.c.
 *   - cc.mallet.types.Dirichlet (MALLET 2.0.8) constructors:
 *       Dirichlet(double[] p): magnitude = sum p (in index order),
 *                              partition[i] = p[i] / magnitude
 *       Dirichlet(int n, double a): magnitude = n*a, partition[i] = 1.0/n
 *     so the gamma shape actually used is partition[i]*magnitude, which is
 *     NOT always bit-equal to p[i].  Call sites: GGS:70,190; UPLDA:1292.
 *   - Philox4x32-10 (Salmon et al., SC'11; Random123 constants).
 */
#include "ggs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* Philox4x32-10                                                             */
/* ------------------------------------------------------------------------ */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    if (r > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Two 32-bit words -> double in [0,1), built the way java.util.Random.nextDouble
 * builds it from next(26) and next(27): ((a26 << 27) + b27) * 2^-53. */
static inline double bits_to_double(uint32_t wa, uint32_t wb) {
  uint64_t a = wa >> 6, b = wb >> 5;
  return (double)((a << 27) + b) * 0x1.0p-53;
}

/* ------------------------------------------------------------------------ */
/* java.util.Random (used only for the seeded initial z, UPLDA:458-460)      */
/* ------------------------------------------------------------------------ */
typedef struct { uint64_t seed; } jrandom;
static const uint64_t JR_MULT = 0x5DEECE66DULL, JR_MASK = (1ULL << 48) - 1;
static void jr_init(jrandom *r, int64_t seed) { r->seed = ((uint64_t)seed ^ JR_MULT) & JR_MASK; }
static int32_t jr_next(jrandom *r, int bits) {
  r->seed = (r->seed * JR_MULT + 0xBULL) & JR_MASK;
  return (int32_t)(int64_t)(r->seed >> (48 - bits)); /* (int)(seed >>> (48-bits)) */
}
static int32_t jr_next_int_bound(jrandom *r, int32_t bound) {
  int32_t rr = jr_next(r, 31);
  int32_t m = bound - 1;
  if ((bound & m) == 0) {
    rr = (int32_t)(((int64_t)bound * (int64_t)rr) >> 31);
  } else {
    /* for (int u = r; u - (r = u % bound) + m < 0; u = next(31)); with int wrap */
    int32_t u = rr;
    for (;;) {
      rr = u % bound;
      int32_t t = (int32_t)((uint32_t)u - (uint32_t)rr + (uint32_t)m);
      if (t >= 0) break;
      u = jr_next(r, 31);
    }
  }
  return rr;
}
static double jr_next_double(jrandom *r) {
  int64_t a = jr_next(r, 26), b = jr_next(r, 27);
  return (double)((a << 27) + b) * 0x1.0p-53;
}
void orc_jrandom_next_ints(int64_t seed, int32_t bound, int64_t n, int32_t *out) {
  jrandom r; jr_init(&r, seed);
  for (int64_t i = 0; i < n; i++) out[i] = jr_next_int_bound(&r, bound);
}
void orc_jrandom_next_int_raw(int64_t seed, int64_t n, int32_t *out) {
  jrandom r; jr_init(&r, seed);
  for (int64_t i = 0; i < n; i++) out[i] = jr_next(&r, 32);
}
void orc_jrandom_next_doubles(int64_t seed, int64_t n, double *out) {
  jrandom r; jr_init(&r, seed);
  for (int64_t i = 0; i < n; i++) out[i] = jr_next_double(&r);
}

/* ------------------------------------------------------------------------ */
/* fdlibm e_log.c / e_pow.c (what StrictMath.log / StrictMath.pow compute)   */
/* ------------------------------------------------------------------------ */
static inline int32_t hi_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (int32_t)(u >> 32); }
static inline uint32_t lo_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
static inline double with_hi(double x, int32_t hi) {
  uint64_t u; memcpy(&u, &x, 8); u = ((uint64_t)(uint32_t)hi << 32) | (u & 0xffffffffULL);
  memcpy(&x, &u, 8); return x;
}
static inline double with_lo(double x, uint32_t lo) {
  uint64_t u; memcpy(&u, &x, 8); u = (u & 0xffffffff00000000ULL) | lo;
  memcpy(&x, &u, 8); return x;
}

double orc_log(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                      two54 = 1.80143985094819840000e+16,
                      Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                      Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                      Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                      Lg7 = 1.479819860511658591e-01;
  double hfsq, f, s, z, R, w, t1, t2, dk;
  int32_t k, hx, i, j;
  uint32_t lx;
  hx = hi_word(x); lx = lo_word(x);
  k = 0;
  if (hx < 0x00100000) {                 /* x < 2**-1022 */
    if (((hx & 0x7fffffff) | lx) == 0) return -INFINITY; /* log(+-0) = -inf */
    if (hx < 0) return NAN;              /* log(-#) = NaN */
    k -= 54; x *= two54;                 /* subnormal: scale up */
    hx = hi_word(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  i = (hx + 0x95f64) & 0x100000;
  x = with_hi(x, hx | (i ^ 0x3ff00000)); /* normalize x or x/2 */
  k += (i >> 20);
  f = x - 1.0;
  if ((0x000fffff & (2 + hx)) < 3) {     /* |f| < 2**-20 */
    if (f == 0.0) {
      if (k == 0) return 0.0;
      dk = (double)k; return dk * ln2_hi + dk * ln2_lo;
    }
    R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    dk = (double)k; return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  s = f / (2.0 + f);
  dk = (double)k;
  z = s * s;
  i = hx - 0x6147a;
  w = z * z;
  j = 0x6b851 - hx;
  t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  R = t2 + t1;
  if (i > 0) {
    hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  } else {
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
  }
}

/* fdlibm s_scalbn.c, reduced to the finite-input case e_pow.c needs */
static double fd_scalbn(double x, int n) {
  static const double two54 = 1.80143985094819840000e+16, twom54 = 5.55111512312578270212e-17,
                      huge = 1.0e+300, tiny = 1.0e-300;
  int32_t k, hx = hi_word(x);
  uint32_t lx = lo_word(x);
  k = (hx & 0x7ff00000) >> 20;
  if (k == 0) {
    if ((lx | (hx & 0x7fffffff)) == 0) return x;
    x *= two54; hx = hi_word(x);
    k = ((hx & 0x7ff00000) >> 20) - 54;
    if (n < -50000) return tiny * x;
  }
  if (k == 0x7ff) return x + x;
  k = k + n;
  if (k > 0x7fe) return huge * copysign(huge, x);
  if (k > 0) return with_hi(x, (hx & (int32_t)0x800fffff) | (k << 20));
  if (k <= -54) {
    if (n > 50000) return huge * copysign(huge, x);
    return tiny * copysign(tiny, x);
  }
  k += 54;
  x = with_hi(x, (hx & (int32_t)0x800fffff) | (k << 20));
  return x * twom54;
}

/* e_pow.c for the domain the sampler uses: x >= +0 finite, y finite.
 * (ParallelRandoms.java:66 calls Math.pow(u, 1.0/alpha) with u in [0,1),
 * alpha in (0,1).)  Negative x, NaN and infinities are outside that domain and
 * return NaN here. */
double orc_pow(double x, double y) {
  static const double bp[2] = {1.0, 1.5},
      dp_h[2] = {0.0, 5.84962487220764160156e-01}, dp_l[2] = {0.0, 1.35003920212974897128e-08},
      two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300,
      L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01,
      L3 = 3.33333329818377432918e-01, L4 = 2.72728123808534006489e-01,
      L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01,
      P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
      P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
      P5 = 4.13813679705723846039e-08,
      lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01,
      lg2_l = -1.90465429995776804525e-09, ovt = 8.0085662595372944372e-0017,
      cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01,
      cp_l = -7.02846165095275826516e-09, ivln2 = 1.44269504088896338700e+00,
      ivln2_h = 1.44269502162933349609e+00, ivln2_l = 1.92596299112661746887e-08;
  double z, ax, z_h, z_l, p_h, p_l, y1, t1, t2, r, t, u, v, w;
  int32_t i, j, k, n, hx, hy, ix, iy;
  uint32_t lx, ly;
  hx = hi_word(x); lx = lo_word(x);
  hy = hi_word(y); ly = lo_word(y);
  ix = hx & 0x7fffffff; iy = hy & 0x7fffffff;
  if ((iy | ly) == 0) return 1.0;                     /* x**0 = 1 */
  if (hx < 0 || ix >= 0x7ff00000 || iy >= 0x7ff00000) return NAN; /* outside domain */
  if ((ix | lx) == 0) return (hy < 0) ? INFINITY : 0.0; /* (+0)**y */
  if (ix == 0x3ff00000 && lx == 0) return 1.0;        /* 1**y */
  if (hy == 0x3ff00000 && ly == 0) return x;          /* x**1 */
  ax = x;
  if (iy > 0x41e00000) {                              /* |y| > 2**31 */
    if (iy > 0x43f00000) {                            /* |y| > 2**64: must o/uflow */
      if (ix <= 0x3fefffff) return (hy < 0) ? huge * huge : tiny * tiny;
      if (ix >= 0x3ff00000) return (hy > 0) ? huge * huge : tiny * tiny;
    }
    if (ix < 0x3fefffff) return (hy < 0) ? huge * huge : tiny * tiny;
    if (ix > 0x3ff00000) return (hy > 0) ? huge * huge : tiny * tiny;
    t = ax - 1.0;                                     /* |1-x| tiny */
    w = (t * t) * (0.5 - t * (0.3333333333333333333333 - t * 0.25));
    u = ivln2_h * t;
    v = t * ivln2_l - w * ivln2;
    t1 = u + v;
    t1 = with_lo(t1, 0);
    t2 = v - (t1 - u);
  } else {
    double ss, s2, s_h, s_l, t_h, t_l;
    n = 0;
    if (ix < 0x00100000) { ax *= two53; n -= 53; ix = hi_word(ax); }
    n += ((ix) >> 20) - 0x3ff;
    j = ix & 0x000fffff;
    ix = j | 0x3ff00000;
    if (j <= 0x3988E) k = 0;
    else if (j < 0xBB67A) k = 1;
    else { k = 0; n += 1; ix -= 0x00100000; }
    ax = with_hi(ax, ix);
    u = ax - bp[k];
    v = 1.0 / (ax + bp[k]);
    ss = u * v;
    s_h = ss;
    s_h = with_lo(s_h, 0);
    t_h = 0.0;
    t_h = with_hi(t_h, ((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18));
    t_l = ax - (t_h - bp[k]);
    s_l = v * ((u - s_h * t_h) - s_h * t_l);
    s2 = ss * ss;
    r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
    r += s_l * (s_h + ss);
    s2 = s_h * s_h;
    t_h = 3.0 + s2 + r;
    t_h = with_lo(t_h, 0);
    t_l = r - ((t_h - 3.0) - s2);
    u = s_h * t_h;
    v = s_l * t_h + t_l * ss;
    p_h = u + v;
    p_h = with_lo(p_h, 0);
    p_l = v - (p_h - u);
    z_h = cp_h * p_h;
    z_l = cp_l * p_h + p_l * cp + dp_l[k];
    t = (double)n;
    t1 = (((z_h + z_l) + dp_h[k]) + t);
    t1 = with_lo(t1, 0);
    t2 = z_l - (((t1 - t) - dp_h[k]) - z_h);
  }
  y1 = y;
  y1 = with_lo(y1, 0);
  p_l = (y - y1) * t1 + y * t2;
  p_h = y1 * t1;
  z = p_l + p_h;
  j = hi_word(z); i = (int32_t)lo_word(z);
  if (j >= 0x40900000) {                              /* z >= 1024 */
    if (((j - 0x40900000) | i) != 0) return huge * huge;
    if (p_l + ovt > z - p_h) return huge * huge;
  } else if ((j & 0x7fffffff) >= 0x4090cc00) {        /* z <= -1075 */
    if (((j - (int32_t)0xc090cc00) | i) != 0) return tiny * tiny;
    if (p_l <= z - p_h) return tiny * tiny;
  }
  i = j & 0x7fffffff;
  k = (i >> 20) - 0x3ff;
  n = 0;
  if (i > 0x3fe00000) {                               /* |z| > 0.5: n = [z+0.5] */
    n = j + (0x00100000 >> (k + 1));
    k = ((n & 0x7fffffff) >> 20) - 0x3ff;
    t = 0.0;
    t = with_hi(t, n & ~(0x000fffff >> k));
    n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
    if (j < 0) n = -n;
    p_h -= t;
  }
  t = p_l + p_h;
  t = with_lo(t, 0);
  u = t * lg2_h;
  v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
  z = u + v;
  w = v - (z - u);
  t = z * z;
  t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  r = (z * t1) / (t1 - 2.0) - (w + z * w);
  z = 1.0 - (r - z);
  j = hi_word(z);
  j += (int32_t)((uint32_t)n << 20);
  if ((j >> 20) <= 0) z = fd_scalbn(z, n);            /* subnormal output */
  else z = with_hi(z, j);
  return z;
}

void orc_log_array(int64_t n, const double *x, double *out) {
  for (int64_t i = 0; i < n; i++) out[i] = orc_log(x[i]);
}
void orc_pow_array(int64_t n, const double *x, const double *y, double *out) {
  for (int64_t i = 0; i < n; i++) out[i] = orc_pow(x[i], y[i]);
}

/* ------------------------------------------------------------------------ */
/* Per-draw random stream with java.util.Random's nextDouble / nextGaussian  */
/* semantics on top of Philox blocks                                          */
/* ------------------------------------------------------------------------ */
typedef struct {
  uint32_t key[2];
  uint32_t ctr0, ctr1, ctr2_base, ctr3;
  uint32_t pos;         /* doubles consumed so far */
  int32_t cached_block; /* -1 = none */
  double buf[2];
  int have_next_gaussian;
  double next_gaussian;
  int exhausted;
  uint32_t max_blocks;  /* ORC_MAX_BLOCKS for a draw (a runaway rejection loop); the held-out stream of a long test document may use
                           the whole 24-bit block field of the counter */
} draw_rng;

static void draw_init(draw_rng *r, uint64_t seed, uint32_t iter, uint32_t purpose, uint64_t elem) {
  r->key[0] = (uint32_t)seed; r->key[1] = (uint32_t)(seed >> 32);
  r->ctr0 = (uint32_t)elem; r->ctr1 = (uint32_t)(elem >> 32);
  r->ctr2_base = purpose << 24; r->ctr3 = iter;
  r->pos = 0; r->cached_block = -1; r->have_next_gaussian = 0; r->next_gaussian = 0.0;
  r->exhausted = 0;
  r->max_blocks = ORC_MAX_BLOCKS;
}
static double draw_next_double(draw_rng *r) {
  uint32_t blk = r->pos >> 1;
  if (blk >= r->max_blocks) { r->exhausted = 1; return 0.5; }
  if ((int32_t)blk != r->cached_block) {
    uint32_t c[4] = {r->ctr0, r->ctr1, r->ctr2_base | blk, r->ctr3}, o[4];
    orc_philox4x32_10(c, r->key, o);
    r->buf[0] = bits_to_double(o[0], o[1]);
    r->buf[1] = bits_to_double(o[2], o[3]);
    r->cached_block = (int32_t)blk;
  }
  double d = r->buf[r->pos & 1];
  r->pos++;
  return d;
}
/* java.util.Random.nextGaussian (JDK 8): polar method, second value cached */
static double draw_next_gaussian(draw_rng *r) {
  if (r->have_next_gaussian) { r->have_next_gaussian = 0; return r->next_gaussian; }
  double v1, v2, s;
  do {
    v1 = 2 * draw_next_double(r) - 1;
    v2 = 2 * draw_next_double(r) - 1;
    s = v1 * v1 + v2 * v2;
    if (r->exhausted) return 0.0;
  } while (s >= 1 || s == 0);
  double multiplier = sqrt(-2 * orc_log(s) / s);
  r->next_gaussian = v2 * multiplier;
  r->have_next_gaussian = 1;
  return v1 * multiplier;
}

/* ParallelRandoms.java:148-159 prgamma(alpha) */
static double prgamma(draw_rng *r, double alpha) {
  double x, v, u;
  double d = alpha - (1.0 / 3.0);
  double c = 1.0 / sqrt(9.0 * d);
  for (;;) {
    do {
      x = draw_next_gaussian(r);
      v = 1.0 + c * x;
      if (r->exhausted) return NAN;
    } while (v <= 0.0);
    v = v * v * v;
    u = draw_next_double(r);
    if (u < (1.0 - 0.0331 * (x * x) * (x * x))) return (d * v);
    if (orc_log(u) < (0.5 * x * x + d * (1.0 - v + orc_log(v)))) return (d * v);
    if (r->exhausted) return NAN;
  }
}
/* ParallelRandoms.java:60-70 rgamma(alpha, beta=1, lambda=0).  The trailing
 * "*beta + lambda" with beta=1, lambda=0 is the identity on non-negative
 * doubles and is kept for the record.
 * Stream layout (ours -- the reference's generator cannot be seeded): the uniform of the
 * alpha < 1 boost is the element's FIRST draw and is taken for every shape, used or not, so
 * that all elements sit at the same position of their Philox stream when the
 * Marsaglia-Tsang loop starts (on the GPU the lanes of a wave then need new blocks together). */
static double rgamma(draw_rng *r, double alpha) {
  double u = draw_next_double(r);
  if (alpha < 1) return ((prgamma(r, 1 + alpha) * orc_pow(u, 1.0 / alpha)) * 1.0) + 0.0;
  return (prgamma(r, alpha) * 1.0) + 0.0;
}

void orc_uniform_array(uint64_t seed, uint32_t iter, uint32_t purpose, uint64_t elem0, int64_t n,
                       double *out) {
  for (int64_t i = 0; i < n; i++) {
    draw_rng r; draw_init(&r, seed, iter, purpose, elem0 + (uint64_t)i);
    out[i] = draw_next_double(&r);
  }
}
void orc_gaussian_array(uint64_t seed, uint32_t iter, uint32_t purpose, uint64_t elem0, int64_t n,
                        double *out) {
  for (int64_t i = 0; i < n; i++) {
    draw_rng r; draw_init(&r, seed, iter, purpose, elem0 + (uint64_t)i);
    out[i] = draw_next_gaussian(&r);
  }
}
int orc_gamma_array(uint64_t seed, uint32_t iter, uint32_t purpose, uint64_t elem0, int64_t n,
                    const double *shape, double *out) {
  int err = ORC_OK;
  for (int64_t i = 0; i < n; i++) {
    if (!(shape[i] > 0)) { out[i] = NAN; err = ORC_ERR_BAD_ARG; continue; }
    draw_rng r; draw_init(&r, seed, iter, purpose, elem0 + (uint64_t)i);
    out[i] = rgamma(&r, shape[i]);
    if (r.exhausted) err = ORC_ERR_RNG_EXHAUSTED;
  }
  return err;
}

/* cc.mallet.types.Dirichlet(double[] p) + ParallelDirichlet.nextDistribution()
 * (ParallelDirichlet.java:46-70).  Element i of the draw uses RNG element
 * elem0 + i. */
int orc_dirichlet(uint64_t seed, uint32_t iter, uint32_t purpose, uint64_t elem0, int64_t n,
                  const double *p, double *out) {
  double magnitude = 0;
  for (int64_t i = 0; i < n; i++) magnitude += p[i];
  int err = ORC_OK;
  double sum = 0;
  for (int64_t i = 0; i < n; i++) {
    double partition = p[i] / magnitude;
    double a = partition * magnitude;
    if (!(a > 0)) { out[i] = NAN; err = ORC_ERR_BAD_ARG; continue; }
    draw_rng r; draw_init(&r, seed, iter, purpose, elem0 + (uint64_t)i);
    out[i] = rgamma(&r, a);
    if (r.exhausted) err = ORC_ERR_RNG_EXHAUSTED;
    sum += out[i];
  }
  if (sum != 0) {
    for (int64_t i = 0; i < n; i++) {
      out[i] /= sum;
      if (out[i] <= 0) out[i] = 4.9e-324; /* Double.MIN_VALUE */
    }
  }
  return err;
}

/* ------------------------------------------------------------------------ */
/* Sampler state                                                             */
/* ------------------------------------------------------------------------ */
struct orc_state {
  int32_t K, V;
  int64_t D, N;
  double *alpha; double beta;
  uint64_t seed;
  int64_t doc_base, tok_base;
  int64_t *doc_ptr; int32_t *tokens; int32_t *z;
  double *phi;        /* [K][V]  UPLDA:69  */
  int32_t *n_kw;      /* topicTypeCountMapping [K][V]  UPLDA:108 */
  int32_t *n_wk;      /* typeTopicCounts [V][K]        MSLDA:73  */
  int32_t *n_k;       /* tokensPerTopic [K]            MSLDA:74  */
  int32_t *delta;     /* batchLocalTopicTypeUpdates [K][V] UPLDA:102 */
  double *theta;      /* thetaMatrix rows [D][K]  GGS:72 */
  double *phi_mean; int32_t n_sampled_phi;
  int save_phi_mean, phi_burn_in, phi_thin;
  int32_t iteration;  /* currentIteration */
  int threads;
  int scheme;         /* 0 = ggs (LDAGroupedGibbsSampler), 1 = pcgs (LDAPartiallyCollapsedGibbsSampler) */
  jrandom collapsed_rng; int collapsed_rng_ready;
  double *phiT;       /* [V][K]: orc_sweep_tuned only (the "tuned CPU" baseline of BASELINE.md section 3) */
  int32_t *perm; int64_t *wptr;   /* token indices sorted by word, word run offsets [V+1]: orc_sweep_tuned only */
  char err[256];
};

static int fail(orc_state *s, int code, const char *msg) {
#pragma omp critical(orc_err)
  { if (!s->err[0]) snprintf(s->err, sizeof s->err, "%s", msg); }
  return code;
}
const char *orc_last_error(const orc_state *s) { return s->err; }

orc_state *orc_create(int32_t K, int32_t V, const double *alpha, double beta, uint64_t seed) {
  if (K <= 0 || V <= 0) return NULL;
  orc_state *s = calloc(1, sizeof *s);
  s->K = K; s->V = V; s->beta = beta; s->seed = seed;
  s->alpha = malloc(sizeof(double) * K);
  memcpy(s->alpha, alpha, sizeof(double) * K);
  size_t kv = (size_t)K * V;
  s->phi = calloc(kv, sizeof(double));
  s->n_kw = calloc(kv, sizeof(int32_t));
  s->n_wk = calloc(kv, sizeof(int32_t));
  s->delta = calloc(kv, sizeof(int32_t));
  s->n_k = calloc(K, sizeof(int32_t));
  s->phi_thin = 1; s->threads = 1;
  return s;
}
void orc_destroy(orc_state *s) {
  if (!s) return;
  free(s->alpha); free(s->phi); free(s->n_kw); free(s->n_wk); free(s->delta); free(s->n_k);
  free(s->doc_ptr); free(s->tokens); free(s->z); free(s->theta); free(s->phi_mean);
  free(s->phiT); free(s->perm); free(s->wptr);
  free(s);
}
int orc_set_corpus(orc_state *s, int64_t D, const int64_t *doc_ptr, const int32_t *tokens,
                   int64_t doc_base, int64_t tok_base) {
  s->err[0] = 0;
  if (D < 0 || doc_ptr[0] != 0) return fail(s, ORC_ERR_BAD_ARG, "bad corpus");
  int64_t N = doc_ptr[D];
  for (int64_t i = 0; i < N; i++)
    if (tokens[i] < 0 || tokens[i] >= s->V) return fail(s, ORC_ERR_BAD_ARG, "token id out of range");
  free(s->doc_ptr); free(s->tokens); free(s->z); free(s->theta);
  free(s->perm); free(s->wptr); s->perm = NULL; s->wptr = NULL;
  s->D = D; s->N = N; s->doc_base = doc_base; s->tok_base = tok_base;
  s->doc_ptr = malloc(sizeof(int64_t) * (D + 1));
  memcpy(s->doc_ptr, doc_ptr, sizeof(int64_t) * (D + 1));
  s->tokens = malloc(sizeof(int32_t) * (N ? N : 1));
  memcpy(s->tokens, tokens, sizeof(int32_t) * N);
  s->z = calloc(N ? N : 1, sizeof(int32_t));
  s->theta = calloc((size_t)(D ? D : 1) * s->K, sizeof(double));
  return ORC_OK;
}
void orc_set_phi_mean_gating(orc_state *s, int save, int burn_in, int thin) {
  s->save_phi_mean = save; s->phi_burn_in = burn_in; s->phi_thin = thin;
  if (save && !s->phi_mean) s->phi_mean = calloc((size_t)s->K * s->V, sizeof(double));
}
void orc_set_scheme(orc_state *s, int scheme) { s->scheme = scheme == 1 ? 1 : 0; }
void orc_set_threads(orc_state *s, int threads) { s->threads = threads > 0 ? threads : 1; }
void orc_set_iteration(orc_state *s, int32_t it) { s->iteration = it; }
int32_t orc_get_iteration(const orc_state *s) { return s->iteration; }
int64_t orc_num_tokens(const orc_state *s) { return s->N; }

/* UPLDA:471-482 updateTypeTopicCount */
static int update_type_topic_count(orc_state *s, int32_t type, int32_t topic, int32_t count) {
  s->n_kw[(size_t)topic * s->V + type] += count;
  s->n_wk[(size_t)type * s->K + topic] += count;
  s->n_k[topic] += count;
  if (s->n_kw[(size_t)topic * s->V + type] < 0) return fail(s, ORC_ERR_NEGATIVE_COUNT, "Negative count for topic");
  return ORC_OK;
}
static void zero_counts(orc_state *s) {
  size_t kv = (size_t)s->K * s->V;
  memset(s->n_kw, 0, kv * 4); memset(s->n_wk, 0, kv * 4); memset(s->delta, 0, kv * 4);
  memset(s->n_k, 0, (size_t)s->K * 4);
}

/* UPLDA:1287-1294 initialSamplePhi -> MarsagliaSparseDirichlet.nextDistribution(int[])
 * (MarsagliaSparseDirichlet.java:31-55) built by Dirichlet(int size, double beta):
 * magnitude = V*beta, partition[i] = 1.0/V. */
int orc_init_phi(orc_state *s) { return orc_init_phi_range(s, 0, s->K); }
/* initialSamplePhi(indices, phi) for the topic batch [k0, k1) (UPLDA:1287-1294 takes the batch's indices) */
int orc_init_phi_range(orc_state *s, int32_t k0, int32_t k1) {
  s->err[0] = 0;
  int err = ORC_OK;
  if (k0 < 0 || k1 > s->K || k0 > k1) return fail(s, ORC_ERR_BAD_ARG, "bad topic range");
  const double magnitude = (double)s->V * s->beta;
  const double partition = 1.0 / (double)s->V;
#pragma omp parallel for num_threads(s->threads) schedule(static)
  for (int32_t k = k0; k < k1; k++) {
    double *row = s->phi + (size_t)k * s->V;
    const int32_t *cnt = s->n_kw + (size_t)k * s->V;
    double sum = 0;
    for (int32_t v = 0; v < s->V; v++) {
      double a = (cnt[v] == 0) ? (partition * magnitude) : ((partition * magnitude) + cnt[v]);
      draw_rng r; draw_init(&r, s->seed, (uint32_t)s->iteration, ORC_PURPOSE_INIT_PHI, (uint64_t)k * s->V + v);
      row[v] = (a > 0) ? rgamma(&r, a) : NAN;
      if (r.exhausted || !(a > 0)) {
#pragma omp atomic write
        err = ORC_ERR_RNG_EXHAUSTED;
      }
      sum += row[v];
    }
    if (sum != 0)
      for (int32_t v = 0; v < s->V; v++) { row[v] /= sum; if (row[v] <= 0) row[v] = 4.9e-324; }
  }
  if (err) return fail(s, err, "gamma draw failed in init phi");
  return ORC_OK;
}

/* The gamma draws of a topic batch WITHOUT the normalisation (GGS:182-191 / UPLDA:1287-1294 up to ParallelDirichlet.java:57):
 * gam [(k1 - k0)][V] and their index-order sums -- what a rank of the multi-GPU exchange puts on the wire (the division of
 * ParallelDirichlet.java:60-66 is done by the receiver: the same IEEE division on the same operands).  s->phi is not
 * touched.  For tests/test_distributed_gloo.py, which restates that protocol over gloo. */
int orc_phi_gammas_range(orc_state *s, int32_t initial, int32_t k0, int32_t k1, double *gam, double *sums) {
  s->err[0] = 0;
  int err = ORC_OK;
  const int32_t V = s->V;
  if (k0 < 0 || k1 > s->K || k0 > k1) return fail(s, ORC_ERR_BAD_ARG, "bad topic range");
#pragma omp parallel for num_threads(s->threads) schedule(static)
  for (int32_t k = k0; k < k1; k++) {
    const int32_t *cnt = s->n_kw + (size_t)k * V;
    double *row = gam + (size_t)(k - k0) * V;
    double magnitude = 0;
    if (!initial)
      for (int32_t v = 0; v < V; v++) magnitude += s->beta + cnt[v];        /* Dirichlet(double[]): magnitude in index order */
    const double pm = (1.0 / (double)V) * ((double)V * s->beta);             /* Dirichlet(int, double): partition * magnitude */
    double sum = 0;
    for (int32_t v = 0; v < V; v++) {
      const double a = initial ? ((cnt[v] == 0) ? pm : pm + cnt[v]) : (((s->beta + cnt[v]) / magnitude) * magnitude);
      draw_rng r; draw_init(&r, s->seed, (uint32_t)s->iteration, initial ? ORC_PURPOSE_INIT_PHI : ORC_PURPOSE_PHI, (uint64_t)k * V + v);
      row[v] = (a > 0) ? rgamma(&r, a) : NAN;
      if (r.exhausted || !(a > 0)) {
#pragma omp atomic write
        err = ORC_ERR_RNG_EXHAUSTED;
      }
      sum += row[v];
    }
    sums[k - k0] = sum;
  }
  if (err) return fail(s, err, "gamma draw failed");
  return ORC_OK;
}

/* UPLDA:398-406,458-460: z0 = Randoms(seed).nextInt(K) in (doc, position) order */
int orc_init_z_java_lcg(orc_state *s, int32_t seed) {
  s->err[0] = 0;
  if (s->tok_base != 0) return fail(s, ORC_ERR_BAD_ARG, "java-LCG init needs the whole corpus (sequential stream)");
  zero_counts(s);
  jrandom r; jr_init(&r, (int64_t)seed);
  for (int64_t d = 0; d < s->D; d++)
    for (int64_t i = s->doc_ptr[d]; i < s->doc_ptr[d + 1]; i++) {
      int32_t topic = jr_next_int_bound(&r, s->K);
      s->z[i] = topic;
      int e = update_type_topic_count(s, s->tokens[i], topic, 1);
      if (e) return e;
    }
  /* SerialCollapsedLDA owns ONE Randoms(seed) (SerialCollapsedLDA.java:60-65): its addInstances draws the initial
   * topics from it (:789) and the sampling loop goes on drawing from the same object (MSLDA:206) -- so the stream of
   * orc_collapsed_sweep continues where this initialisation stopped */
  s->collapsed_rng = r; s->collapsed_rng_ready = 1;
  return ORC_OK;
}
/* UPLDA:1797-1843 setZIndicators */
int orc_set_z(orc_state *s, const int32_t *z, int redraw_phi) {
  s->err[0] = 0;
  zero_counts(s);
  for (int64_t i = 0; i < s->N; i++) {
    if (z[i] < 0 || z[i] >= s->K) return fail(s, ORC_ERR_BAD_ARG, "z out of range");
    s->z[i] = z[i];
    int e = update_type_topic_count(s, s->tokens[i], z[i], 1);
    if (e) return e;
  }
  return redraw_phi ? orc_init_phi(s) : ORC_OK;
}

/* GGS:47-132 sampleTopicAssignmentsParallel for local doc d */
static int ggs_doc_step(orc_state *s, int64_t d, int32_t *localTopicCounts, double *thetaParameter,
                        double *theta, double *topicTermScores) {
  const int32_t K = s->K, V = s->V;
  const int64_t b = s->doc_ptr[d], e = s->doc_ptr[d + 1];
  const int64_t docLength = e - b;
  if (docLength == 0) return ORC_OK;                          /* GGS:52-53 */
  const int32_t *tokenSequence = s->tokens + b;
  int32_t *oneDocTopics = s->z + b;
  memset(localTopicCounts, 0, sizeof(int32_t) * K);
  for (int64_t position = 0; position < docLength; position++)  /* GGS:60-63 */
    localTopicCounts[oneDocTopics[position]]++;
  for (int32_t topic = 0; topic < K; topic++)                   /* GGS:66-69 */
    thetaParameter[topic] = localTopicCounts[topic] + s->alpha[topic];
  int err = orc_dirichlet(s->seed, (uint32_t)s->iteration, ORC_PURPOSE_THETA, /* GGS:70-71 */
                          (uint64_t)(s->doc_base + d) * (uint64_t)K, K, thetaParameter, theta);
  if (err) return fail(s, err, "theta draw failed");
  memcpy(s->theta + (size_t)d * K, theta, sizeof(double) * K);   /* GGS:72 */

  for (int64_t position = 0; position < docLength; position++) { /* GGS:79-130 */
    int32_t type = tokenSequence[position];
    int32_t oldTopic = oneDocTopics[position];
    localTopicCounts[oldTopic]--;
    if (localTopicCounts[oldTopic] < 0) return fail(s, ORC_ERR_NEGATIVE_COUNT, "Invalid count!");
#pragma omp atomic
    s->delta[(size_t)oldTopic * V + type] -= 1;                  /* decrement(), UPLDA:1553-1556 */
    double sum = 0.0;
    for (int32_t topic = 0; topic < K; topic++) {                /* GGS:96-101 */
      double score = theta[topic] * s->phi[(size_t)topic * V + type];
      topicTermScores[topic] = score;
      sum += score;
    }
    draw_rng r; draw_init(&r, s->seed, (uint32_t)s->iteration, ORC_PURPOSE_Z, (uint64_t)(s->tok_base + b + position));
    double U = draw_next_double(&r);                             /* GGS:107 */
    double sample = U * sum;
    int32_t newTopic = -1;
    while (sample > 0.0) {                                       /* GGS:110-113 */
      newTopic++;
      if (newTopic >= K) break;   /* Java: ArrayIndexOutOfBoundsException */
      sample -= topicTermScores[newTopic];
    }
    if (newTopic < 0 || newTopic >= K) {                         /* GGS:116-118 */
      /* keep memory safe, then report like the Java throw */
      newTopic = newTopic < 0 ? 0 : K - 1;
      oneDocTopics[position] = newTopic;
      localTopicCounts[newTopic]++;
#pragma omp atomic
      s->delta[(size_t)newTopic * V + type] += 1;
      return fail(s, ORC_ERR_INVALID_TOPIC, "Topic sampled is invalid!");
    }
    oneDocTopics[position] = newTopic;
    localTopicCounts[newTopic]++;
#pragma omp atomic
    s->delta[(size_t)newTopic * V + type] += 1;                  /* increment(), UPLDA:1547-1551 */
  }
  return ORC_OK;
}

/* UPLDA:1466-1544 sampleTopicAssignmentsParallel: the z loop of scheme=pcgs (and of
 * "uncollapsed").  theta is integrated out: score = (n_dk + alpha_k) * phi[k][w] with the
 * document's counts updated token by token, so the tokens of one document are strictly
 * sequential; documents are independent given Phi.  Same per-token Philox uniform as the GGS
 * path (purpose Z, element = global token index). */
static int pcgs_doc_step(orc_state *s, int64_t d, int32_t *localTopicCounts, double *topicTermScores) {
  const int32_t K = s->K, V = s->V;
  const int64_t b = s->doc_ptr[d], e = s->doc_ptr[d + 1];
  const int64_t docLength = e - b;
  if (docLength == 0) return ORC_OK;                              /* UPLDA:1474 */
  const int32_t *tokenSequence = s->tokens + b;
  int32_t *oneDocTopics = s->z + b;
  memset(localTopicCounts, 0, sizeof(int32_t) * K);
  for (int64_t position = 0; position < docLength; position++)   /* UPLDA:1482-1485 */
    localTopicCounts[oneDocTopics[position]]++;
  for (int64_t position = 0; position < docLength; position++) { /* UPLDA:1491-1543 */
    int32_t type = tokenSequence[position];
    int32_t oldTopic = oneDocTopics[position];
    localTopicCounts[oldTopic]--;
    if (localTopicCounts[oldTopic] < 0) return fail(s, ORC_ERR_NEGATIVE_COUNT, "Counts cannot be negative!");
#pragma omp atomic
    s->delta[(size_t)oldTopic * V + type] -= 1;                  /* decrement(), UPLDA:1553-1556 */
    double sum = 0.0;
    for (int32_t topic = 0; topic < K; topic++) {                /* UPLDA:1509-1513 */
      double score = (localTopicCounts[topic] + s->alpha[topic]) * s->phi[(size_t)topic * V + type];
      topicTermScores[topic] = score;
      sum += score;
    }
    draw_rng r; draw_init(&r, s->seed, (uint32_t)s->iteration, ORC_PURPOSE_Z, (uint64_t)(s->tok_base + b + position));
    double U = draw_next_double(&r);                             /* UPLDA:1519 */
    double sample = U * sum;
    int32_t newTopic = -1;
    while (sample > 0.0) {                                       /* UPLDA:1523-1526 */
      newTopic++;
      if (newTopic >= K) break;   /* Java: ArrayIndexOutOfBoundsException */
      sample -= topicTermScores[newTopic];
    }
    if (newTopic < 0 || newTopic >= K) {                         /* UPLDA:1529-1531 */
      newTopic = newTopic < 0 ? 0 : K - 1;
      oneDocTopics[position] = newTopic;
      localTopicCounts[newTopic]++;
#pragma omp atomic
      s->delta[(size_t)newTopic * V + type] += 1;
      return fail(s, ORC_ERR_INVALID_TOPIC, "UncollapsedParallelLDA: New valid topic not sampled.");
    }
    oneDocTopics[position] = newTopic;
    localTopicCounts[newTopic]++;
#pragma omp atomic
    s->delta[(size_t)newTopic * V + type] += 1;                  /* increment(), UPLDA:1547-1551 */
  }
  return ORC_OK;
}

/* UPLDA:1434-1437 loopOverBatches: the fork-join halving down to
 * document_sampler_split_limit (100) docs is restated as dynamic chunks of 100
 * documents; every result is schedule-independent because the RNG is
 * counter-based. */
int orc_z_step(orc_state *s) {
  s->err[0] = 0;
  int err = ORC_OK;
  const int32_t K = s->K;
#pragma omp parallel num_threads(s->threads)
  {
    /* per-document allocations of GGS:57,66,76 hoisted per thread */
    int32_t *ltc = malloc(sizeof(int32_t) * K);
    double *tp = malloc(sizeof(double) * K), *th = malloc(sizeof(double) * K), *sc = malloc(sizeof(double) * K);
#pragma omp for schedule(dynamic, 100)
    for (int64_t d = 0; d < s->D; d++) {
      int e = s->scheme == 1 ? pcgs_doc_step(s, d, ltc, sc) : ggs_doc_step(s, d, ltc, tp, th, sc);
      if (e) {
#pragma omp atomic write
        err = e;
      }
    }
    free(ltc); free(tp); free(th); free(sc);
  }
  return err;
}

/* UPLDA:1107-1138 -> updateTopics :1203-1221 -> ParallelTopicUpdater.call :1158-1182 */
int orc_update_counts(orc_state *s) {
  s->err[0] = 0;
  int err = ORC_OK;
  const int32_t K = s->K, V = s->V;
  int thr = s->threads < 2 ? s->threads : 2;   /* topicUpdaters pool = 2 threads, UPLDA:1085 */
#pragma omp parallel for num_threads(thr) schedule(dynamic, 1)
  for (int32_t topic = 0; topic < K; topic++) {
    for (int32_t type = 0; type < V; type++) {
      int32_t dlt = s->delta[(size_t)topic * V + type];
      if (dlt != 0) {
        s->delta[(size_t)topic * V + type] = 0;
        s->n_kw[(size_t)topic * V + type] += dlt;
        s->n_wk[(size_t)type * K + topic] += dlt;
        s->n_k[topic] += dlt;
        if (s->n_kw[(size_t)topic * V + type] < 0) {
#pragma omp atomic write
          err = ORC_ERR_NEGATIVE_COUNT;
        }
      }
    }
  }
  if (err) return fail(s, err, "Negative count for topic");
  return ORC_OK;
}

/* UPLDA:1350-1352 */
static int sample_phi_this_iteration(const orc_state *s) {
  return s->phi_burn_in > 0 && s->iteration > s->phi_burn_in && (s->iteration % s->phi_thin) == 0;
}

/* GGS:182-198 loopOverTopics(indices, phi) for the topic batch [k0, k1) -- samplePhi hands one such batch to every
 * PhiSampler thread (GGS:139-171); a multi-GPU run hands one to every rank (EvenSplitTopicBatchBuilder.java:28-39). */
int orc_sample_phi_range(orc_state *s, int32_t k0, int32_t k1) {
  s->err[0] = 0;
  int err = ORC_OK;
  const int32_t K = s->K, V = s->V;
  if (k0 < 0 || k1 > K || k0 > k1) return fail(s, ORC_ERR_BAD_ARG, "bad topic range");
  const int accumulate = s->save_phi_mean && sample_phi_this_iteration(s);
#pragma omp parallel num_threads(s->threads)
  {
    double *dirichletParams = malloc(sizeof(double) * V);
#pragma omp for schedule(dynamic, 1)
    for (int32_t topic = k0; topic < k1; topic++) {
      const int32_t *relevantTypeTopicCounts = s->n_kw + (size_t)topic * V;
      for (int32_t type = 0; type < V; type++)
        dirichletParams[type] = s->beta + relevantTypeTopicCounts[type];
      int e = orc_dirichlet(s->seed, (uint32_t)s->iteration, ORC_PURPOSE_PHI, (uint64_t)topic * V, V,
                            dirichletParams, s->phi + (size_t)topic * V);
      if (e) {
#pragma omp atomic write
        err = e;
      }
      if (accumulate)
        for (int32_t v = 0; v < V; v++) s->phi_mean[(size_t)topic * V + v] += s->phi[(size_t)topic * V + v];
    }
    free(dirichletParams);
  }
  if (err) return fail(s, err, "phi draw failed");
  return ORC_OK;
}

/* GGS:139-171 samplePhi: every topic batch, then the noSampledPhi bookkeeping */
int orc_sample_phi(orc_state *s) {
  int e = orc_sample_phi_range(s, 0, s->K);
  if (e) return e;
  if (s->save_phi_mean && sample_phi_this_iteration(s)) s->n_sampled_phi++;     /* GGS:168-170 */
  return ORC_OK;
}

/* UPLDA:645-687: one iteration = loopOverBatches, updateCounts, samplePhi */
int orc_sweep(orc_state *s, int32_t n_sweeps) {
  for (int32_t it = 0; it < n_sweeps; it++) {
    s->iteration++;                       /* currentIteration = iteration, UPLDA:646 */
    int e = orc_z_step(s);      if (e) return e;
    e = orc_update_counts(s);   if (e) return e;
    e = orc_sample_phi(s);      if (e) return e;
  }
  return ORC_OK;
}

/* ---- the "tuned CPU" baseline (cpu_tuned_mt) ----------------------------------------------
 * The same sweep -- the same draws and the same arithmetic in the same order, hence the same bits as orc_sweep -- with
 * the memory behaviour a CPU implementation would choose rather than the Java program's: Phi transposed to
 * phiT[V][K] so that a token reads one contiguous row (GGS:98 gathers a column of phi[K][V]), no AtomicInteger deltas
 * (UPLDA:1547-1557): the counts are rebuilt from z per word over a word-sorted token index, all threads for every
 * phase (the reference merges on 2 threads, UPLDA:1085).  Timed by bench.py beside the Java-layout port; scheme ggs. */
static int tuned_doc_step(orc_state *s, int64_t d, int32_t *localTopicCounts, double *thetaParameter, double *theta) {
  const int32_t K = s->K;
  const int64_t b = s->doc_ptr[d], e = s->doc_ptr[d + 1];
  if (e == b) return ORC_OK;
  memset(localTopicCounts, 0, sizeof(int32_t) * K);
  for (int64_t i = b; i < e; i++) localTopicCounts[s->z[i]]++;
  for (int32_t topic = 0; topic < K; topic++) thetaParameter[topic] = localTopicCounts[topic] + s->alpha[topic];
  int err = orc_dirichlet(s->seed, (uint32_t)s->iteration, ORC_PURPOSE_THETA, (uint64_t)(s->doc_base + d) * (uint64_t)K, K, thetaParameter, theta);
  if (err) return fail(s, err, "theta draw failed");
  memcpy(s->theta + (size_t)d * K, theta, sizeof(double) * K);
  for (int64_t i = b; i < e; i++) {
    const double *row = s->phiT + (size_t)s->tokens[i] * K;
    double sum = 0.0;
    for (int32_t topic = 0; topic < K; topic++) sum += theta[topic] * row[topic];
    draw_rng r; draw_init(&r, s->seed, (uint32_t)s->iteration, ORC_PURPOSE_Z, (uint64_t)(s->tok_base + i));
    double sample = draw_next_double(&r) * sum;
    int32_t newTopic = -1;
    while (sample > 0.0) {
      newTopic++;
      if (newTopic >= K) break;
      sample -= theta[newTopic] * row[newTopic];       /* the same single IEEE product as in the sum */
    }
    if (newTopic < 0 || newTopic >= K) { s->z[i] = newTopic < 0 ? 0 : K - 1; return fail(s, ORC_ERR_INVALID_TOPIC, "Topic sampled is invalid!"); }
    s->z[i] = newTopic;
  }
  return ORC_OK;
}
/* The Phi draw of the tuned variant: the K * V gamma variates spread over all threads in (topic, 1024-type tile) units
 * instead of one topic per thread (GGS:139-171 hands whole topics to `topic_batches` threads: at K = 100 at most 100 of
 * the host's threads ever work).  Each topic's two V-long sums stay ONE sequential chain in index order (the magnitude of
 * Dirichlet(double[]), ParallelDirichlet.java:53-57), so the result is orc_sample_phi's bit for bit. */
static int tuned_sample_phi(orc_state *s) {
  const int32_t K = s->K, V = s->V;
  const int accumulate = s->save_phi_mean && sample_phi_this_iteration(s);
  int err = ORC_OK;
  double *magnitude = malloc(sizeof(double) * (size_t)K), *sum = malloc(sizeof(double) * (size_t)K);
  const int32_t tile = 1024, ntiles = (V + tile - 1) / tile;
#pragma omp parallel num_threads(s->threads)
  {
#pragma omp for schedule(static)
    for (int32_t k = 0; k < K; k++) {
      double m = 0;
      const int32_t *c = s->n_kw + (size_t)k * V;
      for (int32_t v = 0; v < V; v++) m += s->beta + c[v];
      magnitude[k] = m;
    }
#pragma omp for schedule(dynamic, 4) collapse(2)
    for (int32_t k = 0; k < K; k++)
      for (int32_t t = 0; t < ntiles; t++) {
        const int32_t v1 = (t + 1) * tile < V ? (t + 1) * tile : V;
        for (int32_t v = t * tile; v < v1; v++) {
          const double p = s->beta + s->n_kw[(size_t)k * V + v];
          const double a = (p / magnitude[k]) * magnitude[k];
          double *out = s->phi + (size_t)k * V + v;
          if (!(a > 0)) {
            *out = NAN;
#pragma omp atomic write
            err = ORC_ERR_BAD_ARG;
            continue;
          }
          draw_rng r; draw_init(&r, s->seed, (uint32_t)s->iteration, ORC_PURPOSE_PHI, (uint64_t)k * V + (uint64_t)v);
          *out = rgamma(&r, a);
          if (r.exhausted) {
#pragma omp atomic write
            err = ORC_ERR_RNG_EXHAUSTED;
          }
        }
      }
#pragma omp for schedule(static)
    for (int32_t k = 0; k < K; k++) {
      double t = 0;
      const double *g = s->phi + (size_t)k * V;
      for (int32_t v = 0; v < V; v++) t += g[v];
      sum[k] = t;
    }
#pragma omp for schedule(static) collapse(2)
    for (int32_t k = 0; k < K; k++)
      for (int32_t t = 0; t < ntiles; t++) {
        const int32_t v1 = (t + 1) * tile < V ? (t + 1) * tile : V;
        double *g = s->phi + (size_t)k * V;
        for (int32_t v = t * tile; v < v1; v++) {
          if (sum[k] != 0) {
            g[v] /= sum[k];
            if (g[v] <= 0) g[v] = 4.9e-324;          /* Double.MIN_VALUE */
          }
          if (accumulate) s->phi_mean[(size_t)k * V + v] += g[v];
        }
      }
  }
  free(magnitude); free(sum);
  if (err) return fail(s, err, "phi draw failed");
  if (accumulate) s->n_sampled_phi++;                                          /* GGS:168-170 */
  return ORC_OK;
}

int orc_sweep_tuned(orc_state *s, int32_t n_sweeps) {
  s->err[0] = 0;
  if (s->scheme != 0) return fail(s, ORC_ERR_BAD_ARG, "orc_sweep_tuned: scheme ggs only");
  const int32_t K = s->K, V = s->V;
  if (!s->phiT) s->phiT = malloc(sizeof(double) * (size_t)K * V);
  if (!s->perm) {                                      /* counting sort of the token indices by word, once per corpus */
    s->wptr = calloc((size_t)V + 1, sizeof(int64_t));
    s->perm = malloc(sizeof(int32_t) * (size_t)(s->N ? s->N : 1));
    for (int64_t i = 0; i < s->N; i++) s->wptr[s->tokens[i] + 1]++;
    for (int32_t w = 0; w < V; w++) s->wptr[w + 1] += s->wptr[w];
    int64_t *cur = malloc(sizeof(int64_t) * (size_t)V);
    memcpy(cur, s->wptr, sizeof(int64_t) * (size_t)V);
    for (int64_t i = 0; i < s->N; i++) s->perm[cur[s->tokens[i]]++] = (int32_t)i;
    free(cur);
  }
  for (int32_t it = 0; it < n_sweeps; it++) {
    s->iteration++;
    int err = ORC_OK;
#pragma omp parallel num_threads(s->threads)
    {
#pragma omp for schedule(static)
      for (int32_t v = 0; v < V; v++)                  /* phiT := transpose of the current Phi */
        for (int32_t k = 0; k < K; k++) s->phiT[(size_t)v * K + k] = s->phi[(size_t)k * V + v];
      int32_t *ltc = malloc(sizeof(int32_t) * K);
      double *tp = malloc(sizeof(double) * K), *th = malloc(sizeof(double) * K);
#pragma omp for schedule(dynamic, 100)
      for (int64_t d = 0; d < s->D; d++) {
        int e = tuned_doc_step(s, d, ltc, tp, th);
        if (e) {
#pragma omp atomic write
          err = e;
        }
      }
      free(ltc); free(tp); free(th);
#pragma omp for schedule(dynamic, 64)
      for (int32_t w = 0; w < V; w++) {                /* n_wk row of word w = histogram of its tokens' z */
        int32_t *row = s->n_wk + (size_t)w * K;
        memset(row, 0, sizeof(int32_t) * K);
        for (int64_t j = s->wptr[w]; j < s->wptr[w + 1]; j++) row[s->z[s->perm[j]]]++;
        for (int32_t k = 0; k < K; k++) s->n_kw[(size_t)k * V + w] = row[k];
      }
#pragma omp for schedule(static)
      for (int32_t k = 0; k < K; k++) {
        int32_t t = 0;
        for (int32_t w = 0; w < V; w++) t += s->n_kw[(size_t)k * V + w];
        s->n_k[k] = t;
      }
    }
    if (err) return err;
    int e = tuned_sample_phi(s); if (e) return e;
  }
  return ORC_OK;
}

/* MSLDA:158-226 sampleTopicsForOneDoc, looped as SerialCollapsedLDA.sample does
 * (SerialCollapsedLDA.java:159-172): strictly serial, one java.util.Random
 * stream (MALLET Randoms.nextUniform() is restated as nextDouble(): ASSUMPTION
 * flagged in SURVEY 8c).  The stream is the one orc_init_z_java_lcg left behind
 * (the sampler's single Randoms object); only a state initialised another way
 * (orc_set_z) starts a new Random(seed_if_first). */
int orc_collapsed_sweep(orc_state *s, int32_t seed_if_first, int32_t n_sweeps) {
  s->err[0] = 0;
  const int32_t K = s->K, V = s->V;
  if (!s->collapsed_rng_ready) { jr_init(&s->collapsed_rng, (int64_t)seed_if_first); s->collapsed_rng_ready = 1; }
  int32_t *localTopicCounts = malloc(sizeof(int32_t) * K);
  double *topicTermScores = malloc(sizeof(double) * K);
  const double betaSum = s->beta * V;
  int err = ORC_OK;
  for (int32_t it = 0; it < n_sweeps && !err; it++) {
    s->iteration++;
    for (int64_t d = 0; d < s->D && !err; d++) {
      const int64_t b = s->doc_ptr[d], docLength = s->doc_ptr[d + 1] - b;
      memset(localTopicCounts, 0, sizeof(int32_t) * K);
      for (int64_t p = 0; p < docLength; p++) localTopicCounts[s->z[b + p]]++;
      for (int64_t p = 0; p < docLength; p++) {
        int32_t type = s->tokens[b + p], oldTopic = s->z[b + p];
        int32_t *currentTypeTopicCounts = s->n_wk + (size_t)type * K;
        localTopicCounts[oldTopic]--;
        s->n_k[oldTopic]--;
        currentTypeTopicCounts[oldTopic]--;
        s->n_kw[(size_t)oldTopic * V + type]--;
        double sum = 0.0;
        for (int32_t topic = 0; topic < K; topic++) {
          double score = (s->alpha[topic] + localTopicCounts[topic]) *
                         ((s->beta + currentTypeTopicCounts[topic]) / (betaSum + s->n_k[topic]));
          sum += score;
          topicTermScores[topic] = score;
        }
        double sample = jr_next_double(&s->collapsed_rng) * sum;
        int32_t newTopic = -1;
        while (sample > 0.0) { newTopic++; if (newTopic >= K) break; sample -= topicTermScores[newTopic]; }
        if (newTopic < 0 || newTopic >= K) { err = fail(s, ORC_ERR_INVALID_TOPIC, "SimpleLDA: New topic not sampled."); newTopic = newTopic < 0 ? 0 : K - 1; }
        s->z[b + p] = newTopic;
        localTopicCounts[newTopic]++;
        s->n_k[newTopic]++;
        currentTypeTopicCounts[newTopic]++;
        s->n_kw[(size_t)newTopic * V + type]++;
      }
    }
  }
  free(localTopicCounts); free(topicTermScores);
  return err;
}

/* The count-form conditional of MSLDA:196-203 in the PARALLEL schedule the device runs (scheme=collapsed with
 * documents side by side; the AD-LDA decomposition of ADLDA.java:176-332 with one worker per document): every
 * document is sampled against the type-topic counts and topic totals AS THEY STOOD AT THE START OF THE SWEEP, minus
 * the token being resampled (each AD-LDA worker removes the current token from its copy, MSLDA:186-189), while the
 * document's own topic counts run along its positions exactly as in the serial loop; the merged counts of the sweep
 * are the (word, z) histogram of the new assignments (sum of the workers' deltas, ADLDA.java:302).  Uniform: Philox,
 * purpose Z, element = global token index, like the other schemes.  Not the serial chain of orc_collapsed_sweep --
 * a different (approximate, like ADLDA) sampler that the device kernel must reproduce bit for bit. */
int orc_collapsed_parallel_sweep(orc_state *s, int32_t n_sweeps) {
  s->err[0] = 0;
  const int32_t K = s->K, V = s->V;
  const double betaSum = s->beta * V;
  int32_t *stale_wk = malloc(sizeof(int32_t) * (size_t)K * V), *stale_k = malloc(sizeof(int32_t) * K);
  int err = ORC_OK;
  for (int32_t it = 0; it < n_sweeps && !err; it++) {
    s->iteration++;
    memcpy(stale_wk, s->n_wk, sizeof(int32_t) * (size_t)K * V);
    memcpy(stale_k, s->n_k, sizeof(int32_t) * K);
#pragma omp parallel num_threads(s->threads)
    {
      int32_t *localTopicCounts = malloc(sizeof(int32_t) * K);
      double *topicTermScores = malloc(sizeof(double) * K);
#pragma omp for schedule(dynamic, 100)
      for (int64_t d = 0; d < s->D; d++) {
        const int64_t b = s->doc_ptr[d], docLength = s->doc_ptr[d + 1] - b;
        memset(localTopicCounts, 0, sizeof(int32_t) * K);
        for (int64_t p = 0; p < docLength; p++) localTopicCounts[s->z[b + p]]++;
        for (int64_t p = 0; p < docLength; p++) {
          const int32_t type = s->tokens[b + p], oldTopic = s->z[b + p];
          const int32_t *currentTypeTopicCounts = stale_wk + (size_t)type * K;
          localTopicCounts[oldTopic]--;
          double sum = 0.0;
          for (int32_t topic = 0; topic < K; topic++) {
            const int32_t own = topic == oldTopic;                 /* "Remove this token from all counts", MSLDA:185-190 */
            double score = (s->alpha[topic] + localTopicCounts[topic]) *
                           ((s->beta + (currentTypeTopicCounts[topic] - own)) / (betaSum + (stale_k[topic] - own)));
            sum += score;
            topicTermScores[topic] = score;
          }
          draw_rng r; draw_init(&r, s->seed, (uint32_t)s->iteration, ORC_PURPOSE_Z, (uint64_t)(s->tok_base + b + p));
          double sample = draw_next_double(&r) * sum;
          int32_t newTopic = -1;
          while (sample > 0.0) { newTopic++; if (newTopic >= K) break; sample -= topicTermScores[newTopic]; }
          if (newTopic < 0 || newTopic >= K) {
            newTopic = newTopic < 0 ? 0 : K - 1;
#pragma omp atomic write
            err = ORC_ERR_INVALID_TOPIC;
          }
          s->z[b + p] = newTopic;
          localTopicCounts[newTopic]++;
        }
      }
      free(localTopicCounts); free(topicTermScores);
    }
    /* the merge: counts := histogram of the new assignments */
    zero_counts(s);
    for (int64_t i = 0; i < s->N; i++) {
      const int32_t w = s->tokens[i], k = s->z[i];
      s->n_wk[(size_t)w * K + k]++; s->n_kw[(size_t)k * V + w]++; s->n_k[k]++;
    }
  }
  free(stale_wk); free(stale_k);
  if (err) return fail(s, err, "SimpleLDA: New topic not sampled.");
  return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* getters (layouts of the Java getters: MSLDA:464-477, UPLDA:226-234, ...)  */
/* ------------------------------------------------------------------------ */
/* MALLET 2.0.8 cc.mallet.types.Dirichlet.logGammaStirling (not under /root/reference; restated from its
 * published source): shift z up to >= 2, Stirling series, undo the shift.  Math.log is restated with the
 * fdlibm log used everywhere else here. */
static double log_gamma_stirling(double z) {
  const double HALF_LOG_TWO_PI = 0.91893853320467274178;
  int shift = 0;
  while (z < 2) { z++; shift++; }
  double result = HALF_LOG_TWO_PI + (z - 0.5) * orc_log(z) - z + 1 / (12 * z) - 1 / (360 * z * z * z) + 1 / (1260 * z * z * z * z * z);
  while (shift > 0) { shift--; z--; result -= orc_log(z); }
  return result;
}

/* UPLDA:1644-1758 modelLogLikelihood, in the Java loop order (one running double), split into the part that
 * runs over this state's documents (:1674-1694) and the part that runs over the type-topic counts (:1701-1747). */
void orc_model_log_likelihood(const orc_state *s, double *doc_side, double *topic_side) {
  const int32_t K = s->K, V = s->V;
  double alphaSum = 0;
  for (int32_t k = 0; k < K; k++) alphaSum += s->alpha[k];
  int32_t *topicCounts = calloc((size_t)K, sizeof(int32_t));
  double *topicLogGammas = malloc(sizeof(double) * (size_t)K);
  for (int32_t k = 0; k < K; k++) topicLogGammas[k] = log_gamma_stirling(s->alpha[k]);
  double ll = 0.0;
  for (int64_t d = 0; d < s->D; d++) {
    const int64_t b = s->doc_ptr[d], e = s->doc_ptr[d + 1];
    for (int64_t i = b; i < e; i++) topicCounts[s->z[i]]++;
    for (int32_t k = 0; k < K; k++)
      if (topicCounts[k] > 0) ll += (log_gamma_stirling(s->alpha[k] + topicCounts[k]) - topicLogGammas[k]);
    ll -= log_gamma_stirling(alphaSum + (double)(e - b));
    memset(topicCounts, 0, sizeof(int32_t) * (size_t)K);
  }
  ll += s->D * log_gamma_stirling(alphaSum);
  *doc_side = ll;
  ll = 0.0;
  int64_t nonZeroTypeTopics = 0;
  for (int32_t w = 0; w < V; w++)
    for (int32_t k = 0; k < K; k++) {
      int32_t c = s->n_wk[(size_t)w * K + k];
      if (c == 0) continue;
      nonZeroTypeTopics++;
      ll += log_gamma_stirling(s->beta + c);
    }
  for (int32_t k = 0; k < K; k++) ll -= log_gamma_stirling((s->beta * V) + s->n_k[k]);
  ll += log_gamma_stirling(s->beta * V) * K;
  ll -= log_gamma_stirling(s->beta) * nonZeroTypeTopics;
  *topic_side = ll;
  free(topicCounts); free(topicLogGammas);
}

/* UPLDA:710-714: for every scheme but "ggs" the sampling loop draws a fresh theta for its diagnostics from the
 * document-topic counts, theta_d ~ Dirichlet(n_d. + alpha) (LDAUtils.getDocumentTopicCounts + LDAUtils.drawDirichlets,
 * util/LDAUtils.java:1662-1673).  The reference draws it from MALLET's Dirichlet / Randoms (third-party, clock-seeded
 * gammas); here it is the draw of GGS:57-72 under the same Philox stream ORC_PURPOSE_THETA at the current iteration --
 * the same distribution, reproducible.  Fills s->theta. */
int orc_draw_diagnostic_theta(orc_state *s) {
  const int32_t K = s->K;
  int err = ORC_OK;
  int32_t *cnt = malloc(sizeof(int32_t) * (size_t)K);
  double *par = malloc(sizeof(double) * (size_t)K);
  for (int64_t d = 0; d < s->D && !err; d++) {
    const int64_t b = s->doc_ptr[d], e = s->doc_ptr[d + 1];
    if (e == b) continue;                                     /* an empty document keeps its (zero) row, as in the z step */
    memset(cnt, 0, sizeof(int32_t) * (size_t)K);
    for (int64_t i = b; i < e; i++) cnt[s->z[i]]++;
    for (int32_t k = 0; k < K; k++) par[k] = cnt[k] + s->alpha[k];
    err = orc_dirichlet(s->seed, (uint32_t)s->iteration, ORC_PURPOSE_THETA, (uint64_t)(s->doc_base + d) * (uint64_t)K, K, par,
                        s->theta + (size_t)d * K);
  }
  free(cnt); free(par);
  return err ? fail(s, err, "diagnostic theta draw failed") : ORC_OK;
}

/* UPLDA:1573-1634 computeLogPosterior in the Java loop order (Math.log restated with the fdlibm log).  The
 * dense per-document K x V matrix m_djt is kept as the sorted list of the document's (topic, type) pairs: the Java
 * loop visits k then v ascending and adds count * logPhi for every non-zero cell, and so does this. */
static int cmp_i64(const void *a, const void *b) { int64_t x = *(const int64_t *)a, y = *(const int64_t *)b; return (x > y) - (x < y); }
void orc_log_posterior(const orc_state *s, double *doc_side, double *topic_side) {
  const double EPS = 1e-12;
  const int32_t K = s->K, V = s->V;
  double lp = 0.0;
  double *n_dj = malloc(sizeof(double) * (size_t)K);
  int64_t maxlen = 1;
  for (int64_t d = 0; d < s->D; d++) if (s->doc_ptr[d + 1] - s->doc_ptr[d] > maxlen) maxlen = s->doc_ptr[d + 1] - s->doc_ptr[d];
  int64_t *cell = malloc(sizeof(int64_t) * (size_t)maxlen);
  for (int64_t d = 0; d < s->D; d++) {
    const int64_t b = s->doc_ptr[d], e = s->doc_ptr[d + 1];
    for (int32_t k = 0; k < K; k++) n_dj[k] = 0.0;
    for (int64_t i = b; i < e; i++) { n_dj[s->z[i]] += 1.0; cell[i - b] = (int64_t)s->z[i] * V + s->tokens[i]; }
    qsort(cell, (size_t)(e - b), sizeof(int64_t), cmp_i64);
    for (int64_t i = 0; i < e - b;) {                       /* :1604-1612 */
      int64_t j = i;
      while (j < e - b && cell[j] == cell[i]) j++;
      const int32_t k = (int32_t)(cell[i] / V), v = (int32_t)(cell[i] % V);
      lp += (double)(j - i) * orc_log(s->phi[(size_t)k * V + v] + EPS);
      i = j;
    }
    for (int32_t k = 0; k < K; k++)                          /* :1615-1618 */
      lp += (n_dj[k] + s->alpha[k] - 1.0) * orc_log(s->theta[(size_t)d * K + k] + EPS);
  }
  *doc_side = lp;
  lp = 0.0;
  const double betaMinus1 = s->beta - 1.0;                   /* :1622-1628 */
  for (int32_t k = 0; k < K; k++)
    for (int32_t v = 0; v < V; v++) lp += betaMinus1 * orc_log(s->phi[(size_t)k * V + v] + EPS);
  *topic_side = lp;
  free(n_dj); free(cell);
}

/* ------------------------------------------------------------------------ */
/* topics/MarginalProbEstimatorPlain.java: the left-to-right held-out estimator the sampling loop calls on the
 * test set every diagnostic iteration (UPLDA:604-611,677-682,840-844; 100 particles, UPLDA:615).  Restated in the
 * Java operation order with usingResampling = false (MPE:125, "The resampling implementation is broken").
 *   ctor            MPE:51-79     betaSum = beta * V; smoothingOnlyMass; cachedCoefficients
 *   leftToRight     MPE:123-519   one particle's pass over one document
 *   evaluateLeftToRight MPE:85-121 per position: sum over particles, log, minus log(numParticles)
 * The reference draws from a clock-seeded `new Randoms()` (MPE:64,87): not reproducible, so the stream is ours --
 * purpose ORC_PURPOSE_HELDOUT, element = (doc_base + doc) * numParticles + particle, the uniforms of one particle
 * taken in sequence (one per in-vocabulary token); Randoms.nextUniform is taken to be nextDouble (as for the
 * collapsed path).  Math.log is the fdlibm log used everywhere else here.  Parity unpinned against a JVM run. */
static int left_to_right(orc_state *s, const int32_t *tok, int64_t docLength, uint64_t elem, double betaSum, double smoothingOnlyMass,
                         double alphaSum, double *cachedCoefficients, int32_t *localTopicCounts, int32_t *localTopicIndex,
                         double *topicTermScores, double *wordProbabilities) {
  const int32_t numTopics = s->K;
  const double beta = s->beta;
  const int32_t *tokensPerTopic = s->n_k;
  int rc = ORC_OK;
  int tokensSoFar = 0, nonZeroTopics = 0, denseIndex;
  double topicBetaMass = 0.0, topicTermMass;
  draw_rng r; draw_init(&r, s->seed, (uint32_t)s->iteration, ORC_PURPOSE_HELDOUT, elem);
  r.max_blocks = 1u << 24;                                                /* one uniform per token: the block field's 24 bits = 2^25 tokens */
  memset(localTopicCounts, 0, sizeof(int32_t) * (size_t)numTopics);
  for (int64_t limit = 0; limit < docLength; limit++) wordProbabilities[limit] = 0.0;
  for (int64_t limit = 0; limit < docLength; limit++) {
    const int32_t type = tok[limit];
    if (type >= s->V) continue;                                           /* MPE:341-345 out-of-vocabulary */
    const int32_t *currentTypeTopicCounts = s->n_wk + (size_t)type * numTopics;
    topicTermMass = 0.0;
    for (int32_t index = 0; index < numTopics; index++) {                 /* MPE:352-365 */
      double score = cachedCoefficients[index] * currentTypeTopicCounts[index];
      topicTermMass += score;
      topicTermScores[index] = score;
    }
    double sample = draw_next_double(&r) * (smoothingOnlyMass + topicBetaMass + topicTermMass);   /* MPE:393 */
    wordProbabilities[limit] += (smoothingOnlyMass + topicBetaMass + topicTermMass) / (alphaSum + tokensSoFar);   /* MPE:399-401 */
    tokensSoFar++;
    int32_t newTopic = -1;
    if (sample < topicTermMass) {                                         /* MPE:409-419 */
      int32_t i = -1;
      while (sample > 0) {
        i++;
        if (i >= numTopics) break;                                        /* Java: ArrayIndexOutOfBoundsException */
        sample -= topicTermScores[i];
      }
      newTopic = i;
    } else {
      sample -= topicTermMass;
      if (sample < topicBetaMass) {                                       /* MPE:423-440 */
        sample /= beta;
        for (denseIndex = 0; denseIndex < nonZeroTopics; denseIndex++) {
          int32_t topic = localTopicIndex[denseIndex];
          sample -= localTopicCounts[topic] / (tokensPerTopic[topic] + betaSum);
          if (sample <= 0.0) { newTopic = topic; break; }
        }
      } else {                                                            /* MPE:442-460 */
        sample -= topicBetaMass;
        sample /= beta;
        newTopic = 0;
        sample -= s->alpha[newTopic] / (tokensPerTopic[newTopic] + betaSum);
        while (sample > 0.0) {
          newTopic++;
          if (newTopic >= numTopics) break;
          sample -= s->alpha[newTopic] / (tokensPerTopic[newTopic] + betaSum);
        }
      }
    }
    if (r.exhausted) { rc = fail(s, ORC_ERR_RNG_EXHAUSTED, "held-out stream: document longer than 2^25 tokens"); break; }
    if (newTopic < 0 || newTopic >= numTopics) {                          /* MPE:416,447,455,464-469: IllegalStateException */
      rc = fail(s, ORC_ERR_INVALID_TOPIC, "Sampled invalid topic");
      break;
    }
    topicBetaMass -= beta * localTopicCounts[newTopic] / (tokensPerTopic[newTopic] + betaSum);          /* MPE:474-475 */
    localTopicCounts[newTopic]++;
    if (localTopicCounts[newTopic] == 1) {                                /* MPE:481-499 sorted insert */
      denseIndex = nonZeroTopics;
      while (denseIndex > 0 && localTopicIndex[denseIndex - 1] > newTopic) {
        localTopicIndex[denseIndex] = localTopicIndex[denseIndex - 1];
        denseIndex--;
      }
      localTopicIndex[denseIndex] = newTopic;
      nonZeroTopics++;
    }
    cachedCoefficients[newTopic] = (s->alpha[newTopic] + localTopicCounts[newTopic]) / (tokensPerTopic[newTopic] + betaSum);   /* MPE:502-504 */
    topicBetaMass += beta * localTopicCounts[newTopic] / (tokensPerTopic[newTopic] + betaSum);          /* MPE:506-507 */
  }
  for (denseIndex = 0; denseIndex < nonZeroTopics; denseIndex++) {        /* MPE:514-519 */
    int32_t topic = localTopicIndex[denseIndex];
    cachedCoefficients[topic] = s->alpha[topic] / (tokensPerTopic[topic] + betaSum);
  }
  return rc;
}

int orc_heldout_log_likelihood(orc_state *s, int64_t D, const int64_t *doc_ptr, const int32_t *tokens, int64_t doc_base,
                               int32_t numParticles, double *doc_ll, double *total) {
  if (D < 0 || numParticles < 1 || !doc_ptr || !total) return fail(s, ORC_ERR_BAD_ARG, "bad held-out arguments");
  const int32_t numTopics = s->K;
  double alphaSum = 0;
  for (int32_t k = 0; k < numTopics; k++) alphaSum += s->alpha[k];
  const double betaSum = s->beta * s->V;                                  /* MPE:63 */
  double smoothingOnlyMass = 0;
  double *base = malloc(sizeof(double) * (size_t)numTopics);
  for (int32_t topic = 0; topic < numTopics; topic++) {                   /* MPE:75-78 */
    smoothingOnlyMass += s->alpha[topic] * s->beta / (s->n_k[topic] + betaSum);
    base[topic] = s->alpha[topic] / (s->n_k[topic] + betaSum);
  }
  const double logNumParticles = orc_log(numParticles);
  int rc_all = ORC_OK;
#pragma omp parallel num_threads(s->threads)
  {
    double *cachedCoefficients = malloc(sizeof(double) * (size_t)numTopics);
    memcpy(cachedCoefficients, base, sizeof(double) * (size_t)numTopics);
    int32_t *localTopicCounts = malloc(sizeof(int32_t) * (size_t)numTopics);
    int32_t *localTopicIndex = malloc(sizeof(int32_t) * (size_t)numTopics);
    double *topicTermScores = malloc(sizeof(double) * (size_t)numTopics);
    double *probs = NULL; int64_t cap = 0;
#pragma omp for schedule(dynamic, 4)
    for (int64_t d = 0; d < D; d++) {
      const int64_t docLength = doc_ptr[d + 1] - doc_ptr[d];
      if (docLength * numParticles > cap) { cap = docLength * numParticles; free(probs); probs = malloc(sizeof(double) * (size_t)(cap ? cap : 1)); }
      for (int32_t particle = 0; particle < numParticles; particle++) {   /* MPE:97-100 */
        int rc = left_to_right(s, tokens + doc_ptr[d], docLength, (uint64_t)(doc_base + d) * (uint64_t)numParticles + (uint64_t)particle, betaSum,
                               smoothingOnlyMass, alphaSum, cachedCoefficients, localTopicCounts, localTopicIndex, topicTermScores,
                               probs + (size_t)particle * docLength);
        if (rc) {
#pragma omp critical(orc_heldout_rc)
          rc_all = rc;
        }
      }
      double docLogLikelihood = 0;
      for (int64_t position = 0; position < docLength; position++) {     /* MPE:102-111 */
        double sum = 0;
        for (int32_t particle = 0; particle < numParticles; particle++) sum += probs[(size_t)particle * docLength + position];
        if (sum > 0.0) docLogLikelihood += orc_log(sum) - logNumParticles;
      }
      doc_ll[d] = docLogLikelihood;
    }
    free(cachedCoefficients); free(localTopicCounts); free(localTopicIndex); free(topicTermScores); free(probs);
  }
  double totalLogLikelihood = 0;                                          /* MPE:116: in document order */
  for (int64_t d = 0; d < D; d++) totalLogLikelihood += doc_ll[d];
  *total = totalLogLikelihood;
  free(base);
  return rc_all;
}

void orc_get_z(const orc_state *s, int32_t *z) { memcpy(z, s->z, sizeof(int32_t) * s->N); }
void orc_get_type_topic_counts(const orc_state *s, int32_t *o) { memcpy(o, s->n_wk, sizeof(int32_t) * (size_t)s->K * s->V); }
void orc_get_topic_type_counts(const orc_state *s, int32_t *o) { memcpy(o, s->n_kw, sizeof(int32_t) * (size_t)s->K * s->V); }
void orc_get_topic_totals(const orc_state *s, int32_t *o) { memcpy(o, s->n_k, sizeof(int32_t) * s->K); }
void orc_get_delta(const orc_state *s, int32_t *o) {
  for (int32_t k = 0; k < s->K; k++)
    for (int32_t v = 0; v < s->V; v++) o[(size_t)v * s->K + k] = s->delta[(size_t)k * s->V + v];
}
void orc_add_delta(orc_state *s, const int32_t *d) {
  for (int32_t k = 0; k < s->K; k++)
    for (int32_t v = 0; v < s->V; v++) s->delta[(size_t)k * s->V + v] += d[(size_t)v * s->K + k];
}
void orc_get_phi(const orc_state *s, double *o) { memcpy(o, s->phi, sizeof(double) * (size_t)s->K * s->V); }
/* UPLDA:1897-1902: this.phi = phi; if (savePhiMeans()) phiMean = new double[numTopics][numTypes] -- the running sum
 * restarts at zero while noSampledPhi keeps counting */
void orc_set_phi(orc_state *s, const double *p) {
  memcpy(s->phi, p, sizeof(double) * (size_t)s->K * s->V);
  if (s->save_phi_mean && s->phi_mean) memset(s->phi_mean, 0, sizeof(double) * (size_t)s->K * s->V);
}
int orc_get_phi_mean(const orc_state *s, double *o) {
  if (s->n_sampled_phi == 0 || !s->phi_mean) return 0;
  size_t kv = (size_t)s->K * s->V;
  for (size_t i = 0; i < kv; i++) o[i] = s->phi_mean[i] / s->n_sampled_phi; /* UPLDA:1959-1964 */
  return s->n_sampled_phi;
}
/* rows [k0, k1) of Phi as drawn elsewhere (another rank's topic batch); unlike setPhi this is not a caller's new
 * matrix, so the running phi mean is left alone */
void orc_set_phi_rows(orc_state *s, int32_t k0, int32_t k1, const double *rows) {
  memcpy(s->phi + (size_t)k0 * s->V, rows, sizeof(double) * (size_t)(k1 - k0) * s->V);
}
/* all three count structures from a typeTopicCounts matrix [V][K] (the merged counts of a sharded run); deltas zeroed */
void orc_set_counts(orc_state *s, const int32_t *n_wk) {
  zero_counts(s);
  for (int32_t v = 0; v < s->V; v++)
    for (int32_t k = 0; k < s->K; k++) {
      const int32_t c = n_wk[(size_t)v * s->K + k];
      s->n_wk[(size_t)v * s->K + k] = c; s->n_kw[(size_t)k * s->V + v] = c; s->n_k[k] += c;
    }
}
void orc_get_theta(const orc_state *s, double *o) { memcpy(o, s->theta, sizeof(double) * (size_t)s->D * s->K); }
void orc_get_doc_topic_counts(const orc_state *s, int32_t *o) {
  memset(o, 0, sizeof(int32_t) * (size_t)s->D * s->K);
  for (int64_t d = 0; d < s->D; d++)
    for (int64_t i = s->doc_ptr[d]; i < s->doc_ptr[d + 1]; i++) o[(size_t)d * s->K + s->z[i]]++;
}
