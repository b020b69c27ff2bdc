// ggs_device_math.hpp -- device-side numerics of the Grouped Gibbs Sampler.
//
// Everything here must round exactly as the JVM rounds the reference's double
// arithmetic, so this translation unit is compiled with -ffp-contract=off and
// never with fast-math; products and sums below are single IEEE operations.
//
//   philox4x32_10      Salmon et al. SC'11 (Random123 constants)
//   DrawStream         java.util.Random's nextDouble()/nextGaussian() semantics
//                      (JDK 8: 26+27 bit doubles, polar Gaussian with caching)
//                      fed from the Philox blocks of one (purpose, iteration, elem)
//   strict_log/pow     fdlibm 5.3 e_log.c / e_pow.c == java.lang.StrictMath
//   rgamma             cc/mallet/util/ParallelRandoms.java:60-70,148-159
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ggs_hip.h"

namespace ggs {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one v_mad_u64_u32 per product (measured: 100 ns of a SIMD per wave and block, against 133 with v_mul_hi_u32 + v_mul_lo_u32)
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  return U4{c0, c1, c2, c3};
}

// ((next(26) << 27) + next(27)) * 2^-53 with next(n) = top n bits of a 32-bit word
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  const uint64_t m = ((uint64_t)(a >> 6) << 27) + (uint64_t)(b >> 5);
  return (double)m * 0x1.0p-53;
}

__device__ __forceinline__ int32_t hi32(double x) { return (int32_t)(__double_as_longlong(x) >> 32); }
__device__ __forceinline__ uint32_t lo32(double x) { return (uint32_t)__double_as_longlong(x); }
__device__ __forceinline__ double set_hi(double x, int32_t hi) {
  return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)lo32(x)));
}
__device__ __forceinline__ double clr_lo(double x) {
  return __longlong_as_double((long long)((uint64_t)__double_as_longlong(x) & 0xffffffff00000000ull));
}
__device__ __forceinline__ double mk(int32_t hi, uint32_t lo) {
  return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | lo));
}

// ---- StrictMath.log ---------------------------------------------------------
__device__ __noinline__ double strict_log(double x) {
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                   two54 = 1.80143985094819840000e+16,
                   Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                   Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                   Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                   Lg7 = 1.479819860511658591e-01;
  int32_t hx = hi32(x);
  const uint32_t lx = lo32(x);
  int32_t k = 0;
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lx) == 0) return -__builtin_huge_val();
    if (hx < 0) return __builtin_nan("");
    k -= 54; x *= two54; hx = hi32(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int32_t i = (hx + 0x95f64) & 0x100000;
  x = set_hi(x, hx | (i ^ 0x3ff00000));
  k += (i >> 20);
  const double f = x - 1.0;
  const double dk = (double)k;
  if ((0x000fffff & (2 + hx)) < 3) {
    if (f == 0.0) { if (k == 0) return 0.0; return dk * ln2_hi + dk * ln2_lo; }
    const double R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  const double s = f / (2.0 + f);
  const double z = s * s;
  i = hx - 0x6147a;
  const double w = z * z;
  const int32_t j = 0x6b851 - hx;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  const double R = t2 + t1;
  if (i > 0) {
    const double hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// fdlibm scalbn for finite x (only reached for subnormal pow results)
__device__ __forceinline__ double strict_scalbn(double x, int n) {
  constexpr double two54 = 1.80143985094819840000e+16, twom54 = 5.55111512312578270212e-17,
                   huge = 1.0e+300, tiny = 1.0e-300;
  int32_t hx = hi32(x);
  int32_t k = (hx & 0x7ff00000) >> 20;
  if (k == 0) {
    if ((lo32(x) | (uint32_t)(hx & 0x7fffffff)) == 0) return x;
    x *= two54; hx = hi32(x);
    k = ((hx & 0x7ff00000) >> 20) - 54;
    if (n < -50000) return tiny * x;
  }
  if (k == 0x7ff) return x + x;
  k = k + n;
  if (k > 0x7fe) return huge * __builtin_copysign(huge, x);
  if (k > 0) return set_hi(x, (hx & (int32_t)0x800fffff) | (k << 20));
  if (k <= -54) {
    if (n > 50000) return huge * __builtin_copysign(huge, x);
    return tiny * __builtin_copysign(tiny, x);
  }
  k += 54;
  x = set_hi(x, (hx & (int32_t)0x800fffff) | (k << 20));
  return x * twom54;
}

// ---- StrictMath.pow for x >= +0 finite, y finite (the domain of
// ParallelRandoms.java:66, Math.pow(u, 1.0/alpha)) ----------------------------
__device__ __noinline__ double strict_pow(double x, double y) {
  constexpr double two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300,
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
  const int32_t hx = hi32(x), hy = hi32(y);
  const uint32_t lx = lo32(x), ly = lo32(y);
  int32_t ix = hx & 0x7fffffff;
  const int32_t iy = hy & 0x7fffffff;
  if ((iy | ly) == 0) return 1.0;
  if (hx < 0 || ix >= 0x7ff00000 || iy >= 0x7ff00000) return __builtin_nan("");
  if ((ix | lx) == 0) return (hy < 0) ? __builtin_huge_val() : 0.0;
  if (ix == 0x3ff00000 && lx == 0) return 1.0;
  if (hy == 0x3ff00000 && ly == 0) return x;
  double ax = x, t1, t2;
  if (iy > 0x41e00000) {
    if (iy > 0x43f00000) {
      if (ix <= 0x3fefffff) return (hy < 0) ? huge * huge : tiny * tiny;
      if (ix >= 0x3ff00000) return (hy > 0) ? huge * huge : tiny * tiny;
    }
    if (ix < 0x3fefffff) return (hy < 0) ? huge * huge : tiny * tiny;
    if (ix > 0x3ff00000) return (hy > 0) ? huge * huge : tiny * tiny;
    const double t = ax - 1.0;
    const double w = (t * t) * (0.5 - t * (0.3333333333333333333333 - t * 0.25));
    const double u = ivln2_h * t;
    const double v = t * ivln2_l - w * ivln2;
    t1 = clr_lo(u + v);
    t2 = v - (t1 - u);
  } else {
    int32_t n = 0, k;
    if (ix < 0x00100000) { ax *= two53; n -= 53; ix = hi32(ax); }
    n += (ix >> 20) - 0x3ff;
    const int32_t j = ix & 0x000fffff;
    ix = j | 0x3ff00000;
    if (j <= 0x3988E) k = 0;
    else if (j < 0xBB67A) k = 1;
    else { k = 0; n += 1; ix -= 0x00100000; }
    ax = set_hi(ax, ix);
    const double bpk = k ? 1.5 : 1.0;
    const double dp_hk = k ? 5.84962487220764160156e-01 : 0.0;
    const double dp_lk = k ? 1.35003920212974897128e-08 : 0.0;
    double u = ax - bpk;
    double v = 1.0 / (ax + bpk);
    const double ss = u * v;
    const double s_h = clr_lo(ss);
    double t_h = mk(((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18), 0u);
    double t_l = ax - (t_h - bpk);
    const double s_l = v * ((u - s_h * t_h) - s_h * t_l);
    double s2 = ss * ss;
    double r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
    r += s_l * (s_h + ss);
    s2 = s_h * s_h;
    t_h = clr_lo(3.0 + s2 + r);
    t_l = r - ((t_h - 3.0) - s2);
    u = s_h * t_h;
    v = s_l * t_h + t_l * ss;
    const double p_h = clr_lo(u + v);
    const double p_l = v - (p_h - u);
    const double z_h = cp_h * p_h;
    const double z_l = cp_l * p_h + p_l * cp + dp_lk;
    const double t = (double)n;
    t1 = clr_lo(((z_h + z_l) + dp_hk) + t);
    t2 = z_l - (((t1 - t) - dp_hk) - z_h);
  }
  const double y1 = clr_lo(y);
  double p_l = (y - y1) * t1 + y * t2;
  double p_h = y1 * t1;
  double z = p_l + p_h;
  int32_t j = hi32(z);
  int32_t i = (int32_t)lo32(z);
  if (j >= 0x40900000) {
    if (((j - 0x40900000) | i) != 0) return huge * huge;
    if (p_l + ovt > z - p_h) return huge * huge;
  } else if ((j & 0x7fffffff) >= 0x4090cc00) {
    if (((j - (int32_t)0xc090cc00) | i) != 0) return tiny * tiny;
    if (p_l <= z - p_h) return tiny * tiny;
  }
  i = j & 0x7fffffff;
  int32_t k = (i >> 20) - 0x3ff;
  int32_t n = 0;
  if (i > 0x3fe00000) {
    n = j + (0x00100000 >> (k + 1));
    k = ((n & 0x7fffffff) >> 20) - 0x3ff;
    const double t = mk(n & ~(0x000fffff >> k), 0u);
    n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
    if (j < 0) n = -n;
    p_h -= t;
  }
  double t = clr_lo(p_l + p_h);
  const double u = t * lg2_h;
  const double v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
  z = u + v;
  const double w = v - (z - u);
  t = z * z;
  t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  const double r = (z * t1) / (t1 - 2.0) - (w + z * w);
  z = 1.0 - (r - z);
  j = hi32(z);
  j += (int32_t)((uint32_t)n << 20);
  if ((j >> 20) <= 0) return strict_scalbn(z, n);
  return set_hi(z, j);
}

// ---- per-draw stream ----------------------------------------------------------
struct DrawStream {
  uint32_t k0, k1, c0, c1, c2base, c3;
  uint32_t pos;
  int32_t cached;     // block whose doubles are in d0/d1, -1 = none
  double d0, d1;
  double next_gauss;
  bool have_gauss;
  bool exhausted;

  __device__ __forceinline__ DrawStream(uint64_t seed, uint32_t iter, uint32_t purpose, uint64_t elem)
      : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), c0((uint32_t)elem), c1((uint32_t)(elem >> 32)),
        c2base(purpose << 24), c3(iter), pos(0), cached(-1), d0(0), d1(0), next_gauss(0),
        have_gauss(false), exhausted(false) {}

  __device__ __forceinline__ double next_double() {
    const uint32_t blk = pos >> 1;
    if (blk >= (uint32_t)GGS_MAX_BLOCKS) { exhausted = true; return 0.5; }
    if ((int32_t)blk != cached) {
      const U4 o = philox4x32_10(c0, c1, c2base | blk, c3, k0, k1);
      d0 = u53(o.x, o.y);
      d1 = u53(o.z, o.w);
      cached = (int32_t)blk;
    }
    const double d = (pos & 1u) ? d1 : d0;
    ++pos;
    return d;
  }
  // java.util.Random.nextGaussian
  __device__ __forceinline__ double next_gaussian() {
    if (have_gauss) { have_gauss = false; return next_gauss; }
    double v1, v2, s;
    do {
      v1 = 2 * next_double() - 1;
      v2 = 2 * next_double() - 1;
      s = v1 * v1 + v2 * v2;
      if (exhausted) return 0.0;
    } while (s >= 1 || s == 0);
    const double multiplier = sqrt(-2 * strict_log(s) / s);
    next_gauss = v2 * multiplier;
    have_gauss = true;
    return v1 * multiplier;
  }
};

// ParallelRandoms.java:148-159
__device__ __forceinline__ double prgamma(DrawStream &r, double alpha) {
  const double d = alpha - (1.0 / 3.0);
  const double c = 1.0 / sqrt(9.0 * d);
  for (;;) {
    double x, v;
    do {
      x = r.next_gaussian();
      v = 1.0 + c * x;
      if (r.exhausted) return __builtin_nan("");
    } while (v <= 0.0);
    v = v * v * v;
    const double u = r.next_double();
    if (u < (1.0 - 0.0331 * (x * x) * (x * x))) return d * v;
    if (strict_log(u) < (0.5 * x * x + d * (1.0 - v + strict_log(v)))) return d * v;
    if (r.exhausted) return __builtin_nan("");
  }
}
// ParallelRandoms.java:60-70 with beta = 1, lambda = 0
// One prgamma instance for both branches: a wave usually mixes shapes below and above 1 (sparse
// counts), and two inlined copies would run one after the other.  The boost uniform is the
// element's first draw whatever the shape (stream layout, see oracle/ggs_oracle.c rgamma): every
// lane is then at the same stream position at every call site and new Philox blocks are needed
// by all lanes together.
__device__ __forceinline__ double rgamma(DrawStream &r, double alpha) {
  const bool boost = alpha < 1;
  const double u = r.next_double();                  // drawn BEFORE the Marsaglia-Tsang loop, as at ParallelRandoms.java:62
  const double g = prgamma(r, boost ? 1 + alpha : alpha);
  return boost ? g * strict_pow(u, 1.0 / alpha) : g;
}

// The same draw when nothing is rejected twice -- straight-line code for a whole wave.  rgamma's loops cost a wave the
// MAXIMUM over its 64 lanes: the polar method rejects 21 % of its pairs (3.3 rounds until all lanes hold one), and one
// lane in seven fails the squeeze of Marsaglia-Tsang, so both log tests and a second round of the outer loop are
// executed by practically every wave.  Here every lane evaluates polar attempts 0 and 1 (stream positions 1-2 and
// 3-4: Philox blocks 0-2) and the squeeze test once, with exactly the operations rgamma would perform on that path;
// a lane for which that does not settle the draw (both pairs rejected 4.6 %, squeeze failed ~9 %, v <= 0) returns false
// and its element is drawn by rgamma from the start of its stream -- by the caller, gathered into full waves.
__device__ __forceinline__ bool rgamma_first_try(uint64_t seed, uint32_t iter, uint32_t purpose, uint64_t elem, double alpha, double &g) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), c0 = (uint32_t)elem, c1 = (uint32_t)(elem >> 32), c2 = purpose << 24;
  const U4 b0 = philox4x32_10(c0, c1, c2, iter, k0, k1), b1 = philox4x32_10(c0, c1, c2 | 1u, iter, k0, k1),
           b2 = philox4x32_10(c0, c1, c2 | 2u, iter, k0, k1);
  const double u0 = u53(b0.x, b0.y);                                  // position 0: the boost uniform
  const double a1 = 2 * u53(b0.z, b0.w) - 1, a2 = 2 * u53(b1.x, b1.y) - 1, sa = a1 * a1 + a2 * a2;    // positions 1, 2
  const double w = u53(b1.z, b1.w);                                   // position 3: the acceptance uniform, or the next pair's first
  const double e1 = 2 * w - 1, e2 = 2 * u53(b2.x, b2.y) - 1, sb = e1 * e1 + e2 * e2;                  // positions 3, 4
  const bool ok_a = !(sa >= 1 || sa == 0), ok_b = !(sb >= 1 || sb == 0);
  if (!(ok_a || ok_b)) return false;
  const double v1 = ok_a ? a1 : e1, s = ok_a ? sa : sb;
  const double u = ok_a ? w : u53(b2.z, b2.w);                        // position 3 or 5
  const double multiplier = sqrt(-2 * strict_log(s) / s);
  const double x = v1 * multiplier;
  const bool boost = alpha < 1;
  const double shape = boost ? 1 + alpha : alpha;
  const double d = shape - (1.0 / 3.0);
  const double c = 1.0 / sqrt(9.0 * d);
  double v = 1.0 + c * x;
  if (!(v > 0.0)) return false;                                       // rgamma: the cached second Gaussian comes next
  v = v * v * v;
  if (!(u < (1.0 - 0.0331 * (x * x) * (x * x)))) return false;        // rgamma: the log test, perhaps another round
  const double r = d * v;
  g = boost ? r * strict_pow(u0, 1.0 / alpha) : r;
  return true;
}

}  // namespace ggs
