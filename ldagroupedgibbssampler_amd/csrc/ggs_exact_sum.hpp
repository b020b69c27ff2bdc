// ggs_exact_sum.hpp -- the two V-long normalisers of the Phi draw, bit-exact and parallel.
//
// Java adds V doubles per topic in index order: the Dirichlet(double[]) magnitude
// sum_v (beta + n_kv) (MALLET Dirichlet constructor, used at GGS:190) and the sum of the
// gamma draws (ParallelDirichlet.java:53-57).  Each is ONE dependent rounding chain per topic --
// 0.28 ms at V = 50k when walked element by element, whatever the hardware.  But all addends
// are >= 0, so the running sum s only grows, and while s stays inside one binade
// [2^e, 2^(e+1)) its ulp u = 2^(e-52) is constant and
//
//     fl(s + x) = s + R_u(x),   R_u(x) = x rounded to the nearest multiple of u,
//
// independent of s except when x mod u == u/2 exactly (a tie: then the result is the EVEN
// multiple).  Inside a binade the chain is therefore integer arithmetic on quantised addends,
// which any number of lanes can add up in any order.  Per topic:
//
//   sum_seg      plain (order-free) sums of 64-row segments                 -> a guess only
//   sum_prefix   exclusive prefix of those                                  -> a guess only
//   sum_segfn    per segment, assuming the binade e of the guessed start:  D1 = u * sum of R_u(x)
//                up to the first tie (floor part of the tie included), H = u/2 if there was a
//                tie, D2 = u * sum after it (later ties resolved by parity: the sum is even
//                right after a tie).  Then  s_out = ((s_in + D1) + H) + D2  with every add exact
//                except the "+ H", which the hardware rounds to even exactly as the chain would.
//   sum_walk     one wave per topic walks the segments.  A step is ACCEPTED only if s_in and
//                s_out both lie in binade e -- s is monotone, so then every intermediate sum did
//                too and the assumption held; the guess never decides a result.  Anything else
//                (binade crossings, the start at s = 0, wrong guesses, NaN) is re-done the Java
//                way, element by element, from the raw data.  64 consecutive clean tie-free
//                segments of one binade collapse into a single exact add (wave reduction).
//
// About ten segments of 782 take the element-by-element path at V = 50k (s doubles ~25 times,
// mostly inside the first segment).
#pragma once
#include "ggs_device_math.hpp"

namespace ggs {

constexpr int kSumSegRows = 64;        // rows per segment == lanes per wave (sum_walk loads a segment with one load per lane)
constexpr int kSumBlock = 128;         // topics per workgroup in the row-streaming kernels
constexpr double kSumDirty = -5000.0;  // fn[3] marker (binades are -900..1000): no segment function, walk the raw elements
constexpr double kSumPastEnd = -6000.0;

struct SumParams {
  const void *src;        // int32 [V][pitch] counts (MAGNITUDE) or fp64 [V][pitch] gamma draws
  double *pref;           // [nseg + 1][K] guessed running sum at each segment start
  double *fn;             // [nseg][K][4]: D1, H, D2, e (as a double; kSumDirty = none)
  double *out;            // [K]
  int32_t *n_k;           // MAGNITUDE only, optional: += column sums of the counts (zeroed by the caller)
  double beta;
  int32_t pitch, K, V, nseg;
};

// These kernels are short, latency-bound links of the chain between the z step and the next one, and they run while
// the theta draw of the side stream keeps every SIMD busy: their waves ask for instruction issue ahead of its waves.
__device__ __forceinline__ void chain_priority() { __builtin_amdgcn_s_setprio(3); }

__device__ __forceinline__ int binade_of(double x) { return ((hi32(x) >> 20) & 0x7ff) - 1023; }   // 1024 for NaN/inf, -1023 for 0

template <typename T, bool MAGNITUDE>
__device__ __forceinline__ double sum_elem(const T *src, size_t idx, double beta) {
  return MAGNITUDE ? (beta + (double)src[idx]) : (double)src[idx];   // GGS:188: beta + count, one rounding
}

template <typename T, bool MAGNITUDE>
__global__ __launch_bounds__(kSumBlock) void sum_seg_kernel(SumParams p) {
  chain_priority();
  const int k = blockIdx.y * kSumBlock + threadIdx.x;
  if (k >= p.K) return;
  const int i = blockIdx.x;
  const T *src = static_cast<const T *>(p.src);
  const int v0 = i * kSumSegRows;
  double a = 0.0;
  int32_t cnt = 0;
  // 16 loads in flight; rows past V are clamped and contribute + 0.0
#pragma unroll 1
  for (int r0 = 0; r0 < kSumSegRows; r0 += 16) {
    T xs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) xs[j] = src[(size_t)min(v0 + r0 + j, p.V - 1) * p.pitch + k];
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (v0 + r0 + j < p.V) {
        a += MAGNITUDE ? (p.beta + (double)xs[j]) : (double)xs[j];
        if (MAGNITUDE) cnt += (int32_t)xs[j];
      }
  }
  p.pref[(size_t)i * p.K + k] = a;
  if (MAGNITUDE && p.n_k && cnt) atomicAdd(&p.n_k[k], cnt);      // tokensPerTopic on the way (integers: any order)
}

// one wave per topic: exclusive prefix over the segment sums, in place; pref[nseg][k] = total
__global__ __launch_bounds__(64) void sum_prefix_kernel(SumParams p) {
  chain_priority();
  const int k = blockIdx.x, lane = threadIdx.x;
  const int per = (p.nseg + 63) / 64;
  const int i0 = min(lane * per, p.nseg), i1 = min(i0 + per, p.nseg);
  double mine = 0.0;
  for (int i = i0; i < i1; ++i) mine += p.pref[(size_t)i * p.K + k];
  double incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  double run = incl - mine;
  for (int i = i0; i < i1; ++i) {
    const double t = p.pref[(size_t)i * p.K + k];
    p.pref[(size_t)i * p.K + k] = run;
    run += t;
  }
  if (lane == 63) p.pref[(size_t)p.nseg * p.K + k] = incl;
}

template <typename T, bool MAGNITUDE>
__global__ __launch_bounds__(kSumBlock) void sum_segfn_kernel(SumParams p) {
  chain_priority();
  const int k = blockIdx.y * kSumBlock + threadIdx.x;
  if (k >= p.K) return;
  const int i = blockIdx.x;
  const T *src = static_cast<const T *>(p.src);
  double *fn = p.fn + ((size_t)i * p.K + k) * 4;
  const double s0 = p.pref[(size_t)i * p.K + k], s1 = p.pref[(size_t)(i + 1) * p.K + k];
  const int e = binade_of(s0);
  // the guess must leave a margin on both sides of the binade, or the step would mostly be rejected
  const bool plausible = s0 > 0.0 && e >= -900 && e <= 1000 && binade_of(s0 * (1.0 - 1e-9)) == e && binade_of(s1 * (1.0 + 1e-9)) == e;
  if (!plausible) { fn[3] = kSumDirty; return; }
  const double u = mk((1023 + e - 52) << 20, 0), scale = mk((1023 + 52 - e) << 20, 0);   // 2^(e-52), 2^(52-e)
  const int v0 = i * kSumSegRows;
  double pre = 0.0, post = 0.0;      // integer-valued doubles (< 2^53 whenever the step ends up accepted)
  bool tie = false;
  int32_t bad = 0;                   // sign bit of any addend: the monotonicity argument needs x >= 0
#pragma unroll 1
  for (int r0 = 0; r0 < kSumSegRows; r0 += 16) {
    double xs[16];                   // 16 loads in flight; rows past V are clamped and count as + 0.0
#pragma unroll
    for (int j = 0; j < 16; ++j) xs[j] = sum_elem<T, MAGNITUDE>(src, (size_t)min(v0 + r0 + j, p.V - 1) * p.pitch + k, p.beta);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double x = (v0 + r0 + j < p.V) ? xs[j] : 0.0;
      bad |= hi32(x);
      const double y = x * scale;    // exact (power of two): x / u
      const double f = floor(y);
      const double r = y - f;        // exact
      if (r == 0.5) {
        if (!tie) {                  // the first tie: hardware rounds "+ u/2" to even in the walk
          tie = true;
          pre += f;
        } else {                     // s is even right after a tie, so parity(s/u) == parity(post) from here on
          const double z = post + f, h = z * 0.5;
          post = z + (h != floor(h) ? 1.0 : 0.0);
        }
      } else {
        const double q = r > 0.5 ? f + 1.0 : f;
        if (tie) post += q; else pre += q;
      }
    }
  }
  if (bad < 0) { fn[3] = kSumDirty; return; }
  fn[0] = pre * u;
  fn[1] = tie ? 0.5 * u : 0.0;
  fn[2] = post * u;
  fn[3] = (double)e;
}

// One wave per topic.  Segments are taken 256 at a time (4 groups of 64, lane j of a group owns
// segment j): all their segment functions and the raw rows of every segment already known to need
// the element-by-element path are fetched up front, so memory latency is paid once, not per step.
// Within a group the walk advances by RUNS: an inclusive wave scan of D1 over the consecutive clean,
// tie-free segments of one binade gives every candidate s_out at once (sums of multiples of u below
// 2^(e+1) are exact in any order; a partial sum that reached 2^(e+1) may have rounded, but only
// upward of 2^(e+1), so its binade test still fails); the longest prefix whose s_out stays in the
// binade is accepted in one step.  Whatever stops a run -- a tie, a dirty segment, a crossing -- is
// then taken on its own.
// LDS: 8 KiB of segment functions + 6 KiB of raw rows per (single-wave) workgroup.  Kept small on purpose: the walk
// runs while the next theta draw fills the CUs from the side stream, and a workgroup that asks for more LDS than one
// theta workgroup frees is starved until the theta draw ends (measured with 58 KiB: 80 us instead of 25 at K = 100,
// 5 ms instead of 0.06 at K = 1024).
constexpr int kWalkSuper = 256, kWalkGroups = kWalkSuper / 64, kWalkRawSlots = 12;

template <typename T, bool MAGNITUDE>
__global__ __launch_bounds__(64) void sum_walk_kernel(SumParams p) {
  chain_priority();
  __shared__ double tup[kWalkSuper][4];                     // 8 KiB
  __shared__ double raw[kWalkRawSlots][kSumSegRows];        // 6 KiB
  __shared__ int16_t dirty_list[kWalkSuper];
  const int k = blockIdx.x, lane = threadIdx.x;
  const T *src = static_cast<const T *>(p.src);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  double s = 0.0;                                           // identical in every lane
  auto load_row = [&](int seg) {                            // this lane's element of segment seg (+ 0.0 past V)
    const int v = seg * kSumSegRows + lane;
    const double x = sum_elem<T, MAGNITUDE>(src, (size_t)min(v, p.V - 1) * p.pitch + k, p.beta);
    return v < p.V ? x : 0.0;
  };
  for (int sg = 0; sg < p.nseg; sg += kWalkSuper) {
    const int nhere = min(kWalkSuper, p.nseg - sg), groups = (nhere + 63) / 64;
    __syncthreads();
    {                                                       // all segment functions of the super-group -> LDS
      double t[kWalkGroups][4];
#pragma unroll
      for (int g = 0; g < kWalkGroups; ++g) {
        const int i = min(sg + g * 64 + lane, p.nseg - 1);
        const double *fn = p.fn + ((size_t)i * p.K + k) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) t[g][c] = fn[c];
      }
#pragma unroll
      for (int g = 0; g < kWalkGroups; ++g) {
        const bool valid = g * 64 + lane < nhere;
#pragma unroll
        for (int c = 0; c < 3; ++c) tup[g * 64 + lane][c] = t[g][c];
        tup[g * 64 + lane][3] = valid ? t[g][3] : kSumPastEnd;
      }
    }
    __syncthreads();
    int ndirty = 0;                                         // list of the predicted-dirty segments, in order
    for (int g = 0; g < groups; ++g) {
      const bool dirty = tup[g * 64 + lane][3] == kSumDirty;
      const unsigned long long m = __ballot(dirty);
      if (dirty) dirty_list[ndirty + __popcll(m & lt_mask)] = (int16_t)(g * 64 + lane);
      ndirty += __popcll(m);
    }
    __syncthreads();
    const int nslots = min(ndirty, kWalkRawSlots);
    for (int b = 0; b < nslots; b += 8) {                   // their raw rows, 8 loads in flight
      double x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = load_row(sg + dirty_list[min(b + j, nslots - 1)]);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (b + j < nslots) raw[b + j][lane] = x[j];
    }
    __syncthreads();

    int slot_base = 0;                                      // dirty segments before the current group
    for (int g = 0; g < groups; ++g) {
      const double d1 = tup[g * 64 + lane][0], hh = tup[g * 64 + lane][1], d2 = tup[g * 64 + lane][2], ee = tup[g * 64 + lane][3];
      const bool valid = ee != kSumPastEnd, dirty = ee == kSumDirty;
      const unsigned long long dirty_mask = __ballot(dirty);
      unsigned long long remaining = __ballot(valid);
      while (remaining) {
        const int j0 = __ffsll((long long)remaining) - 1;
        const double e0 = __shfl(ee, j0);
        if (e0 != kSumDirty && binade_of(s) == (int)e0) {
          // the run of clean tie-free segments of binade e0 starting at j0
          const unsigned long long good = __ballot(valid && ee == e0 && hh == 0.0) >> j0;
          const int run = good == ~0ull ? 64 - j0 : __ffsll((long long)~good) - 1;
          if (run > 0) {
            double inc = (lane >= j0 && lane < j0 + run) ? d1 : 0.0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
              const double o = __shfl_up(inc, d);
              if (lane >= d) inc += o;
            }
            const double t = s + inc;
            const unsigned long long okm = __ballot(lane >= j0 && lane < j0 + run && binade_of(t) == (int)e0) >> j0;
            const int acc = okm == ~0ull ? 64 - j0 : __ffsll((long long)~okm) - 1;   // leading accepted segments (s_out is monotone)
            if (acc > 0) {
              s = __shfl(t, j0 + acc - 1);
              remaining &= ~(((acc >= 64) ? ~0ull : ((1ull << acc) - 1ull)) << j0);
              continue;
            }
          }
        }
        // segment j0 on its own
        bool ok = false;
        if (e0 != kSumDirty) {
          const int e = (int)e0;
          const double t = ((s + __shfl(d1, j0)) + __shfl(hh, j0)) + __shfl(d2, j0);
          ok = binade_of(s) == e && binade_of(t) == e;
          if (ok) s = t;
        }
        if (!ok) {
          const int slot = slot_base + __popcll(dirty_mask & ((1ull << j0) - 1ull));
          if ((dirty_mask >> j0) & 1ull && slot < kWalkRawSlots) {
#pragma unroll 16
            for (int r = 0; r < kSumSegRows; ++r) s += raw[slot][r];
          } else {                                          // a rejected guess, or more dirty segments than slots: fetch now
            const double x = load_row(sg + g * 64 + j0);
#pragma unroll 16
            for (int r = 0; r < kSumSegRows; ++r) s += __shfl(x, r);
          }
        }
        remaining &= ~(1ull << j0);
      }
      slot_base += __popcll(dirty_mask);
    }
  }
  if (lane == 0) p.out[k] = s;
}

}  // namespace ggs
