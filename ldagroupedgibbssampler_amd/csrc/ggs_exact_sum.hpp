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
// independent of s except when x mod u == u/2 exactly (a tie: the result is then the EVEN
// multiple, which depends on s).  Inside a binade and away from ties the chain is therefore
// integer arithmetic on quantised addends, which any number of lanes can add up in any order.
//
// Per topic and 64-row segment a SEGMENT FUNCTION is computed under a GUESS of the binade the
// running sum has there, for two candidate binades e_lo and e_lo + 1 (the guess may be off by
// +-41 %): D1(e) = u * sum of R_u(x) over the segment, or "unusable" if an addend hits a tie.
// Then one wave per topic WALKS the segments: a step s -> s + D1(e) is ACCEPTED only if s and the
// result both lie in binade e -- s is monotone, so then every intermediate sum did too and the
// assumption held.  The guess never decides a result: anything it gets wrong (and every binade
// crossing, the start at s = 0, ties, NaN, negative addends) is re-done the Java way, element by
// element, from the raw data.  Consecutive usable segments of one binade collapse into one wave
// scan (sums of multiples of u below 2^(e+1) are exact in any order).
//
// Where the guess comes from (`guess` [nseg + 1][K], the running sum at each segment start):
//   * the magnitude sum of sweep t: the EXACT running sums the walk of sweep t-1 wrote out -- counts move
//     slowly between sweeps; the first time (or after ggs_set_z) two extra launches make an order-free
//     guess (sum_seg, sum_prefix);
//   * the gamma sum of sweep t: the exact running MAGNITUDE sums of the same sweep -- the expectation of
//     a Gamma(a) draw is a, so the two chains run within a fraction of a percent of each other; its
//     segment functions are computed by the workgroup that drew the segment's gammas
//     (phi_gamma_kernel), no launch of their own.
// A Phi phase is then  segfn -> walk -> gamma draw (+ segfn) -> walk -> normalise:  5 launches where the
// round-2 form took 10 and a memset (seg, prefix, segfn, walk for each sum).
//
// About ten segments of 782 take the element-by-element path at V = 50k (s doubles ~25 times,
// mostly inside the first segment).
#pragma once
#include "ggs_device_math.hpp"

namespace ggs {

constexpr int kSumSegRows = 64;        // rows per segment == lanes per wave (sum_walk loads a segment with one load per lane)
constexpr int kSumBlock = 128;         // topics per workgroup in the row-streaming kernels
constexpr double kSumDirty = -5000.0;  // fn[0] marker (binades are -900..1000): no segment function, walk the raw elements
constexpr double kSumPastEnd = -6000.0;

struct SumParams {
  const void *src;        // int32 [V][pitch] counts (MAGNITUDE) or fp64 [V][pitch] gamma draws
  double *guess;          // [nseg + 1][K] running sum at each segment start: a guess on the way in (any quality);
                          //   the walk overwrites it with the exact values when write_pref is set
  double *fn;             // [nseg][K][4]: e_lo (+ 0.5: fetch the raw rows ahead), D1(e_lo), D1(e_lo + 1) (< 0: unusable), column count
  double *out;            // [K]
  int32_t *n_k;           // MAGNITUDE only, optional: the integer column sums of the counts (tokensPerTopic)
  double beta;
  int32_t pitch, K, V, nseg;
  int32_t write_pref;
};

// These kernels are short, latency-bound links of the chain between the z step and the next one, and they run while
// the theta draw of the side stream keeps every SIMD busy: their waves ask for instruction issue ahead of its waves.
__device__ __forceinline__ void chain_priority() { __builtin_amdgcn_s_setprio(3); }

// Cross-lane moves of a double by DPP (data-parallel primitives: the move rides on a VALU instruction, a few cycles)
// instead of ds_bpermute (an LDS-crossbar round trip, ~100 cycles): a 64-lane inclusive scan is 6 dependent steps, and
// the walk below does one per run.  Lanes a step does not write (row_mask) or that have no source (row_shr across the
// row's start, bound_ctrl off) read +0.0, the identity.  Needs all 64 lanes active.
template <int CTRL, int ROW_MASK, bool BOUND_ZERO>
__device__ __forceinline__ double dpp_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)lo32(x), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)hi32(x), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
  return mk(hi, (uint32_t)lo);
}
__device__ __forceinline__ double wave_inclusive_scan(double x) {
  x += dpp_f64<0x111, 0xf, true>(x);    // row_shr:1 (a lane without a source reads 0: no preset of the destination)
  x += dpp_f64<0x112, 0xf, true>(x);    // row_shr:2
  x += dpp_f64<0x114, 0xf, true>(x);    // row_shr:4
  x += dpp_f64<0x118, 0xf, true>(x);    // row_shr:8   -- every row of 16 lanes holds its own inclusive scan
  x += dpp_f64<0x142, 0xa, false>(x);   // row_bcast:15 into rows 1 and 3 (rows 0 and 2 keep the preset 0)
  x += dpp_f64<0x143, 0xc, false>(x);   // row_bcast:31 into rows 2 and 3
  return x;
}
__device__ __forceinline__ double read_lane(double x, int lane) {   // lane: wave-uniform
  return mk(__builtin_amdgcn_readlane((int)hi32(x), lane), (uint32_t)__builtin_amdgcn_readlane((int)lo32(x), lane));
}

__device__ __forceinline__ uint32_t udiv_magic32(uint32_t d) { return d > 1 ? (uint32_t)((0x100000000ull + d - 1) / d) : 0u; }   // ceil(2^32 / d): x / d = umulhi(x, m) for x * d < 2^32; 0 stands for d = 1
__device__ __forceinline__ int binade_of(double x) { return ((hi32(x) >> 20) & 0x7ff) - 1023; }   // 1024 for NaN/inf, -1023 for 0

template <typename T, bool MAGNITUDE>
__device__ __forceinline__ double sum_elem(const T *src, size_t idx, double beta) {
  return MAGNITUDE ? (beta + (double)src[idx]) : (double)src[idx];   // GGS:188: beta + count, one rounding
}

// ---- the cold guess: order-free segment sums and their exclusive prefix (two launches, only when no better guess exists) ----
template <typename T, bool MAGNITUDE>
__global__ __launch_bounds__(kSumBlock) void sum_seg_kernel(SumParams p) {
  chain_priority();
  const int k = blockIdx.y * kSumBlock + threadIdx.x;
  if (k >= p.K) return;
  const int i = blockIdx.x;
  const T *src = static_cast<const T *>(p.src);
  const int v0 = i * kSumSegRows;
  double a = 0.0;
  // 16 loads in flight; rows past V are clamped and contribute + 0.0
#pragma unroll 1
  for (int r0 = 0; r0 < kSumSegRows; r0 += 16) {
    T xs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) xs[j] = src[(size_t)min(v0 + r0 + j, p.V - 1) * p.pitch + k];
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (v0 + r0 + j < p.V) a += MAGNITUDE ? (p.beta + (double)xs[j]) : (double)xs[j];
  }
  p.guess[(size_t)i * p.K + k] = a;
}

// one wave per topic: exclusive prefix over the segment sums, in place; guess[nseg][k] = total
__global__ __launch_bounds__(64) void sum_prefix_kernel(SumParams p) {
  chain_priority();
  const int k = blockIdx.x, lane = threadIdx.x;
  const int per = (p.nseg + 63) / 64;
  const int i0 = min(lane * per, p.nseg), i1 = min(i0 + per, p.nseg);
  double mine = 0.0;
  for (int i = i0; i < i1; ++i) mine += p.guess[(size_t)i * p.K + k];
  double incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  double run = incl - mine;
  for (int i = i0; i < i1; ++i) {
    const double t = p.guess[(size_t)i * p.K + k];
    p.guess[(size_t)i * p.K + k] = run;
    run += t;
  }
  if (lane == 63) p.guess[(size_t)p.nseg * p.K + k] = incl;
}

// ---- the segment function of one (segment, topic), element-parallel ----
// Which lanes look at which of the 64 addends does not matter: D1(e) = u * sum of R_u(x) is a sum of INTEGERS
// (x / u rounded to nearest), accumulated with atomic adds in LDS by whatever lanes hold the addends -- the lanes
// that have just drawn the gammas (phi_gamma_kernel), or the lanes of a streaming pass over the counts
// (sum_segfn_kernel).  Per (segment, topic): the candidate binades from the guess, two accumulators, flags.
constexpr int kSegNoGuess = -100000;       // e_lo marker: no usable guess, the walk takes the raw rows
constexpr int kSegTieLo = 1, kSegTieHi = 2, kSegBad = 4;

struct SegCand {
  int e_lo;        // candidates e_lo and e_lo + 1, or kSegNoGuess
  bool ahead;      // the running sum probably crosses a binade inside the segment: have the raw rows fetched ahead
};
// g0 / g1: the guessed running sum at the segment's start / end
__device__ __forceinline__ SegCand seg_candidates(double g0, double g1) {
  const int e0 = binade_of(g0), e1 = binade_of(g1);
  // no usable guess (the start of the chain, a sum that is not finite, a multi-binade segment): raw rows
  if (!(g0 > 0.0) || e0 < -890 || e1 > 990 || e1 > e0 + 1 || e1 < e0) return SegCand{kSegNoGuess, true};
  // the second candidate sits on the side of e0 the guess is nearer to (in ratio): below sqrt(2) * 2^e0 the true sum
  // may still be one binade down, above it one binade up; a guess that sees a crossing inside the segment says {e0, e1}
  const bool lower_half = e1 == e0 && (hi32(g0) & 0x000fffff) < 0x6a09e;    // mantissa of sqrt(2) = 1.6a09e...
  return SegCand{lower_half ? e0 - 1 : e0, binade_of(g0 * (1.0 - 0x1p-13)) != binade_of(g1 * (1.0 + 0x1p-13))};
}
// R_u(x) / u for both candidates (integer-valued doubles); flags what makes a candidate unusable
__device__ __forceinline__ void seg_quantise(double x, int e_lo, double &q_lo, double &q_hi, int &flags) {
  q_lo = q_hi = 0.0; flags = 0;
  if (!(x >= 0.0)) { flags = kSegBad; return; }                 // negative or NaN: the monotonicity argument needs x >= 0
  const double y_lo = x * mk((1023 + 52 - e_lo) << 20, 0);      // exact (power of two): x / u
  const double y_hi = y_lo * 0.5;                                // ... for the binade above (exact: y_lo is far from subnormal, or 0)
  {
    const double f = floor(y_lo), r = y_lo - f;                  // exact (an integer for y_lo >= 2^52: r = 0)
    if (r == 0.5) flags |= kSegTieLo;                            // a tie rounds to EVEN, which depends on the running sum
    q_lo = r > 0.5 ? f + 1.0 : f;
  }
  {
    const double f = floor(y_hi), r = y_hi - f;
    if (r == 0.5) flags |= kSegTieHi;
    q_hi = r > 0.5 ? f + 1.0 : f;
  }
}
// The accumulators are doubles in LDS, added to with the LDS unit's fp64 atomic add: every addend is an integer, so as
// long as the sum stays below 2^53 every add is exact and the order the lanes arrive in does not matter; a sum that
// reaches 2^53 (it may have rounded on the way, but never back below 2^53: adds of non-negative values are monotone)
// gives D1 >= 2^(e+1), which the walk's binade test rejects.
__device__ __forceinline__ void seg_accumulate(double *acc_lo, double *acc_hi, int32_t *flag, double q_lo, double q_hi, int fl) {
  if (q_lo != 0.0) __hip_atomic_fetch_add(acc_lo, q_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (q_hi != 0.0) __hip_atomic_fetch_add(acc_hi, q_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (fl) atomicOr(flag, fl);
}
// the four doubles the walk reads: e_lo (+ 0.5: fetch ahead), D1(e_lo), D1(e_lo + 1) (< 0: unusable), column count
__device__ __forceinline__ void seg_compose(SegCand c, double a_lo, double a_hi, int flags, double count, double *fn) {
  fn[3] = count;
  if (c.e_lo == kSegNoGuess || (flags & kSegBad)) { fn[0] = kSumDirty; fn[1] = fn[2] = -1.0; return; }
  // rows ahead also when a candidate is unusable (a tie: typically one addend as large as the running sum itself)
  fn[0] = (double)c.e_lo + ((c.ahead || (flags & (kSegTieLo | kSegTieHi))) ? 0.5 : 0.0);
  fn[1] = (flags & kSegTieLo) ? -1.0 : a_lo * mk((1023 + c.e_lo - 52) << 20, 0);
  fn[2] = (flags & kSegTieHi) ? -1.0 : a_hi * mk((1023 + c.e_lo - 51) << 20, 0);
}

// A workgroup takes one 64-row segment x up to kSegFnCols adjacent topics and streams the cells with all lanes.
constexpr int kSegFnCols = 32;
template <typename T, bool MAGNITUDE>
__global__ __launch_bounds__(256) void sum_segfn_kernel(SumParams p) {
  chain_priority();
  __shared__ double acc_lo[kSegFnCols], acc_hi[kSegFnCols];
  __shared__ int32_t e_los[kSegFnCols], aheads[kSegFnCols], flag_s[kSegFnCols], cnt_s[kSegFnCols];
  const int tid = threadIdx.x, seg = blockIdx.x, kb = blockIdx.y * kSegFnCols, kw = min(kSegFnCols, p.K - kb);
  const int v0 = seg * kSumSegRows, rows = min(kSumSegRows, p.V - v0);
  const T *src = static_cast<const T *>(p.src);
  if (tid < kw) {
    const SegCand c = seg_candidates(p.guess[(size_t)seg * p.K + kb + tid], p.guess[(size_t)(seg + 1) * p.K + kb + tid]);
    e_los[tid] = c.e_lo; aheads[tid] = c.ahead ? 1 : 0; acc_lo[tid] = 0; acc_hi[tid] = 0; flag_s[tid] = 0; cnt_s[tid] = 0;
  }
  __syncthreads();
  const uint32_t m_kw = udiv_magic32((uint32_t)kw);
  for (int j = tid; j < rows * kw; j += 256) {
    const int dv = m_kw ? (int)__umulhi((uint32_t)j, m_kw) : j, c = j - dv * kw;
    const T raw = src[(size_t)(v0 + dv) * p.pitch + kb + c];
    if (MAGNITUDE && raw) atomicAdd(&cnt_s[c], (int32_t)raw);
    const int e_lo = e_los[c];
    if (e_lo == kSegNoGuess) continue;
    double q_lo, q_hi;
    int fl;
    seg_quantise(MAGNITUDE ? (p.beta + (double)raw) : (double)raw, e_lo, q_lo, q_hi, fl);   // GGS:188: beta + count, one rounding
    seg_accumulate(&acc_lo[c], &acc_hi[c], &flag_s[c], q_lo, q_hi, fl);
  }
  __syncthreads();
  if (tid < kw)
    seg_compose(SegCand{e_los[tid], aheads[tid] != 0}, acc_lo[tid], acc_hi[tid], flag_s[tid], (double)cnt_s[tid], p.fn + ((size_t)seg * p.K + kb + tid) * 4);
}

// One wave per topic.  Segments are taken 256 at a time (4 groups of 64, lane j of a group owns segment j), in a
// two-deep software pipeline: while super-group i is walked, the segment functions of super-group i + 2 and the raw rows
// of the segments of super-group i + 1 that asked for them (a predicted crossing, no usable guess) are in flight -- memory
// latency is paid once at the start, not per super-group.
// Within a group the walk advances by RUNS: an inclusive wave scan of D1 over the consecutive
// segments usable in the binade e of the running sum gives every candidate s_out at once (sums of
// multiples of u below 2^(e+1) are exact in any order; a partial sum that reached 2^(e+1) may have
// rounded, but only upward of 2^(e+1), so its binade test still fails); the longest prefix whose
// s_out stays in the binade is accepted in one step.  Whatever stops a run -- a tie, a dirty
// segment, a crossing, a segment function for other binades -- is walked element by element.
// A lone wave issues one instruction per ~8 cycles, so what the walk costs is its instruction count: cross-lane moves by
// DPP and v_readlane, row fetches only for the slots in use, 16-byte loads of the segment functions.
// LDS: 2 x 6 KiB of segment functions + 8 KiB of raw rows per (single-wave) workgroup (the rows of the next super-group
// wait in registers until the current one is walked).  Kept small on purpose: the
// walk runs while the next theta draw fills the CUs from the side stream, and a workgroup that asks for more LDS than
// one theta workgroup frees is starved until the theta draw ends (measured with 58 KiB: 80 us instead of 25 at K = 100,
// 5 ms instead of 0.06 at K = 1024).
constexpr int kWalkSuper = 256, kWalkGroups = kWalkSuper / 64, kWalkRawSlots = 16;

#ifdef GGS_WALK_TRACE                                         // scripts/probes/walk_probe.hip: where the walk's time goes
__device__ unsigned long long g_walk_trace[128];
#define GGS_WT(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 64) g_walk_trace[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#if GGS_WALK_TRACE > 1                                         // step counters too: each is a global read-modify-write, they distort the times
#define GGS_WC(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 128) g_walk_trace[i] += 1; } while (0)
#else
#define GGS_WC(i) do { } while (0)
#endif
#else
#define GGS_WT(i) do { } while (0)
#define GGS_WC(i) do { } while (0)
#endif

template <typename T, bool MAGNITUDE>
__global__ __launch_bounds__(64) void sum_walk_kernel(SumParams p) {
  chain_priority();
  GGS_WT(0);
  __shared__ double tup[2][kWalkSuper][3];                  // 12 KiB
  __shared__ double raw[kWalkRawSlots][kSumSegRows];        // 8 KiB
  __shared__ int16_t dirty_list[kWalkSuper];
  const int k = blockIdx.x, lane = threadIdx.x;
  const T *src = static_cast<const T *>(p.src);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  double s = 0.0;                                           // identical in every lane
  double cnt = 0.0;                                         // this lane's share of the column's integer sum
  auto load_row = [&](int seg) {                            // this lane's element of segment seg (+ 0.0 past V)
    const int v = seg * kSumSegRows + lane;
    const double x = sum_elem<T, MAGNITUDE>(src, (size_t)min(v, p.V - 1) * p.pitch + k, p.beta);
    return v < p.V ? x : 0.0;
  };
  auto wants_rows = [](double ee) { return ee == kSumDirty || (ee > kSumDirty && ee != floor(ee)); };
  double *pref = p.write_pref ? p.guess : nullptr;          // exact running sums out: the next sweep's guess
  if (pref && lane == 0) pref[k] = 0.0;

  double tn[kWalkGroups][4];                                // segment functions in flight (the lane's segment of each group)
  auto fetch_fn = [&](int sg) {
#pragma unroll
    for (int g = 0; g < kWalkGroups; ++g) {
      const int i = min(sg + g * 64 + lane, p.nseg - 1);
      const double2 *fn = reinterpret_cast<const double2 *>(p.fn + ((size_t)i * p.K + k) * 4);   // 32-byte records
      const double2 a = fn[0], b = fn[1];
      tn[g][0] = a.x; tn[g][1] = a.y; tn[g][2] = b.x; tn[g][3] = b.y;
    }
  };
  // tn -> tup[buf]; the list of the super-group's segments that want their raw rows; returns how many get a slot
  auto stage_fn = [&](int sg, int buf) {
    const int nhere = min(kWalkSuper, p.nseg - sg);
    int ndirty = 0;
#pragma unroll
    for (int g = 0; g < kWalkGroups; ++g) {
      const bool valid = g * 64 + lane < nhere;
      const double ee = valid ? tn[g][0] : kSumPastEnd;
      tup[buf][g * 64 + lane][0] = ee;
      tup[buf][g * 64 + lane][1] = tn[g][1];
      tup[buf][g * 64 + lane][2] = tn[g][2];
      if (valid) cnt += tn[g][3];
      const bool ahead = wants_rows(ee);
      const unsigned long long m = __ballot(ahead);
      if (ahead) dirty_list[ndirty + __popcll(m & lt_mask)] = (int16_t)(g * 64 + lane);
      ndirty += __popcll(m);
    }
    __syncthreads();
    return min(ndirty, kWalkRawSlots);
  };
  double rx[kWalkRawSlots];                                 // raw rows in flight
  int nslots_staged = 0;
  auto fetch_rows = [&](int sg, int nslots) {
    nslots_staged = nslots;
#pragma unroll
    for (int j = 0; j < kWalkRawSlots; ++j)
      if (j < nslots) rx[j] = load_row(sg + (int)dirty_list[j]);           // wave-uniform: unused slots cost nothing
  };
  auto stage_rows = [&]() {
#pragma unroll
    for (int j = 0; j < kWalkRawSlots; ++j)
      if (j < nslots_staged) raw[j][lane] = rx[j];
  };

  // prologue: super-group 0 staged, super-group 1's segment functions in flight
  fetch_fn(0);
  int nslots = stage_fn(0, 0);
  fetch_rows(0, nslots);
  stage_rows();
  if (kWalkSuper < p.nseg) fetch_fn(kWalkSuper);
  __syncthreads();
  GGS_WT(1);

  for (int sg = 0, it = 0; sg < p.nseg; sg += kWalkSuper, ++it) {
    const int buf = it & 1, nhere = min(kWalkSuper, p.nseg - sg), groups = (nhere + 63) / 64;
    const bool more = sg + kWalkSuper < p.nseg;
    if (more) {                                             // stage the next super-group, start the one after
      nslots = stage_fn(sg + kWalkSuper, buf ^ 1);
      fetch_rows(sg + kWalkSuper, nslots);
      if (sg + 2 * kWalkSuper < p.nseg) fetch_fn(sg + 2 * kWalkSuper);
    }
    int slot_base = 0;                                      // fetched-ahead segments before the current group
    for (int g = 0; g < groups; ++g) {
      const double ee = tup[buf][g * 64 + lane][0], d_lo = tup[buf][g * 64 + lane][1], d_hi = tup[buf][g * 64 + lane][2];
      const bool valid = ee != kSumPastEnd;
      const bool ahead = wants_rows(ee);
      const int e_lo = ee > kSumDirty ? (int)floor(ee) : -100000;
      const unsigned long long ahead_mask = __ballot(ahead);
      unsigned long long remaining = __ballot(valid);
      while (remaining) {
        const int j0 = __ffsll((long long)remaining) - 1;
        const int e0 = binade_of(s);
        // this lane's segment function for the binade the running sum is in (< 0: none)
        const double d1 = !valid ? -1.0 : e0 == e_lo ? d_lo : e0 == e_lo + 1 ? d_hi : -1.0;
        const unsigned long long good = __ballot(d1 >= 0.0) >> j0;       // NaN compares false
        const int run = good == ~0ull ? 64 - j0 : __ffsll((long long)~good) - 1;
        if (run > 0) {
          const double t = s + wave_inclusive_scan((lane >= j0 && lane < j0 + run) ? d1 : 0.0);
          const bool in_run = lane >= j0 && lane < j0 + run;
          const unsigned long long okm = __ballot(in_run && binade_of(t) == e0) >> j0;
          const int acc = okm == ~0ull ? 64 - j0 : __ffsll((long long)~okm) - 1;   // leading accepted segments (s_out is monotone)
          if (acc > 0) {
            GGS_WC(40); GGS_WC(64 + it);
            if (pref && lane >= j0 && lane < j0 + acc) pref[(size_t)(sg + g * 64 + lane + 1) * p.K + k] = t;
            s = read_lane(t, j0 + acc - 1);
            remaining &= ~(((acc >= 64) ? ~0ull : ((1ull << acc) - 1ull)) << j0);
            continue;
          }
        }
        // segment j0 the Java way, element by element
        const int slot = slot_base + __popcll(ahead_mask & ((1ull << j0) - 1ull));
        if ((ahead_mask >> j0) & 1ull && slot < kWalkRawSlots) {
          GGS_WC(41); GGS_WC(80 + it);
#pragma unroll 16
          for (int r = 0; r < kSumSegRows; ++r) s += raw[slot][r];
        } else {                                            // not foreseen, or more than the slots hold: fetch now
          GGS_WC(42); GGS_WC(96 + it);
          const double x = load_row(sg + g * 64 + j0);
#pragma unroll 16
          for (int r = 0; r < kSumSegRows; ++r) s += read_lane(x, r);
        }
        if (pref && lane == 0) pref[(size_t)(sg + g * 64 + j0 + 1) * p.K + k] = s;
        remaining &= ~(1ull << j0);
      }
      slot_base += __popcll(ahead_mask);
    }
    GGS_WT(2 + 2 * it);
    if (more) stage_rows();                                 // the rows fetched at the top have had the whole walk to arrive
    __syncthreads();
    GGS_WT(3 + 2 * it);
  }
  if (MAGNITUDE && p.n_k) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);         // integers below 2^31: exact in any order
    if (lane == 0) p.n_k[k] = (int32_t)cnt;
  }
  if (lane == 0) p.out[k] = s;
}

}  // namespace ggs
