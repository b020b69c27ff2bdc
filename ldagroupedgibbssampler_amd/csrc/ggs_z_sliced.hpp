// ggs_z_sliced.hpp -- K3, the token loop (GGS:79-130), for K <= kSlicedMaxTopics: the K scores
// of a token live in REGISTERS, exactly like the Java topicTermScores[] array.
//
// Given theta and Phi every token's draw is independent of every other (GGS:79-130 touches only
// its own z), and its random number is addressed by the token's index -- so tokens may be
// processed in any grouping.  The host cuts the corpus into CHUNKS of up to 64 tokens drawn from
// at most two consecutive documents (lane t owns token t and knows which of the chunk's two theta
// rows is its own), and keeps two chunk lists:
//
//   hot chunks   tokens of the few dozen most frequent words (Zipf: ~45 % of all tokens).  Their
//                phiT rows are copied into LDS once per launch -- one table per workgroup
//                (4 waves, one per SIMD, one workgroup per CU) -- and a hot chunk reads phi there:
//                no global traffic at all.
//   cold chunks  everything else.  Their phiT rows are streamed through a per-wave 3-slot LDS
//                ring in slices of 16 topics (128 B per row, 8 KiB per slice) by LDS-DMA, kept
//                kAhead slices ahead of the arithmetic, across chunk boundaries.
//
// Why the split: measured on MI355X, this kernel is bound by instruction ISSUE, and the issue of
// the LDS-DMA instructions (64 lanes x 16 B each through the CU's one address unit, shared by
// the 4 waves) does not overlap the wave's own arithmetic -- a chunk costs its VALU time PLUS
// its DMA issue time, whether or not the data is waited for.  Hot chunks pay no DMA issue.
// Odd waves take their hot chunks first, even waves their cold chunks, so that fewer waves
// queue at the address unit at any time.
//
// Per chunk:
//   for each slice s:   [cold: issue the LDS-DMA of slice s+kAhead (of this or the next chunk), wait for slice s]
//                       score[k] = theta[k]*phi[k][w_t], sum += score[k] for its 16 topics, k ascending (GGS:96-101)
//   U from Philox, sample = U*sum                                            (GGS:107-108)
//   walk: count the k with sample > 0, sample -= score[k], k ascending       (GGS:109-113)
//   store z (document order and word-sorted order)
//
// The two fp64 chains per token (sum, then walk) are the reference's sequential chains; the
// products are computed once and kept, as in the reference.
//
// DMA shape: one global_load_lds_dwordx4 wave-instruction fills 1 KiB = 8 rows x 128 B, lane l
// -> row 8m + l/8, LDS slot l%8.  Slot j of row r holds source unit (j - r/2) mod 8 (a per-row
// rotation chosen through the per-lane SOURCE address), so lane t's 16-byte read of unit u, at
// slot (u + t/2) mod 8 of row t, is bank-conflict free across the wave.  Per chunk every lane
// computes its 8 source row addresses once; the slice offset is the DMA instruction's immediate
// -- which the hardware adds to the LDS destination as well as to the global source, hence the
// "- s*128" on the destination and a ring that does not start at LDS address 0.  The ring slot
// is wave-uniform and reaches M0 through scalar registers.
// Lanes with nothing useful to fetch (rows past the chunk, units past the row) read whatever
// finite bytes sit there (word 0's row; the next row's first entries; the zeroed tail pad of
// phiT): those scores are multiplied by theta = 0 or belong to lanes that store nothing.
#pragma once
#include <type_traits>

#include "ggs_z_kernel.hpp"

// timing-only experiments, compile time (results are wrong on purpose): 1 no DMA, 8 DMA issued but never waited for
#ifndef GGS_ABL
#define GGS_ABL 0
#endif

namespace ggs {

constexpr int kSliceTopics = 16;
constexpr int kSliceUnits = 8;            // 16-byte units per row per slice
constexpr int kSliceBytes = 64 * 128;     // 64 rows x 16 topics x 8 B
// Two slots (one slice in flight beyond the one being scored) since round 4: the LDS a third slot takes is worth more as
// hot-word table -- measured at BASELINE config 2, sweep / z step in ms: 2 slots 1.496 / 0.922 (96 hot rows), 3 slots
// 1.548 / 0.968 (57 rows), 4 slots 1.622 / 1.048 (19 rows).  (Round 1 chose 3 when the hot chunks still cost twice as much.)
#ifndef GGS_RING_SLOTS
#define GGS_RING_SLOTS 2
#endif
constexpr int kRingSlots = GGS_RING_SLOTS;
constexpr int kSlicedMaxTopics = 192;     // 384 score registers (VGPR + AGPR) + working set < 512
constexpr int kSlicedDefaultTopics = 160; // ... and the largest K the host gives to these kernels unasked (ggs_api.hip: measured break-even)
constexpr int kPhiTailPadBytes = 1024;    // zeroed bytes after the last phiT row (see above; the pcgs kernel pads K to 48, the one-pass stream kernel to a multiple of 64)
constexpr int kSlicedWaves = 4;           // waves per workgroup (one per SIMD), sharing the hot-word table
constexpr int kChunkDocs = 2;             // documents a chunk may draw tokens from
constexpr int kHotTailBytes = 64;          // zeroed bytes after z_hot_kernel's table: what a lane refining the last slice reads past the last row
constexpr int kSlotShift = 30;            // chunk token word: value | (which of the chunk's documents) << 30

// ---- the warm tiers (z_warm_kernel, below)
constexpr int kWarmMaxTiers = 8;
constexpr int kWarmSlotShift = 16;         // warm chunk token word: table row | (which of the chunk's documents) << 16
constexpr int kWarmDocSlots = 8;           // document ids stored per warm chunk (one 32-byte scalar load), whatever warm_docs_for() says
// LDS pitch of a warm chunk's theta rows = whole slices + 16 bytes: the rows are a multiple of 128 bytes, so without the pad
// lanes of different documents reading the same topics hit the same banks (measured: 54 % of the warm kernel's LDS cycles
// were bank conflicts against the hot kernel's 38 %); with it the 16-byte reads of 8 rows at one offset touch 8 disjoint
// groups of 4 banks.
constexpr int kWarmThetaPad = 16;
#ifndef GGS_WARM_UNITS
#define GGS_WARM_UNITS 4                   // hot_token's read-ahead in z_warm_kernel (16-byte units of a slice)
#endif
#ifndef GGS_WARM_LANE_DOUBLES
#define GGS_WARM_LANE_DOUBLES 12           // registers (pairs) a lane spends on the theta rows of ONE of the two chunks whose operands are in flight
#endif
constexpr int warm_docs_for(const int kmax) {
  const int rowd = (kmax + kSliceTopics - 1) / kSliceTopics * kSliceTopics;
  // a lane holds 16-byte pieces, 128 topics per piece and row; what keeps every instance of the kernel out of scratch (a
  // register of a load in flight must never be spilled)
  const int d = rowd <= 112 ? GGS_WARM_LANE_DOUBLES / 2 : rowd <= 128 ? GGS_WARM_LANE_DOUBLES / 2 - 1 : 2;
  return d < 2 ? 2 : d > 8 ? 8 : d;
}
struct alignas(8) D2u { double a, b; };    // two doubles at an 8-byte aligned address (a theta row starts at d*K*8)

template <int S, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (S < N) {
    f(std::integral_constant<int, S>{});
    static_for<S + 1, N>(f);
  }
}

// KMAX = K rounded up to a multiple of 8: the size of the score register file.  Topics
// K..KMAX-1 are scored too, with theta = 0 (the LDS theta rows are zero-padded) against finite
// phi bytes, so they add +0.0 to the sum and subtract 0.0 in the walk: no per-topic guards.
template <int KMAX>
__global__ __launch_bounds__(kSlicedWaves * 64) void z_sliced_kernel(ZParams p) {
  constexpr int NS = (KMAX + kSliceTopics - 1) / kSliceTopics;     // slices per chunk
  constexpr int kAhead = NS < kRingSlots - 1 ? NS : kRingSlots - 1;   // slices in flight beyond the one being scored
  constexpr int NT = (KMAX + 63) / 64;                             // 64-topic pieces of a theta row
  constexpr int kThetaRow = KMAX * 8;                              // bytes of one theta row in LDS
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = p.K;
  unsigned char *thb = smem + wave * p.wave_lds;                   // the chunk's kChunkDocs theta rows
  unsigned char *ring = thb + p.ring_base;
  const unsigned char *phib = reinterpret_cast<const unsigned char *>(p.phiT);
  const size_t rowbytes = (size_t)p.Kp * 8;
  const const_i32_t *cdocs = (const const_i32_t *)p.c_docs;
  const int64_t stride = (int64_t)gridDim.x * kSlicedWaves;
  const int64_t wid = (int64_t)blockIdx.x * kSlicedWaves + wave;

  // the hot rows: KMAX doubles each (like the ring, the tail past K is finite filler scored with theta = 0)
  {
    constexpr int upr = KMAX / 2;                                  // 16-byte units per row
    for (int i = threadIdx.x; i < p.num_hot * upr; i += kSlicedWaves * 64) {
      const int r = i / upr, u = i - r * upr;
      *reinterpret_cast<D2 *>(smem + p.hot_off + r * p.hot_pitch + u * 16) =
          *reinterpret_cast<const D2 *>(phib + (size_t)p.hot_words[r] * rowbytes + (size_t)u * 16);
    }
  }
  __syncthreads();                                                 // the only barrier; no DMA is in flight yet

  const int lrow = lane >> 3, lslot = lane & 7;
  // byte offset of unit u of this lane's row inside a ring slot: lane*128 + ((u + lane/2) & 7)*16
  const unsigned char *my_row = ring + lane * 128;
  const int rot = lane >> 1;

  // theta rows of a chunk's two documents: lane l holds entries l, l + 64, ... of each (0.0 beyond K)
  auto load_theta = [&](const int d0, const int d1, double (&tv)[kChunkDocs][NT]) {
    const double *t0 = p.theta + (size_t)d0 * K, *t1 = p.theta + (size_t)d1 * K;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      tv[0][t] = (t * 64 + lane < K) ? t0[t * 64 + lane] : 0.0;
      tv[1][t] = (t * 64 + lane < K) ? t1[t * 64 + lane] : 0.0;
    }
  };
  auto stage_theta = [&](const double (&tv)[kChunkDocs][NT]) {
#pragma unroll
    for (int r = 0; r < kChunkDocs; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t * 64 + lane < KMAX) reinterpret_cast<double *>(thb + r * kThetaRow)[t * 64 + lane] = tv[r][t];
  };
  // U, the walk and the stores (GGS:107-130) of one chunk
  auto finish = [&](const double (&sc)[KMAX], const double sum, const int idx, const int ip, const int word) {
    if (idx < 0) return;
    const uint64_t gtok = (uint64_t)(p.tok_base + idx);
    const U4 o = philox4x32_10((uint32_t)gtok, (uint32_t)(gtok >> 32), (uint32_t)GGS_PURPOSE_Z << 24, p.iteration,
                               (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
    const double U = u53(o.x, o.y);
    // The walk of GGS:108-113, negated and in counting form.  t = -(sample): t_0 = 0 - U*sum and
    // t_{j+1} = t_j + score[j] round exactly as sample_{j+1} = sample_j - score[j] does (round to
    // nearest is symmetric), x + (-x) gives +0, and scores are >= +0, so t is never -0 and
    //     sample_j > 0   <=>   the sign bit of t_j is set.
    // Scores are >= 0, so once the sign clears it stays clear, and
    // newTopic + 1 == #{j : sample_j > 0} == number of sign bits collected before each subtraction.
    // One funnel shift per topic collects them; no compare, no carry add.
    double t = 0.0 - U * sum;
    int cnt = 0;
    bool live = true;
#pragma unroll
    for (int kb = 0; kb < KMAX; kb += 16) {
      if (live) {                                                  // wave-uniform
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (kb + j < KMAX) {
            bits = __builtin_amdgcn_alignbit(bits, (uint32_t)hi32(t), 31);   // (bits << 1) | sign(t)
            t += sc[kb + j];
          }
        cnt += __popc(bits);
        live = __any(hi32(t) < 0);
      }
    }
    int new_topic = cnt - 1;
    if (new_topic < 0 || hi32(t) < 0) {                            // GGS:116-118 (and the index past K Java would throw on;
                                                                   // only then can cnt have run past K through the padding)
      if (!GGS_ABL) atomicOr(p.status, ST_INVALID_TOPIC);
      new_topic = new_topic < 0 ? 0 : K - 1;
    }
    p.z[idx] = new_topic;
    p.zw[ip] = new_topic;
    // cold tokens only: the few hot words take half of all tokens, and atomics on their few hundred cells serialise in L2
    // (measured with them: the z step of one rank in eight 0.41 ms instead of 0.14) -- the hot words' segments are counted
    // by count_sorted_kernel on their own
    if (p.cnt_send && word >= 0) __hip_atomic_fetch_add(&p.cnt_send[slice_cell(p.smap, new_topic, word)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // ------------------------------------------------------------------ cold chunks
  auto cold_chunks = [&]() {
    const int64_t C = p.num_cold;
    // Source addresses of this lane's 8 DMA rows (m = 0..7): row 8m + lrow of the chunk, unit
    // (lslot - row/2) mod 8 of slice 0.
    auto row_addresses = [&](const int w, const unsigned char *(&ra)[8]) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int row = 8 * m + lrow;
        const int wm = __shfl(w, row) & ((1 << kSlotShift) - 1);   // 0 for rows past the chunk
        ra[m] = phib + (size_t)wm * rowbytes + (size_t)(((lslot - (row >> 1)) & 7) << 4);
      }
    };
    auto issue_slice = [&](auto sc, const int slot, const unsigned char *const (&ra)[8]) {
      constexpr int s = decltype(sc)::value;
      if (GGS_ABL & 1) return;
#pragma unroll
      for (int m = 0; m < 8; ++m)
        __builtin_amdgcn_global_load_lds((glb_cvoid_t *)ra[m], (lds_void_t *)(ring + slot * kSliceBytes + m * 1024 - s * 128), 16, s * 128, 0);
    };

    // ---- prologue.  Suffix 0 = this chunk, 1 = next, 2 = the one after.
    int64_t c = wid;
    if (c >= C) return;
    int w0 = p.ct_tok[c * 64 + lane], id0 = p.ct_idx[c * 64 + lane], ip0 = p.ct_ip[c * 64 + lane];
    int w1 = 0, id1 = -1, ip1 = 0, w2 = 0, id2 = -1, ip2 = 0;
    double tv0[kChunkDocs][NT], tv1[kChunkDocs][NT], tv2[kChunkDocs][NT];
    load_theta(cdocs[2 * c], cdocs[2 * c + 1], tv0);
    if (c + stride < C) {
      const int64_t c1 = c + stride;
      w1 = p.ct_tok[c1 * 64 + lane]; id1 = p.ct_idx[c1 * 64 + lane]; ip1 = p.ct_ip[c1 * 64 + lane];
      load_theta(cdocs[2 * c1], cdocs[2 * c1 + 1], tv1);
    } else {
      load_theta(0, 0, tv1);
    }
#pragma unroll
    for (int r = 0; r < kChunkDocs; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t) tv2[r][t] = 0.0;
    const unsigned char *ra[8], *ran[8];                           // DMA row addresses: this chunk, next chunk
    row_addresses(w0, ra);
    row_addresses(w1, ran);
    int g = 0;                                                     // ring slot of this chunk's slice 0
    static_for<0, kAhead>([&](auto sc) { issue_slice(sc, decltype(sc)::value % kRingSlots, ra); });

    for (;;) {
      const bool has1 = c + stride < C, has2 = c + 2 * stride < C;
      g = __builtin_amdgcn_readfirstlane(g);                       // wave-uniform by construction: keep it scalar
      stage_theta(tv0);                                            // requested two chunks ago
      const unsigned char *trow = thb + ((unsigned)w0 >> kSlotShift) * kThetaRow;   // this lane's document

      double sc[KMAX];
      double sum = 0.0;
      static_for<0, NS>([&](auto sidx) {
        constexpr int s = decltype(sidx)::value;
        const int cur = (g + s) % kRingSlots;
        const int nxt = (g + s + kAhead) % kRingSlots;        // freed by the slice scored last step
        // keep kAhead slices in flight: slice s + kAhead of this chunk, or of the next one
        if constexpr (s + kAhead < NS) issue_slice(std::integral_constant<int, s + kAhead>{}, nxt, ra);
        else if (has1) issue_slice(std::integral_constant<int, s + kAhead - NS>{}, nxt, ran);
        // LDS-DMA completion is tracked by vmcnt in issue order.  With kAhead slices (8 DMAs each)
        // issued after slice s, "at most 8*kAhead outstanding" means slice s has landed; at the tail
        // of the last chunk fewer slices follow.  The waves of the workgroup never meet in the
        // chunk loop: no hardware barrier, only a compiler fence so that no LDS read moves above the wait.
        if (GGS_ABL & (1 | 8)) {}
        else if (has1 || s + kAhead < NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * kAhead) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NS - 1 - s)) : "memory");
        if (id0 >= 0) {
          const unsigned char *rb = my_row + cur * kSliceBytes;
          // unit u = topics (k, k+1).  All 16 LDS reads of the slice are issued up front (they return
          // in order), so the chain below starts after one LDS latency and never waits again.
          D2 ph[kSliceUnits], th[kSliceUnits];
#pragma unroll
          for (int u = 0; u < kSliceUnits; ++u)
            if (s * kSliceTopics + 2 * u + 1 < KMAX) {             // compile time (KMAX is even)
              ph[u] = lds_d2(rb + (((u + rot) & 7) << 4));
              th[u] = lds_d2(trow + s * (kSliceTopics * 8) + u * 16);
            }
#pragma unroll
          for (int u = 0; u < kSliceUnits; ++u) {
            constexpr int k0 = s * kSliceTopics;
            const int k = k0 + 2 * u;
            if (k + 1 < KMAX) {
              sc[k] = th[u].a * ph[u].a;
              sum += sc[k];
              sc[k + 1] = th[u].b * ph[u].b;
              sum += sc[k + 1];
            }
          }
        }
        asm volatile("" ::: "memory");                             // every read of this ring slot is issued before it is refilled
      });
      g = (g + NS) % kRingSlots;                             // ring slot of the next chunk's slice 0

      // requests for two chunks ahead: younger than every DMA above, so they are only waited for
      // at the next chunk's slices, after the walk below.
      if (has2) {
        const int64_t c2 = c + 2 * stride;
        w2 = p.ct_tok[c2 * 64 + lane]; id2 = p.ct_idx[c2 * 64 + lane]; ip2 = p.ct_ip[c2 * 64 + lane];
        load_theta(cdocs[2 * c2], cdocs[2 * c2 + 1], tv2);
      } else {
        w2 = 0; id2 = -1; ip2 = 0;
      }

      finish(sc, sum, id0, ip0, w0 & ((1 << kSlotShift) - 1));
      if (!has1) break;
      c += stride;
      w0 = w1; id0 = id1; ip0 = ip1;
      w1 = w2; id1 = id2; ip1 = ip2;
#pragma unroll
      for (int r = 0; r < kChunkDocs; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) { tv0[r][t] = tv1[r][t]; tv1[r][t] = tv2[r][t]; }
#pragma unroll
      for (int m = 0; m < 8; ++m) ra[m] = ran[m];
      row_addresses(w1, ran);                                      // w1 was requested a chunk ago
    }
  };

  // ------------------------------------------------------------------ hot chunks: phi from the LDS table
  auto hot_chunks = [&]() {
    const int64_t C = p.num_chunks;
    int64_t c = p.num_cold + wid;
    if (c >= C) return;
    int w0 = p.ct_tok[c * 64 + lane], id0 = p.ct_idx[c * 64 + lane], ip0 = p.ct_ip[c * 64 + lane];
    double tv0[kChunkDocs][NT];
    load_theta(cdocs[2 * c], cdocs[2 * c + 1], tv0);
    for (;;) {
      const bool has1 = c + stride < C;
      int w1 = 0, id1 = -1, ip1 = 0;
      double tv1[kChunkDocs][NT];
      if (has1) {                                                  // the next chunk's operands land during this chunk's arithmetic
        const int64_t c1 = c + stride;
        w1 = p.ct_tok[c1 * 64 + lane]; id1 = p.ct_idx[c1 * 64 + lane]; ip1 = p.ct_ip[c1 * 64 + lane];
        load_theta(cdocs[2 * c1], cdocs[2 * c1 + 1], tv1);
      } else {
        load_theta(0, 0, tv1);
      }
      stage_theta(tv0);
      const unsigned char *trow = thb + ((unsigned)w0 >> kSlotShift) * kThetaRow;
      const unsigned char *hrow = smem + p.hot_off + (w0 & ((1 << kSlotShift) - 1)) * p.hot_pitch;   // table row (plain unit order)

      double sc[KMAX];
      double sum = 0.0;
      if (id0 >= 0) {
        static_for<0, NS>([&](auto sidx) {
          constexpr int s = decltype(sidx)::value;
          D2 ph[kSliceUnits], th[kSliceUnits];
#pragma unroll
          for (int u = 0; u < kSliceUnits; ++u)
            if (s * kSliceTopics + 2 * u + 1 < KMAX) {
              ph[u] = lds_d2(hrow + s * (kSliceTopics * 8) + u * 16);
              th[u] = lds_d2(trow + s * (kSliceTopics * 8) + u * 16);
            }
#pragma unroll
          for (int u = 0; u < kSliceUnits; ++u) {
            constexpr int k0 = s * kSliceTopics;
            const int k = k0 + 2 * u;
            if (k + 1 < KMAX) {
              sc[k] = th[u].a * ph[u].a;
              sum += sc[k];
              sc[k + 1] = th[u].b * ph[u].b;
              sum += sc[k + 1];
            }
          }
        });
      }
      finish(sc, sum, id0, ip0, -1);
      if (!has1) break;
      c += stride;
      w0 = w1; id0 = id1; ip0 = ip1;
#pragma unroll
      for (int r = 0; r < kChunkDocs; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) tv0[r][t] = tv1[r][t];
    }
  };

  // odd waves start with their hot chunks: fewer waves at the address unit at any time
#pragma unroll 1
  for (int phase = 0; phase < 2; ++phase) {
    if ((phase ^ wave) & 1) hot_chunks();
    else cold_chunks();
  }
}

// The LDS table of z_hot_kernel / z_warm_kernel: `nrows` phiT rows (KMAX doubles each, plain unit order), a zeroed pad
// unit behind each (hot_pitch) and kHotTailBytes of zeros behind the last.  A row per wave at a time, its word id a scalar
// load, six rows in flight per wave (one row per thread-loop iteration with the id fetched by the lanes was two dependent
// memory latencies per 16 bytes and thread: ~30 us per table).  The caller puts the barrier behind it.
template <int KMAX>
__device__ __forceinline__ void load_phi_table(unsigned char *table, const int pitch, const unsigned char *phib, const size_t rowbytes,
                                               const int32_t *words_g, const int nrows, const int wave, const int lane, const int tid) {
  constexpr int upr = KMAX / 2 + 1, kRowsAhead = 6, NP = (upr + 63) / 64;
  const const_i32_t *words = (const const_i32_t *)words_g;
  for (int r0 = wave; r0 < nrows; r0 += kSlicedWaves * kRowsAhead) {
    D2 v[kRowsAhead][NP];
#pragma unroll
    for (int a = 0; a < kRowsAhead; ++a) {
      const int r = r0 + kSlicedWaves * a;
      const unsigned char *src = phib + (size_t)words[r < nrows ? r : r0] * rowbytes;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int u = q * 64 + lane;
        v[a][q] = u < upr - 1 ? *reinterpret_cast<const D2 *>(src + (size_t)u * 16) : D2{0.0, 0.0};
      }
    }
#pragma unroll
    for (int a = 0; a < kRowsAhead; ++a) {
      const int r = r0 + kSlicedWaves * a;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int u = q * 64 + lane;
        if (r < nrows && u < upr) *reinterpret_cast<D2 *>(table + r * pitch + u * 16) = v[a][q];
      }
    }
  }
  if (tid < kHotTailBytes / 8) reinterpret_cast<double *>(table + nrows * pitch)[tid] = 0.0;
}

// One token of a hot (or warm) chunk, as z_hot_kernel's comment below describes it: phi from the table row `hrow`, theta
// from the LDS row `trow` (both zero-padded to whole slices).  Returns the new topic.
// UB = 16-byte units of a slice read ahead of the chain at a time: 8 (a whole slice: 64 registers of operands) in z_hot_kernel,
// 4 in z_warm_kernel, whose next chunk's theta rows take 32 registers of the same 128.
template <int KMAX, int UB = kSliceUnits>
__device__ __forceinline__ int hot_token(const ZParams &p, const int K, const unsigned char *hrow, const unsigned char *trow, const int id0) {
  constexpr int NS = (KMAX + kSliceTopics - 1) / kSliceTopics;
  double sum = 0.0, ck[NS];
  static_for<0, NS>([&](auto sidx) {                           // GGS:96-101
    constexpr int s = decltype(sidx)::value;
#pragma unroll
    for (int b = 0; b < kSliceUnits / UB; ++b) {
      D2 ph[UB], th[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (s * kSliceTopics + 2 * (b * UB + u) + 1 < KMAX) {
          ph[u] = lds_d2(hrow + s * (kSliceTopics * 8) + (b * UB + u) * 16);
          th[u] = lds_d2(trow + s * (kSliceTopics * 8) + (b * UB + u) * 16);
        }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (s * kSliceTopics + 2 * (b * UB + u) + 1 < KMAX) {
          sum += th[u].a * ph[u].a;
          sum += th[u].b * ph[u].b;
        }
      asm volatile("" : "+v"(sum) : : "memory");               // one batch of reads in flight at a time (the register budget is 128): the chain is pinned before the next batch's reads
    }
    ck[s] = sum;
  });
  const uint64_t gtok = (uint64_t)(p.tok_base + id0);
  const U4 o = philox4x32_10((uint32_t)gtok, (uint32_t)(gtok >> 32), (uint32_t)GGS_PURPOSE_Z << 24, p.iteration,
                             (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
  const double t0 = u53(o.x, o.y) * sum;                       // GGS:107-108
  const double delta = (sum * (double)K) * 0x1p-51 * p.margin_scale;
  // the first slice whose closing checkpoint proves the walk has stopped (d = t0 - s is monotone)
  int gsel = NS;
  static_for<0, NS>([&](auto sidx) {
    constexpr int s = NS - 1 - decltype(sidx)::value;
    if (t0 - ck[s] < -delta) gsel = s;
  });
  bool undecided = gsel == NS;                                 // the walk would not end inside the row, NaN, or too close to call
  if (undecided) gsel = 0;
  double s = 0.0;                                              // the chain as it stood when pass 1 entered the slice
  static_for<0, NS - 1>([&](auto sidx) {
    constexpr int q = decltype(sidx)::value;
    if (gsel == q + 1) s = ck[q];
  });
  undecided |= fabs(t0 - s) <= delta;                          // d_{r-1} > delta for r at the head of the slice (t_0 > delta for r = 0)
  int ahead = 0;                                               // topics of the slice the walk is proved to pass
  {
    const unsigned char *hs = hrow + gsel * (kSliceTopics * 8), *ts = trow + gsel * (kSliceTopics * 8);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                              // half a slice's reads in flight at a time
      D2 ph[kSliceUnits / 2], th[kSliceUnits / 2];
#pragma unroll
      for (int u = 0; u < kSliceUnits / 2; ++u) {
        ph[u] = lds_d2(hs + (h * (kSliceUnits / 2) + u) * 16);
        th[u] = lds_d2(ts + (h * (kSliceUnits / 2) + u) * 16);
      }
#pragma unroll
      for (int u = 0; u < kSliceUnits / 2; ++u) {
        s += th[u].a * ph[u].a;
        double d = t0 - s;
        ahead += d >= -delta;
        undecided |= fabs(d) <= delta;
        s += th[u].b * ph[u].b;
        d = t0 - s;
        ahead += d >= -delta;
        undecided |= fabs(d) <= delta;
      }
      asm volatile("" ::: "memory");
    }
  }
  int new_topic = gsel * kSliceTopics + ahead;
  if (undecided || ahead >= kSliceTopics || new_topic >= K) {
    // the exact replay: GGS:108-113 element by element from the table row (rare; see ggs_z_stream.hpp)
    const double *prow = reinterpret_cast<const double *>(hrow), *trw = reinterpret_cast<const double *>(trow);
    double sample = t0;
    new_topic = -1;
    while (sample > 0.0) {
      ++new_topic;
      if (new_topic >= K) break;
      sample -= trw[new_topic] * prow[new_topic];
    }
    if (new_topic < 0 || new_topic >= K) {                     // GGS:116-118 (and the index past K Java would throw on)
      atomicOr(p.status, ST_INVALID_TOPIC);
      new_topic = new_topic < 0 ? 0 : K - 1;
    }
  }
  return new_topic;
}

// ------------------------------------------------------------------------------------------------
// z_hot_kernel: the hot chunks on their own, launched on a second stream BESIDE z_sliced_kernel (which
// then takes the cold chunks only).  A lone wave issues one VALU instruction per ~8 cycles and sits out
// its own DMA issue; a second wave on the SIMD fills those slots (two waves per SIMD get 1.7x the issue
// rate of one).  The cold kernel's waves hold most of a SIMD's 512 registers, so the guest must live in
// <= 128 (amdgpu_waves_per_eu(4)): it keeps no score registers.  Pass 1 sums a token's K products from the
// LDS table (GGS:96-101) and keeps the chain at every slice boundary (NS <= 12 checkpoints); where the walk of
// GGS:108-113 stops is then DECIDED from that chain by the margin argument of z_stream1_kernel (ggs_z_stream.hpp:
// the same delta, the same test; the slice the draw falls into is walked once more from the table, continuing
// the chain from its checkpoint -- the same additions in the same order), and the token in 10^10 that is too
// close to call replays the Java walk element by element.  One 4-wave workgroup per CU; its LDS (the table + 2
// theta rows per wave) is what the cold kernel's rings leave.  Theta rows are zero-padded to whole slices and
// 64 zero bytes follow the table, so a lane refining the last slice multiplies finite bytes by 0.
// If the two kernels happen not to share the CUs the results are the same and the hot chunks simply run
// before or after the cold ones.
// The chunk loop of the two table kernels (z_hot_kernel: DOCS = 2 documents per chunk; z_warm_kernel: warm_docs_for(KMAX)).
// Its loads run two chunks ahead and are counted by hand.  Measured (in-kernel cycle counters and kernel timelines): beside the
// cold kernel -- whose row gather keeps the CU's vector-memory pipeline full -- a guest wave waits hundreds of cycles to
// ISSUE each vector-memory instruction, its loads come back after ~4 us where a chunk's arithmetic takes 2-3, and the
// compiler's wait insertion puts s_waitcnt vmcnt(0) at the head of such a loop (the z stores of the chunk before are
// waited for as well).  Hence: few and wide loads (one 16-byte list entry per lane, one 16-byte piece per theta row and
// lane); every load of the loop issued by inline assembly (the compiler neither sees nor waits for it), for the chunk
// after next, into one of two register sets; and the wait in front of a set's first use counts what was issued behind
// that set's loads and may stay in flight: the other set's kLoads loads and one chunk's two z stores (vector memory
// operations complete in issue order on gfx9, stores included -- the assumption the compiler's own counts rest on; every
// chunk has an active lane, so its stores are issued; anything issued beyond that, a rare replay's status atomic or the
// count update, only makes the wait stricter).  A set's registers are handed to the compiler through empty asm statements
// behind the wait, so no use can move above it -- and no instance of these kernels may spill (a register of a load in
// flight must never be copied): tests/test_abi_symbols.py checks the build's resource summary.
typedef int i3v_t __attribute__((ext_vector_type(3)));
typedef int i4v_t __attribute__((ext_vector_type(4)));
// The wait and, IN THE SAME asm statement, the copies of the list entry's three words into registers of their own: those
// words outlive the set (it is refilled while the chunk is sampled), and copies the compiler makes for that reason it may
// place anywhere behind the load statement -- in front of the wait, reading registers the load has not written yet (it
// did: a memory access fault on the first run).  The theta registers are only masked and stored to LDS right behind the
// wait; they go through empty asm statements there.  tests/test_abi_symbols.py reads the generated assembly of every
// instance: outside the asm statements nothing but those masking selects may read a register an asm load writes.
template <int N>
__device__ __forceinline__ void wait_vmcnt_take(const i3v_t &e, int &w, int &id, int &ip) {
  asm volatile("s_waitcnt vmcnt(%6)\n\tv_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5"
               : "=&v"(w), "=&v"(id), "=&v"(ip) : "v"(e.x), "v"(e.y), "v"(e.z), "n"(N) : "memory");
}

// pack: per lane {table row | (which of the chunk's documents) << kWarmSlotShift, local token index or -1, its word-sorted
// position, 0}; docs: kWarmDocSlots local documents per chunk (the first DOCS in use); chunks first + wid, + stride, ... < C.
// count_words: null, or the word ids of the table's rows -- then every token adds its cell into p.cnt_send.
template <int KMAX, int DOCS, int PAD, int UB>
__device__ __forceinline__ void table_chunks(const ZParams &p, unsigned char *smem, unsigned char *thb, const int4 *pack, const int32_t *docs_g,
                                             const int64_t first, const int64_t C, const int64_t stride, const int lane, const int32_t *count_words) {
  constexpr int NS = (KMAX + kSliceTopics - 1) / kSliceTopics;
  constexpr int kRowD = NS * kSliceTopics;                         // doubles of a theta row in LDS (zero-padded to whole slices)
  constexpr int kThetaRow = kRowD * 8 + PAD;
  constexpr int NQ = (kRowD + 127) / 128;                          // 16-byte pieces per lane of a theta row
  constexpr int kLoads = 1 + DOCS * NQ;                            // vector loads per chunk: its list entry and its theta rows
  constexpr int kStores = 2;                                       // ... and stores: z in document order and in word order
  const int K = p.K;
  const const_i32_t *cdocs = (const const_i32_t *)docs_g;
  const uint32_t loff = (uint32_t)lane * 16;
  // Lane l holds topics 2l, 2l + 1 (+ 128q) of each of the chunk's DOCS rows: one 16-byte load per row and lane from a scalar
  // row base (the document ids stay in SGPRs), unconditional (a lane past the row re-reads its head; the last lane of an odd K
  // reads 8 bytes past its row -- the next row, or the slack ggs_set_corpus leaves behind the theta buffers) and masked to
  // 0.0 beyond K when it is staged.
  uint32_t toff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { const int k0 = 2 * (q * 64 + lane); toff[q] = (uint32_t)(k0 < K ? k0 : 0) * 8; }
  struct Set { i3v_t e; i4v_t t[DOCS][NQ]; };
  auto issue = [&](const int64_t c, Set &S) {
    typedef int docs8_t __attribute__((ext_vector_type(8)));
    const docs8_t dd = *reinterpret_cast<const __attribute__((address_space(4))) docs8_t *>(cdocs + c * kWarmDocSlots);   // one s_load_dwordx8
    const int4 *lp = pack + c * 64;
    asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(S.e) : "v"(loff), "s"(lp) : "memory");
#pragma unroll
    for (int r = 0; r < DOCS; ++r) {
      const double *tr = p.theta + (size_t)dd[r] * K;
#pragma unroll
      for (int q = 0; q < NQ; ++q) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(S.t[r][q]) : "v"(toff[q]), "s"(tr) : "memory");
    }
  };
  // `after`: what was issued behind this set's loads and may stay in flight -- 0 nothing known, 1 the other set's loads, 2
  // those and the stores of the chunk sampled in between
  auto arrive = [&](Set &S, const int after, int &w0, int &id0, int &ip0) {
    if (after == 2) wait_vmcnt_take<kLoads + kStores>(S.e, w0, id0, ip0);
    else if (after == 1) wait_vmcnt_take<kLoads>(S.e, w0, id0, ip0);
    else wait_vmcnt_take<0>(S.e, w0, id0, ip0);
#pragma unroll
    for (int r = 0; r < DOCS; ++r)
#pragma unroll
      for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(S.t[r][q]));
  };
  // one chunk: its operands have arrived in S; S is refilled for the chunk after next as soon as it is staged
  auto chunk = [&](Set &S, const int64_t c, const int w0, const int id0, const int ip0) {
#pragma unroll
    for (int r = 0; r < DOCS; ++r)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (2 * (q * 64 + lane) < kRowD) {                         // 0.0 beyond K: the table's filler there is multiplied by it
          const int k0 = 2 * (q * 64 + lane);
          const D2 v = __builtin_bit_cast(D2, S.t[r][q]);
          *reinterpret_cast<D2 *>(thb + r * kThetaRow + k0 * 8) = D2{k0 < K ? v.a : 0.0, k0 + 1 < K ? v.b : 0.0};
        }
    if (c + 2 * stride < C) issue(c + 2 * stride, S);
    const int row = w0 & ((1 << kWarmSlotShift) - 1);
    const unsigned char *trow = thb + ((unsigned)w0 >> kWarmSlotShift) * kThetaRow;
    const unsigned char *hrow = smem + p.hot_off + row * p.hot_pitch;
    if (id0 >= 0) {
      const int new_topic = hot_token<KMAX, UB>(p, K, hrow, trow, id0);
      p.z[id0] = new_topic;
      p.zw[ip0] = new_topic;
      if (count_words)
        __hip_atomic_fetch_add(&p.cnt_send[slice_cell(p.smap, new_topic, count_words[row])], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  int64_t c = first;
  if (c >= C) return;
  Set A, B;
  issue(c, A);
  if (c + stride < C) issue(c + stride, B);
  int done = 0;                                                    // chunks sampled behind the loads a wait is for: 0, then steady state
  for (;;) {
    int w0, id0, ip0;
    arrive(A, c + stride < C ? (done ? 2 : 1) : 0, w0, id0, ip0);
    chunk(A, c, w0, id0, ip0);
    c += stride;
    if (c >= C) break;
    arrive(B, c + stride < C ? 2 : 0, w0, id0, ip0);
    chunk(B, c, w0, id0, ip0);
    c += stride;
    if (c >= C) break;
    done = 1;
  }
}

template <int KMAX>
__global__ __launch_bounds__(kSlicedWaves * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void z_hot_kernel(ZParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  // Raised priority (round 4): beside the cold kernel each vector-memory instruction of a guest wave waited ~800 cycles to be
  // ISSUED (in-kernel cycle counters: a third of z_warm_kernel's time went into issuing 7 loads per chunk, a tenth into its 2
  // stores); with priority over the cold waves at the arbiter it is ~350 (z_warm_kernel 641 000 -> 451 000 cycles per wave;
  // z step 0.92 -> 0.875 ms with both guests raised).  The cold kernel loses nothing it can use: it waits for its rows.
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char *thb = smem + wave * p.wave_lds;                   // this wave's kChunkDocs theta rows
  load_phi_table<KMAX>(smem + p.hot_off, p.hot_pitch, reinterpret_cast<const unsigned char *>(p.phiT), (size_t)p.Kp * 8, p.hot_words, p.num_hot, wave, lane, threadIdx.x);
  __syncthreads();
  // (from 129 topics on a theta row is two pieces per lane: half a slice of operands read ahead, like z_warm_kernel, keeps the kernel out of scratch)
  table_chunks<KMAX, kChunkDocs, 0, (KMAX <= 128 ? kSliceUnits : kSliceUnits / 2)>(p, smem, thb, p.ht_pack, p.h_docs, (int64_t)blockIdx.x * kSlicedWaves + wave, p.num_chunks - p.num_cold,
                                                 (int64_t)gridDim.x * kSlicedWaves, lane, nullptr);
}

// ------------------------------------------------------------------------------------------------
// z_warm_kernel: the WARM tiers (round 4).  A token whose row sits in an LDS table costs about 0.4 of one whose row is
// gathered (z_hot_kernel alone: 27 M tokens per ms, the cold chunks alone: 10.6), and the split z step ends when the COLD
// kernel does (0.94 ms; the hot kernel beside it is through after 0.57) -- so the words next in frequency after the hot
// table's get tables of their own, one after the other: tier t = the next `warm_rows` words, its table loaded by every
// workgroup (a few us from L2), its chunks taken like hot chunks, then a barrier and the next tier.  Such words are rarer
// per document (benchmark corpus: 12 tokens per document in the first tier after the hot table's 110, then 7, 5, ...), so
// a warm chunk draws its 64 tokens from up to DOCS = warm_docs_for(KMAX) documents (6 up to K = 112) instead of 2: lane t
// reads theta from its own document's LDS row, as it always did.  The theta rows of the NEXT chunk travel through
// registers (DOCS rows x NS slices = at most 14 doubles per lane) while this chunk is sampled.  Everything else is
// z_hot_kernel: the same hot_token(), hence the same z bit for bit whichever list a token is in.  Launched behind
// z_hot_kernel on its stream (split form) or behind z_sliced_kernel (fused form).  With an exchange whose send buffer the
// z kernels fill themselves (ZParams::cnt_send) a warm token adds its own cell, like a cold one.

template <int KMAX>
__global__ __launch_bounds__(kSlicedWaves * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void z_warm_kernel(ZParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);                                   // a guest beside the cold kernel: see z_hot_kernel
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char *thb = smem + wave * p.wave_lds;                   // this wave's DOCS theta rows
  const const_i64_t *meta = (const const_i64_t *)p.warm_meta;
  const int64_t stride = (int64_t)gridDim.x * kSlicedWaves;
  const int64_t wid = (int64_t)blockIdx.x * kSlicedWaves + wave;
  for (int tier = 0; tier < p.warm_tiers; ++tier) {
    if (tier) __syncthreads();                                     // every wave is through with the previous table
    const int32_t *words = p.warm_words + (size_t)tier * p.warm_rows;
    load_phi_table<KMAX>(smem + p.hot_off, p.hot_pitch, reinterpret_cast<const unsigned char *>(p.phiT), (size_t)p.Kp * 8, words, (int)meta[p.warm_tiers + 1 + tier], wave, lane, threadIdx.x);
    __syncthreads();                                               // (also: nothing of this wave's is in flight here -- the table's loads have been waited for)
    table_chunks<KMAX, warm_docs_for(KMAX), kWarmThetaPad, GGS_WARM_UNITS>(p, smem, thb, p.wt_pack, p.w_docs, meta[tier] + wid, meta[tier + 1], stride, lane,
                                                                          p.cnt_send ? words : nullptr);
  }
}

}  // namespace ggs
