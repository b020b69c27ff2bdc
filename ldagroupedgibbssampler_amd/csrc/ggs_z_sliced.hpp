// ggs_z_sliced.hpp -- K3, the token loop (GGS:79-130), for K <= kSlicedMaxTopics: the K scores
// of a token live in REGISTERS, exactly like the Java topicTermScores[] array.
//
// Persistent single-wave workgroups stride the chunk table; one chunk = up to 64 consecutive
// tokens of ONE document, lane t owns token t.  The phiT rows of the chunk are streamed
// through a 4-slot LDS ring in slices of 16 topics (128 B per row, 8 KiB per slice), kept
// kAhead slices ahead of the arithmetic, across chunk boundaries:
//
//   for each slice s:   issue the LDS-DMA of slice s+kAhead (of this or the next chunk)
//                       wait for slice s; score[k] = theta[k]*phi[k][w_t], sum += score[k]
//                       for its 16 topics, k ascending                       (GGS:96-101)
//   U from Philox, sample = U*sum                                            (GGS:107-108)
//   walk: cnt += (sample > 0); sample -= score[k], k ascending               (GGS:109-113)
//   store z
//
// The two fp64 chains per token (sum, then walk) are the reference's sequential chains; the
// products are computed once and kept, as in the reference.  HBM->LDS traffic is one pass over
// each row.  Word ids and theta rows are requested two chunks ahead, chunk descriptors three.
//
// DMA shape: one global_load_lds_dwordx4 wave-instruction fills 1 KiB = 8 rows x 128 B, lane l
// -> row 8m + l/8, LDS slot l%8.  Slot j of row r holds source unit (j - r/2) mod 8 (a per-row
// rotation chosen through the per-lane SOURCE address), so lane t's 16-byte read of unit u, at
// slot (u + t/2) mod 8 of row t, is bank-conflict free across the wave.  Per chunk every lane
// computes its 8 source row addresses once; each DMA then only adds an immediate slice offset.
// Lanes with nothing useful to fetch (rows past the chunk, units past the row) read whatever
// finite bytes sit there (word 0's row; the next row's first entries; the zeroed tail pad of
// phiT): those scores are multiplied by theta = 0 or belong to lanes that store nothing.
#pragma once
#include <type_traits>

#include "ggs_z_kernel.hpp"

// timing-only experiments, compile time (results are wrong on purpose): 1 no DMA, 2 no walk, 4 no score pass
#ifndef GGS_ABL
#define GGS_ABL 0
#endif

namespace ggs {

constexpr int kSliceTopics = 16;
constexpr int kSliceUnits = 8;            // 16-byte units per row per slice
constexpr int kSliceBytes = 64 * 128;     // 64 rows x 16 topics x 8 B
constexpr int kRingSlots = 4;
constexpr int kSlicedMaxTopics = 192;     // 384 score registers (VGPR + AGPR) + working set < 512
constexpr int kPhiTailPadBytes = 256;     // zeroed bytes after the last phiT row (see above)

constexpr int kRingBase = 2048;           // ring offset inside the wave's LDS: the theta row (<= 1536 B) sits below it

template <int S, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (S < N) {
    f(std::integral_constant<int, S>{});
    static_for<S + 1, N>(f);
  }
}

// KMAX = K rounded up to a multiple of 8: the size of the score register file.  Topics
// K..KMAX-1 are scored too, with theta = 0 (the LDS theta row is zero-padded) against finite
// phi bytes, so they add +0.0 to the sum and subtract 0.0 in the walk: no per-topic guards.
//
// Issue economy (one wave per SIMD: every VALU instruction costs ~8 cycles whatever it does):
// the slice index is a compile-time constant, so a DMA's slice offset is the instruction's
// immediate -- which the hardware adds to the LDS destination as well as to the global source,
// hence the "- s*128" on the destination and the ring starting at kRingBase, not 0; the ring
// slot is wave-uniform and goes to M0 through scalar registers.
template <int KMAX>
__global__ __launch_bounds__(64) void z_sliced_kernel(ZParams p) {
  constexpr int NS = (KMAX + kSliceTopics - 1) / kSliceTopics;     // slices per chunk
  constexpr int kAhead = NS < 3 ? NS : 3;                          // slices in flight beyond the one being scored
  constexpr int NT = (KMAX + 63) / 64;                             // 64-topic pieces of a theta row
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int K = p.K, Kp = p.Kp;
  unsigned char *thb = smem;                                       // theta row, KMAX doubles
  unsigned char *ring = smem + kRingBase;
  const unsigned char *phib = reinterpret_cast<const unsigned char *>(p.phiT);
  const size_t rowbytes = (size_t)Kp * 8;
  const const_i64_t *cstart = (const const_i64_t *)p.chunk_start;
  const const_i32_t *clen = (const const_i32_t *)p.chunk_len;
  const const_i32_t *cdoc = (const const_i32_t *)p.chunk_doc;
  const int64_t stride = gridDim.x;
  const int64_t C = p.num_chunks;

  const int lrow = lane >> 3, lslot = lane & 7;
  // byte offset of unit u of this lane's row inside a ring slot: lane*128 + ((u + lane/2) & 7)*16
  const unsigned char *my_row = ring + lane * 128;
  const int rot = lane >> 1;

  // Source addresses of this lane's 8 DMA rows (m = 0..7): row 8m + lrow of the chunk, unit
  // (lslot - row/2) mod 8 of slice 0.  Slice s adds s*128 bytes as an immediate.
  auto row_addresses = [&](const int w, const unsigned char *(&ra)[8]) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int row = 8 * m + lrow;
      const int wm = __shfl(w, row);                               // 0 for rows past the chunk
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
  auto load_theta = [&](const int doc, double (&tv)[NT]) {
    const double *thg = p.theta + (size_t)doc * K;
#pragma unroll
    for (int t = 0; t < NT; ++t) tv[t] = (t * 64 + lane < K) ? thg[t * 64 + lane] : 0.0;
  };

  // ---- prologue.  Suffix 0 = this chunk, 1 = next, 2 = the one after; descriptors reach to 3.
  int64_t c = blockIdx.x;
  if (c >= C) return;
  int64_t start0 = cstart[c], start1 = 0, start2 = 0, start3 = 0;
  int len0 = clen[c], len1 = 0, len2 = 0, len3 = 0;
  int doc0 = cdoc[c], doc1 = 0, doc2 = 0, doc3 = 0;
  if (c + stride < C) { start1 = cstart[c + stride]; len1 = clen[c + stride]; doc1 = cdoc[c + stride]; }
  if (c + 2 * stride < C) { start2 = cstart[c + 2 * stride]; len2 = clen[c + 2 * stride]; doc2 = cdoc[c + 2 * stride]; }
  int w0 = (lane < len0) ? p.tok[start0 + lane] : 0;
  int w1 = (lane < len1) ? p.tok[start1 + lane] : 0;              // len1 == 0 when there is no next chunk
  int w2 = 0;
  // word-sorted slot of each lane's token (for the second z store), requested with the word ids
  int ip0 = (lane < len0) ? p.inv_perm[start0 + lane] : 0;
  int ip1 = (lane < len1) ? p.inv_perm[start1 + lane] : 0;
  int ip2 = 0;
  double tv0[NT], tv1[NT], tv2[NT];
  load_theta(doc0, tv0);
  load_theta(doc1, tv1);                                           // doc1 == 0 (a valid row) when there is no next chunk
#pragma unroll
  for (int t = 0; t < NT; ++t) tv2[t] = 0.0;
  const unsigned char *ra[8], *ran[8];                             // DMA row addresses: this chunk, next chunk
  row_addresses(w0, ra);
  row_addresses(w1, ran);
  int g = 0;                                                       // ring slot of this chunk's slice 0
  static_for<0, kAhead>([&](auto sc) { issue_slice(sc, decltype(sc)::value & (kRingSlots - 1), ra); });

  for (;;) {
    const bool has1 = c + stride < C, has2 = c + 2 * stride < C;
    g = __builtin_amdgcn_readfirstlane(g);                         // wave-uniform by construction: keep it scalar
    // this chunk's theta row: registers -> LDS (requested two chunks ago)
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t * 64 + lane < KMAX) reinterpret_cast<double *>(thb)[t * 64 + lane] = tv0[t];   // 0.0 beyond K

    double sc[KMAX];
    double sum = 0.0;
    if (GGS_ABL & 4) {
      sum = 1.0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) sc[k] = 0.01;
    }
    static_for<0, NS>([&](auto sidx) {
      constexpr int s = decltype(sidx)::value;
      const int cur = (g + s) & (kRingSlots - 1);
      const int nxt = (g + s + kAhead) & (kRingSlots - 1);          // freed by the slice scored last step
      // keep kAhead slices in flight: slice s + kAhead of this chunk, or of the next one
      if constexpr (s + kAhead < NS) issue_slice(std::integral_constant<int, s + kAhead>{}, nxt, ra);
      else if (has1) issue_slice(std::integral_constant<int, s + kAhead - NS>{}, nxt, ran);
      // LDS-DMA completion is tracked by vmcnt in issue order.  With kAhead slices (8 DMAs each)
      // issued after slice s, "at most 8*kAhead outstanding" means slice s has landed; at the tail
      // of the last chunk fewer slices follow.  One wave per workgroup: no hardware barrier, only
      // a compiler fence so that no LDS read moves above the wait.
      if (GGS_ABL & 1) {}
      else if (has1 || s + kAhead < NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * kAhead) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NS - 1 - s)) : "memory");
      if ((GGS_ABL & 4) ? false : lane < len0) {
        const unsigned char *rb = my_row + cur * kSliceBytes;
        const unsigned char *tb = thb + s * kSliceTopics * 8;
        // unit u = topics (k, k+1).  All 16 LDS reads of the slice are issued up front (they return
        // in order), so the chain below starts after one LDS latency and never waits again.
        D2 ph[kSliceUnits], th[kSliceUnits];
#pragma unroll
        for (int u = 0; u < kSliceUnits; ++u)
          if (s * kSliceTopics + 2 * u + 1 < KMAX) {               // compile time (KMAX is even)
            ph[u] = lds_d2(rb + (((u + rot) & 7) << 4));
            th[u] = lds_d2(tb + u * 16);
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
      asm volatile("" ::: "memory");                               // every read of this ring slot is issued before it is refilled
    });
    g = (g + NS) & (kRingSlots - 1);                               // ring slot of the next chunk's slice 0

    // requests for two chunks ahead: younger than every DMA above, so they are only waited for
    // at the next chunk's slices, after the walk below.
    if (has2) {
      if (c + 3 * stride < C) { start3 = cstart[c + 3 * stride]; len3 = clen[c + 3 * stride]; doc3 = cdoc[c + 3 * stride]; }
      else { start3 = 0; len3 = 0; doc3 = 0; }
      w2 = (lane < len2) ? p.tok[start2 + lane] : 0;
      ip2 = (lane < len2) ? p.inv_perm[start2 + lane] : 0;
      load_theta(doc2, tv2);
    }

    if (lane < len0) {
      const uint64_t gtok = (uint64_t)(p.tok_base + start0 + lane);
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
        if (live && !(GGS_ABL & 2)) {                              // wave-uniform
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
      if (new_topic < 0 || hi32(t) < 0) {                          // GGS:116-118 (and the index past K Java would throw on;
                                                                   // only then can cnt have run past K through the padding)
        atomicOr(p.status, ST_INVALID_TOPIC);
        new_topic = new_topic < 0 ? 0 : K - 1;
      }
      p.z[start0 + lane] = new_topic;
      p.zw[ip0] = new_topic;
    }
    if (!has1) break;
    c += stride;
    start0 = start1; len0 = len1; doc0 = doc1; w0 = w1; ip0 = ip1;
    start1 = start2; len1 = len2; doc1 = doc2; w1 = w2; ip1 = ip2;
    start2 = start3; len2 = len3; doc2 = doc3;
#pragma unroll
    for (int t = 0; t < NT; ++t) { tv0[t] = tv1[t]; tv1[t] = tv2[t]; }
#pragma unroll
    for (int m = 0; m < 8; ++m) ra[m] = ran[m];
    row_addresses(w1, ran);                                        // w1 was requested a chunk ago
  }
}

}  // namespace ggs
