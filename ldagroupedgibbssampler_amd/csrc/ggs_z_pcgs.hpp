// ggs_z_pcgs.hpp -- the z loop of scheme=pcgs (LDAPartiallyCollapsedGibbsSampler; the loop itself
// is UncollapsedParallelLDA.sampleTopicAssignmentsParallel, UPLDA:1466-1544): theta is integrated
// out, score[k] = (n_dk + alpha_k) * phi[k][w] with the document's topic counts n_d. updated after
// every token.  The tokens of a document are therefore strictly sequential; documents are
// independent given Phi.  SURVEY.md section 8(f), item 1 ("next"); same Phi draw, same count
// rebuild and the same per-token Philox uniform (purpose Z, element = global token index) as GGS.
//
// One LANE per document: a single-wave workgroup takes 64 documents (the host sorts documents
// by length, longest first, so the 64 of a group are equally long) and walks their tokens in
// lockstep -- step t handles token t of each of the 64 documents, which is exactly the access
// pattern of a GGS chunk (64 tokens, 64 different phiT rows).  The rows stream through the
// slice ring of ggs_z_stream.hpp (3 slots here) twice per step (pass 1 sums, pass 2 re-multiplies and
// walks; the product (n + alpha) * phi is the same two roundings in both passes: int + double,
// then the multiply), the next step's first slices are in flight while this step ends.
// Per-lane state: the document's K topic counts, int16, in LDS as [k][lane] (lane-contiguous,
// conflict-free), decremented before the scores and incremented after the draw (UPLDA:1494,1535).
#pragma once
#include "ggs_z_stream.hpp"

namespace ggs {

struct PcgsParams {
  const int32_t *tok;
  const int32_t *inv_perm;
  int32_t *z, *zw;
  const int64_t *doc_ptr;
  const int32_t *order;        // [D] local documents, longest first
  const double *alpha;
  const double *phiT;
  uint32_t *status;
  int64_t num_docs, tok_base;
  uint64_t seed;
  uint32_t iteration;
  int32_t K, Kp;
  // scheme=collapsed (pcgs_z_kernel<true>, ggs_z_collapsed.hpp): phiT holds psi[w][k] = (beta + n_wk)/(betaSum + n_k) of the
  // sweep-start counts; the entry of the token's own old topic is recomputed with the token removed
  const int32_t *n_wk;         // [V][K] sweep-start counts
  const int32_t *n_k;          // [K]
  double beta, beta_sum;
};

constexpr int kPcgsMaxDocLen = 32767;  // counts are int16
constexpr int kPcgsRingSlots = 3;      // two slices ahead: 24 KiB of ring, so that four waves fit a CU at K = 100

template <bool COLLAPSED>
__global__ __launch_bounds__(64) void pcgs_z_kernel(PcgsParams p) {
  constexpr int kAhead = kPcgsRingSlots - 1;
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int K = p.K;
  const int NS = max(kAhead, (K + kSliceTopics - 1) / kSliceTopics);   // slices per pass; at least kAhead (padding topics score 0)
  const int KT = NS * kSliceTopics;
  double *alb = reinterpret_cast<double *>(smem + kPcgsRingSlots * kSliceBytes);   // alpha, zero padded to KT
  int16_t *cnt = reinterpret_cast<int16_t *>(alb + KT);                              // [KT][64]
  const unsigned char *phib = reinterpret_cast<const unsigned char *>(p.phiT);
  const size_t rowbytes = (size_t)p.Kp * 8;
  const int lrow = lane >> 3, lslot = lane & 7;
  const unsigned char *my_row = smem + lane * 128;
  const int rot = lane >> 1;

  auto row_addresses = [&](const int w, const unsigned char *(&ra)[8]) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int row = 8 * m + lrow;
      const int wm = __shfl(w, row);
      ra[m] = phib + (size_t)wm * rowbytes + (size_t)(((lslot - (row >> 1)) & 7) << 4);
    }
  };
  auto issue_slice = [&](const int s, const int slot, const unsigned char *const (&ra)[8]) {
    const size_t off = (size_t)s * 128;
#pragma unroll
    for (int m = 0; m < 8; ++m)
      __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(ra[m] + off), (lds_void_t *)(smem + slot * kSliceBytes + m * 1024), 16, 0, 0);
  };

  for (int k = lane; k < KT; k += 64) alb[k] = k < K ? p.alpha[k] : 0.0;

  const int64_t groups = (p.num_docs + 63) / 64;
  for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const int64_t di = g * 64 + lane;
    const int d = di < p.num_docs ? p.order[di] : -1;
    const int64_t beg = d >= 0 ? p.doc_ptr[d] : 0;
    const int len = d >= 0 ? (int)(p.doc_ptr[d + 1] - beg) : 0;
    const int steps = __shfl(len, 0);                              // lane 0 holds the group's longest document
    if (steps == 0) break;                                         // sorted: every later group is empty too
    // UPLDA:1482-1485 localTopicCounts
    for (int k = 0; k < KT; ++k) cnt[k * 64 + lane] = 0;
    for (int t = 0; t < len; ++t) cnt[p.z[beg + t] * 64 + lane] += 1;

    int w = len > 0 ? p.tok[beg] : 0;
    const unsigned char *ra[8], *ran[8];
    row_addresses(w, ra);
    int gs = 0;                                                    // ring slot of this step's first slice
#pragma unroll
    for (int s = 0; s < kAhead; ++s) issue_slice(s, s, ra);
    // the token's old topic and (COLLAPSED) psi of that topic with the token itself removed (MSLDA:185-190), both
    // fetched one step ahead: z[beg + t + 1] is not written before step t + 1, the counts are the sweep-start ones
    int znext = len > 0 ? p.z[beg] : 0;
    auto own_of = [&](const int word, const int topic) {
      return (p.beta + (double)(p.n_wk[(size_t)word * K + topic] - 1)) / (p.beta_sum + (double)(p.n_k[topic] - 1));
    };
    double own_next = (COLLAPSED && len > 0) ? own_of(w, znext) : 0.0;

    for (int t = 0; t < steps; ++t) {
      const bool active = t < len, has1 = t + 1 < steps;
      const int zold = znext;
      const double own = own_next;
      const int ip = active ? p.inv_perm[beg + t] : 0;
      const int w1 = (t + 1 < len) ? p.tok[beg + t + 1] : 0;
      znext = (t + 1 < len) ? p.z[beg + t + 1] : 0;
      if (COLLAPSED && t + 1 < len) own_next = own_of(w1, znext);
      if (has1) row_addresses(w1, ran);
      if (active) cnt[zold * 64 + lane] -= 1;                      // UPLDA:1494 (a count below zero cannot arise: it was built from z above)
      asm volatile("" ::: "memory");

      double sum = 0.0, tt = 0.0;
      int newc = 0;
      for (int j = 0; j < 2 * NS; ++j) {
        const int s = j < NS ? j : j - NS;
        const int cur = (gs + j) % kPcgsRingSlots;
        const int nxt = (gs + j + kAhead) % kPcgsRingSlots;
        const int ja = j + kAhead;
        if (ja < 2 * NS) issue_slice(ja < NS ? ja : ja - NS, nxt, ra);
        else if (has1) issue_slice(ja - 2 * NS, nxt, ran);
        // all but the youngest 8*kAhead DMAs done => slice j has landed; at the tail of the last step fewer follow
        if (has1 || ja < 2 * NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * kAhead) : "memory");
        else if (2 * NS - 1 - j == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (active) {
          const unsigned char *rb = my_row + cur * kSliceBytes;
          const unsigned char *ab = reinterpret_cast<const unsigned char *>(alb) + s * kSliceTopics * 8;
          const int16_t *cb = cnt + (s * kSliceTopics) * 64 + lane;
          D2 ph[kSliceUnits], al[kSliceUnits];
          int n[kSliceTopics];
#pragma unroll
          for (int u = 0; u < kSliceUnits; ++u) {
            ph[u] = lds_d2(rb + (((u + rot) & 7) << 4));
            al[u] = lds_d2(ab + u * 16);
            n[2 * u] = cb[(2 * u) * 64];
            n[2 * u + 1] = cb[(2 * u + 1) * 64];
          }
          if (COLLAPSED) {
            const int rel = zold - s * kSliceTopics;               // position of the old topic inside this slice, if any
#pragma unroll
            for (int u = 0; u < kSliceUnits; ++u) {
              if (rel == 2 * u) ph[u].a = own;
              if (rel == 2 * u + 1) ph[u].b = own;
            }
          }
          if (j < NS) {                                            // UPLDA:1509-1513 / MSLDA:196-203
#pragma unroll
            for (int u = 0; u < kSliceUnits; ++u) {
              sum += ((double)n[2 * u] + al[u].a) * ph[u].a;
              sum += ((double)n[2 * u + 1] + al[u].b) * ph[u].b;
            }
            if (j == NS - 1) {                                     // UPLDA:1519-1520; the walk runs negated, see ggs_z_sliced.hpp
              const uint64_t gtok = (uint64_t)(p.tok_base + beg + t);
              const U4 o = philox4x32_10((uint32_t)gtok, (uint32_t)(gtok >> 32), (uint32_t)GGS_PURPOSE_Z << 24, p.iteration,
                                         (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
              tt = 0.0 - u53(o.x, o.y) * sum;
            }
          } else {                                                 // UPLDA:1522-1526
            uint32_t bits = 0;
#pragma unroll
            for (int u = 0; u < kSliceUnits; ++u) {
              bits = __builtin_amdgcn_alignbit(bits, (uint32_t)hi32(tt), 31);
              tt += ((double)n[2 * u] + al[u].a) * ph[u].a;
              bits = __builtin_amdgcn_alignbit(bits, (uint32_t)hi32(tt), 31);
              tt += ((double)n[2 * u + 1] + al[u].b) * ph[u].b;
            }
            newc += __popc(bits);
          }
        }
        asm volatile("" ::: "memory");
      }
      gs = (gs + 2 * NS) % kPcgsRingSlots;

      if (active) {
        int new_topic = newc - 1;
        if (new_topic < 0 || hi32(tt) < 0) {                       // UPLDA:1529-1531
          atomicOr(p.status, ST_INVALID_TOPIC);
          new_topic = new_topic < 0 ? 0 : K - 1;
        }
        cnt[new_topic * 64 + lane] += 1;                           // UPLDA:1535
        p.z[beg + t] = new_topic;
        p.zw[ip] = new_topic;
      }
      asm volatile("" ::: "memory");
      w = w1;
#pragma unroll
      for (int m = 0; m < 8; ++m) ra[m] = ran[m];
    }
  }
}

}  // namespace ggs

// ------------------------------------------------------------------------------------------------
// pcgs_sliced_kernel<KMAX>: the same z loop for K <= kSlicedMaxTopics with the token's K scores kept in
// registers (as the cold-chunk loop of ggs_z_sliced.hpp keeps them), so the rows stream through the ring
// ONCE per step: half the LDS-DMA instructions of pcgs_z_kernel, and those are what a step costs.
// Same lane-per-document groups, same int16 [k][lane] counts in LDS, same 3-slot ring with the slice
// offset as the DMA immediate (the ring therefore starts above the alpha row and the counts).
namespace ggs {

template <int KMAX, bool COLLAPSED = false>
__global__ __launch_bounds__(64) void pcgs_sliced_kernel(PcgsParams p) {
  constexpr int NS = (KMAX + kSliceTopics - 1) / kSliceTopics;
  constexpr int kAhead = NS < kPcgsRingSlots - 1 ? NS : kPcgsRingSlots - 1;
  constexpr int kHead = (KMAX * 8 + KMAX * 128 + 255) / 256 * 256;  // alpha row + counts, below the ring (>= NS*128)
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int K = p.K;
  double *alb = reinterpret_cast<double *>(smem);                  // alpha, zero padded to KMAX
  int16_t *cnt = reinterpret_cast<int16_t *>(smem + KMAX * 8);     // [KMAX][64]
  unsigned char *ring = smem + kHead;
  const unsigned char *phib = reinterpret_cast<const unsigned char *>(p.phiT);
  const size_t rowbytes = (size_t)p.Kp * 8;
  const int lrow = lane >> 3, lslot = lane & 7;
  const unsigned char *my_row = ring + lane * 128;
  const int rot = lane >> 1;
  const int16_t *my_cnt = cnt + lane;

  auto row_addresses = [&](const int w, const unsigned char *(&ra)[8]) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int row = 8 * m + lrow;
#ifdef GGS_PCGS_ABL                                             // timing-only experiment (results wrong): every row the same one -- the kernel without its gather
      const int wm = __shfl(w, row) & (GGS_PCGS_ABL - 1);
#else
      const int wm = __shfl(w, row);
#endif
      ra[m] = phib + (size_t)wm * rowbytes + (size_t)(((lslot - (row >> 1)) & 7) << 4);
    }
  };
  auto issue_slice = [&](auto sc, const int slot, const unsigned char *const (&ra)[8]) {
    constexpr int s = decltype(sc)::value;
#pragma unroll
    for (int m = 0; m < 8; ++m)
      __builtin_amdgcn_global_load_lds((glb_cvoid_t *)ra[m], (lds_void_t *)(ring + slot * kSliceBytes + m * 1024 - s * 128), 16, s * 128, 0);
  };

  for (int k = lane; k < KMAX; k += 64) alb[k] = k < K ? p.alpha[k] : 0.0;

  const int64_t groups = (p.num_docs + 63) / 64;
  for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const int64_t di = g * 64 + lane;
    const int d = di < p.num_docs ? p.order[di] : -1;
    const int64_t beg = d >= 0 ? p.doc_ptr[d] : 0;
    const int len = d >= 0 ? (int)(p.doc_ptr[d + 1] - beg) : 0;
    const int steps = __shfl(len, 0);                              // lane 0 holds the group's longest document
    if (steps == 0) break;                                         // sorted: every later group is empty too
    for (int k = 0; k < KMAX; ++k) cnt[k * 64 + lane] = 0;         // UPLDA:1482-1485 localTopicCounts
    for (int t = 0; t < len; ++t) cnt[p.z[beg + t] * 64 + lane] += 1;

    int w = len > 0 ? p.tok[beg] : 0;
    const unsigned char *ra[8], *ran[8];
    row_addresses(w, ra);
    int gs = 0;
    static_for<0, kAhead>([&](auto sc) { issue_slice(sc, decltype(sc)::value % kPcgsRingSlots, ra); });
    int znext = len > 0 ? p.z[beg] : 0;                            // old topic and (COLLAPSED) its own-token psi, one step ahead (see pcgs_z_kernel)
    auto own_of = [&](const int word, const int topic) {
      return (p.beta + (double)(p.n_wk[(size_t)word * K + topic] - 1)) / (p.beta_sum + (double)(p.n_k[topic] - 1));
    };
    double own_next = (COLLAPSED && len > 0) ? own_of(w, znext) : 0.0;

    for (int t = 0; t < steps; ++t) {
      const bool active = t < len, has1 = t + 1 < steps;
      gs = __builtin_amdgcn_readfirstlane(gs);
      const int zold = znext;
      const double own = own_next;
      const int ip = active ? p.inv_perm[beg + t] : 0;
      const int w1 = (t + 1 < len) ? p.tok[beg + t + 1] : 0;
      znext = (t + 1 < len) ? p.z[beg + t + 1] : 0;
      if (COLLAPSED && t + 1 < len) own_next = own_of(w1, znext);
      if (has1) row_addresses(w1, ran);
      if (active) cnt[zold * 64 + lane] -= 1;                      // UPLDA:1494
      asm volatile("" ::: "memory");

      double sc[KMAX];
      double sum = 0.0;
      static_for<0, NS>([&](auto sidx) {                           // UPLDA:1509-1513
        constexpr int s = decltype(sidx)::value;
        const int cur = (gs + s) % kPcgsRingSlots;
        const int nxt = (gs + s + kAhead) % kPcgsRingSlots;
        if constexpr (s + kAhead < NS) issue_slice(std::integral_constant<int, s + kAhead>{}, nxt, ra);
        else if (has1) issue_slice(std::integral_constant<int, s + kAhead - NS>{}, nxt, ran);
        if (has1 || s + kAhead < NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * kAhead) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NS - 1 - s)) : "memory");
        if (active) {
          const unsigned char *rb = my_row + cur * kSliceBytes;
          D2 ph[kSliceUnits], al[kSliceUnits];
          int n[kSliceTopics];
#pragma unroll
          for (int u = 0; u < kSliceUnits; ++u)
            if (s * kSliceTopics + 2 * u + 1 < KMAX) {
              ph[u] = lds_d2(rb + (((u + rot) & 7) << 4));
              al[u] = lds_d2(reinterpret_cast<const unsigned char *>(alb) + (s * kSliceTopics + 2 * u) * 8);
              n[2 * u] = my_cnt[(s * kSliceTopics + 2 * u) * 64];
              n[2 * u + 1] = my_cnt[(s * kSliceTopics + 2 * u + 1) * 64];
            }
          if constexpr (COLLAPSED) {
            const int rel = zold - s * kSliceTopics;               // position of the old topic inside this slice, if any
#pragma unroll
            for (int u = 0; u < kSliceUnits; ++u) {
              if (rel == 2 * u) ph[u].a = own;
              if (rel == 2 * u + 1) ph[u].b = own;
            }
          }
#pragma unroll
          for (int u = 0; u < kSliceUnits; ++u) {
            constexpr int k0 = s * kSliceTopics;
            const int k = k0 + 2 * u;
            if (k + 1 < KMAX) {
              sc[k] = ((double)n[2 * u] + al[u].a) * ph[u].a;
              sum += sc[k];
              sc[k + 1] = ((double)n[2 * u + 1] + al[u].b) * ph[u].b;
              sum += sc[k + 1];
            }
          }
        }
        asm volatile("" ::: "memory");
      });
      gs = (gs + NS) % kPcgsRingSlots;

      if (active) {
        const uint64_t gtok = (uint64_t)(p.tok_base + beg + t);
        const U4 o = philox4x32_10((uint32_t)gtok, (uint32_t)(gtok >> 32), (uint32_t)GGS_PURPOSE_Z << 24, p.iteration,
                                   (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
        double tt = 0.0 - u53(o.x, o.y) * sum;                     // UPLDA:1519-1526, negated walk (see ggs_z_sliced.hpp)
        int newc = 0;
        bool live = true;
#pragma unroll
        for (int kb = 0; kb < KMAX; kb += 16) {
          if (live) {
            uint32_t bits = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j)
              if (kb + j < KMAX) {
                bits = __builtin_amdgcn_alignbit(bits, (uint32_t)hi32(tt), 31);
                tt += sc[kb + j];
              }
            newc += __popc(bits);
            live = __any(hi32(tt) < 0);
          }
        }
        int new_topic = newc - 1;
        if (new_topic < 0 || hi32(tt) < 0) {                       // UPLDA:1529-1531
          atomicOr(p.status, ST_INVALID_TOPIC);
          new_topic = new_topic < 0 ? 0 : K - 1;
        }
        cnt[new_topic * 64 + lane] += 1;                           // UPLDA:1535
        p.z[beg + t] = new_topic;
        p.zw[ip] = new_topic;
      }
      asm volatile("" ::: "memory");
      w = w1;
#pragma unroll
      for (int m = 0; m < 8; ++m) ra[m] = ran[m];
    }
  }
}

}  // namespace ggs
