// ggs_z_pcgs_wave.hpp -- the z loop of scheme=pcgs / scheme=collapsed for WIDE topic rows and LONG documents:
// one WAVE per document, the K topics spread over the 64 lanes.
//
// The lane-per-document kernels of ggs_z_pcgs.hpp keep the 64 documents' topic counts in LDS as int16 [K][64]: 128 bytes
// of LDS per topic and wave (K = 1024 would need 128 KiB), documents below 32 768 tokens.  The reference loops have no
// such limits (UPLDA:1466-1545, MSLDA:158-226).  Here a document owns a wave: its counts are int32 [K] (4 bytes per
// topic), every lane holds the phiT entries of K/64 topics of the current token's word, and the tokens of the
// document are taken one after the other -- which is what the sampler demands (n_d. moves with every token);
// documents are independent given Phi, so waves run side by side over a length-sorted list.
//
// Java's per-token arithmetic is two sequential fp64 chains over k (UPLDA:1509-1526): sum += score[k], then
// sample = U * sum and sample -= score[k] until it is <= 0.  With the topics spread over the lanes these become a
// wave reduction and a search over a wave scan, which round differently -- so they only PROPOSE the topic, by the
// margin argument of ggs_z_stream.hpp (z_stream1_kernel), restated for this kernel:
//   every partial sum of the K scores, in whatever association, is within K * 2^-53 * S of the real partial sum
//   (S = the real sum of the scores), and so are Java's chain values s_j; Java's `sample` after subtracting scores
//   0..k is within (2K + 1) * 2^-53 * S of (U * S - P_{k+1}) (P = real prefix).  With d_k = T' - C'_{k+1} (T' = U * S',
//   C' = this kernel's prefix; T', C' and S' carry at most NB + 9 roundings each beyond the real values) and the margin
//   delta = (K + 16) * 2^-51 * S' = (4 K + 64) * 2^-53 * S' -- more than the sum of all these error bounds --
//   d_k < -delta proves Java's sample <= 0 after topic k and d_k > delta proves it > 0; the real prefixes are monotone,
//   so the first k with d_k < -delta is Java's topic provided d_{k-1} > delta.
// A token is undecided only if U * S falls within delta of one of the K prefixes (probability ~ K^2 * 2^-50); such a
// token, one whose U * sum is 0 and one whose walk would leave [0, K) are replayed exactly as Java does it, element by
// element over the scores (read lane by lane out of the registers that hold them), and raise what Java raises.  GGS_DEBUG_MARGIN scales delta up: the tests send
// nearly every token through the replay, and both ways give the oracle's bits.
// The scores themselves, ((double)n_dk + alpha_k) * phi[k][w], are the same two roundings as in Java, per element.
//
// Layout: a lane owns the 16-byte units u = lane + 64 j (topics 2u, 2u + 1) of a row, j < NB: every load instruction of
// the wave reads 1 KiB of consecutive bytes of the row, and block j (topics 128 j .. 128 j + 127) is in lane order.
// The rows of the next one or two tokens are loaded into further register sets while the current token is computed
// (two ahead up to K = 1024: round 4; the wave then waits for a row only when the memory system is more than two
// tokens' arithmetic behind).
// LDS per wave: counts int32 [128 NB] and alpha fp64 [128 NB]: 12 KiB at K = 1024.
#pragma once
#include "ggs_z_pcgs.hpp"

namespace ggs {

// How far ahead of the token being sampled its phiT row is requested: as many tokens as register sets of K/64 doubles
// fit beside the working set WITH the resident waves kept (what the z step lives on): two up to K = 512, one at K = 1024
// (three waves per SIMD; measured 15.0 ms against 16.9 with two sets ahead and two waves).  The sets rotate
// by position in a loop unrolled kDepth + 1 times, so no row is ever copied.  Depth 0 (K > 2048: a row is 128 registers
// per lane, two of them plus the working set spilled 450 bytes per lane to scratch and the collapsed z step took 637 ms
// instead of 116; K = 2048 by measurement: pcgs 41.0 ms with depth 0 and six waves per CU against 48.8 with depth 1 and
// four, collapsed 47.9 against 42.5) keeps ONE set: the token's row is
// dead once the block the draw falls into has been picked out of it, and the next token's row is requested into the same
// registers right there, under the rest of the step (the scan of the deciding block, the stores, the next token's start).
#ifndef GGS_PCGS_WAVE_GROUPS
#define GGS_PCGS_WAVE_GROUPS 8
#endif
#ifndef GGS_PCGS_WAVE_DEPTH_AT_16
#define GGS_PCGS_WAVE_DEPTH_AT_16 0
#endif
#ifndef GGS_PCGS_WAVE_DEPTH_AT_8
#define GGS_PCGS_WAVE_DEPTH_AT_8 1
#endif
template <int NB, bool COLLAPSED>
constexpr int pcgs_wave_depth() { return NB < 8 ? 2 : NB == 8 ? GGS_PCGS_WAVE_DEPTH_AT_8 : NB == 16 ? (COLLAPSED ? 1 : GGS_PCGS_WAVE_DEPTH_AT_16) : 0; }

// waves per SIMD the register allocation aims for: three at K = 1024 (12 waves of 12 KiB LDS per CU), where resident waves
// decide the speed (measured, z step at K = 1024: 15.0 ms with 3 waves per SIMD and rows one token ahead, 16.9 with 2 waves
// and rows two tokens ahead)
template <int NB>
constexpr int pcgs_wave_min_waves() { return NB == 8 ? 3 : 1; }

template <int NB, bool COLLAPSED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(pcgs_wave_min_waves<NB>()))) void pcgs_wave_kernel(PcgsParams p, double margin_scale) {
  constexpr int KT = NB * 128;
  constexpr int kDepth = pcgs_wave_depth<NB, COLLAPSED>(), kSets = kDepth + 1;
  constexpr int kNG = NB < GGS_PCGS_WAVE_GROUPS ? NB : GGS_PCGS_WAVE_GROUPS, kG = NB / kNG;   // groups of the block search, blocks per group
  extern __shared__ __align__(16) unsigned char smem[];
  int32_t *cnt = reinterpret_cast<int32_t *>(smem);                        // [KT]
  double *alb = reinterpret_cast<double *>(smem + (size_t)KT * 4);         // [KT] alpha, zero padded
  const int lane = threadIdx.x, K = p.K;
  const int units = p.Kp / 2;                                              // 16-byte units per phiT row (Kp is even; a padding column holds 0)

  for (int k = lane; k < KT; k += 64) alb[k] = k < K ? p.alpha[k] : 0.0;

  struct Row { double a[NB], b[NB]; };
  auto load_row = [&](int w, Row &r) __attribute__((always_inline)) {
    // UNCONDITIONAL loads (a lane past the row's last unit reads that unit again: its topics are >= K and score 0 by the
    // guards of `scores`): a load under a lane mask or a branch makes the compiler's s_waitcnt accounting fall back to
    // vmcnt(0) at the next use of ANY row -- which waited for the rows requested ahead as well and undid the prefetch
    const double2 *row = reinterpret_cast<const double2 *>(p.phiT + (size_t)w * p.Kp);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const double2 v = row[min(lane + 64 * j, units - 1)];
      r.a[j] = v.x; r.b[j] = v.y;
    }
  };
  auto wave_sum = [&](double x) { return read_lane(wave_inclusive_scan(x), 63); };   // any association: a proposal only

  for (int64_t di = blockIdx.x; di < p.num_docs; di += gridDim.x) {
    const int d = p.order[di];
    if (d < 0) continue;                                                   // padding of the lane-per-document kernels' list
    const int64_t beg = p.doc_ptr[d];
    const int len = (int)(p.doc_ptr[d + 1] - beg);
    if (len == 0) continue;
    __syncthreads();
    for (int k = lane; k < KT; k += 64) cnt[k] = 0;                        // UPLDA:1482-1485 localTopicCounts
    __syncthreads();
    for (int t = lane; t < len; t += 64) atomicAdd(&cnt[p.z[beg + t]], 1);
    __syncthreads();

    // Per 64 tokens, lane-parallel and one chunk ahead of the sequential loop: the word, the old topic, the position in the
    // word-sorted order, the token's uniform (Philox) and, COLLAPSED, the own-topic psi -- none of them depends on what the
    // loop does to earlier tokens (z[beg + t] is only written at step t; the psi's counts are the sweep-start ones), so
    // the loop itself waits for no memory but the phiT rows, and those are kDepth tokens ahead.
    struct Chunk { int w, zold, ip; double U, own; };
    auto load_chunk = [&](int t0) {
      Chunk c{0, 0, 0, 0.0, 0.0};
      const int t = t0 + lane;
      if (t < len) {
        c.w = p.tok[beg + t]; c.zold = p.z[beg + t]; c.ip = p.inv_perm[beg + t];
        const uint64_t gtok = (uint64_t)(p.tok_base + beg + t);
        const U4 o = philox4x32_10((uint32_t)gtok, (uint32_t)(gtok >> 32), (uint32_t)GGS_PURPOSE_Z << 24, p.iteration, (uint32_t)p.seed,
                                   (uint32_t)(p.seed >> 32));
        c.U = u53(o.x, o.y);
        if (COLLAPSED)                                                     // psi of the old topic with the token itself removed, MSLDA:185-190
          c.own = (p.beta + (double)(p.n_wk[(size_t)c.w * K + c.zold] - 1)) / (p.beta_sum + (double)(p.n_k[c.zold] - 1));
      }
      return c;
    };
    Chunk ch = load_chunk(0), chn = len > 64 ? load_chunk(64) : ch;
    // word of token t (wave-uniform), t within the current chunk or the next one
    auto word_of = [&](int t_in_chunk) {
      return t_in_chunk < 64 ? __builtin_amdgcn_readlane(ch.w, t_in_chunk) : __builtin_amdgcn_readlane(chn.w, t_in_chunk - 64);
    };
    Row rows[kSets];
    static_for<0, (kDepth > 0 ? kDepth : 1)>([&](auto i) {
      constexpr int u = decltype(i)::value;
      load_row(word_of(u), rows[u]);                                       // u >= len: lane u of the chunk holds word 0
    });

    // one token: `cur` holds its row, the row of token t + kDepth goes into `tgt` (the set the previous token has just left)
    auto step = [&](const int t, const Row &cur, Row &tgt) __attribute__((always_inline)) {
      const int tl = t & 63;
      const int zold = __builtin_amdgcn_readlane(ch.zold, tl), ip = __builtin_amdgcn_readlane(ch.ip, tl);
      const double U = read_lane(ch.U, tl), own = COLLAPSED ? read_lane(ch.own, tl) : 0.0;
      if constexpr (kDepth > 0) load_row(word_of(tl + kDepth), tgt);       // in flight while this token and the next are computed; past the document's end: some valid row, never used
      bool refilled = false;                                               // kDepth == 0: `tgt` IS `cur`, refilled in mid-step (below)
      if (lane == 0) cnt[zold] -= 1;                                       // UPLDA:1494
      // One wave per workgroup: its LDS operations execute in program order, so all that is needed between lane 0's update
      // and the other lanes' reads is that the COMPILER keeps that order.  (__syncthreads() here also meant
      // s_waitcnt vmcnt(0): every token waited for the rows requested ahead.)
      __builtin_amdgcn_wave_barrier();

      // scores of this lane's two topics of block j (UPLDA:1509-1513 / MSLDA:196-203): the same two roundings as in Java.
      // They are recomputed where they are needed again (the deciding block; the replay): keeping K doubles per wave in
      // LDS for that cost half of the resident waves, and waves in flight are what this kernel's row traffic lives on
      // (K = 1024: 27.0 ms per z step with 7 waves per CU, 20.2 with 12).
      auto scores = [&](int j, double pa, double pb, double &qa, double &qb) __attribute__((always_inline)) {
        const int k = 2 * (lane + 64 * j);
        const int2 n = *reinterpret_cast<const int2 *>(&cnt[k]);
        const double2 al = *reinterpret_cast<const double2 *>(&alb[k]);
        if (COLLAPSED) { if (k == zold) pa = own; if (k + 1 == zold) pb = own; }
        qa = k < K ? ((double)n.x + al.x) * pa : 0.0;
        qb = k + 1 < K ? ((double)n.y + al.y) * pb : 0.0;
      };
      int new_topic = -1;
      // Given the block the draw falls into (js), the prefix before it and the inclusive scan of the block's lane sums:
      // the proposal and its proof (header).  a, b = this lane's two scores of that block.
      auto decide = [&](const int js, const double before, const double a, const double b, const double scan_ab, const double T,
                        const double delta) __attribute__((always_inline)) {
        (void)a;
        const double c_ab = before + scan_ab, c_a = c_ab - b;              // prefixes after this lane's first / second topic
        const double d_a = T - c_a, d_ab = T - c_ab;
        const unsigned long long m_a = __ballot(d_a < -delta), m_ab = __ballot(d_ab < -delta);
        if (m_a | m_ab) {
          const int l = __ffsll((long long)(m_a | m_ab)) - 1;
          const bool at_a = (m_a >> l) & 1ull;
          // the topic before the proposed one must be surely NOT yet past the sample
          const double d_prev = at_a ? (l == 0 ? T - before : read_lane(d_ab, max(l - 1, 0))) : read_lane(d_a, l);
          const int k = 128 * js + 2 * l + (at_a ? 0 : 1);
          const bool first = js == 0 && l == 0 && at_a;                    // topic 0: nothing before it (T > delta holds)
          if ((first || d_prev > delta) && k < K) new_topic = k;
        }
      };
      if constexpr (NB <= 2) {
        // One or two blocks: the inclusive scan of each block gives its prefixes AND (lane 63) its total -- NB wave scans do
        // the whole token where total + descent + deciding scan would take NB + 1 or NB + 2.
        double qa[NB], qb[NB], sc[NB], tot[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          scores(j, cur.a[j], cur.b[j], qa[j], qb[j]);
          sc[j] = wave_inclusive_scan(qa[j] + qb[j]);
          tot[j] = read_lane(sc[j], 63);
        }
        const double s_hat = NB == 2 ? tot[0] + tot[NB - 1] : tot[0];
        const double T = U * s_hat;
        const double delta = ((double)(K + 16) * s_hat) * 0x1p-51 * margin_scale;
        if (T > delta && s_hat < __builtin_huge_val()) {
          const bool second = NB == 2 && !(T - tot[0] <= delta);           // wave-uniform
          decide(second ? 1 : 0, second ? tot[0] : 0.0, second ? qa[NB - 1] : qa[0], second ? qb[NB - 1] : qb[0], second ? sc[NB - 1] : sc[0], T, delta);
        }
      } else {
      // The lane's share of every GROUP of kG adjacent blocks (at most 8 groups) and of every aligned range of groups (a
      // binary tree, lane-local adds only); wave reductions then run ONLY along the path of the block search: 2 (the total,
      // as left half + right half) + log2(groups) - 1 (the rest of the descent) + at most kG - 1 (inside the group, its
      // blocks' scores recomputed) instead of one per block -- sums in any association, which is all a proposal needs (header).
      double gv[kNG];                                                      // the lane's share of group g; ranges of groups are added up where the search asks for them
#pragma unroll
      for (int g = 0; g < kNG; ++g) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < kG; ++i) {
          double qa, qb;
          scores(g * kG + i, cur.a[g * kG + i], cur.b[g * kG + i], qa, qb);
          acc = i == 0 ? qa + qb : acc + (qa + qb);
        }
        gv[g] = acc;
        // wide rows: one group's LDS reads (counts, alpha) in flight at a time -- hoisted all at once they cost more
        // registers than two rows of K/64 doubles leave (NB = 32: 147 dwords per lane spilled to scratch, the z step 3x slower)
        if constexpr (NB >= 16) asm volatile("" ::: "memory");
      }
      auto range_sum = [&](auto lo_c, auto width_c) __attribute__((always_inline)) {   // lane-local, pairwise
        constexpr int lo = decltype(lo_c)::value, width = decltype(width_c)::value;
        double x[width];
#pragma unroll
        for (int i = 0; i < width; ++i) x[i] = gv[lo + i];
#pragma unroll
        for (int w = width; w > 1; w /= 2)
#pragma unroll
          for (int i = 0; i < w / 2; ++i) x[i] = x[2 * i] + x[2 * i + 1];
        return x[0];
      };
      // the total as left half + right half: the first level of the descent then has its `left` already
      const double root_left = wave_sum(range_sum(std::integral_constant<int, 0>{}, std::integral_constant<int, kNG / 2>{}));
      const double s_hat = root_left + wave_sum(range_sum(std::integral_constant<int, kNG / 2>{}, std::integral_constant<int, kNG / 2>{}));

      const double T = U * s_hat;
      const double delta = ((double)(K + 16) * s_hat) * 0x1p-51 * margin_scale;

      if (T > delta && s_hat < __builtin_huge_val()) {
        // descend: at a node covering groups [lo, lo + width) with C' = `before` at its start, the crossing is in the left
        // half iff the prefix at the left half's end is past T or too close to call; inside the group, block by block
        double before = 0.0;
        int js = 0;
        auto descend = [&](auto self, auto lo_c, auto width_c) __attribute__((always_inline)) -> void {
          constexpr int lo = decltype(lo_c)::value, width = decltype(width_c)::value;
          if constexpr (width == 1) {
            js = lo * kG + kG - 1;                                         // the group's last block unless an earlier one takes it
            bool found = false;
            static_for<0, kG - 1>([&](auto ic) {
              constexpr int j = lo * kG + decltype(ic)::value;
              if (!found) {                                                // wave-uniform
                double qa, qb;
                scores(j, cur.a[j], cur.b[j], qa, qb);
                const double blk = wave_sum(qa + qb);
                if (T - (before + blk) <= delta) { js = j; found = true; }
                else before += blk;
              }
            });
          } else {
            const double left = (lo == 0 && width == kNG) ? root_left
                                                          : wave_sum(range_sum(std::integral_constant<int, lo>{}, std::integral_constant<int, width / 2>{}));
            if (T - (before + left) <= delta) {
              self(self, std::integral_constant<int, lo>{}, std::integral_constant<int, width / 2>{});
            } else {
              before += left;
              self(self, std::integral_constant<int, lo + width / 2>{}, std::integral_constant<int, width / 2>{});
            }
          }
        };
        descend(descend, std::integral_constant<int, 0>{}, std::integral_constant<int, kNG>{});
        double pa = 0.0, pb = 0.0;                                         // this lane's row entries of block js (a static select: registers are not indexable)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          if (j == js) { pa = cur.a[j]; pb = cur.b[j]; }
        if constexpr (kDepth == 0) { load_row(word_of(tl + 1), tgt); refilled = true; }   // the row is dead: the next token's takes its registers
        double a, b;
        scores(js, pa, pb, a, b);
        decide(js, before, a, b, wave_inclusive_scan(a + b), T, delta);
      }
      }
      if (__builtin_expect(new_topic < 0, 0)) {
        // undecided (or Java would throw): replay the token as Java runs it, UPLDA:1509-1531 (negated walk in counting form,
        // see ggs_z_sliced.hpp): the scores in k order -- block by block, lane by lane, first then second topic -- read
        // out of the lanes that hold them; every lane runs the same two chains.  (As a function of its own, reading the
        // row from memory again so that the main path need not keep it for this: measured slower, 32.7 ms against 20.2
        // at K = 1024 -- the compiler then holds the kernel at 128 registers and spills scalars in the token loop.)
        if constexpr (kDepth == 0) {
          if (refilled) load_row(word_of(tl), tgt);                        // the token's own row once more (this path is taken by one token in 10^9)
          refilled = false;
        }
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          double qa, qb;
          scores(j, cur.a[j], cur.b[j], qa, qb);
#pragma unroll 8
          for (int l = 0; l < 64; ++l) { sum += read_lane(qa, l); sum += read_lane(qb, l); }   // topics past K score 0: + 0.0
        }
        double tt = 0.0 - U * sum;
        int newc = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          double qa, qb;
          scores(j, cur.a[j], cur.b[j], qa, qb);
#pragma unroll 8
          for (int l = 0; l < 64; ++l) {
            const bool ina = 128 * j + 2 * l < K, inb = 128 * j + 2 * l + 1 < K;   // the walk's counting form stops at K
            newc += (ina && hi32(tt) < 0) ? 1 : 0;
            tt += read_lane(qa, l);
            newc += (inb && hi32(tt) < 0) ? 1 : 0;
            tt += read_lane(qb, l);
          }
        }
        new_topic = newc - 1;
        if (new_topic < 0 || hi32(tt) < 0) {                               // UPLDA:1529-1531
          if (lane == 0) atomicOr(p.status, ST_INVALID_TOPIC);
          new_topic = new_topic < 0 ? 0 : K - 1;
        }
      }
      if constexpr (kDepth == 0) { if (!refilled) load_row(word_of(tl + 1), tgt); }
      __builtin_amdgcn_wave_barrier();                                     // every lane's reads of the counts are issued before the update
      if (lane == 0) {
        cnt[new_topic] += 1;                                               // UPLDA:1535
        p.z[beg + t] = new_topic;
        p.zw[ip] = new_topic;
      }
      if (tl == 63) {                                                      // the next chunk becomes the current one, the one after it is requested
        ch = chn;
        if (t + 1 + 64 < len) chn = load_chunk(t + 1 + 64);
      }
    };
    for (int t0 = 0; t0 < len; t0 += kSets)
      static_for<0, kSets>([&](auto i) {
        constexpr int u = decltype(i)::value;
        if (t0 + u < len) step(t0 + u, rows[u], rows[(u + kDepth) % kSets]);
      });
  }
}

}  // namespace ggs
