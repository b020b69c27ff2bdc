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
// The next token's row is loaded into a second register set while the current token is computed.
// LDS per wave: counts int32 [128 NB] and alpha fp64 [128 NB]: 12 KiB at K = 1024.
#pragma once
#include "ggs_z_pcgs.hpp"

namespace ggs {

template <int NB, bool COLLAPSED>
__global__ __launch_bounds__(64) void pcgs_wave_kernel(PcgsParams p, double margin_scale) {
  constexpr int KT = NB * 128;
  extern __shared__ __align__(16) unsigned char smem[];
  int32_t *cnt = reinterpret_cast<int32_t *>(smem);                        // [KT]
  double *alb = reinterpret_cast<double *>(smem + (size_t)KT * 4);         // [KT] alpha, zero padded
  const int lane = threadIdx.x, K = p.K;
  const int units = p.Kp / 2;                                              // 16-byte units per phiT row (Kp is even; a padding column holds 0)

  for (int k = lane; k < KT; k += 64) alb[k] = k < K ? p.alpha[k] : 0.0;

  struct Row { double a[NB], b[NB]; };
  auto load_row = [&](int w, Row &r) {
    const double2 *row = reinterpret_cast<const double2 *>(p.phiT + (size_t)w * p.Kp);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int u = lane + 64 * j;
      const double2 v = u < units ? row[u] : double2{0.0, 0.0};
      r.a[j] = v.x; r.b[j] = v.y;
    }
  };

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
    // the loop itself waits for no memory but the phiT rows, and those are a token ahead.
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
    Chunk ch = load_chunk(0), chn = ch;
    Row cur, nxt;
    load_row(__builtin_amdgcn_readlane(ch.w, 0), cur);
    for (int t = 0; t < len; ++t) {
      const int tl = t & 63;
      if (tl == 0 && t + 64 < len) chn = load_chunk(t + 64);               // the chunk after this one: in flight for 64 tokens
      const int w = __builtin_amdgcn_readlane(ch.w, tl), zold = __builtin_amdgcn_readlane(ch.zold, tl), ip = __builtin_amdgcn_readlane(ch.ip, tl);
      const double U = read_lane(ch.U, tl), own = COLLAPSED ? read_lane(ch.own, tl) : 0.0;
      (void)w;
      if (t + 1 < len) load_row(tl == 63 ? __builtin_amdgcn_readlane(chn.w, 0) : __builtin_amdgcn_readlane(ch.w, tl + 1), nxt);   // in flight while this token is computed
      if (lane == 0) cnt[zold] -= 1;                                       // UPLDA:1494
      __syncthreads();

      // scores of this lane's two topics of block j (UPLDA:1509-1513 / MSLDA:196-203): the same two roundings as in Java.
      // They are recomputed where they are needed again (the deciding block; the replay): keeping K doubles per wave in
      // LDS for that cost half of the resident waves, and waves in flight are what this kernel's row traffic lives on
      // (K = 1024: 27.0 ms per z step with 7 waves per CU, 20.2 with 12).
      auto scores = [&](int j, double pa, double pb, double &qa, double &qb) {
        const int k = 2 * (lane + 64 * j);
        const int2 n = *reinterpret_cast<const int2 *>(&cnt[k]);
        const double2 al = *reinterpret_cast<const double2 *>(&alb[k]);
        if (COLLAPSED) { if (k == zold) pa = own; if (k + 1 == zold) pb = own; }
        qa = k < K ? ((double)n.x + al.x) * pa : 0.0;
        qb = k + 1 < K ? ((double)n.y + al.y) * pb : 0.0;
      };
      // block sums in any order: a proposal only
      double tot[NB];                                                      // wave-uniform (scalar registers)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        double qa, qb;
        scores(j, cur.a[j], cur.b[j], qa, qb);
        tot[j] = read_lane(wave_inclusive_scan(qa + qb), 63);
      }
      double s_hat = 0.0;
#pragma unroll
      for (int j = 0; j < NB; ++j) s_hat += tot[j];

      const double T = U * s_hat;
      const double delta = ((double)(K + 16) * s_hat) * 0x1p-51 * margin_scale;

      int new_topic = -1;
      if (T > delta && s_hat < __builtin_huge_val()) {
        double before = 0.0;                                               // C' at the start of the block
        int js = -1;
#pragma unroll
        for (int j = 0; j < NB; ++j)
          if (js < 0) {
            if (T - (before + tot[j]) <= delta) js = j;                    // the crossing is in this block, or too close to its end to call
            else before += tot[j];
          }
        if (js >= 0) {
          double pa = 0.0, pb = 0.0;                                       // this lane's row entries of block js (a static select: registers are not indexable)
#pragma unroll
          for (int j = 0; j < NB; ++j)
            if (j == js) { pa = cur.a[j]; pb = cur.b[j]; }
          double a, b;
          scores(js, pa, pb, a, b);
          const double c_ab = before + wave_inclusive_scan(a + b), c_a = c_ab - b;   // prefixes after this lane's first / second topic
          const double d_a = T - c_a, d_ab = T - c_ab;
          const unsigned long long m_a = __ballot(d_a < -delta), m_ab = __ballot(d_ab < -delta);
          if (m_a | m_ab) {
            const int l = __ffsll((long long)(m_a | m_ab)) - 1;
            const bool at_a = (m_a >> l) & 1ull;
            // the topic before the proposed one must be surely NOT yet past the sample
            const double d_prev = at_a ? (l == 0 ? T - before : read_lane(d_ab, max(l - 1, 0))) : read_lane(d_a, l);
            const int k = 128 * js + 2 * l + (at_a ? 0 : 1);
            const bool first = js == 0 && l == 0 && at_a;                  // topic 0: nothing before it (T > delta holds)
            if ((first || d_prev > delta) && k < K) new_topic = k;
          }
        }
      }
      if (new_topic < 0) {
        // undecided (or Java would throw): replay the token as Java runs it, UPLDA:1509-1531 (negated walk in counting form,
        // see ggs_z_sliced.hpp): the scores in k order -- block by block, lane by lane, first then second topic -- read
        // out of the lanes that hold them; every lane runs the same two chains.  (As a function of its own, reading the
        // row from memory again so that the main path need not keep it for this: measured slower, 32.7 ms against 20.2
        // at K = 1024 -- the compiler then holds the kernel at 128 registers and spills scalars in the token loop.)
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
      __syncthreads();                                                     // every lane has read the counts
      if (lane == 0) {
        cnt[new_topic] += 1;                                               // UPLDA:1535
        p.z[beg + t] = new_topic;
        p.zw[ip] = new_topic;
      }
      cur = nxt;
      if (tl == 63) ch = chn;
    }
  }
}

}  // namespace ggs
