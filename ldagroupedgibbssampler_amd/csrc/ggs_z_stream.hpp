// ggs_z_stream.hpp -- K3, the token loop (GGS:79-130), for any number of topics: the wide-row
// variant (K > kSlicedMaxTopics, e.g. BASELINE config 3 with K = 1024), where a token's K scores
// fit neither the register file nor LDS.
//
// Same machinery as ggs_z_sliced.hpp -- persistent single-wave workgroups striding the chunk
// table, lane t owns token t of a <= 64-token chunk of ONE document, phiT rows streamed by
// LDS-DMA through a 2-slot ring of 16-topic slices (8 KiB), one slice ahead (measured: a small ring
// that lets 6-8 waves share a CU beats a deep one with 4 -- two waves per SIMD issue 1.7x what one does), the same
// per-row rotation for conflict-free reads -- but the row is streamed TWICE per chunk:
//
//   pass 1 (slices 0..NS-1):  sum += theta[k]*phi[k][w_t], k ascending          (GGS:96-101)
//   U from Philox, sample = U*sum                                               (GGS:107-108)
//   pass 2 (slices 0..NS-1):  cnt += (sample > 0); sample -= theta[k]*phi[k][w_t] (GGS:109-113)
//
// The product theta[k]*phi[k][w] is the same single IEEE multiplication in both passes, so
// pass 2 subtracts exactly the scores pass 1 summed (what Java keeps in topicTermScores[]).
// The 2*NS slices of a chunk and the first slices of the next chunk form one regular DMA
// stream (no early exit: the latest lane of a 64-token chunk almost always walks to the end),
// so the vmcnt arithmetic stays exact.  Second-pass reads are mostly L2 / Infinity-Cache hits.
#pragma once
#include "ggs_z_sliced.hpp"

namespace ggs {

#ifndef GGS_STREAM_RING
#define GGS_STREAM_RING 2
#endif
constexpr int kStreamRingSlots = GGS_STREAM_RING;

__global__ __launch_bounds__(64) void z_stream_kernel(ZParams p) {
  constexpr int kAhead = kStreamRingSlots - 1;
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int K = p.K, Kp = p.Kp;
  const int NS = (K + kSliceTopics - 1) / kSliceTopics;            // slices per pass (host guarantees NS >= 3)
  const int KT = NS * kSliceTopics;                                // theta row in LDS, zero padded
  double *thb = reinterpret_cast<double *>(smem + kStreamRingSlots * kSliceBytes);
  const unsigned char *phib = reinterpret_cast<const unsigned char *>(p.phiT);
  const size_t rowbytes = (size_t)Kp * 8;
  const const_i64_t *cstart = (const const_i64_t *)p.chunk_start;
  const const_i32_t *clen = (const const_i32_t *)p.chunk_len;
  const const_i32_t *cdoc = (const const_i32_t *)p.chunk_doc;
  const int64_t C = p.num_chunks;

  const int lrow = lane >> 3, lslot = lane & 7;
  const unsigned char *my_row = smem + lane * 128;
  const int rot = lane >> 1;

  auto row_addresses = [&](const int w, const unsigned char *(&ra)[8]) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int row = 8 * m + lrow;
      const int wm = __shfl(w, row);                               // 0 for rows past the chunk
      ra[m] = phib + (size_t)wm * rowbytes + (size_t)(((lslot - (row >> 1)) & 7) << 4);
    }
  };
  auto issue_slice = [&](const int s, const int slot, const unsigned char *const (&ra)[8]) {
    const size_t off = (size_t)s * 128;
#pragma unroll
    for (int m = 0; m < 8; ++m)
      __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(ra[m] + off), (lds_void_t *)(smem + slot * kSliceBytes + m * 1024), 16, 0, 0);
  };

  // a contiguous range of chunks per wave (every chunk costs the same whatever its length), so
  // the chunks of one document follow each other and its theta row is staged once
  const int64_t cend = C * (int64_t)(blockIdx.x + 1) / (int64_t)gridDim.x;
  int64_t c = C * (int64_t)blockIdx.x / (int64_t)gridDim.x;
  if (c >= cend) return;
  int doc_in_lds = -1;
  int64_t start0 = cstart[c], start1 = 0;
  int len0 = clen[c], len1 = 0, doc0 = cdoc[c], doc1 = 0;
  if (c + 1 < cend) { start1 = cstart[c + 1]; len1 = clen[c + 1]; doc1 = cdoc[c + 1]; }
  int w0 = (lane < len0) ? p.tok[start0 + lane] : 0;
  int ip0 = (lane < len0) ? p.inv_perm[start0 + lane] : 0;
  const unsigned char *ra[8], *ran[8];
  row_addresses(w0, ra);
  int g = 0;                                                       // ring slot of this chunk's first slice
#pragma unroll
  for (int s = 0; s < kAhead; ++s) issue_slice(s, (g + s) % kStreamRingSlots, ra);

  for (;;) {
    const bool has1 = c + 1 < cend;
    // This chunk's theta row -> LDS, and the next chunk's word ids / addresses.  The loads are
    // waited for right here (they drain the DMA queue once per chunk: a small bubble against
    // 2*NS slices of work).
    if (doc0 != doc_in_lds) {                                      // wave-uniform: consecutive chunks share the document
      const double *thg = p.theta + (size_t)doc0 * K;
      for (int base = 0; base < KT; base += 1024) {                // 16 loads in flight per lane, one round trip per 1024 topics
        double t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = base + i * 64 + lane;
          t[i] = thg[k < K ? k : K - 1];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = base + i * 64 + lane;
          if (k < KT) thb[k] = (k < K) ? t[i] : 0.0;
        }
      }
      doc_in_lds = doc0;
    }
    int w1 = 0, ip1 = 0;
    int64_t start2 = 0;
    int len2 = 0, doc2 = 0;
    if (has1) {
      w1 = (lane < len1) ? p.tok[start1 + lane] : 0;
      ip1 = (lane < len1) ? p.inv_perm[start1 + lane] : 0;
      if (c + 2 < cend) { start2 = cstart[c + 2]; len2 = clen[c + 2]; doc2 = cdoc[c + 2]; }
      row_addresses(w1, ran);
    }
    asm volatile("" ::: "memory");

    double sum = 0.0, sample = 0.0, U = 0.0;
    int cnt = 0;
    for (int j = 0; j < 2 * NS; ++j) {
      const int s = j < NS ? j : j - NS;
      const int cur = (g + j) % kStreamRingSlots;
      const int nxt = (g + j + kAhead) % kStreamRingSlots;
      const int ja = j + kAhead;
      if (ja < 2 * NS) issue_slice(ja < NS ? ja : ja - NS, nxt, ra);
      else if (has1) issue_slice(ja - 2 * NS, nxt, ran);
      // all but the youngest 8*kAhead DMAs done => slice j has landed; at the tail of the last
      // chunk fewer slices follow it
      if (has1 || ja < 2 * NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * kAhead) : "memory");
      else {
        const int rem = 2 * NS - 1 - j;                            // slices still behind this one: fewer than kAhead
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (lane < len0) {
        const unsigned char *rb = my_row + cur * kSliceBytes;
        const unsigned char *tb = reinterpret_cast<const unsigned char *>(thb) + s * kSliceTopics * 8;
        D2 ph[kSliceUnits], th[kSliceUnits];
#pragma unroll
        for (int u = 0; u < kSliceUnits; ++u) {
          ph[u] = lds_d2(rb + (((u + rot) & 7) << 4));
          th[u] = lds_d2(tb + u * 16);
        }
        if (j < NS) {
#pragma unroll
          for (int u = 0; u < kSliceUnits; ++u) {
            sum += th[u].a * ph[u].a;
            sum += th[u].b * ph[u].b;
          }
          if (j == NS - 1) {
            const uint64_t gtok = (uint64_t)(p.tok_base + start0 + lane);
            const U4 o = philox4x32_10((uint32_t)gtok, (uint32_t)(gtok >> 32), (uint32_t)GGS_PURPOSE_Z << 24, p.iteration,
                                       (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
            U = u53(o.x, o.y);
            sample = U * sum;
          }
        } else {
#pragma unroll
          for (int u = 0; u < kSliceUnits; ++u) {
            cnt += (sample > 0.0); sample -= th[u].a * ph[u].a;
            cnt += (sample > 0.0); sample -= th[u].b * ph[u].b;
          }
        }
      }
      asm volatile("" ::: "memory");                               // every read of this ring slot is issued before it is refilled
    }
    g = (g + 2 * NS) % kStreamRingSlots;

    if (lane < len0) {
      int new_topic = cnt - 1;
      if (new_topic < 0 || sample > 0.0) {                         // GGS:116-118 (and the index past K Java would throw on)
        atomicOr(p.status, ST_INVALID_TOPIC);
        new_topic = new_topic < 0 ? 0 : K - 1;
      }
      p.z[start0 + lane] = new_topic;
      p.zw[ip0] = new_topic;
    }
    if (!has1) break;
    c += 1;
    start0 = start1; len0 = len1; doc0 = doc1; w0 = w1; ip0 = ip1;
    start1 = start2; len1 = len2; doc1 = doc2;
#pragma unroll
    for (int m = 0; m < 8; ++m) ra[m] = ran[m];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// z_stream1_kernel -- the same token loop with every phiT row streamed ONCE.
//
// The second pass of z_stream_kernel exists to replay the walk of GGS:108-113, sample -= score[k] until sample <= 0,
// a rounding chain of its own.  But where that chain stops can almost always be decided from the chain the first pass
// computes anyway, the running sum s_j = fl(s_{j-1} + p_j) of the scores p_j = theta[j]*phi[j][w] (GGS:96-101):
//   * every step of either chain rounds a value of magnitude <= sum, so each is off by at most ulp(sum)/2, and after
//     j steps  |t_j - (t_0 - s_j)| <= j * 2^-52 * sum,  with t_0 = fl(U*sum) and t_j the Java walk's `sample` after
//     subtracting p_0..p_{j-1};
//   * the walk returns the first topic r with t_{r+1} <= 0.  With d_j = t_0 - s_{j+1} and the margin
//     delta = K * 2^-51 * sum (twice the bound):  d_j < -delta proves t_{j+1} < 0,  d_j > delta proves t_{j+1} > 0.
//     Both chains are monotone, so if r is the first index with d_r < -delta and also d_{r-1} > delta (t_0 > delta for
//     r = 0), the walk stops at r -- whatever its own roundings were.
// A token is undecided only if U*sum falls within delta of one of K partial sums: probability ~ K^2 * 2^-50 (1e-9 at
// K = 1024).  Such a token -- and any whose walk would not end inside [0, K) -- is replayed exactly as Java does it,
// element by element from the raw row (the chunk's lanes wait for it; GGS_DEBUG_MARGIN scales delta up so that the
// tests drive thousands of tokens through the replay).
//
// To find r without keeping K partial sums per token, pass 1 stores the chain at every 64th topic (a checkpoint per
// group of 4 slices, [group][lane] in LDS); the first group whose checkpoint satisfies d < -delta contains r, and only
// that group's 512 bytes of the row are streamed again (each lane its own group: the group offset rides on the per-lane
// DMA source address), continuing the chain from the previous checkpoint -- the same additions in the same order, hence
// the same s_j.  Row traffic per token: 8*K + 512 bytes instead of 16*K.
// LDS per wave: the 2-slot ring, the theta row zero-padded to whole groups, the checkpoints.
// ------------------------------------------------------------------------------------------------------------------
// Slices per checkpoint group G (a template parameter): the group is what is streamed again, so small groups cost less
// refinement traffic and more checkpoints.  With the checkpoints in registers (at most kRegCheckpoints groups) the host
// takes the smallest G that fits: 1 up to K = 256 (16 B... one 128-byte slice per token again), 2 up to 512, 4 up to 1024.
constexpr int kMaxGroupSlices = 4;
#ifndef GGS_STREAM1_RING
#define GGS_STREAM1_RING 2
#endif
constexpr int kStream1RingSlots = GGS_STREAM1_RING;                // measured at K = 1024 with 4 waves per CU: 2 slots 12.7 ms, 3 slots 12.9; waves per CU matter more
constexpr int kRegCheckpoints = 16;                                // REGCK: checkpoints in registers for up to 16 groups (K <= 1024): 8 KiB less LDS per wave at K = 1024

template <bool REGCK, int kGroupSlices>
__global__ __launch_bounds__(64) void z_stream1_kernel(ZParams p) {
  constexpr int R = kStream1RingSlots, kAhead = R - 1;
  static_assert(kGroupSlices == 1 || kGroupSlices == 2 || kGroupSlices == 4, "the checkpoint test masks with G - 1");
  static_assert(kAhead >= 1 && kAhead <= 3, "wait_younger covers up to 3 slices in flight behind the current one");
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int K = p.K, Kp = p.Kp;
  const int NS = (K + kSliceTopics - 1) / kSliceTopics;            // slices of pass 1
  const int NG = (NS + kGroupSlices - 1) / kGroupSlices;           // checkpoint groups
  const int KTG = NG * kGroupSlices * kSliceTopics;                // theta row in LDS, zero padded to whole groups
  double *thb = reinterpret_cast<double *>(smem + R * kSliceBytes);   // theta row(s): one, or two with p.two_rows (a chunk across a document boundary)
  double *ckb = thb + (p.two_rows ? 2 : 1) * KTG;                  // [NG][64] (!REGCK)
  double ck[REGCK ? kRegCheckpoints : 1];                           // REGCK: the lane's checkpoints
  const unsigned char *phib = reinterpret_cast<const unsigned char *>(p.phiT);
  const size_t rowbytes = (size_t)Kp * 8;
  const const_i64_t *cstart = (const const_i64_t *)p.chunk_start;
  const const_i32_t *clen = (const const_i32_t *)p.chunk_len;
  const const_i32_t *cdoc = (const const_i32_t *)p.chunk_doc;
  const const_i32_t *cdoc1 = (const const_i32_t *)p.chunk_doc1;
  const int64_t C = p.num_chunks;

  const int lrow = lane >> 3, lslot = lane & 7;
  const unsigned char *my_row = smem + lane * 128;
  const int rot = lane >> 1;

  auto row_addresses = [&](const int w, const unsigned char *(&ra)[8]) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int row = 8 * m + lrow;
      const int wm = __shfl(w, row);                               // 0 for rows past the chunk
      ra[m] = phib + (size_t)wm * rowbytes + (size_t)(((lslot - (row >> 1)) & 7) << 4);
    }
  };
  auto issue_slice = [&](const int s, const int slot, const unsigned char *const (&ra)[8]) {
    const size_t off = (size_t)s * 128;
#pragma unroll
    for (int m = 0; m < 8; ++m)
      __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(ra[m] + off), (lds_void_t *)(smem + slot * kSliceBytes + m * 1024), 16, 0, 0);
  };

  const int64_t cend = C * (int64_t)(blockIdx.x + 1) / (int64_t)gridDim.x;
  int64_t c = C * (int64_t)blockIdx.x / (int64_t)gridDim.x;
  if (c >= cend) return;
  int rd0 = -1, rd1 = -1;                                          // the documents whose theta rows sit in LDS rows 0 and 1
  int64_t start0 = cstart[c], start1 = 0;
  int len0 = clen[c], len1 = 0, doc0 = cdoc[c], doc1 = 0;          // len: tokens | tokens of the first document << 8 (0: all of them)
  int docb0 = p.two_rows ? cdoc1[c] : doc0, docb1 = 0;             // the chunk's second document (= the first if it has one only)
  if (c + 1 < cend) { start1 = cstart[c + 1]; len1 = clen[c + 1]; doc1 = cdoc[c + 1]; docb1 = p.two_rows ? cdoc1[c + 1] : doc1; }
  const int split_shift = 8;
  auto tokens_of = [](const int v) { return v & 0xff; };
  auto split_of = [&](const int v) { const int sp = v >> split_shift; return sp ? sp : (v & 0xff); };
  int w0 = (lane < tokens_of(len0)) ? p.tok[start0 + lane] : 0;
  int ip0 = (lane < tokens_of(len0)) ? p.inv_perm[start0 + lane] : 0;
  const unsigned char *ra[8], *ran[8], *rr[8];
  row_addresses(w0, ra);
  // all but the youngest n slices (8 DMA instructions each) have landed
  auto wait_younger = [](const int n) {
    if (n >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  int g = 0;                                                       // ring slot of this chunk's first slice
#pragma unroll
  for (int s = 0; s < kAhead; ++s) issue_slice(s, (g + s) % R, ra);

  for (;;) {
    const bool has1 = c + 1 < cend;
    // theta rows: the chunk's first document goes where it already is, or into row 0; the second one (if any) into the
    // other row.  Wave-uniform: consecutive chunks share documents, so most chunks stage one row or none.
    const int n0 = tokens_of(len0), sp0 = split_of(len0);
    auto stage_row = [&](const int doc, const int rowi) {
      const double *thg = p.theta + (size_t)doc * K;
      double *dst = thb + rowi * KTG;
      for (int base = 0; base < KTG; base += 1024) {               // 16 loads in flight per lane, one round trip per 1024 topics
        double t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = base + i * 64 + lane;
          t[i] = thg[k < K ? k : K - 1];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = base + i * 64 + lane;
          if (k < KTG) dst[k] = (k < K) ? t[i] : 0.0;
        }
      }
      if (rowi) rd1 = doc; else rd0 = doc;
    };
    int rowa = 0;
    if (p.two_rows && doc0 == rd1) rowa = 1;
    else if (doc0 != rd0) stage_row(doc0, 0);
    if (sp0 < n0 && docb0 != (rowa ? rd0 : rd1)) stage_row(docb0, rowa ^ 1);
    const double *thl = thb + ((lane < sp0) ? rowa : (rowa ^ 1)) * KTG;   // this lane's theta row
    int w1 = 0, ip1 = 0;
    int64_t start2 = 0;
    int len2 = 0, doc2 = 0;
    int docb2 = 0;
    if (has1) {
      w1 = (lane < tokens_of(len1)) ? p.tok[start1 + lane] : 0;
      ip1 = (lane < tokens_of(len1)) ? p.inv_perm[start1 + lane] : 0;
      if (c + 2 < cend) { start2 = cstart[c + 2]; len2 = clen[c + 2]; doc2 = cdoc[c + 2]; docb2 = p.two_rows ? cdoc1[c + 2] : doc2; }
      row_addresses(w1, ran);
    }
    asm volatile("" ::: "memory");

    // ---- pass 1: the sum chain (GGS:96-101), checkpointed every kGroupSlices slices
    double sum = 0.0;
    auto pass1_slice = [&](const int j) {
      const int cur = (g + j) % R;
      if (j + kAhead < NS) issue_slice(j + kAhead, (g + j + kAhead) % R, ra);
      wait_younger(min(kAhead, NS - 1 - j));                       // the pipeline drains at the end: which group to stream next is not known yet
      if (lane < n0) {
        const unsigned char *rb = my_row + cur * kSliceBytes;
        const unsigned char *tb = reinterpret_cast<const unsigned char *>(thl) + j * kSliceTopics * 8;
        D2 ph[kSliceUnits], th[kSliceUnits];
#pragma unroll
        for (int u = 0; u < kSliceUnits; ++u) {
          ph[u] = lds_d2(rb + (((u + rot) & 7) << 4));
          th[u] = lds_d2(tb + u * 16);
        }
#pragma unroll
        for (int u = 0; u < kSliceUnits; ++u) {
          sum += th[u].a * ph[u].a;
          sum += th[u].b * ph[u].b;
        }
      }
      asm volatile("" ::: "memory");                               // every read of this ring slot is issued before it is refilled
    };
    if constexpr (REGCK) {
#pragma unroll
      for (int q = 0; q < kRegCheckpoints; ++q)
        if (q < NG) {                                              // wave-uniform
#pragma unroll 1
          for (int j = q * kGroupSlices; j < min((q + 1) * kGroupSlices, NS); ++j) pass1_slice(j);
          ck[q] = sum;
        }
    } else {
      for (int j = 0; j < NS; ++j) {
        pass1_slice(j);
        if (lane < n0 && ((j & (kGroupSlices - 1)) == kGroupSlices - 1 || j == NS - 1)) ckb[(j / kGroupSlices) * 64 + lane] = sum;
      }
    }

    // ---- the threshold and the group it falls into
    double t0 = 0.0, delta = 0.0, s = 0.0;
    int gsel = 0;
    bool undecided = true;
    if (lane < n0) {
      const uint64_t gtok = (uint64_t)(p.tok_base + start0 + lane);
      const U4 o = philox4x32_10((uint32_t)gtok, (uint32_t)(gtok >> 32), (uint32_t)GGS_PURPOSE_Z << 24, p.iteration,
                                 (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
      t0 = u53(o.x, o.y) * sum;                                    // GGS:107-108
      delta = (sum * (double)K) * 0x1p-51 * p.margin_scale;
      int gi = NG;
      if constexpr (REGCK) {
#pragma unroll
        for (int q = kRegCheckpoints - 1; q >= 0; --q)
          if (q < NG && t0 - ck[q] < -delta) gi = q;               // d is monotone: ends at the first group that satisfies it
      } else {
        for (int q = NG - 1; q >= 0; --q)
          if (t0 - ckb[q * 64 + lane] < -delta) gi = q;
      }
      undecided = gi == NG;                                        // the walk would not end inside the row (or too close to call)
      gsel = undecided ? 0 : gi;
      if constexpr (REGCK) {                                       // the chain as it stood when pass 1 entered the group
#pragma unroll
        for (int q = 0; q < kRegCheckpoints - 1; ++q)
          if (gsel == q + 1) s = ck[q];
      } else {
        s = gsel > 0 ? ckb[(gsel - 1) * 64 + lane] : 0.0;
      }
    }
    // ---- refinement: the 4 slices of each lane's own group
#pragma unroll
    for (int m = 0; m < 8; ++m) rr[m] = ra[m] + (size_t)__shfl(gsel, 8 * m + lrow) * (kGroupSlices * 128);
    const int gr = (g + NS) % R;                                   // ring slot of the first refinement slice
#pragma unroll
    for (int i = 0; i < kAhead; ++i) {
      if (i < kGroupSlices) issue_slice(i, (gr + i) % R, rr);
      else if (has1) issue_slice(i - kGroupSlices, (gr + i) % R, ran);
    }
    int found = -1;
    double dprev = 0.0;
#pragma unroll 1
    for (int i = 0; i < kGroupSlices; ++i) {
      const int cur = (gr + i) % R;
      const int ia = i + kAhead;
      if (ia < kGroupSlices) issue_slice(ia, (gr + ia) % R, rr);
      else if (has1) issue_slice(ia - kGroupSlices, (gr + ia) % R, ran);   // the next chunk's first slices
      wait_younger(has1 ? kAhead : min(kAhead, kGroupSlices - 1 - i));
      if (lane < n0) {
        const unsigned char *rb = my_row + cur * kSliceBytes;
        const unsigned char *tb = reinterpret_cast<const unsigned char *>(thl) + ((size_t)gsel * kGroupSlices + i) * kSliceTopics * 8;
        D2 ph[kSliceUnits], th[kSliceUnits];
#pragma unroll
        for (int u = 0; u < kSliceUnits; ++u) {
          ph[u] = lds_d2(rb + (((u + rot) & 7) << 4));
          th[u] = lds_d2(tb + u * 16);
        }
#pragma unroll
        for (int u = 0; u < kSliceUnits; ++u) {
          const int kk = (gsel * kGroupSlices + i) * kSliceTopics + 2 * u;
          double sp = s;
          s += th[u].a * ph[u].a;
          if (found < 0 && t0 - s < -delta) { found = kk; dprev = t0 - sp; }
          sp = s;
          s += th[u].b * ph[u].b;
          if (found < 0 && t0 - s < -delta) { found = kk + 1; dprev = t0 - sp; }
        }
      }
      asm volatile("" ::: "memory");
    }
    g = (g + NS + kGroupSlices) % R;

    if (lane < n0) {
      int new_topic = found;
      if (undecided || found < 0 || found >= K || !(dprev > delta)) {
        // the exact replay: GGS:108-113 element by element from the raw row (rare; see the header)
        const double *row = p.phiT + (size_t)w0 * Kp;
        double sample = t0;
        new_topic = -1;
        while (sample > 0.0) {
          ++new_topic;
          if (new_topic >= K) break;
          sample -= thl[new_topic] * row[new_topic];
        }
        if (new_topic < 0 || new_topic >= K) {                     // GGS:116-118 (and the index past K Java would throw on)
          atomicOr(p.status, ST_INVALID_TOPIC);
          new_topic = new_topic < 0 ? 0 : K - 1;
        }
      }
      p.z[start0 + lane] = new_topic;
      p.zw[ip0] = new_topic;
    }
    if (!has1) break;
    c += 1;
    start0 = start1; len0 = len1; doc0 = doc1; docb0 = docb1; w0 = w1; ip0 = ip1;
    start1 = start2; len1 = len2; doc1 = doc2; docb1 = docb2;
#pragma unroll
    for (int m = 0; m < 8; ++m) ra[m] = ran[m];
  }
}

}  // namespace ggs
