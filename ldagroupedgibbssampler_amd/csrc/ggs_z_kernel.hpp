// ggs_z_kernel.hpp -- K3: the token loop of the Grouped Gibbs sampler (GGS:79-130).
//
// Persistent waves: the grid is (CUs x waves that fit in LDS) single-wave workgroups, each
// walking the chunk table with a grid stride.  One chunk = up to `tile_tokens` (<= 64)
// consecutive tokens of ONE document; lane t owns token t.  Per chunk:
//   stage   one LDS-DMA wave-instruction per token row (global_load_lds_dwordx4, lanes < row
//           units active): phiT row w_t, Kp*8 contiguous bytes, HBM -> LDS row t; the
//           document's theta row sits in LDS behind the tile
//   pass 1  lane t: sum = sum_k theta[k]*phi[k][w_t], k ascending (GGS:96-101)
//   draw    U from Philox (GGS:107), sample = U*sum
//   pass 2  lane t: the "while (sample > 0) sample -= score[++k]" walk (GGS:108-113)
//   store   z (the count updates of GGS:93,129 are rebuilt by count_sorted_kernel)
// Both passes are the Java code's sequential fp64 chains, one per lane.  theta[k] is a
// broadcast LDS read (every lane reads the same 16 bytes), phi[k][w_t] a per-lane LDS read of
// row t; the reads of the next 8 topics are in flight while the chain of the current 8 runs.
// The next chunk's word ids and theta row are requested BEFORE this chunk's DMA is issued, so
// their latency hides behind the tile's; chunk descriptors are fetched two chunks ahead.
// LDS row pitch = pitch16*16 bytes with pitch16 odd, so the 16-byte per-lane reads of the
// passes (lane t reads row t) are bank-conflict free.
#pragma once
#include "ggs_kernels.hpp"

namespace ggs {

// Topic k of word w -> its cell in the slice-major send buffer of the count reduce-scatter, [nranks][V][Ksm] (rank r owns
// the topics of slice r: sizes size + 1 for the first `rem` ranks, size for the rest -- EvenSplitTopicBatchBuilder.java:28-39).
// Arithmetic only (two multiplies by a reciprocal): the z kernels call it once per token, and a table lookup there would be
// a dependent load in front of the atomic.
struct SliceMap {
  int32_t rem, size, ksm;      // ksm = the widest slice = the row pitch of a rank's part
  uint32_t m_size, m_size1;    // udiv_magic(size), udiv_magic(size + 1)
  int64_t rank_stride;         // V * ksm
};
__device__ __forceinline__ int64_t slice_cell(const SliceMap &m, int k, int w) {
  const int cut = m.rem * (m.size + 1);
  int r, c;
  if (k < cut) { r = udiv_small(k, m.m_size1); c = k - r * (m.size + 1); }
  else { const int q = udiv_small(k - cut, m.m_size); r = m.rem + q; c = k - cut - q * m.size; }
  return (int64_t)r * m.rank_stride + (int64_t)w * m.ksm + c;
}

__device__ __forceinline__ void slice_rank_col(const SliceMap &m, int k, int &r, int &c) {
  const int cut = m.rem * (m.size + 1);
  if (k < cut) { r = udiv_small(k, m.m_size1); c = k - r * (m.size + 1); }
  else { const int q = udiv_small(k - cut, m.m_size); r = m.rem + q; c = k - cut - q * m.size; }
}

struct ZParams {
  const int32_t *tok;
  const int32_t *inv_perm;     // position of token i in the word-sorted order
  int32_t *z;                  // document (CSR) order
  int32_t *zw;                 // word-sorted order, zw[inv_perm[i]] = z[i]
  const int64_t *chunk_start;  // local token index of the chunk's first token
  const int32_t *chunk_doc;    // local document index
  const int32_t *chunk_len;
  const double *theta;
  const double *phiT;
  uint32_t *status;
  int64_t num_chunks;
  int64_t tok_base;
  uint64_t seed;
  uint32_t iteration;
  int32_t K, Kp, pitch16, tile_tokens;
  int32_t ablate;   // timing-only experiments (env GGS_DEBUG_ABLATE): 2 no walk, 4 no staging, 8 no sum pass
  // z_sliced_kernel only (ggs_z_sliced.hpp): its own chunk lists -- cold chunks [0, num_cold), hot chunks
  // [num_cold, num_chunks) -- stored chunk-major, 64 entries per chunk
  const int32_t *ct_tok;       // cold: word id, hot: row of the LDS table; | (0 or 1: which of the chunk's documents) << 30
  const int32_t *ct_idx;       // local token index, -1 = idle lane
  const int32_t *ct_ip;        // inv_perm[ct_idx]
  const int32_t *c_docs;       // [num_chunks][2] local documents of a chunk (the second repeats the first if there is one)
  int64_t num_cold;
  const int32_t *hot_words;    // [num_hot] word ids of the table rows, most frequent first
  int32_t num_hot, hot_pitch;  // rows and row pitch (bytes) of the LDS table
  int32_t hot_off, wave_lds;   // LDS byte offset of the table; bytes of LDS per wave (theta rows + ring)
  int32_t ring_base;           // offset of the ring inside a wave's LDS
  double margin_scale;         // z_stream1_kernel: 1.0; tests scale the certainty margin up to force its exact replay path
  // z_stream1_kernel with two theta rows per wave (moderate K): a chunk may run across ONE document boundary;
  // chunk_len then carries len | split << 8 (split = tokens of the first document) and chunk_doc1 the second document
  const int32_t *chunk_doc1;
  int32_t two_rows;
  // With an exchange attached (z_sliced_kernel's cold chunks): the kernel adds every COLD token's new (word, topic) cell into
  // the zeroed send buffer of the count reduce-scatter itself -- one fire-and-forget atomic per token beside its two z
  // stores (UPLDA:1547-1557 did the same with AtomicIntegers; integer sums are order-free) -- and only the few hot words'
  // segments are left to count_sorted_kernel between the z step and the collective (a rank in eight: 35 us -> a few).
  // Null: the counts are rebuilt by count_sorted_kernel alone.
  int32_t *cnt_send;
  SliceMap smap;
  // z_warm_kernel (ggs_z_sliced.hpp): the WARM tiers -- the words next in frequency after the hot table's, a table load
  // each, their chunks drawn from up to warm_docs_for(KMAX) documents.  Chunk lists like ct_*, tier after tier.
  const int4 *ht_pack;         // z_hot_kernel's chunks (the hot chunks of ct_*, in order) in the packed form of wt_pack
  const int32_t *h_docs;       // ... and their documents, kWarmDocSlots per chunk (two in use)
  const int4 *wt_pack;         // per lane {row of the tier's table | (which of the chunk's documents) << kWarmSlotShift, local token
                               // index or -1, its word-sorted position, 0}: ONE 16-byte load per lane and chunk
  const int32_t *w_docs;       // [warm chunks][kWarmDocSlots] local documents, the first warm_docs_for(KMAX) in use (unused slots repeat the first)
  const int32_t *warm_words;   // [warm_tiers][warm_rows] word ids of the tables' rows
  const int64_t *warm_meta;    // [warm_tiers + 1] first chunk of a tier, then [warm_tiers] rows of its table
  int32_t warm_tiers, warm_rows;
  long long *dbg;              // -DGGS_WARM_TRACE builds only: per wave, cycles by phase of z_warm_kernel's chunk loop
};

struct alignas(16) D2 { double a, b; };

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;
// Read-only tables, read through the constant address space so that wave-uniform accesses
// stay scalar (SMEM) loads in a loop that also contains LDS-DMA writes.
typedef __attribute__((address_space(4))) const int64_t const_i64_t;
typedef __attribute__((address_space(4))) const int32_t const_i32_t;

__device__ __forceinline__ D2 lds_d2(const unsigned char *p) { return *reinterpret_cast<const D2 *>(p); }

// NT = number of 64-topic slices of a theta row; the launcher picks the smallest NT with 64*NT >= K.
template <int NT>
__global__ __launch_bounds__(64) void z_kernel(ZParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int K = p.K, Kp = p.Kp;
  const int upr = Kp >> 1;                 // 16-byte units per phi row
  const int pitch = p.pitch16 * 16;        // LDS row pitch, bytes
  unsigned char *thb = smem + (size_t)p.tile_tokens * pitch;   // theta row, Kp doubles
  const unsigned char *phib = reinterpret_cast<const unsigned char *>(p.phiT);
  const size_t rowbytes = (size_t)Kp * 8;
  const const_i64_t *cstart = (const const_i64_t *)p.chunk_start;
  const const_i32_t *clen = (const const_i32_t *)p.chunk_len;
  const const_i32_t *cdoc = (const const_i32_t *)p.chunk_doc;
  const int64_t stride = gridDim.x;
  const int64_t C = p.num_chunks;

  // prologue: this wave's first chunk in full, the second chunk's descriptor
  int64_t c = blockIdx.x;
  if (c >= C) return;
  int64_t start = cstart[c];
  int len = clen[c], doc = cdoc[c];
  int64_t nstart = 0;
  int nlen = 0, ndoc = 0;
  if (c + stride < C) { nstart = cstart[c + stride]; nlen = clen[c + stride]; ndoc = cdoc[c + stride]; }
  int w = (lane < len) ? p.tok[start + lane] : 0;
  double tv[NT];
  {
    const double *thg = p.theta + (size_t)doc * K;
#pragma unroll
    for (int t = 0; t < NT; ++t) tv[t] = (t * 64 + lane < K) ? thg[t * 64 + lane] : 0.0;
  }

  for (;;) {
    // this chunk's theta row: registers -> LDS (loaded one iteration ago)
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t * 64 + lane < Kp) reinterpret_cast<double *>(thb)[t * 64 + lane] = tv[t];

    // ---- requests for the chunks ahead, issued before the tile so they complete with it
    const bool has_next = c + stride < C;
    int64_t nnstart = 0;
    int nnlen = 0, nndoc = 0, wn = 0;
    if (has_next) {
      if (c + 2 * stride < C) { nnstart = cstart[c + 2 * stride]; nnlen = clen[c + 2 * stride]; nndoc = cdoc[c + 2 * stride]; }
      wn = (lane < nlen) ? p.tok[nstart + lane] : 0;
      const double *thg = p.theta + (size_t)ndoc * K;
#pragma unroll
      for (int t = 0; t < NT; ++t) tv[t] = (t * 64 + lane < K) ? thg[t * 64 + lane] : 0.0;
    }

    // ---- stage: one DMA per token row
    if (!(p.ablate & 4)) {
      for (int i = 0; i < len; ++i) {
        const int wi = __builtin_amdgcn_readlane(w, i);              // wave-uniform word id of row i
        const unsigned char *rb = phib + (size_t)wi * rowbytes;
        unsigned char *dst = smem + (size_t)i * pitch;
        for (int u0 = 0; u0 < upr; u0 += 64) {
          const int u = u0 + lane;
          if (u < upr)
            __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(rb + (size_t)u * 16), (lds_void_t *)(dst + (size_t)u0 * 16), 16, 0, 0);
        }
      }
    }
    // LDS-DMA completion is tracked by vmcnt and the compiler does not know the LDS reads
    // below depend on it.
    asm volatile("s_waitcnt vmcnt(0)");
    __syncthreads();

    if (lane < len) {
      const unsigned char *rowp = smem + lane * pitch;

#define GGS_LD8(A, P) { A##0 = lds_d2(P); A##1 = lds_d2((P) + 16); A##2 = lds_d2((P) + 32); A##3 = lds_d2((P) + 48); }
      // ---- pass 1: sum of the K scores, k ascending (GGS:96-101).  Register sets a/b (phi)
      // and s/t (theta), 8 topics each, ping-pong: the next 8 are in flight during the chain.
#define GGS_SUM8(A, S) { \
        sum += (S##0).a * (A##0).a; sum += (S##0).b * (A##0).b; sum += (S##1).a * (A##1).a; sum += (S##1).b * (A##1).b; \
        sum += (S##2).a * (A##2).a; sum += (S##2).b * (A##2).b; sum += (S##3).a * (A##3).a; sum += (S##3).b * (A##3).b; }
      double sum = 0.0;
      if (p.ablate & 8) sum = 1.0;
      else {
        int j = 0;
        if (K >= 8) {
          D2 a0, a1, a2, a3, b0, b1, b2, b3, s0, s1, s2, s3, t0, t1, t2, t3;
          GGS_LD8(a, rowp) GGS_LD8(s, thb)
          while (j + 24 <= K) {
            GGS_LD8(b, rowp + (j + 8) * 8) GGS_LD8(t, thb + (j + 8) * 8)
            GGS_SUM8(a, s)
            GGS_LD8(a, rowp + (j + 16) * 8) GGS_LD8(s, thb + (j + 16) * 8)
            GGS_SUM8(b, t)
            j += 16;
          }
          if (j + 16 <= K) {
            GGS_LD8(b, rowp + (j + 8) * 8) GGS_LD8(t, thb + (j + 8) * 8)
            GGS_SUM8(a, s)
            GGS_SUM8(b, t)
            j += 16;
          } else {
            GGS_SUM8(a, s)
            j += 8;
          }
        }
        for (; j < K; ++j)
          sum += *reinterpret_cast<const double *>(thb + j * 8) * *reinterpret_cast<const double *>(rowp + j * 8);
      }
#undef GGS_SUM8

      const uint64_t gtok = (uint64_t)(p.tok_base + start + lane);
      const U4 o = philox4x32_10((uint32_t)gtok, (uint32_t)(gtok >> 32), (uint32_t)GGS_PURPOSE_Z << 24, p.iteration,
                                 (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
      const double U = u53(o.x, o.y);
      double sample = U * sum;

      // ---- pass 2: the walk of GGS:108-113,
      //   newTopic = -1; while (sample > 0) { newTopic++; sample -= score[newTopic]; }
      // in counting form: scores are >= 0, so once sample <= 0 it stays <= 0 and
      // newTopic + 1 == #{k : sample before subtracting score[k] was > 0}.  The subtraction
      // itself is the same sequential fp64 chain; the wave leaves when no lane is still > 0.
      int cnt = 0;
      bool live = !(p.ablate & 2);
      if (!live) { cnt = 1 + (int)(U * K); sample = 0.0; }
#define GGS_STEP(TH, PHI) { cnt += (sample > 0.0); sample -= (TH) * (PHI); }
#define GGS_WALK8(A, S) { \
        GGS_STEP((S##0).a, (A##0).a) GGS_STEP((S##0).b, (A##0).b) GGS_STEP((S##1).a, (A##1).a) GGS_STEP((S##1).b, (A##1).b) \
        GGS_STEP((S##2).a, (A##2).a) GGS_STEP((S##2).b, (A##2).b) GGS_STEP((S##3).a, (A##3).a) GGS_STEP((S##3).b, (A##3).b) }
      {
        int j = 0;
        if (live && K >= 8) {
          D2 a0, a1, a2, a3, b0, b1, b2, b3, s0, s1, s2, s3, t0, t1, t2, t3;
          GGS_LD8(a, rowp) GGS_LD8(s, thb)
          while (live && j + 24 <= K) {
            GGS_LD8(b, rowp + (j + 8) * 8) GGS_LD8(t, thb + (j + 8) * 8)
            GGS_WALK8(a, s)
            GGS_LD8(a, rowp + (j + 16) * 8) GGS_LD8(s, thb + (j + 16) * 8)
            GGS_WALK8(b, t)
            j += 16;
            live = __any(sample > 0.0);
          }
          if (live) {
            if (j + 16 <= K) {
              GGS_LD8(b, rowp + (j + 8) * 8) GGS_LD8(t, thb + (j + 8) * 8)
              GGS_WALK8(a, s)
              GGS_WALK8(b, t)
              j += 16;
            } else {
              GGS_WALK8(a, s)
              j += 8;
            }
            live = __any(sample > 0.0);
          }
        }
        if (live)
          for (; j < K; ++j)
            GGS_STEP(*reinterpret_cast<const double *>(thb + j * 8), *reinterpret_cast<const double *>(rowp + j * 8))
      }
#undef GGS_WALK8
#undef GGS_STEP
#undef GGS_LD8
      int new_topic = cnt - 1;
      if (new_topic < 0 || sample > 0.0) {        // GGS:116-118 (and the index past K Java would throw on)
        atomicOr(p.status, ST_INVALID_TOPIC);
        new_topic = new_topic < 0 ? 0 : K - 1;
      }
      p.z[start + lane] = new_topic;
      p.zw[p.inv_perm[start + lane]] = new_topic;
    }
    if (!has_next) break;
    __syncthreads();                               // every LDS read of this tile is done before the next tile lands
    c += stride;
    start = nstart; len = nlen; doc = ndoc; w = wn;
    nstart = nnstart; nlen = nnlen; ndoc = nndoc;
  }
}

}  // namespace ggs
