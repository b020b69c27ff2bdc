// ggs_kernels.hpp -- the HIP kernels of the Grouped Gibbs sweep (gfx950).
//
// Device layouts (private to the library):
//   tok[N]            int32   word ids, CSR order
//   z[N]              int32   topic assignments
//   chunk_*[C]                work items of the K > 192 z kernels: <=64 consecutive tokens of ONE document
//   ct_*[Cs][64], c_docs      work items of the sliced z kernels: <=64 tokens of <=2 documents, cold chunks first,
//                             then the chunks of the hot words (ggs_z_sliced.hpp)
//   perm[N]           int32   token indices sorted by word id (built once on the host)
//   inv_perm[N]       int32   its inverse: position of token i in the word-sorted order
//   zw[N]             int32   z in word-sorted order (zw[inv_perm[i]] == z[i]), written by the
//                             z kernel beside z so that the count rebuild streams instead of gathering
//   seg_*[S]                  count-kernel work items: <=4096 consecutive entries of perm
//                             that all carry the same word
//   theta[D][K]       fp64    thetaMatrix rows (GGS:72)
//   phiT[V][Kp]       fp64    TRANSPOSE of the Java phi[K][V] (UPLDA:69), row pitch Kp = K
//                             rounded up to even, so one token reads one contiguous row
//   n_wk[V][K]        int32   typeTopicCounts layout (MSLDA:73)
//   n_k[K]            int32   tokensPerTopic
//
// Java keeps every running sum sequential in index order (sum += ...), and these kernels produce
// the same bits: the K-long sums (scores of a token, gammas of a document) are walked by ONE lane
// each, in that order, many side by side (one lane per token / document); the two V-long sums of
// the Phi draw are computed in parallel by ggs_exact_sum.hpp, which proves step by step that its
// result is the sequential one.
//
// The reference's AtomicInteger delta matrix + merge (UPLDA:1547-1557, 1107-1221) exists to
// let JVM threads share counts; its net effect per sweep is n_wk = histogram of (word, z).
// Integer sums do not depend on order, so the device rebuilds that histogram from z with a
// word-sorted segmented pass (count_sorted_kernel) instead of 2 contended global atomics per
// token: identical counts, no hot-word serialisation.
#pragma once
#include "ggs_device_math.hpp"
#include "ggs_exact_sum.hpp"

namespace ggs {

constexpr double kJavaMinValue = 4.9e-324;  // Double.MIN_VALUE, ParallelDirichlet.java:64

// x / d for 0 <= x, 0 < d with x * d < 2^32, by one multiply (m = udiv_magic(d)): the element loops below split a
// small linear index into (row, column) once per element (measured: 1 % of the Phi draw's VALU instructions).
__device__ __forceinline__ uint32_t udiv_magic(uint32_t d) { return d > 1 ? (uint32_t)((0x100000000ull + d - 1) / d) : 0u; }   // ceil(2^32 / d); 0 stands for d = 1
__device__ __forceinline__ int udiv_small(int x, uint32_t m) { return m ? (int)__umulhi((uint32_t)x, m) : x; }

enum StatusBits : uint32_t {
  ST_NEGATIVE_COUNT = 1u << 0,
  ST_INVALID_TOPIC = 1u << 1,
  ST_RNG_EXHAUSTED = 1u << 2,
  ST_BAD_SHAPE = 1u << 3,
};

// ------------------------------------------------------------------------------
// K1+K2: per-document theta draw (GGS:57-72 + ParallelDirichlet.java:46-70).
// One workgroup owns `docs_per_block` consecutive documents.
//   phase 1  histogram of the current z per document (LDS atomics)
//   phase 2  one lane per document: magnitude = sum_k (n_dk + alpha_k), in k order
//   phase 3  all lanes: Gamma(partition*magnitude) draws, one (doc, k) pair each: straight-line first try, then the
//            general rejection loops for the elements it left over, gathered into full waves
//   phase 4  one lane per document: sum of the gammas, in k order
//   phase 5  all lanes: normalise, clamp <=0 to Double.MIN_VALUE, coalesced store
// LDS: one 8-byte cell per (k, document), [K][BP] with BP = docs_per_block | 1 (odd => the
// k-major walks of phases 2, 4, 5 are bank-conflict free): the count n_dk (int32, low word) until
// its lane has drawn the gamma that replaces it.
// ------------------------------------------------------------------------------
struct ThetaParams {
  const int64_t *doc_ptr;
  const int32_t *z;
  const double *alpha;
  double *theta;
  uint32_t *status;
  int64_t num_docs, doc_base;
  uint64_t seed;
  uint32_t iteration;
  int32_t K, docs_per_block;
  int32_t queue_cap;   // <= kGammaQueue (smaller only in tests: the draw-on-the-spot path of a full queue)
};

constexpr int kGammaQueue = 768;             // uint16 element indices: a workgroup's cells number < 65536 (LDS / 8)
constexpr int kThetaQueueBytes = 8 + 2 * kGammaQueue;

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void theta_kernel(ThetaParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int K = p.K, B = p.docs_per_block, BP = B | 1;
  double *gam = reinterpret_cast<double *>(smem);                    // [K][BP]
  int32_t *hist = reinterpret_cast<int32_t *>(smem);                 // the same cells: count in word 0 of cell i = hist[2 * i]
  double *mag = gam + (size_t)K * BP;                                // [B]
  double *tot = mag + B;                                             // [B]
  int32_t *len = reinterpret_cast<int32_t *>(tot + B);               // [B]
  int32_t *qn = len + B;                                             // [2] (one used) fill of the queue below
  uint16_t *queue = reinterpret_cast<uint16_t *>(qn + 2);            // [kGammaQueue] elements left to the general draw

  const int tid = threadIdx.x;
  const int64_t d0 = (int64_t)blockIdx.x * B;
  const int nb = (int)min((int64_t)B, p.num_docs - d0);

  for (int i = tid; i < K * BP; i += BLOCK) hist[2 * i] = 0;
  if (tid == 0) *qn = 0;
  __syncthreads();

  constexpr int NW = BLOCK / 64;
  const int wave = tid >> 6, lane = tid & 63;
  for (int b = wave; b < nb; b += NW) {
    const int64_t beg = p.doc_ptr[d0 + b], end = p.doc_ptr[d0 + b + 1];
    if (lane == 0) len[b] = (int32_t)(end - beg);
    for (int64_t i = beg + lane; i < end; i += 64) atomicAdd(&hist[2 * (p.z[i] * BP + b)], 1);
  }
  __syncthreads();

  if (tid < nb && len[tid] > 0) {
    double m = 0;
    for (int k = 0; k < K; ++k) m += (double)hist[2 * (k * BP + tid)] + p.alpha[k];  // Dirichlet(double[]): magnitude
    mag[tid] = m;
  }
  __syncthreads();

  // phase 3a: every lane tries the straight-line draw (rgamma_first_try); the ~14 % of elements it does not settle are
  // queued and drawn by the general rgamma in full waves (3b) -- or on the spot when the queue is full
  auto shape_of = [&](int k, int b) {
    const double pk = (double)hist[2 * (k * BP + b)] + p.alpha[k];   // GGS:68
    const double m = mag[b];
    return (pk / m) * m;                                             // partition[i] * magnitude
  };
  auto draw_general = [&](int k, int b, double shape) {
    DrawStream rs(p.seed, p.iteration, GGS_PURPOSE_THETA, (uint64_t)(p.doc_base + d0 + b) * (uint64_t)K + (uint64_t)k);
    const double g = rgamma(rs, shape);
    if (rs.exhausted) atomicOr(p.status, ST_RNG_EXHAUSTED);
    return g;
  };
  const uint32_t m_nb = udiv_magic((uint32_t)max(nb, 1)), m_K = udiv_magic((uint32_t)K);     // nb * K cells: < 2^16
  for (int i = tid; i < nb * K; i += BLOCK) {
    const int k = udiv_small(i, m_nb), b = i - k * nb;
    if (len[b] == 0) continue;                                       // GGS:52-53
    const double shape = shape_of(k, b);
    double g;
    if (shape > 0) {
      if (!rgamma_first_try(p.seed, p.iteration, GGS_PURPOSE_THETA, (uint64_t)(p.doc_base + d0 + b) * (uint64_t)K + (uint64_t)k, shape, g)) {
        const int slot = atomicAdd(qn, 1);
        if (slot < p.queue_cap) { queue[slot] = (uint16_t)i; continue; }   // the cell keeps its count until 3b
        g = draw_general(k, b, shape);
      }
    } else {
      g = __builtin_nan("");
      atomicOr(p.status, ST_BAD_SHAPE);
    }
    gam[k * BP + b] = g;
  }
  __syncthreads();
  for (int q = tid, m = min(*qn, p.queue_cap); q < m; q += BLOCK) {
    const int i = queue[q], k = udiv_small(i, m_nb), b = i - k * nb;
    gam[k * BP + b] = draw_general(k, b, shape_of(k, b));
  }
  __syncthreads();

  if (tid < nb && len[tid] > 0) {
    double s = 0;
    for (int k = 0; k < K; ++k) s += gam[k * BP + tid];              // ParallelDirichlet.java:53-57
    tot[tid] = s;
  }
  __syncthreads();

  for (int i = tid; i < nb * K; i += BLOCK) {
    const int b = udiv_small(i, m_K), k = i - b * K;
    if (len[b] == 0) continue;
    double v = gam[k * BP + b];
    const double s = tot[b];
    if (s != 0) {                                                    // ParallelDirichlet.java:60-66
      v = v / s;
      if (v <= 0) v = kJavaMinValue;
    }
    p.theta[(size_t)(d0 + b) * K + k] = v;
  }
}

// K3, the token loop (GGS:79-130), lives in ggs_z_kernel.hpp.

// ------------------------------------------------------------------------------
// K4+K5: type-topic counts.  One workgroup = one segment = up to 4096 consecutive entries
// of the word-sorted topic assignments zw, all with the same word w: topic histogram in LDS,
// then K integer adds onto row w (n_wk is zeroed before the launch; several segments of a
// frequent word add to the same row).  Short segments (rare words) add straight to HBM.
// ------------------------------------------------------------------------------
struct CountParams {
  const int32_t *zw;         // topic assignments in word-sorted order
  const int32_t *seg_word;   // [S]
  const int32_t *seg_begin;  // [S+1] offsets into zw
  int32_t *n_wk;             // cell (w, k) lives at n_wk[w * row_stride + (koff ? koff[k] : k)]
  const int64_t *koff;       // null: the plain [V][K] layout.  With an exchange attached: the slice-major
                             // [nranks][V][Ksm] send buffer of the count reduce-scatter, koff[k] = slice(k)*V*Ksm + (k - k0(slice))
  int32_t K, num_segs, row_stride;
  const int32_t *seg_end;    // null: segment s ends where s + 1 begins.  Given: a list of segments that are not adjacent
                             // (the hot words' segments, counted on their own when the z kernels count the rest)
  int32_t segs_per_block;    // kCountSegsPerBlock for the whole corpus (most segments are tiny), 1 for a list of full ones
};

constexpr int kCountSegsPerBlock = 8;   // most words are rare: a workgroup per <= 256-token segment would be mostly dispatch overhead

__global__ __launch_bounds__(256) void count_sorted_kernel(CountParams p) {
  __builtin_amdgcn_s_setprio(3);   // a link of the chain to the next z step: issue ahead of the theta draw beside it (ggs_exact_sum.hpp)
  extern __shared__ __align__(16) unsigned char smem[];
  int32_t *hist = reinterpret_cast<int32_t *>(smem);
  const int tid = threadIdx.x, K = p.K;
  const int seg0 = blockIdx.x * p.segs_per_block, seg1 = min(seg0 + p.segs_per_block, p.num_segs);
  for (int seg = seg0; seg < seg1; ++seg) {
    const int beg = p.seg_begin[seg], end = p.seg_end ? p.seg_end[seg] : p.seg_begin[seg + 1];
    int32_t *row = p.n_wk + (size_t)p.seg_word[seg] * p.row_stride;
    if (end - beg <= 256) {                    // uniform per block
      if (beg + tid < end) {
        const int k = p.zw[beg + tid];
        atomicAdd(&row[p.koff ? p.koff[k] : (int64_t)k], 1);
      }
      continue;
    }
    for (int k = tid; k < K; k += 256) hist[k] = 0;
    __syncthreads();
    int i = beg + tid;
    for (; i + 768 < end; i += 1024) {
      const int k0 = p.zw[i], k1 = p.zw[i + 256], k2 = p.zw[i + 512], k3 = p.zw[i + 768];
      atomicAdd(&hist[k0], 1); atomicAdd(&hist[k1], 1); atomicAdd(&hist[k2], 1); atomicAdd(&hist[k3], 1);
    }
    for (; i < end; i += 256) atomicAdd(&hist[p.zw[i]], 1);
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
      const int32_t cnt = hist[k];
      if (cnt) atomicAdd(&row[p.koff ? p.koff[k] : (int64_t)k], cnt);
    }
    __syncthreads();                           // hist is reused by the next segment
  }
}

// ------------------------------------------------------------------------------
// The SPARSE count exchange (include/ggs_hip.h, ggs_set_count_exchange): instead of the dense [nranks][V][Ksm] histogram a
// rank ships the non-zero cells of its (word, topic) histogram as (cell, count) pairs, cell = v * Ksm + column in the
// destination rank's slice.  The reference tracks exactly this touched-cell set per topic (globalDeltaNUpdates,
// UPLDA:1166-1176).  Two passes over the word-sorted z, each a per-segment LDS histogram [K]:
//   EMIT = false   how many pairs go to every destination (dest_count [nranks])
//   EMIT = true    the pairs, appended at dest_cursor[r] (initialised with the destination's offset in the send buffer)
// A word with several segments emits a cell several times: the receiver adds them up.  Pair order is whatever the
// atomics make it; the sums are order-free.
// ------------------------------------------------------------------------------
struct SparseCountParams {
  const int32_t *zw, *seg_word, *seg_begin;
  int32_t K, num_segs, nranks;
  int32_t ksm;
  int32_t *wg_count;          // [workgroups][nranks] pairs per destination (EMIT = false: written)
  const int64_t *wg_off;      // [workgroups][nranks] exclusive prefix of wg_count over the workgroups (EMIT = true: read)
  const int64_t *dest_base;   // [nranks] element offset of a destination's block in `pairs` (EMIT = true)
  int32_t *pairs;             // (cell, count) int32 pairs
  // topic -> (rank, column): the even split of EvenSplitTopicBatchBuilder.java:28-39
  int32_t rem, size;
  uint32_t m_size, m_size1;
};
constexpr int kSparseMaxRanks = 64;
constexpr int kSparseSegsPerBlock = 32;   // segments per workgroup: one row of the offset table per workgroup

// No global atomics: pass A leaves every workgroup's pair count per destination, a scan turns them into offsets, and pass B
// writes each workgroup's pairs into its own stretch of the destination's block (positions inside it by an LDS cursor).
// (A first version appended every pair at a global cursor per destination: 28 M returning atomics on 8 addresses, 571 ms.)
template <bool EMIT>
__global__ __launch_bounds__(256) void sparse_count_kernel(SparseCountParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  int32_t *hist = reinterpret_cast<int32_t *>(smem);                       // [K]
  __shared__ int32_t dest_n[kSparseMaxRanks];                              // pass A: pairs so far; pass B: pairs written so far
  __shared__ long long dest_at[kSparseMaxRanks];                           // pass B: this workgroup's first element in each destination's block
  const int tid = threadIdx.x, K = p.K;
  const int seg0 = blockIdx.x * kSparseSegsPerBlock, seg1 = min(seg0 + kSparseSegsPerBlock, p.num_segs);
  if (tid < kSparseMaxRanks) {
    dest_n[tid] = 0;
    if (EMIT && tid < p.nranks) dest_at[tid] = p.dest_base[tid] + 2 * p.wg_off[(size_t)blockIdx.x * p.nranks + tid];
  }
  auto owner = [&](int k, int &r, int &c) {
    const int cut = p.rem * (p.size + 1);
    if (k < cut) { r = udiv_small(k, p.m_size1); c = k - r * (p.size + 1); }
    else { const int q = udiv_small(k - cut, p.m_size); r = p.rem + q; c = k - cut - q * p.size; }
  };
  for (int seg = seg0; seg < seg1; ++seg) {
    const int beg = p.seg_begin[seg], end = p.seg_begin[seg + 1], w = p.seg_word[seg];
    for (int k = tid; k < K; k += 256) hist[k] = 0;
    __syncthreads();
    for (int i = beg + tid; i < end; i += 256) atomicAdd(&hist[p.zw[i]], 1);
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
      const int32_t cnt = hist[k];
      if (!cnt) continue;
      int r, c;
      owner(k, r, c);
      const int slot = atomicAdd(&dest_n[r], 1);                         // LDS
      if (EMIT) {
        const long long at = dest_at[r] + 2ll * slot;
        p.pairs[at] = w * p.ksm + c;                                     // V * Ksm < 2^31 (checked by the host)
        p.pairs[at + 1] = cnt;
      }
    }
    __syncthreads();
  }
  if (!EMIT && tid < p.nranks) p.wg_count[(size_t)blockIdx.x * p.nranks + tid] = dest_n[tid];
}

// wg_off[g][r] = sum over g' < g of wg_count[g'][r]; total[r] = the sum over all workgroups.  One workgroup per destination.
__global__ __launch_bounds__(256) void sparse_scan_kernel(const int32_t *wg_count, int32_t num_wg, int32_t nranks, int64_t *wg_off, int64_t *total) {
  __shared__ long long part[256];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int per = (num_wg + 255) / 256, g0 = tid * per, g1 = min(g0 + per, num_wg);
  long long s = 0;
  for (int g = g0; g < g1; ++g) s += wg_count[(size_t)g * nranks + r];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    long long run = 0;
    for (int i = 0; i < 256; ++i) { const long long v = part[i]; part[i] = run; run += v; }
    total[r] = run;
  }
  __syncthreads();
  long long run = part[tid];
  for (int g = g0; g < g1; ++g) { wg_off[(size_t)g * nranks + r] = run; run += wg_count[(size_t)g * nranks + r]; }
}

// cnt_own[cell] += count for the received pairs (cnt_own zeroed before the launch)
__global__ __launch_bounds__(256) void scatter_add_pairs_kernel(const int32_t *pairs, int64_t num_pairs, int32_t *cnt_own) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < num_pairs; i += stride) {
    const int2 pc = reinterpret_cast<const int2 *>(pairs)[i];
    atomicAdd(&cnt_own[pc.x], pc.y);
  }
}

// ------------------------------------------------------------------------------
// K6/K8: Phi draw (GGS:182-198 / MarsagliaSparseDirichlet.java:31-55).
//   column sum <int32,true>    magnitude_k = sum_v (beta + n_kv), v ascending (the
//                              Dirichlet(double[]) constructor); also tokensPerTopic
//   phi_gamma                  lane per (v,k): Gamma(partition*magnitude) -> phiT (unnormalised)
//   column sum <double,false>  sum_v gamma, v ascending (ParallelDirichlet.java:53-57)
//   phi_normalise              lane per (v,k): divide, clamp, optional running phiMean +=
// The column sums are the exact parallel ones of ggs_exact_sum.hpp.  column_chain_kernel below is
// what they replaced, kept as the element-by-element cross-check (GGS_DEBUG_CHAIN=1):
//
// column_chain: a sum over V in index order walked as one dependent fp64 add chain per topic
// (8 cycles per add for a lone wave: >= 0.17 ms at V = 50k, 0.28 ms measured).
// One workgroup owns 8 adjacent topics (a 64-byte / 32-byte slice of every row).  Waves 1-3
// are loaders: they stream [384 rows x 8 topics] tiles into a double-buffered LDS ring with
// plain coalesced loads and do the per-element part of the sum (beta + count) on the way;
// 8 lanes of wave 0 do nothing but walk the previous tile row by row.  48 KiB of LDS: small
// enough to slot in beside the theta draw running on the side stream.
// ------------------------------------------------------------------------------
template <typename T, bool MAGNITUDE>
__global__ __launch_bounds__(256) void column_chain_kernel(const T *src, int32_t pitch, int32_t K, int32_t V, double beta,
                                                           double *out_sum) {
  constexpr int TPB = 8, ROWS = 384, LOADERS = 192, PER_THREAD = ROWS * TPB / LOADERS;   // 16
  __shared__ double buf[2][ROWS * TPB];
  const int tid = threadIdx.x;
  const int k0 = blockIdx.x * TPB;
  const bool loader = tid >= 64;
  const int lt = tid - 64;                           // loader thread id
  const int t = lt & 7, r0 = lt >> 3;               // element (row r0 + 24 j, topic t)
  // Loads are unconditional (addresses clamped into the matrix) so that all PER_THREAD of
  // them are in flight together; rows >= V and topics >= K are loaded but never consumed.
  const T *col = src + min(k0 + (t & 7), K - 1);

  T regs[PER_THREAD];
  auto load_tile = [&](int v0) {
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
      const int v = min(v0 + r0 + 24 * j, V - 1);
      regs[j] = col[(size_t)v * pitch];
    }
  };
  // GGS:188 dirichletParams[type] = beta + count (int -> double, one rounding), in parallel
  auto store_tile = [&](int b) {
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j)
      buf[b][(r0 + 24 * j) * TPB + t] = MAGNITUDE ? (beta + (double)regs[j]) : (double)regs[j];
  };

  double acc = 0;
  if (loader) { load_tile(0); store_tile(0); }
  __syncthreads();
  int b = 0;
  for (int v0 = 0; v0 < V; v0 += ROWS, b ^= 1) {
    const bool more = v0 + ROWS < V;
    if (loader) {
      if (more) { load_tile(v0 + ROWS); store_tile(b ^ 1); }   // lands while the chain below runs
    } else if (tid < TPB) {
      const int rows = min(ROWS, V - v0);
      const double *bp = &buf[b][tid];
      if (rows == ROWS) {
        // two register sets of 16 rows, no copies: the LDS reads of the next 16 rows are in
        // flight while this set's dependent add chain runs
        double x[16], y[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = bp[j * TPB];
#pragma unroll
        for (int r = 0; r < ROWS; r += 32) {
#pragma unroll
          for (int j = 0; j < 16; ++j) y[j] = bp[(r + 16 + j) * TPB];
#pragma unroll
          for (int j = 0; j < 16; ++j) acc += x[j];
          if (r + 32 < ROWS) {
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = bp[(r + 32 + j) * TPB];
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) acc += y[j];
        }
      } else {
        for (int r = 0; r < rows; ++r) acc += bp[r * TPB];   // the last, partial tile
      }
    }
    __syncthreads();
  }
  if (tid < TPB && k0 + tid < K) out_sum[k0 + tid] = acc;
}

// zw[i] = z[perm[i]]: word-sorted copy of z after a host upload (set_z, seeded initial z)
__global__ __launch_bounds__(256) void permute_z_kernel(const int32_t *perm, const int32_t *z, int32_t *zw, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) zw[i] = z[perm[i]];
}

// tokensPerTopic n_k = sum_v n_wk[v][k]: integers, any order (rows of `pitch` ints, the first K columns).
__global__ __launch_bounds__(256) void topic_totals_kernel(const int32_t *n_wk, int32_t K, int32_t pitch, int32_t V, int32_t *n_k) {
  extern __shared__ __align__(16) unsigned char smem[];
  int32_t *part = reinterpret_cast<int32_t *>(smem);             // [K]
  for (int k = threadIdx.x; k < K; k += 256) part[k] = 0;
  __syncthreads();
  const int64_t n = (int64_t)V * K, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int v = (int)(i / K), k = (int)(i - (int64_t)v * K);
    const int32_t c = n_wk[(size_t)v * pitch + k];
    if (c) atomicAdd(&part[k], c);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256)
    if (part[k]) atomicAdd(&n_k[k], part[k]);
}

// The Phi draw works on a TOPIC SLICE [k0, k0 + Ks): the whole matrix for one GPU (k0 = 0, Ks = K, counts [V][K],
// output phiT [V][Kp]); with an exchange attached the rank's own topics (counts [V][Ksm] from the reduce-scatter,
// output [V][Ksm] for the all-gather).  The Philox element id carries the GLOBAL topic, so a slice draws what the
// one-GPU run draws for those topics.
struct PhiGammaParams {
  const int32_t *n_wk; // [V][cnt_pitch], column j = topic k0 + j
  const double *mag;   // [Ks] per topic of the slice (sweep draw) -- unused for the initial draw
  double *phiT;        // [V][Kp], column j = topic k0 + j
  uint32_t *status;
  uint64_t seed;
  uint32_t iteration, purpose;
  int32_t K, Kp, V;    // K = Ks: the slice width
  int32_t cnt_pitch, k0;
  double beta;         // sweep draw: shape = ((beta+n)/mag)*mag
  double prior_pm;     // initial draw: partition*magnitude = (1.0/V)*(V*beta)
  int32_t initial;
  int32_t kc, ncg;     // a tile = one 64-row segment x kc <= kPhiCols adjacent topics (ncg = ceil(K / kc) column groups)
  int32_t seg_begin, seg_end;   // the segments this launch draws (with an exchange the draw is cut in two, the first
                                // half's all-gather running under the second half's draw)
  int32_t queue_cap;   // <= kPhiQueue (smaller only in tests)
  int32_t prio;        // 1: ask for instruction issue ahead of the theta draw beside it
  // the segment functions of the gammas' column sums (ggs_exact_sum.hpp), computed by the workgroup that drew the tile;
  // guess = the EXACT running magnitude sums [nseg + 1][K] (E Gamma(a) = a).  Null: not wanted.
  const double *guess;
  double *fn;          // [nseg][K][4]
};

// ONE WAVE per workgroup takes tiles of one 64-row segment x kc <= kPhiCols adjacent topics: the straight-line first try
// for all of the tile's elements, then the general rejection loops for the elements it left over (queued in LDS, so that
// they fill the wave: ~14 % of 64 * 6 elements is one round; drawn on the spot when the queue is full).  Every lane
// that has a gamma in hand also adds its quantised value to the tile's segment functions (LDS atomic adds of integers,
// ggs_exact_sum.hpp): the column sums of the gammas need no pass of their own over the matrix.  Single waves because
// they balance: a topic slice of one rank in eight is 391 segments x 3 column groups per launch.
constexpr int kPhiCols = 6, kPhiQueue = 128;

__global__ __launch_bounds__(64) void phi_gamma_kernel(PhiGammaParams p) {
  // With an exchange the slice's draw is on the critical path and the next theta has the whole Phi phase to finish: issue
  // ahead of the theta draw on the side stream (behind the chain's short kernels).  One GPU: the next z step waits for
  // both draws, which share the VALUs -- no priority, so that they end together (with it the theta draw ended 0.17 ms
  // after the Phi chain and the sweep was 0.02 ms longer).
  if (p.prio) __builtin_amdgcn_s_setprio(1);
  __shared__ uint16_t queue[kPhiQueue];
  __shared__ int32_t qn;
  __shared__ double acc_lo[kPhiCols], acc_hi[kPhiCols];
  __shared__ int32_t e_los[kPhiCols], aheads[kPhiCols], flag_s[kPhiCols];
  const int lane = threadIdx.x;
  const int64_t tiles = (int64_t)(p.seg_end - p.seg_begin) * p.ncg;
  auto shape_of = [&](int v, int k) {
    const int32_t cnt = p.n_wk[(size_t)v * p.cnt_pitch + k];
    if (p.initial) return (cnt == 0) ? p.prior_pm : p.prior_pm + (double)cnt;   // MarsagliaSparseDirichlet.java:37-41
    const double pk = p.beta + (double)cnt;                                     // GGS:188
    const double m = p.mag[k];
    return (pk / m) * m;
  };
  auto draw_general = [&](int v, int k, double shape) {
    DrawStream rs(p.seed, p.iteration, p.purpose, (uint64_t)(p.k0 + k) * (uint64_t)p.V + (uint64_t)v);
    const double g = rgamma(rs, shape);
    if (rs.exhausted) atomicOr(p.status, ST_RNG_EXHAUSTED);
    return g;
  };
  const uint32_t m_ncg = udiv_magic((uint32_t)p.ncg);
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int si = udiv_small((int)tile, m_ncg), cg = (int)tile - si * p.ncg;     // tiles < 2^31 / ncg (V < 2^31)
    const int seg = p.seg_begin + si, v0 = seg * 64, kb = cg * p.kc, kw = min(p.kc, p.K - kb);
    const int rows = min(64, p.V - v0), n = rows * kw;
    const uint32_t m_kw = udiv_magic((uint32_t)kw);
    __syncthreads();                   // the previous tile's composition has read the accumulators
    if (lane < kw) {
      SegCand c{kSegNoGuess, true};
      if (p.guess) c = seg_candidates(p.guess[(size_t)seg * p.K + kb + lane], p.guess[(size_t)(seg + 1) * p.K + kb + lane]);
      e_los[lane] = c.e_lo; aheads[lane] = c.ahead ? 1 : 0; acc_lo[lane] = 0; acc_hi[lane] = 0; flag_s[lane] = 0;
    }
    if (lane == 0) qn = 0;
    __syncthreads();
    auto emit = [&](int v, int c, double g) {                            // the gamma to memory, its quantised value to the segment functions
      p.phiT[(size_t)v * p.Kp + kb + c] = g;
      const int e_lo = e_los[c];
      if (e_lo == kSegNoGuess) return;
      double q_lo, q_hi;
      int fl;
      seg_quantise(g, e_lo, q_lo, q_hi, fl);
      seg_accumulate(&acc_lo[c], &acc_hi[c], &flag_s[c], q_lo, q_hi, fl);
    };
#pragma unroll 1
    for (int j = lane; j < n; j += 64) {
      const int dv = udiv_small(j, m_kw), v = v0 + dv, c = j - dv * kw, k = kb + c;
      const double shape = shape_of(v, k);
      double g;
      if (shape > 0) {
        if (!rgamma_first_try(p.seed, p.iteration, p.purpose, (uint64_t)(p.k0 + k) * (uint64_t)p.V + (uint64_t)v, shape, g)) {
          const int slot = atomicAdd(&qn, 1);
          if (slot < p.queue_cap) { queue[slot] = (uint16_t)j; continue; }
          g = draw_general(v, k, shape);
        }
      } else {
        g = __builtin_nan("");
        atomicOr(p.status, ST_BAD_SHAPE);
      }
      emit(v, c, g);
    }
    __syncthreads();
    for (int q = lane, m = min(qn, p.queue_cap); q < m; q += 64) {
      const int j = queue[q], dv = udiv_small(j, m_kw), v = v0 + dv, c = j - dv * kw;
      emit(v, c, draw_general(v, kb + c, shape_of(v, kb + c)));
    }
    __syncthreads();
    if (p.guess && lane < kw)
      seg_compose(SegCand{e_los[lane], aheads[lane] != 0}, acc_lo[lane], acc_hi[lane], flag_s[lane], 0.0, p.fn + ((size_t)seg * p.K + kb + lane) * 4);
  }
}

__global__ __launch_bounds__(256) void phi_normalise_kernel(double *phiT, const double *tot, int32_t K, int32_t Kp, int32_t V,
                                                            double *phi_mean /* [V][K] or null */) {
  __builtin_amdgcn_s_setprio(3);
  const int64_t n = (int64_t)V * K;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int v = (int)(i / K), k = (int)(i - (int64_t)v * K);
    double x = phiT[(size_t)v * Kp + k];
    const double s = tot[k];
    if (s != 0) {
      x = x / s;
      if (x <= 0) x = kJavaMinValue;
      phiT[(size_t)v * Kp + k] = x;
    }
    if (phi_mean) phi_mean[i] += x;                                  // GGS:193-197
  }
}

// ---- exchange layout <-> device layout (one GPU of several; see include/ggs_hip.h, "multi-GPU") ----
// The all-gathered UNNORMALISED gamma slices -> phiT [V][Kp], normalised on the way: rank r's slice arrives in two
// halves, all0 [nranks][c0] = its rows below v_split ([v][Ksm]) and all1 [nranks][c1] = the rows from v_split on followed
// by its Ksm column sums.  The division and the clamp are those of phi_normalise_kernel (ParallelDirichlet.java:60-66),
// on the same operands: the same bits.  The running phiMean += of GGS:193-197 rides along.
struct PhiRepackParams {
  const double *all0, *all1;
  const int32_t *krank, *kcol;   // [K] owner rank of topic k, its column in that rank's slice
  double *phiT, *phi_mean;       // phi_mean [V][K] or null
  int64_t c0, c1;
  int32_t K, Kp, V, Ksm, v_split;
};
__global__ __launch_bounds__(256) void phi_repack_kernel(PhiRepackParams p) {
  __builtin_amdgcn_s_setprio(3);
  const int64_t n = (int64_t)p.V * p.K;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tot_off = (int64_t)(p.V - p.v_split) * p.Ksm;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int v = (int)(i / p.K), k = (int)(i - (int64_t)v * p.K);
    const int r = p.krank[k], j = p.kcol[k];
    const double *h1 = p.all1 + (int64_t)r * p.c1;
    double x = v < p.v_split ? p.all0[(int64_t)r * p.c0 + (int64_t)v * p.Ksm + j] : h1[(int64_t)(v - p.v_split) * p.Ksm + j];
    const double s = h1[tot_off + j];
    if (s != 0) {
      x = x / s;
      if (x <= 0) x = kJavaMinValue;
    }
    p.phiT[(size_t)v * p.Kp + k] = x;
    if (p.phi_mean) p.phi_mean[i] += x;
  }
}
// cnt_all [nranks][V][Ksm] (the all-gathered count slices) -> n_wk [V][K]
__global__ __launch_bounds__(256) void counts_unslice_kernel(const int32_t *cnt_all, const int64_t *koff, int32_t Ksm, int32_t *n_wk, int32_t K,
                                                             int32_t V) {
  const int64_t n = (int64_t)V * K;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int v = (int)(i / K), k = (int)(i - (int64_t)v * K);
    n_wk[i] = cnt_all[koff[k] + (int64_t)v * Ksm];
  }
}
// n_wk [V][K] -> the slice-major send layout (ggs_set_type_topic_counts-style uploads; unused columns stay zero)
__global__ __launch_bounds__(256) void counts_slice_kernel(const int32_t *n_wk, const int64_t *koff, int32_t Ksm, int32_t *cnt_send, int32_t K,
                                                           int32_t V) {
  const int64_t n = (int64_t)V * K;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int v = (int)(i / K), k = (int)(i - (int64_t)v * K);
    cnt_send[koff[k] + (int64_t)v * Ksm] = n_wk[i];
  }
}

// host-layout <-> device-layout transposes for get_phi / set_phi
__global__ __launch_bounds__(256) void phiT_to_phi_kernel(const double *phiT, double *phi, int32_t K, int32_t Kp, int32_t V, double scale) {
  const int64_t n = (int64_t)V * K;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int k = (int)(i / V), v = (int)(i - (int64_t)k * V);
    const double x = phiT[(size_t)v * Kp + k];
    phi[i] = (scale == 1.0) ? x : x / scale;
  }
}
__global__ __launch_bounds__(256) void phi_to_phiT_kernel(const double *phi, double *phiT, int32_t K, int32_t Kp, int32_t V) {
  const int64_t n = (int64_t)V * K;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int v = (int)(i / K), k = (int)(i - (int64_t)v * K);
    phiT[(size_t)v * Kp + k] = phi[(size_t)k * V + v];
  }
}

// per-document topic histogram for getDocumentTopicMatrix (MSLDA:536-547)
__global__ __launch_bounds__(64) void doc_topic_kernel(const int64_t *doc_ptr, const int32_t *z, int64_t d_begin, int32_t K, int32_t *n_dk) {
  const int64_t d = d_begin + blockIdx.x;
  int32_t *row = n_dk + (size_t)blockIdx.x * K;
  for (int64_t i = doc_ptr[d] + threadIdx.x; i < doc_ptr[d + 1]; i += 64) atomicAdd(&row[z[i]], 1);
}

// paranoid invariants (UPLDA:299-338): counts >= 0, column sums == n_k, total == N
__global__ __launch_bounds__(256) void invariants_kernel(const int32_t *n_wk, int64_t n, int32_t K, unsigned long long *total,
                                                         int32_t *colsum, uint32_t *flags) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long t = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t c = n_wk[i];
    if (c < 0) atomicOr(flags, 1u);
    if (c) { t += (unsigned long long)c; atomicAdd(&colsum[i % K], c); }
  }
  if (t) atomicAdd(total, t);
}

// ---- debug kernels (parity tests of the primitives) ----------------------------
__global__ void debug_philox_kernel(int64_t n, const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U4 o = philox4x32_10(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], key[2 * i], key[2 * i + 1]);
  out[4 * i] = o.x; out[4 * i + 1] = o.y; out[4 * i + 2] = o.z; out[4 * i + 3] = o.w;
}
__global__ void debug_math_kernel(int op, int64_t n, const double *x, const double *y, double *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r;
  switch (op) {
    case 0: r = strict_log(x[i]); break;
    case 1: r = strict_pow(x[i], y[i]); break;
    case 2: r = sqrt(x[i]); break;
    default: r = x[i] / y[i]; break;
  }
  out[i] = r;
}
__global__ void debug_draw_kernel(int kind, uint64_t seed, uint32_t iteration, uint32_t purpose, uint64_t elem0, int64_t n,
                                  const double *shape, double *out, uint32_t *status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  DrawStream rs(seed, iteration, purpose, elem0 + (uint64_t)i);
  double r;
  if (kind == 0) r = rs.next_double();
  else if (kind == 1) r = rs.next_gaussian();
  else if (kind == 2 || !(shape[i] > 0)) {
    if (shape[i] > 0) r = rgamma(rs, shape[i]);
    else { r = __builtin_nan(""); atomicOr(status, ST_BAD_SHAPE); }
  } else {                                                           // kind 3: as the theta and Phi kernels draw
    if (!rgamma_first_try(seed, iteration, purpose, elem0 + (uint64_t)i, shape[i], r)) r = rgamma(rs, shape[i]);
    else if (kind == 4) r = -r;                                      // kind 4 marks the draws the first try settled
  }
  if (rs.exhausted) atomicOr(status, ST_RNG_EXHAUSTED);
  out[i] = r;
}

}  // namespace ggs

#include "ggs_z_kernel.hpp"
