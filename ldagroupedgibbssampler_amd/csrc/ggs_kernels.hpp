// ggs_kernels.hpp -- the HIP kernels of the Grouped Gibbs sweep (gfx950).
//
// Device layouts (private to the library):
//   tok[N]            int32   word ids, CSR order
//   z[N]              int32   topic assignments
//   chunk_*[C]                z-kernel work items: <=64 consecutive tokens of ONE document
//   theta[D][K]       fp64    thetaMatrix rows (GGS:72)
//   phiT[V][Kp]       fp64    TRANSPOSE of the Java phi[K][V] (UPLDA:69), row pitch Kp = K
//                             rounded up to even, so one token reads one contiguous row
//   n_wk[V][K]        int32   typeTopicCounts layout (MSLDA:73)
//   delta[V][K]       int32   batchLocalTopicTypeUpdates (UPLDA:102), transposed
//   n_k[K]            int32   tokensPerTopic
//
// Java keeps every running sum sequential in index order (sum += ...), and so do
// these kernels: wherever the reference adds K or V doubles one after another, ONE
// lane walks them in that order.  Parallelism comes from doing many such walks side
// by side (one lane per token / document / topic), never from re-associating a sum.
#pragma once
#include "ggs_device_math.hpp"

namespace ggs {

constexpr double kJavaMinValue = 4.9e-324;  // Double.MIN_VALUE, ParallelDirichlet.java:64

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
//   phase 3  all lanes: Gamma(partition*magnitude) draws, one (doc, k) pair each
//   phase 4  one lane per document: sum of the gammas, in k order
//   phase 5  all lanes: normalise, clamp <=0 to Double.MIN_VALUE, coalesced store
// LDS: hist int32 [K][BP], gam fp64 [K][BP], BP = docs_per_block | 1 (odd => the
// k-major walks of phases 2, 4, 5 are bank-conflict free).
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
};

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void theta_kernel(ThetaParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int K = p.K, B = p.docs_per_block, BP = B | 1;
  double *gam = reinterpret_cast<double *>(smem);                    // [K][BP]
  double *mag = gam + (size_t)K * BP;                                // [B]
  double *tot = mag + B;                                             // [B]
  int32_t *hist = reinterpret_cast<int32_t *>(tot + B);              // [K][BP]
  int32_t *len = hist + (size_t)K * BP;                              // [B]

  const int tid = threadIdx.x;
  const int64_t d0 = (int64_t)blockIdx.x * B;
  const int nb = (int)min((int64_t)B, p.num_docs - d0);

  for (int i = tid; i < K * BP; i += BLOCK) hist[i] = 0;
  __syncthreads();

  constexpr int NW = BLOCK / 64;
  const int wave = tid >> 6, lane = tid & 63;
  for (int b = wave; b < nb; b += NW) {
    const int64_t beg = p.doc_ptr[d0 + b], end = p.doc_ptr[d0 + b + 1];
    if (lane == 0) len[b] = (int32_t)(end - beg);
    for (int64_t i = beg + lane; i < end; i += 64) atomicAdd(&hist[p.z[i] * BP + b], 1);
  }
  __syncthreads();

  if (tid < nb && len[tid] > 0) {
    double m = 0;
    for (int k = 0; k < K; ++k) m += (double)hist[k * BP + tid] + p.alpha[k];  // Dirichlet(double[]): magnitude
    mag[tid] = m;
  }
  __syncthreads();

  for (int i = tid; i < nb * K; i += BLOCK) {
    const int k = i / nb, b = i - k * nb;
    if (len[b] == 0) continue;                                       // GGS:52-53
    const double pk = (double)hist[k * BP + b] + p.alpha[k];         // GGS:68
    const double m = mag[b];
    const double shape = (pk / m) * m;                               // partition[i] * magnitude
    double g;
    if (shape > 0) {
      DrawStream rs(p.seed, p.iteration, GGS_PURPOSE_THETA, (uint64_t)(p.doc_base + d0 + b) * (uint64_t)K + (uint64_t)k);
      g = rgamma(rs, shape);
      if (rs.exhausted) atomicOr(p.status, ST_RNG_EXHAUSTED);
    } else {
      g = __builtin_nan("");
      atomicOr(p.status, ST_BAD_SHAPE);
    }
    gam[k * BP + b] = g;
  }
  __syncthreads();

  if (tid < nb && len[tid] > 0) {
    double s = 0;
    for (int k = 0; k < K; ++k) s += gam[k * BP + tid];              // ParallelDirichlet.java:53-57
    tot[tid] = s;
  }
  __syncthreads();

  for (int i = tid; i < nb * K; i += BLOCK) {
    const int b = i / K, k = i - b * K;
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

// ------------------------------------------------------------------------------
// K3+K4: the token loop (GGS:79-130).  One wave = one chunk of <=64 tokens of one
// document; lane t owns token t.
//   stage   the wave copies the chunk's phiT rows HBM -> LDS with 16-byte,
//           fully coalesced loads (a row is Kp*8 contiguous bytes)
//   pass 1  lane t: sum = sum_k theta[k]*phi[k][w_t], k ascending (GGS:96-101)
//   draw    U from Philox (GGS:107), sample = U*sum
//   pass 2  lane t: the "while (sample > 0) sample -= score[++k]" walk (GGS:108-113)
//   update  z store; -1/+1 on delta[w][old/new] (GGS:93,129 -> UPLDA:1547-1557)
// LDS row pitch = pitch16*16 bytes with pitch16 odd, so the 16-byte per-lane reads
// of passes 1-2 (lane t reads row t) are bank-conflict free.
// ------------------------------------------------------------------------------
struct ZParams {
  const int32_t *tok;
  int32_t *z;
  const int64_t *chunk_start;  // local token index of the chunk's first token
  const int32_t *chunk_doc;    // local document index
  const int32_t *chunk_len;
  const double *theta;
  const double *phiT;
  int32_t *delta;
  uint32_t *status;
  int64_t tok_base;
  uint64_t seed;
  uint32_t iteration;
  int32_t K, Kp, pitch16;
};

struct alignas(16) D2 { double a, b; };

__global__ __launch_bounds__(64) void z_kernel(ZParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  const int64_t start = p.chunk_start[c];
  const int len = p.chunk_len[c];
  const int doc = p.chunk_doc[c];
  const int K = p.K, Kp = p.Kp;
  const int upr = Kp >> 1;                 // 16-byte units per phi row
  const int pitch = p.pitch16 * 16;        // LDS row pitch, bytes

  int w = 0, old_topic = 0;
  if (lane < len) { w = p.tok[start + lane]; old_topic = p.z[start + lane]; }

  // ---- stage rows: unit u = (row, col), row-major over the chunk ----
  {
    const int total = len * upr;
    int row = lane / upr, col = lane - row * upr;
    const int drow = 64 / upr, dcol = 64 - drow * upr;
    constexpr int UNR = 8;
    for (int base = 0; base < total; base += 64 * UNR) {   // wave-uniform trip count: every lane takes part in the shuffles
      D2 v[UNR];
      int dst[UNR];
#pragma unroll
      for (int j = 0; j < UNR; ++j) {
        const int u = base + 64 * j + lane;
        const int wr = __shfl(w, row < 64 ? row : 63);
        dst[j] = (u < total) ? row * pitch + col * 16 : -1;
        if (u < total) v[j] = *reinterpret_cast<const D2 *>(p.phiT + (size_t)wr * Kp + 2 * col);
        row += drow; col += dcol;
        if (col >= upr) { col -= upr; ++row; }
      }
#pragma unroll
      for (int j = 0; j < UNR; ++j)
        if (dst[j] >= 0) *reinterpret_cast<D2 *>(smem + dst[j]) = v[j];
    }
  }
  __syncthreads();

  if (lane < len) {
    const double *__restrict__ th = p.theta + (size_t)doc * K;   // wave-uniform row
    const unsigned char *rowp = smem + lane * pitch;
    const int Ke = K & ~1;
    double sum = 0.0;
    for (int k = 0; k < Ke; k += 2) {
      const D2 ph = *reinterpret_cast<const D2 *>(rowp + k * 8);
      const double s0 = th[k] * ph.a;
      sum += s0;
      const double s1 = th[k + 1] * ph.b;
      sum += s1;
    }
    if (K & 1) { const double s0 = th[K - 1] * *reinterpret_cast<const double *>(rowp + (K - 1) * 8); sum += s0; }

    const U4 o = philox4x32_10((uint32_t)((uint64_t)(p.tok_base + start + lane)),
                               (uint32_t)((uint64_t)(p.tok_base + start + lane) >> 32),
                               (uint32_t)GGS_PURPOSE_Z << 24, p.iteration, (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
    const double U = u53(o.x, o.y);
    double sample = U * sum;
    int new_topic = -1;
    for (int k = 0; k < K; ++k) {
      if (!(sample > 0.0)) break;
      new_topic = k;
      sample -= th[k] * *reinterpret_cast<const double *>(rowp + k * 8);
    }
    if (new_topic < 0 || sample > 0.0) {        // GGS:116-118 (and running past K)
      atomicOr(p.status, ST_INVALID_TOPIC);
      new_topic = new_topic < 0 ? 0 : K - 1;
    }
    p.z[start + lane] = new_topic;
    if (new_topic != old_topic) {
      atomicAdd(&p.delta[(size_t)w * K + old_topic], -1);
      atomicAdd(&p.delta[(size_t)w * K + new_topic], 1);
    }
  }
}

// ------------------------------------------------------------------------------
// K5: updateCounts (UPLDA:1158-1182): n_wk += delta, delta = 0, negative check.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_kernel(int32_t *n_wk, int32_t *delta, int64_t n, uint32_t *status) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      int4 d = *reinterpret_cast<int4 *>(delta + i);
      if ((d.x | d.y | d.z | d.w) != 0) {
        int4 c = *reinterpret_cast<int4 *>(n_wk + i);
        c.x += d.x; c.y += d.y; c.z += d.z; c.w += d.w;
        if ((c.x | c.y | c.z | c.w) < 0) atomicOr(status, ST_NEGATIVE_COUNT);
        *reinterpret_cast<int4 *>(n_wk + i) = c;
        *reinterpret_cast<int4 *>(delta + i) = make_int4(0, 0, 0, 0);
      }
    } else {
      for (int64_t j = i; j < n; ++j) {
        const int32_t d = delta[j];
        if (d) { const int32_t c = n_wk[j] + d; n_wk[j] = c; delta[j] = 0; if (c < 0) atomicOr(status, ST_NEGATIVE_COUNT); }
      }
    }
  }
}

// count rebuild for set_z / init (UPLDA:471-474, 1821-1825)
__global__ __launch_bounds__(256) void count_kernel(const int32_t *tok, const int32_t *z, int64_t n, int32_t K, int32_t *n_wk) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    atomicAdd(&n_wk[(size_t)tok[i] * K + z[i]], 1);
}

// ------------------------------------------------------------------------------
// K6/K8: Phi draw (GGS:182-198 / MarsagliaSparseDirichlet.java:31-55).
//   phi_magnitude  lane per topic: magnitude_k = sum_v (beta + n_kv), v ascending;
//                  also tokensPerTopic n_k = sum_v n_kv
//   phi_gamma      lane per (v,k): Gamma(partition*magnitude) -> phiT (unnormalised)
//   phi_total      lane per topic: sum_v gamma, v ascending
//   phi_normalise  lane per (v,k): divide, clamp, optional running phiMean +=
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void phi_magnitude_kernel(const int32_t *n_wk, int32_t K, int32_t V, double beta,
                                                           double *mag, int32_t *n_k) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= K) return;
  double m = 0;
  int32_t nk = 0;
  const int32_t *col = n_wk + k;
  int v = 0;
  for (; v + 8 <= V; v += 8) {
    int32_t c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = col[(size_t)(v + j) * K];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m += beta + (double)c[j]; nk += c[j]; }
  }
  for (; v < V; ++v) { const int32_t c = col[(size_t)v * K]; m += beta + (double)c; nk += c; }
  mag[k] = m;
  n_k[k] = nk;
}

struct PhiGammaParams {
  const int32_t *n_wk;
  const double *mag;   // per topic (sweep draw) -- unused for the initial draw
  double *phiT;
  uint32_t *status;
  uint64_t seed;
  uint32_t iteration, purpose;
  int32_t K, Kp, V;
  double beta;         // sweep draw: shape = ((beta+n)/mag)*mag
  double prior_pm;     // initial draw: partition*magnitude = (1.0/V)*(V*beta)
  int32_t initial;
};

__global__ __launch_bounds__(256) void phi_gamma_kernel(PhiGammaParams p) {
  const int64_t n = (int64_t)p.V * p.K;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int v = (int)(i / p.K), k = (int)(i - (int64_t)v * p.K);
    const int32_t cnt = p.n_wk[i];
    double shape;
    if (p.initial) {
      shape = (cnt == 0) ? p.prior_pm : p.prior_pm + (double)cnt;   // MarsagliaSparseDirichlet.java:37-41
    } else {
      const double pk = p.beta + (double)cnt;                       // GGS:188
      const double m = p.mag[k];
      shape = (pk / m) * m;
    }
    double g;
    if (shape > 0) {
      DrawStream rs(p.seed, p.iteration, p.purpose, (uint64_t)k * (uint64_t)p.V + (uint64_t)v);
      g = rgamma(rs, shape);
      if (rs.exhausted) atomicOr(p.status, ST_RNG_EXHAUSTED);
    } else {
      g = __builtin_nan("");
      atomicOr(p.status, ST_BAD_SHAPE);
    }
    p.phiT[(size_t)v * p.Kp + k] = g;
  }
}

__global__ __launch_bounds__(64) void phi_total_kernel(const double *phiT, int32_t K, int32_t Kp, int32_t V, double *tot) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= K) return;
  double s = 0;
  const double *col = phiT + k;
  int v = 0;
  for (; v + 8 <= V; v += 8) {
    double g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = col[(size_t)(v + j) * Kp];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += g[j];
  }
  for (; v < V; ++v) s += col[(size_t)v * Kp];
  tot[k] = s;
}

__global__ __launch_bounds__(256) void phi_normalise_kernel(double *phiT, const double *tot, int32_t K, int32_t Kp, int32_t V,
                                                            double *phi_mean /* [V][K] or null */) {
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

// paranoid invariants (UPLDA:299-338): counts >= 0, column sums == n_k, total == N, deltas all zero
__global__ __launch_bounds__(256) void invariants_kernel(const int32_t *n_wk, const int32_t *delta, int64_t n, int32_t K,
                                                         unsigned long long *total, int32_t *colsum, uint32_t *flags) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long t = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t c = n_wk[i];
    if (c < 0) atomicOr(flags, 1u);
    if (delta[i] != 0) atomicOr(flags, 2u);
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
  else {
    if (shape[i] > 0) r = rgamma(rs, shape[i]);
    else { r = __builtin_nan(""); atomicOr(status, ST_BAD_SHAPE); }
  }
  if (rs.exhausted) atomicOr(status, ST_RNG_EXHAUSTED);
  out[i] = r;
}

}  // namespace ggs
