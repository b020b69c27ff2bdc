// ggs_loglik.hpp -- UncollapsedParallelLDA.modelLogLikelihood (UPLDA:1644-1758) on the device: the
// Dirichlet-multinomial log likelihood of the topic assignments from the count matrices.  The Java
// driver prints it every `interval` iterations; computed on the host it costs a copy-back of 60 MB
// and ~1 s of lgamma calls at the benchmark size, three hundred sweeps' worth.  SURVEY 8(f)-2.
//
//   document side  sum_d [ sum_{k: n_dk > 0} (lgS(alpha_k + n_dk) - lgS(alpha_k)) - lgS(alphaSum + N_d) ]   UPLDA:1674-1691
//   topic side     sum_{(w,k): n_wk > 0} (lgS(beta + n_wk) - lgS(beta)) - sum_k lgS(V*beta + n_k)           UPLDA:1701-1747
//   + D*lgS(alphaSum) on the document side, + K*lgS(V*beta) on the topic side
//
// lgS is MALLET's Dirichlet.logGammaStirling (MALLET 2.0.8, not in the reference tree; restated from
// its published source: shift z up to >= 2, Stirling series to 1/(1260 z^5), undo the shift).
// A diagnostic, not sampler state: the reference adds ~15 M terms of mixed sign in one running
// double; here partial sums are reduced in a FIXED tree (run-to-run identical, not order-identical
// to Java), so against the oracle's sequential sum the result agrees to ~1e-12 relative (the sequential sum's own
// rounding error grows with the number of terms: 5e-11 measured at K=1024 and 18 M tokens), and the
// test states that tolerance.
#pragma once
#include "ggs_device_math.hpp"

namespace ggs {

__device__ __forceinline__ double log_gamma_stirling(double z) {
  constexpr double kHalfLogTwoPi = 0.91893853320467274178;           // Math.log(2 * Math.PI) / 2
  int shift = 0;
  while (z < 2) { z += 1.0; ++shift; }
  double result = kHalfLogTwoPi + (z - 0.5) * strict_log(z) - z + 1 / (12 * z) - 1 / (360 * z * z * z) + 1 / (1260 * z * z * z * z * z);
  while (shift > 0) { --shift; z -= 1.0; result -= strict_log(z); }
  return result;
}

constexpr int kLLBlock = 256;

// block-wide sum in a fixed order: lane tree, then wave 0 adds the wave partials in index order
__device__ __forceinline__ double ll_block_sum(double v, double *wave_part) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
  if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < kLLBlock / 64; ++w) t += wave_part[w];
  __syncthreads();
  return t;                                                           // valid in thread 0
}

// one wave per document (4 documents per block): LDS histogram of its z, then the document's term
__global__ __launch_bounds__(kLLBlock) void ll_docs_kernel(const int64_t *doc_ptr, const int32_t *z, const double *alpha, double alpha_sum,
                                                            int64_t num_docs, int32_t K, double *block_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double wave_part[kLLBlock / 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int32_t *hist = reinterpret_cast<int32_t *>(smem) + (size_t)wave * K;
  const int64_t d = (int64_t)blockIdx.x * (kLLBlock / 64) + wave;
  double v = 0.0;
  if (d < num_docs) {
    const int64_t beg = doc_ptr[d], end = doc_ptr[d + 1];
    for (int k = lane; k < K; k += 64) hist[k] = 0;
    for (int64_t i = beg + lane; i < end; i += 64) atomicAdd(&hist[z[i]], 1);   // one wave: LDS operations are ordered
    for (int k = lane; k < K; k += 64) {
      const int32_t n = hist[k];
      if (n > 0) v += log_gamma_stirling(alpha[k] + n) - log_gamma_stirling(alpha[k]);
    }
    if (lane == 0) v -= log_gamma_stirling(alpha_sum + (double)(end - beg));
  }
  const double t = ll_block_sum(v, wave_part);
  if (threadIdx.x == 0) block_out[blockIdx.x] = t;
}

// grid-stride over n_wk; also counts the non-zero cells (their -lgS(beta) is added once by the host)
__global__ __launch_bounds__(kLLBlock) void ll_types_kernel(const int32_t *n_wk, int64_t n, double beta, double *block_out, unsigned long long *nonzero) {
  __shared__ double wave_part[kLLBlock / 64];
  double v = 0.0;
  unsigned int nz = 0;
  const int64_t stride = (int64_t)gridDim.x * kLLBlock;
  for (int64_t i = (int64_t)blockIdx.x * kLLBlock + threadIdx.x; i < n; i += stride) {
    const int32_t c = n_wk[i];
    if (c != 0) { v += log_gamma_stirling(beta + c); ++nz; }
  }
  const double t = ll_block_sum(v, wave_part);
  if (threadIdx.x == 0) block_out[blockIdx.x] = t;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) nz += __shfl_down(nz, d);
  if ((threadIdx.x & 63) == 0 && nz) atomicAdd(nonzero, (unsigned long long)nz);
}

// one block: out[0] = document side, out[1] = topic side (each with its constant term), every sum in index order
// over a fixed number of partials
__global__ __launch_bounds__(kLLBlock) void ll_finish_kernel(const double *doc_part, int64_t n_doc_part, const double *type_part, int64_t n_type_part,
                                                              const int32_t *n_k, int32_t K, double vbeta, double alpha_sum, double beta,
                                                              int64_t num_docs, const unsigned long long *nonzero, double *out) {
  __shared__ double wave_part[kLLBlock / 64];
  double v = 0.0;
  for (int64_t i = threadIdx.x; i < n_doc_part; i += kLLBlock) v += doc_part[i];
  double t = ll_block_sum(v, wave_part);
  if (threadIdx.x == 0) out[0] = t + (double)num_docs * log_gamma_stirling(alpha_sum);                    // UPLDA:1694
  v = 0.0;
  for (int64_t i = threadIdx.x; i < n_type_part; i += kLLBlock) v += type_part[i];
  for (int k = threadIdx.x; k < K; k += kLLBlock) v -= log_gamma_stirling(vbeta + n_k[k]);                 // UPLDA:1724-1728
  t = ll_block_sum(v, wave_part);
  if (threadIdx.x == 0)
    out[1] = t + log_gamma_stirling(vbeta) * K - log_gamma_stirling(beta) * (double)*nonzero;              // UPLDA:1742-1747
}

// ------------------------------------------------------------------------------------------------
// computeLogPosterior (UPLDA:1573-1634; the LDA log posterior of Doss and George 2025, the quantity the
// reference's paper tracks).  The Java loop builds a dense K x V count matrix per document; its sum over
// (k, v) of m_djt * logPhi is simply a sum over the document's tokens:
//
//   document side  sum_tokens log(phi[z][w] + EPS) + sum_d sum_k (n_dk + alpha_k - 1) * log(theta[d][k] + EPS)   :1604-1619
//   topic side     (beta - 1) * sum_{k,v} log(phi[k][v] + EPS)                                                    :1622-1628
//
// with EPS = 1e-12 and theta = the rows the last z step used.  Same reduction as above: fixed tree, ~1e-12
// relative against the Java-order loop.  Documents without tokens have no theta row in the reference either
// (GGS:52-53 leaves the row as allocated, zeros): their theta term uses log(0 + EPS), as in Java.
constexpr double kLogPostEps = 1e-12;

__global__ __launch_bounds__(kLLBlock) void lp_docs_kernel(const int64_t *doc_ptr, const int32_t *tok, const int32_t *z, const double *alpha,
                                                            const double *theta, const double *phiT, int64_t num_docs, int32_t K, int32_t Kp,
                                                            double *block_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double wave_part[kLLBlock / 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int32_t *hist = reinterpret_cast<int32_t *>(smem) + (size_t)wave * K;
  const int64_t d = (int64_t)blockIdx.x * (kLLBlock / 64) + wave;
  double v = 0.0;
  if (d < num_docs) {
    const int64_t beg = doc_ptr[d], end = doc_ptr[d + 1];
    for (int k = lane; k < K; k += 64) hist[k] = 0;
    for (int64_t i = beg + lane; i < end; i += 64) {
      const int32_t t = z[i];
      atomicAdd(&hist[t], 1);
      v += strict_log(phiT[(size_t)tok[i] * Kp + t] + kLogPostEps);
    }
    for (int k = lane; k < K; k += 64)
      v += ((double)hist[k] + alpha[k] - 1.0) * strict_log(theta[(size_t)d * K + k] + kLogPostEps);
  }
  const double t = ll_block_sum(v, wave_part);
  if (threadIdx.x == 0) block_out[blockIdx.x] = t;
}

__global__ __launch_bounds__(kLLBlock) void lp_phi_kernel(const double *phiT, int64_t V, int32_t K, int32_t Kp, double *block_out) {
  __shared__ double wave_part[kLLBlock / 64];
  double v = 0.0;
  const int64_t n = V * K, stride = (int64_t)gridDim.x * kLLBlock;
  for (int64_t i = (int64_t)blockIdx.x * kLLBlock + threadIdx.x; i < n; i += stride) {
    const int64_t w = i / K;
    v += strict_log(phiT[w * Kp + (i - w * K)] + kLogPostEps);
  }
  const double t = ll_block_sum(v, wave_part);
  if (threadIdx.x == 0) block_out[blockIdx.x] = t;
}

__global__ __launch_bounds__(kLLBlock) void lp_finish_kernel(const double *doc_part, int64_t n_doc_part, const double *phi_part, int64_t n_phi_part,
                                                              double beta, double *out) {
  __shared__ double wave_part[kLLBlock / 64];
  double v = 0.0;
  for (int64_t i = threadIdx.x; i < n_doc_part; i += kLLBlock) v += doc_part[i];
  double t = ll_block_sum(v, wave_part);
  if (threadIdx.x == 0) out[0] = t;
  v = 0.0;
  for (int64_t i = threadIdx.x; i < n_phi_part; i += kLLBlock) v += phi_part[i];
  t = ll_block_sum(v, wave_part);
  if (threadIdx.x == 0) out[1] = (beta - 1.0) * t;
}

}  // namespace ggs
