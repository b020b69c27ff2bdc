// ggs_z_collapsed.hpp -- the count-form conditional (scheme=collapsed): ModifiedSimpleLDA.sampleTopicsForOneDoc,
// MSLDA:158-226, the in-tree twin of the MALLET SimpleLDA method that SerialCollapsedLDA runs (SerialCollapsedLDA.java:33,
// 159-172):   score[k] = (alpha_k + n_dk) * ((beta + n_wk) / (betaSum + n_k)),   sample = U * sum,   the same walk.
// SURVEY.md 8(a) row 9 / 8(f) item 4.  Two schedules:
//
//   serial    collapsed_serial_kernel: the reference's own schedule -- one chain over all tokens in (document, position)
//             order, the three count structures updated in place after every token, uniforms from ONE
//             java.util.Random stream (MALLET Randoms(seed).nextUniform(), restated as nextDouble(): SURVEY 8c).  A
//             single wave: the lanes compute the K scores of the token side by side, lane 0 adds them in k order,
//             draws, walks and moves the counts.  No parallelism across tokens by construction; it exists so that
//             BASELINE config 1 (the bundled cats corpus, K = 20) runs on the device bit for bit as the Java chain does.
//   parallel  pcgs_z_kernel<true> (ggs_z_pcgs.hpp): documents side by side, one lane per document, each sampled against
//             the counts AS THEY STOOD AT THE START OF THE SWEEP minus the token being resampled -- the AD-LDA
//             decomposition (ADLDA.java:176-332: per-worker copies, sampled independently, summed and copied back)
//             with one worker per document; the sweep's merge is the (word, z) histogram of the new assignments
//             (count_sorted_kernel; summed across GPUs by the exchange exactly as for ggs).  The sweep-start ratios
//             psi[w][k] = (beta + n_wk)/(betaSum + n_k) are materialised once per sweep (psi_kernel) in the phiT buffer,
//             so the z loop is the pcgs loop over a different matrix, plus one recomputed entry per token (its own old
//             topic, with the token removed).  Approximate in the way AD-LDA is; bit-identical to the oracle's
//             restatement of exactly this schedule, and within the north_star's +-1 % held-out log likelihood of the
//             serial chain (tests/test_collapsed_gpu.py).
#pragma once
#include "ggs_z_pcgs.hpp"

namespace ggs {

// psi[w][k] = (beta + n_wk[w][k]) / (betaSum + n_k[k]) into phiT [V][Kp]: the ratio of MSLDA:199-201, one division
__global__ __launch_bounds__(256) void psi_kernel(const int32_t *n_wk, const int32_t *n_k, double beta, double beta_sum, double *phiT, int32_t K,
                                                  int32_t Kp, int32_t V) {
  const int64_t n = (int64_t)V * K;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int v = (int)(i / K), k = (int)(i - (int64_t)v * K);
    phiT[(size_t)v * Kp + k] = (beta + (double)n_wk[i]) / (beta_sum + (double)n_k[k]);
  }
}

struct CollapsedSerialParams {
  const int64_t *doc_ptr;
  const int32_t *tok, *inv_perm;
  int32_t *z, *zw, *n_wk, *n_k;
  const double *alpha;
  uint64_t *lcg;               // the 48-bit state of java.util.Random, carried from sweep to sweep
  uint32_t *status;
  int64_t num_docs;
  double beta, beta_sum;
  int32_t K;
};

// LDS: K scores, K document counts, K topic totals.  The type-topic row of the current word is the one piece of state
// that lives in global memory while lanes other than the writer read it: lane 0 moves it with atomics (they execute at
// L2) and the readers use agent-scope loads (L1 bypassed), so a count changed one token ago is seen.
__global__ __launch_bounds__(64) void collapsed_serial_kernel(CollapsedSerialParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int K = p.K, lane = threadIdx.x;
  double *topicTermScores = reinterpret_cast<double *>(smem);
  int32_t *localTopicCounts = reinterpret_cast<int32_t *>(topicTermScores + K);
  int32_t *tokensPerTopic = localTopicCounts + K;
  uint64_t seed = *p.lcg;
  constexpr uint64_t kMult = 0x5DEECE66DULL, kMask = (1ULL << 48) - 1;
  for (int k = lane; k < K; k += 64) tokensPerTopic[k] = p.n_k[k];
  for (int64_t d = 0; d < p.num_docs; ++d) {
    const int64_t b = p.doc_ptr[d], e = p.doc_ptr[d + 1];
    for (int k = lane; k < K; k += 64) localTopicCounts[k] = 0;
    __syncthreads();
    if (lane == 0)
      for (int64_t i = b; i < e; ++i) localTopicCounts[p.z[i]]++;                  // MSLDA:167-169
    __syncthreads();
    for (int64_t i = b; i < e; ++i) {
      const int type = p.tok[i], oldTopic = p.z[i];
      int32_t *currentTypeTopicCounts = p.n_wk + (size_t)type * K;
      if (lane == 0) {                                                             // MSLDA:185-190
        localTopicCounts[oldTopic]--;
        tokensPerTopic[oldTopic]--;
        atomicSub(&currentTypeTopicCounts[oldTopic], 1);
      }
      __threadfence();
      __syncthreads();
      for (int k = lane; k < K; k += 64) {                                         // MSLDA:196-203, the K scores side by side
        const int32_t c = __hip_atomic_load(&currentTypeTopicCounts[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        topicTermScores[k] = (p.alpha[k] + (double)localTopicCounts[k]) * ((p.beta + (double)c) / (p.beta_sum + (double)tokensPerTopic[k]));
      }
      __syncthreads();
      if (lane == 0) {
        double sum = 0.0;
        for (int k = 0; k < K; ++k) sum += topicTermScores[k];                     // in k order, as the Java loop adds them
        seed = (seed * kMult + 0xBULL) & kMask;                                    // Random.nextDouble(): next(26), next(27)
        const uint64_t hi26 = seed >> 22;
        seed = (seed * kMult + 0xBULL) & kMask;
        const uint64_t lo27 = seed >> 21;
        double sample = (double)((hi26 << 27) + lo27) * 0x1.0p-53 * sum;           // MSLDA:206
        int newTopic = -1;
        while (sample > 0.0) {                                                     // MSLDA:209-213
          newTopic++;
          if (newTopic >= K) break;
          sample -= topicTermScores[newTopic];
        }
        if (newTopic < 0 || newTopic >= K) {                                       // MSLDA:216-218 (and the index past K Java would throw on)
          atomicOr(p.status, ST_INVALID_TOPIC);
          newTopic = newTopic < 0 ? 0 : K - 1;
        }
        p.z[i] = newTopic;                                                         // MSLDA:221-225
        p.zw[p.inv_perm[i]] = newTopic;
        localTopicCounts[newTopic]++;
        tokensPerTopic[newTopic]++;
        atomicAdd(&currentTypeTopicCounts[newTopic], 1);
      }
      __threadfence();
      __syncthreads();
    }
  }
  __syncthreads();
  for (int k = lane; k < K; k += 64) p.n_k[k] = tokensPerTopic[k];
  if (lane == 0) *p.lcg = seed;
}

}  // namespace ggs
