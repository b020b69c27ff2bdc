// ggs_heldout.hpp -- MarginalProbEstimatorPlain.evaluateLeftToRight (topics/MarginalProbEstimatorPlain.java:85-121,
// 123-519; usingResampling = false, :125) on the device: the left-to-right held-out log likelihood the sampling
// loop computes on the test set every diagnostic iteration (UPLDA:604-611,677-682,840-844; 100 particles, :615).
// In Java it is numParticles x test tokens x K serial work per evaluation.  SURVEY 8(f)-2.
//
// One particle's pass over one document is sequential (its topic counts run along the positions); particles and
// documents are independent.  A wave = 64 particles of one document, walking the positions together, so the word's
// typeTopicCounts row is wave-uniform: it is read once, its non-zero cells found with a ballot, and only those enter
// the score loop -- a zero count contributes +0.0 to topicTermMass and nothing to the walk (MPE:352-365,409-415), so
// skipping it is exact.  cachedCoefficients[k] is a function of (k, this particle's count of k) alone (MPE:78,502-504,
// 514-519): (alpha_k + n) / (tokensPerTopic_k + betaSum) -- a block-wide LDS table for small n (as deep as LDS allows
// without costing a wave: 48 at K=100) and the same IEEE division beyond, which leaves the count itself, 1 or 2 bytes
// per (particle, topic), as the only per-particle state in LDS.
//
// The reference draws from a clock-seeded Randoms (MPE:64,87): the stream is ours -- purpose GGS_PURPOSE_HELDOUT,
// element = global test document * numParticles + particle, one uniform per in-vocabulary token, in sequence.
// Bit-identical to oracle/ggs_oracle.c:orc_heldout_log_likelihood, per document and in total (the total is added in
// document order on the host from the per-document values).
#pragma once
#include "ggs_kernels.hpp"

namespace ggs {

struct HeldoutParams {
  const int64_t *doc_ptr;   // test documents [D+1]
  const int32_t *tok;       // test tokens; ids >= V are out of vocabulary and skipped (MPE:341-345)
  const int32_t *n_wk;      // typeTopicCounts [V][K]
  const double *tab;        // [0] smoothingOnlyMass, then alpha[K], then denom[K] = tokensPerTopic + betaSum
  double *probs;            // wordProbabilities of the batch: (doc_ptr[d] - doc_ptr[d0]) * P + particle * len_d + position
  double *doc_ll;           // [D]
  uint32_t *status;
  double beta, alpha_sum;
  uint64_t seed;
  uint32_t iteration;
  const int32_t *docs;      // the particle kernel's documents (ids within [d0, d1)), n_docs of them
  int64_t n_docs;
  int64_t d0, d1, doc_base; // this batch covers test documents [d0, d1); the reduce kernel takes all of them
  int32_t K, V, P, blocks_per_doc, waves;
  int32_t cap;              // counts below cap take their coefficient from the LDS table [K][cap]
  void *cnt_spill;          // SPILL: the per-particle topic counts [resident wave][K][64] in global memory
};

constexpr int kHeldoutMaxWaves = 16;
#ifndef GGS_HELDOUT_BATCH
#define GGS_HELDOUT_BATCH 8
#endif
constexpr int kHeldoutCoefCaps[] = {64, 56, 48, 40, 32, 24, 16, 8, 4, 1};   // table depths the host chooses from (the deepest that costs no wave)
constexpr int kHeldoutBatch = GGS_HELDOUT_BATCH;   // cells per round of LDS reads (must divide 64; 4 and 8 measure the same, 16 slower)

// alpha, denominators and smoothingOnlyMass (MPE:63,75-78), one thread: the mass is one running double
__global__ void heldout_setup_kernel(const double *alpha, const int32_t *n_k, double beta, double beta_sum, int32_t K, double *tab) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double smoothing = 0;
  for (int k = 0; k < K; ++k) {
    const double denom = (double)n_k[k] + beta_sum;
    smoothing += alpha[k] * beta / denom;
    tab[1 + k] = alpha[k];
    tab[1 + K + k] = denom;
  }
  tab[0] = smoothing;
}

// CntT: the particle's per-topic counts -- one byte each for documents of at most 255 tokens, two up to 65 535, four
// beyond.  The kernel is bound by instruction issue (one instruction per ~8 cycles and wave), so what matters is how many
// waves share a SIMD, and that is set by these counts' LDS footprint.  SPILL: where K * 64 counts per wave do not fit LDS
// at all (more than 1704 topics, or more than 1024 with two-byte counts) they live in global memory instead, one
// [K][64] block per RESIDENT wave (the grid is then persistent and a wave strides over its units); same arithmetic.
template <typename CntT, bool SPILL = false>
__global__ __launch_bounds__(kHeldoutMaxWaves * 64) void heldout_particles_kernel(HeldoutParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int K = p.K;
  double *alpha_s = reinterpret_cast<double *>(smem);
  double *denom_s = alpha_s + K;
  const int cap = p.cap;
  double *coef_s = denom_s + K;                                         // [K][cap]: (alpha_k + n) / denom_k for n < cap
  for (int k = threadIdx.x; k < 2 * K; k += blockDim.x) alpha_s[k] = p.tab[1 + k];
  __syncthreads();
  for (int i = threadIdx.x; i < K * cap; i += blockDim.x) {
    const int k = i / cap, n = i - k * cap;
    coef_s[i] = (alpha_s[k] + (double)n) / denom_s[k];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int Kpad = (K + 63) & ~63;
  // the current word's non-zero (topic, count) cells, compacted in topic order: what both passes iterate
  int2 *list_s = reinterpret_cast<int2 *>(coef_s + (size_t)K * cap) + (size_t)wave * Kpad;
  CntT *cnt_s = SPILL ? static_cast<CntT *>(p.cnt_spill) + ((size_t)blockIdx.x * p.waves + wave) * (size_t)K * 64
                      : reinterpret_cast<CntT *>(reinterpret_cast<int2 *>(coef_s + (size_t)K * cap) + (size_t)p.waves * Kpad) + (size_t)wave * K * 64;
  // The score loops are bound by instruction issue (a wave issues one instruction per ~8 cycles; SQ counters in
  // profiles/): the cells are compacted once per word so that the loops index them by a counter instead of peeling
  // bits off a ballot mask, and are taken kHeldoutBatch at a time -- all counts, then all coefficients, then the
  // ordered adds -- so that the dependent LDS reads (count -> coefficient) overlap.  The last batch is padded with cells
  // of count 0, which add +0.0 to the mass and subtract 0.0 in the walk.
  int kk[kHeldoutBatch];
  int32_t cc[kHeldoutBatch];
  double cf[kHeldoutBatch];
  auto next_cells = [&](const int2 cell, int j0) {                      // cells j0 .. j0 + kHeldoutBatch - 1 of `cell` (lane j = j-th cell)
#pragma unroll
    for (int j = 0; j < kHeldoutBatch; ++j) {
      kk[j] = __builtin_amdgcn_readlane(cell.x, j0 + j);
      cc[j] = __builtin_amdgcn_readlane(cell.y, j0 + j);
    }
    // cachedCoefficients of this lane's particle: from the table while the count is small; where some lane's count of
    // a cell is beyond the table, the same division for that cell (the same value for the lanes the table covers)
    int n[kHeldoutBatch];
    bool big = false;
#pragma unroll
    for (int j = 0; j < kHeldoutBatch; ++j) { n[j] = cnt_s[kk[j] * 64 + lane]; big |= n[j] >= cap; }
#pragma unroll
    for (int j = 0; j < kHeldoutBatch; ++j) cf[j] = coef_s[kk[j] * cap + (n[j] < cap ? n[j] : cap - 1)];
    if (__ballot(big)) {
#pragma unroll
      for (int j = 0; j < kHeldoutBatch; ++j)
        if (__ballot(n[j] >= cap)) cf[j] = (alpha_s[kk[j]] + (double)n[j]) / denom_s[kk[j]];
    }
  };
  const int64_t unit0 = (int64_t)blockIdx.x * p.waves + wave, n_units = p.n_docs * p.blocks_per_doc;
  // no block-wide barrier below: a wave simply leaves when it has no unit (left)
  for (int64_t unit = unit0; unit < n_units; unit += SPILL ? (int64_t)gridDim.x * p.waves : n_units) {
  const int64_t d = p.docs[unit / p.blocks_per_doc];
  const int particle = (int)(unit % p.blocks_per_doc) * 64 + lane;
  const bool live = particle < p.P;                                     // dead lanes compute along, store nothing
  const int64_t beg = p.doc_ptr[d], len = p.doc_ptr[d + 1] - beg;
  double *out = p.probs + (beg - p.doc_ptr[p.d0]) * p.P + (int64_t)particle * len;   // [particle][position]: the reduce's reads coalesce
  for (int k = 0; k < K; ++k) cnt_s[k * 64 + lane] = 0;
  const double smoothing = p.tab[0], beta = p.beta;
  const uint64_t elem = (uint64_t)(p.doc_base + d) * (uint64_t)p.P + (uint64_t)particle;
  double beta_mass = 0.0, u_odd = 0.0;
  int so_far = 0;                                                       // tokensSoFar: wave-uniform
  bool bad = false;
  for (int64_t limit = 0; limit < len; ++limit) {
    const int32_t type = __builtin_amdgcn_readfirstlane(p.tok[beg + limit]);
    if (type >= p.V) { if (live) out[limit] = 0.0; continue; }
    const int32_t *row = p.n_wk + (size_t)type * K;
    // the row's non-zero cells in topic order (zero cells add +0.0 to the mass and nothing to the walk: skipped)
    int cells = 0;
    for (int k0 = 0; k0 < K; k0 += 64) {
      const int32_t v = k0 + lane < K ? row[k0 + lane] : 0;
      const uint64_t m = __ballot(v != 0);
      const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (v != 0) list_s[cells + rank] = make_int2(k0 + lane, v);
      cells += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    // MPE:352-365: topicTermMass in topic order; lanes past the last cell hold (topic 0, count 0): +0.0
    double mass = 0.0;
    for (int c0 = 0; c0 < cells; c0 += 64) {
      const int2 cell = c0 + lane < cells ? list_s[c0 + lane] : make_int2(0, 0);
      const int lim = cells - c0 < 64 ? cells - c0 : 64;
      for (int j0 = 0; j0 < lim; j0 += kHeldoutBatch) {
        next_cells(cell, j0);
#pragma unroll
        for (int j = 0; j < kHeldoutBatch; ++j) mass += cf[j] * (double)cc[j];
      }
    }
    const double total = smoothing + beta_mass + mass;
    double u;
    if ((so_far & 1) == 0) {
      const U4 o = philox4x32_10((uint32_t)elem, (uint32_t)(elem >> 32), ((uint32_t)GGS_PURPOSE_HELDOUT << 24) | (uint32_t)(so_far >> 1),
                                 p.iteration, (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
      u = u53(o.x, o.y);
      u_odd = u53(o.z, o.w);
    } else {
      u = u_odd;
    }
    double sample = u * total;                                          // MPE:393
    if (live) out[limit] = total / (p.alpha_sum + (double)so_far);      // MPE:399-401
    ++so_far;
    int newTopic = -1;
    const bool in_term = sample < mass;
    if (__ballot(in_term)) {                                            // MPE:409-419, the same products again
      bool walking = in_term && sample > 0;
      for (int c0 = 0; c0 < cells && __ballot(walking); c0 += 64) {
        const int2 cell = c0 + lane < cells ? list_s[c0 + lane] : make_int2(0, 0);
        const int lim = cells - c0 < 64 ? cells - c0 : 64;
        for (int j0 = 0; j0 < lim && __ballot(walking); j0 += kHeldoutBatch) {
          next_cells(cell, j0);
#pragma unroll
          for (int j = 0; j < kHeldoutBatch; ++j) {
            const double score = cf[j] * (double)cc[j];
            if (walking) {
              sample -= score;
              if (!(sample > 0)) { newTopic = kk[j]; walking = false; }
            }
          }
        }
      }
    }
    if (!in_term) {
      sample -= mass;
      const bool in_beta = sample < beta_mass;
      if (in_beta) sample /= beta; else { sample -= beta_mass; sample /= beta; }
      if (__ballot(in_beta)) {                                          // MPE:423-440: this particle's topics, ascending
        bool walking = in_beta;
        for (int k0 = 0; k0 < K && __ballot(walking); k0 += 8) {
          int n[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) n[j] = k0 + j < K ? cnt_s[(k0 + j) * 64 + lane] : 0;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (walking && n[j] > 0) {
              sample -= (double)n[j] / denom_s[k0 + j];
              if (sample <= 0.0) { newTopic = k0 + j; walking = false; }
            }
        }
      }
      if (__ballot(!in_beta)) {                                         // MPE:442-460: the smoothing-only bucket
        bool walking = !in_beta;
        if (walking) { newTopic = 0; sample -= alpha_s[0] / denom_s[0]; }
        for (int k = 1; __ballot(walking && sample > 0.0); ++k) {
          if (k >= K) break;
          if (walking && sample > 0.0) { newTopic = k; sample -= alpha_s[k] / denom_s[k]; }
        }
        if (walking && sample > 0.0) newTopic = -1;                     // ran past the last topic (MPE:455)
      }
    }
    if (newTopic < 0) { bad = true; newTopic = 0; }                     // MPE:416,447,455,464-469 throw
    const int n_old = cnt_s[newTopic * 64 + lane];
    beta_mass -= beta * (double)n_old / denom_s[newTopic];              // MPE:474-475
    cnt_s[newTopic * 64 + lane] = (CntT)(n_old + 1);
    beta_mass += beta * (double)(n_old + 1) / denom_s[newTopic];        // MPE:506-507
  }
  if (bad && live) atomicOr(p.status, ST_INVALID_TOPIC);
  }
}

// MPE:102-116: per position the sum over the particles in particle order, log, minus log(numParticles); per document
// the sum over the positions in order.  One wave per document.
__global__ __launch_bounds__(256) void heldout_reduce_kernel(HeldoutParams p) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t d = p.d0 + (int64_t)blockIdx.x * 4 + wave;
  if (d >= p.d1) return;
  const int64_t beg = p.doc_ptr[d], len = p.doc_ptr[d + 1] - beg;
  const double *in = p.probs + (beg - p.doc_ptr[p.d0]) * p.P;
  const double log_particles = strict_log((double)p.P);
  double doc_ll = 0.0;
  for (int64_t pos0 = 0; pos0 < len; pos0 += 64) {
    const int64_t pos = pos0 + lane;
    double term = 0.0;
    if (pos < len) {
      double sum = 0.0;
      for (int q = 0; q < p.P; ++q) sum += in[(int64_t)q * len + pos];
      if (sum > 0.0) term = strict_log(sum) - log_particles;            // else: skipped, and x + 0.0 == x
    }
    const int n = (int)(len - pos0 < 64 ? len - pos0 : 64);
    for (int j = 0; j < n; ++j) doc_ll += __shfl(term, j);               // every lane runs the same chain
  }
  if (lane == 0) p.doc_ll[d] = doc_ll;
}

}  // namespace ggs
