/*
 * ggs_oracle.h -- CPU ORACLE for the Grouped Gibbs Sampler hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke
 * check in __graft_entry__.py and bench.py's cpu_baseline leg may load it.
 * The product path (ldagroupedgibbssampler_amd/, libggs_hip.so) never links,
 * imports or falls back to anything in this directory.
 *
 * It is a plain-C restatement, in double precision and in the Java operation
 * order, of the reference's arithmetic (paths relative to the reference
 * checkout, src/main/java/cc/mallet/...):
 *   topics/LDAGroupedGibbsSampler.java:47-132   one document's GGS step
 *   topics/LDAGroupedGibbsSampler.java:182-198  Phi re-draw
 *   types/ParallelDirichlet.java:46-70          gamma-normalise Dirichlet draw
 *   util/ParallelRandoms.java:60-70,148-159     Marsaglia-Tsang gamma
 *   types/MarsagliaSparseDirichlet.java:31-55   initial Phi
 *   topics/UncollapsedParallelLDA.java:357-482  init z, count build
 *   topics/UncollapsedParallelLDA.java:1107-1221 delta merge
 *   topics/ModifiedSimpleLDA.java:158-226       count-form (collapsed) step
 *   topics/UncollapsedParallelLDA.java:1466-1544 the z loop of scheme=pcgs (partially collapsed)
 *   topics/MarginalProbEstimatorPlain.java:51-121,123-519 left-to-right held-out log likelihood
 *
 * PARITY STATUS: the reference's GGS path draws from
 * ThreadLocalRandom / a nanoTime-seeded xorshift and therefore has no
 * reproducible output and no golden vectors; no JVM exists in the build
 * container.  What is pinned: Philox4x32-10 against the Random123 known-answer
 * vectors, the java.util.Random LCG against published values, fdlibm log/pow
 * against libm to <=1 ulp, ModifiedSimpleLDATest's known answers.  The
 * end-to-end GGS sweep is "parity unpinned" against a JVM run; it is pinned
 * against this restatement only.
 */
#ifndef GGS_ORACLE_H
#define GGS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG stream addressing (the build's own definition; the reference RNG is
 * unseedable).  One Philox4x32-10 block = 128 bits = two 53-bit doubles.
 *   key  = (seed_lo, seed_hi)
 *   ctr  = (elem_lo, elem_hi, purpose<<24 | block, iteration)
 * A "draw" (one z uniform, one gamma variate) owns the sub-stream of blocks
 * 0,1,2,... for its (purpose, iteration, elem). */
enum {
  ORC_PURPOSE_Z = 1,        /* elem = global token index                     */
  ORC_PURPOSE_THETA = 2,    /* elem = global_doc * K + k                     */
  ORC_PURPOSE_PHI = 3,      /* elem = k * V + v                              */
  ORC_PURPOSE_INIT_PHI = 4, /* elem = k * V + v                              */
  ORC_PURPOSE_HELDOUT = 5   /* elem = global_test_doc * numParticles + particle; uniforms in sequence */
};

#define ORC_MAX_BLOCKS 4096 /* per-draw cap on consumed Philox blocks        */

enum {
  ORC_OK = 0,
  ORC_ERR_NEGATIVE_COUNT = 1, /* GGS:84-85, UPLDA:475-481                    */
  ORC_ERR_INVALID_TOPIC = 2,  /* GGS:116-118 (and the AIOOBE past K)         */
  ORC_ERR_RNG_EXHAUSTED = 3,
  ORC_ERR_BAD_ARG = 4
};

/* ---- primitives, exposed so tests can pin each layer separately ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_jrandom_next_ints(int64_t seed, int32_t bound, int64_t n, int32_t *out);
void orc_jrandom_next_int_raw(int64_t seed, int64_t n, int32_t *out);
void orc_jrandom_next_doubles(int64_t seed, int64_t n, double *out);
double orc_log(double x);
double orc_pow(double x, double y);
void orc_log_array(int64_t n, const double *x, double *out);
void orc_pow_array(int64_t n, const double *x, const double *y, double *out);
void orc_uniform_array(uint64_t seed, uint32_t iter, uint32_t purpose,
                       uint64_t elem0, int64_t n, double *out);
void orc_gaussian_array(uint64_t seed, uint32_t iter, uint32_t purpose,
                        uint64_t elem0, int64_t n, double *out);
int  orc_gamma_array(uint64_t seed, uint32_t iter, uint32_t purpose,
                     uint64_t elem0, int64_t n, const double *shape, double *out);
int  orc_dirichlet(uint64_t seed, uint32_t iter, uint32_t purpose,
                   uint64_t elem0, int64_t n, const double *p, double *out);

/* ---- sampler state (mirrors the Java fields) ---- */
typedef struct orc_state orc_state;

orc_state *orc_create(int32_t K, int32_t V, const double *alpha /*K*/, double beta,
                      uint64_t seed);
void orc_destroy(orc_state *s);
/* doc_base/tok_base: global index of this shard's first doc / token (0 for a
 * whole corpus); they only enter the RNG element ids. */
int orc_set_corpus(orc_state *s, int64_t D, const int64_t *doc_ptr, const int32_t *tokens,
                   int64_t doc_base, int64_t tok_base);
int orc_init_z_java_lcg(orc_state *s, int32_t seed);   /* UPLDA:398-406,458-460 */
int orc_set_z(orc_state *s, const int32_t *z, int redraw_phi); /* UPLDA:1797-1843 */
int orc_init_phi(orc_state *s);                        /* UPLDA:1287-1294       */
void orc_set_phi_mean_gating(orc_state *s, int save, int burn_in, int thin);
/* UPLDA:1573-1634 computeLogPosterior in the Java loop order; the value is doc_side + topic_side */
void orc_log_posterior(const orc_state *s, double *doc_side, double *topic_side);
int orc_draw_diagnostic_theta(orc_state *s);   /* UPLDA:710-714, the non-ggs schemes' theta for the diagnostics */
/* UPLDA:1644-1758 modelLogLikelihood in the Java loop order; the model's value is doc_side + topic_side */
void orc_model_log_likelihood(const orc_state *s, double *doc_side, double *topic_side);
/* MarginalProbEstimatorPlain.evaluateLeftToRight (MPE:85-121) on the state's current counts; doc_ll[D] is required */
int orc_heldout_log_likelihood(orc_state *s, int64_t D, const int64_t *doc_ptr, const int32_t *tokens, int64_t doc_base,
                               int32_t numParticles, double *doc_ll, double *total);
/* 0 = ggs (default), 1 = pcgs: UPLDA:1466-1544 z loop (theta integrated out), same Phi draw */
void orc_set_scheme(orc_state *s, int scheme);
void orc_set_threads(orc_state *s, int threads);
void orc_set_iteration(orc_state *s, int32_t it);
int32_t orc_get_iteration(const orc_state *s);

/* one full sweep = z step for every doc + updateCounts + samplePhi */
int orc_sweep(orc_state *s, int32_t n_sweeps);
int orc_sample_phi_range(orc_state *s, int32_t k0, int32_t k1);   /* GGS:182-198 loopOverTopics for one topic batch */
int orc_init_phi_range(orc_state *s, int32_t k0, int32_t k1);     /* UPLDA:1287-1294 for one topic batch */
void orc_set_phi_rows(orc_state *s, int32_t k0, int32_t k1, const double *rows);
int orc_phi_gammas_range(orc_state *s, int32_t initial, int32_t k0, int32_t k1, double *gam, double *sums);   /* the batch's unnormalised gammas + their sums */
void orc_set_counts(orc_state *s, const int32_t *n_wk /* [V][K] */);
/* the same sweep, bit for bit, with CPU-friendly memory behaviour (transposed Phi, no atomics): bench.py's cpu_tuned_mt */
int orc_sweep_tuned(orc_state *s, int32_t n_sweeps);
/* pieces, for sharded / staged tests */
int orc_z_step(orc_state *s);       /* GGS:47-132 for every local doc; leaves deltas */
int orc_update_counts(orc_state *s);/* UPLDA:1107-1221 */
int orc_sample_phi(orc_state *s);   /* GGS:139-198 */
int orc_collapsed_sweep(orc_state *s, int32_t seed_if_first, int32_t n_sweeps); /* MSLDA:158-226 */
int orc_collapsed_parallel_sweep(orc_state *s, int32_t n_sweeps);   /* the same conditional, documents side by side on sweep-start counts (AD-LDA) */

int64_t orc_num_tokens(const orc_state *s);
void orc_get_z(const orc_state *s, int32_t *z);
void orc_get_type_topic_counts(const orc_state *s, int32_t *n_wk /*[V][K]*/);
void orc_get_topic_type_counts(const orc_state *s, int32_t *n_kw /*[K][V]*/);
void orc_get_topic_totals(const orc_state *s, int32_t *n_k);
void orc_get_delta(const orc_state *s, int32_t *delta_wk /*[V][K]*/);
void orc_add_delta(orc_state *s, const int32_t *delta_wk /*[V][K]*/);
void orc_get_phi(const orc_state *s, double *phi /*[K][V]*/);
void orc_set_phi(orc_state *s, const double *phi /*[K][V]*/);
int  orc_get_phi_mean(const orc_state *s, double *phi_mean /*[K][V]*/);
void orc_get_theta(const orc_state *s, double *theta /*[D][K]*/);
void orc_get_doc_topic_counts(const orc_state *s, int32_t *n_dk /*[D][K]*/);
const char *orc_last_error(const orc_state *s);

#ifdef __cplusplus
}
#endif
#endif
