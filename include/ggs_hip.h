/*
 * ggs_hip.h -- C-ABI of libggs_hip.so, the MI355X (gfx950) implementation of
 * the LDA Grouped Gibbs Sampler hot path.
 *
 * The reference (clintpgeorge/LDAGroupedGibbsSampler) is 100 % Java and has no
 * FFI; this header is the boundary a thin JNI shim binds (INTEGRATION.md shows
 * the shim).  Every entry point names the reference method(s) whose work it
 * replaces; paths are relative to src/main/java/cc/mallet/ in the reference:
 *   GGS   = topics/LDAGroupedGibbsSampler.java
 *   UPLDA = topics/UncollapsedParallelLDA.java
 *   MSLDA = topics/ModifiedSimpleLDA.java
 *
 * Conventions: plain pointers and sizes only; every function returns
 * GGS_OK (0) or a GGS_ERR_* code, with text available from ggs_last_error();
 * the caller owns every host buffer, which is only touched during the call; a
 * handle is driven by ONE coordinator thread (as the Java sampler is,
 * UPLDA:552-943).  Host layouts mirror the Java getters: counts are
 * int32 [V][K] (getTypeTopicMatrix), Phi is double [K][V] (getPhi).
 * The library reads no environment variable in normal use; its GGS_DEBUG_* knobs (kernel selection, proof margins,
 * overlaps: tests and experiments) are honoured only when GGS_DEBUG=1 is set as well.
 *
 * One handle = one GPU.  Several live handles on the SAME device in one process are correct but slow: each brings
 * three streams and the runtime multiplexes all of them onto the device's few hardware queues (measured: a second
 * handle's sweeps ran 6x slower beside an idle first one) -- destroy a handle before building its successor.
 *
 * Several GPUs: documents are sharded contiguously, one handle per shard (doc_base / tok_base of ggs_set_corpus), and
 * the handles are joined by an EXCHANGE (section "multi-GPU" below): ggs_attach_rccl* (RCCL over xGMI, one process per
 * GPU or one process for all), or ggs_attach_exchange (caller-supplied transport).  With an exchange attached the
 * sweep itself contains the per-sweep merge of the reference (UPLDA:1107-1221; ADLDA.java:302-332), in the form
 * SURVEY.md 8e prefers: reduce-scatter of the int32 counts by TOPIC SLICE (even-split rule of
 * randomscan/topic/EvenSplitTopicBatchBuilder.java:28-39), Phi drawn for the rank's own topics only (the topic-parallel
 * samplePhi of GGS:139-171 with one batch per GPU), all-gather of the fp64 Phi slices.  Results are bit-identical to
 * one handle over the whole corpus: the Philox element ids are global (token, document*K+k, k*V+v).
 */
#ifndef GGS_HIP_H
#define GGS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGS_ABI_VERSION 5

typedef struct ggs_handle ggs_handle;

enum {
  GGS_OK = 0,
  GGS_ERR_NEGATIVE_COUNT = 1, /* IllegalStateException "Invalid count!" GGS:84-85; UPLDA:475-481 */
  GGS_ERR_INVALID_TOPIC = 2,  /* IllegalStateException "Topic sampled is invalid!" GGS:116-118    */
  GGS_ERR_RNG_EXHAUSTED = 3,  /* a rejection loop used more than GGS_MAX_BLOCKS Philox blocks      */
  GGS_ERR_BAD_ARG = 4,        /* IllegalArgumentException (e.g. ParallelRandoms.java:61-63)        */
  GGS_ERR_HIP = 5,            /* a HIP runtime call failed; text has the hipError string           */
  GGS_ERR_STATE = 6,          /* call order violated (e.g. sweep before set_corpus)                */
  GGS_ERR_UNSUPPORTED = 7,
  GGS_ERR_INVARIANT = 8       /* paranoid check failed (UPLDA:299-338)                             */
};

enum {
  GGS_FLAG_PARANOID = 1 << 0,     /* run ggs_check_invariants after every sweep
                                     (ParanoidUncollapsedParallelLDA.java:14-55)   */
  GGS_FLAG_SAVE_PHI_MEAN = 1 << 1,/* cfg key save_phi_mean, UPLDA:205,1331-1335    */
  GGS_FLAG_COLLAPSED = 1 << 3,    /* scheme=collapsed (SerialCollapsedLDA; the conditional is MSLDA:158-226): no theta, no Phi:
                                     score = (alpha_k + n_dk)*((beta + n_wk)/(betaSum + n_k)).  ggs_sweep runs the PARALLEL
                                     schedule -- documents side by side against the sweep-start counts minus the token
                                     itself, merged after the sweep (AD-LDA semantics, ADLDA.java:176-332; approximate as
                                     ADLDA is, and what a doc-sharded run exchanges is again the count buffer);
                                     ggs_collapsed_serial_sweep runs the reference's own serial chain.  ggs_get_phi returns
                                     the point estimate (beta + n_wk)/(betaSum + n_k); ggs_get_theta, ggs_log_posterior and
                                     ggs_sample_z_given_phi do not apply.  K up to 4096, any document length (as pcgs). */
  GGS_FLAG_PCGS = 1 << 2          /* scheme=pcgs (LDAPartiallyCollapsedGibbsSampler): the z step is UPLDA:1466-1544,
                                     score = (n_dk + alpha_k)*phi[k][w], sequential inside a document; no theta draw;
                                     counts, Phi draw and exchange exactly as for ggs.  Any K up to 4096 and any
                                     document length: up to 176 topics (scheme collapsed: 96 -- the measured break-even
                                     points) one LANE owns a document (64 documents per wave, int16 counts in LDS); above
                                     that, or when a document has 32768 tokens or more, one WAVE owns a document (int32
                                     counts, the topics spread over the lanes). */
};

/* RNG stream addressing.  The reference draws from ThreadLocalRandom and a
 * nanoTime-seeded xorshift (GGS:107, util/XORShiftRandom.java:7) and is not
 * reproducible; this library defines a counter-based stream instead:
 * Philox4x32-10, key = (seed lo, seed hi),
 * counter = (elem lo, elem hi, purpose << 24 | block, iteration). */
enum {
  GGS_PURPOSE_Z = 1,       /* elem = global token index          */
  GGS_PURPOSE_THETA = 2,   /* elem = global doc index * K + k    */
  GGS_PURPOSE_PHI = 3,     /* elem = k * V + v                   */
  GGS_PURPOSE_INIT_PHI = 4,/* elem = k * V + v                   */
  GGS_PURPOSE_HELDOUT = 5  /* elem = global test doc * numParticles + particle; one uniform per in-vocabulary token */
};
#define GGS_MAX_BLOCKS 4096

typedef struct ggs_config {
  int32_t struct_size;    /* = sizeof(ggs_config), for ABI growth                              */
  int32_t num_topics;     /* cfg key "topics"; MSLDA:126-127                                   */
  int32_t num_types;      /* alphabet size V; UPLDA:361                                        */
  int32_t device_id;      /* HIP device ordinal this handle lives on                           */
  const double *alpha;    /* K values, or NULL to use alpha_scalar for every topic (MSLDA:129-135) */
  double alpha_scalar;    /* cfg key "alpha"                                                   */
  double beta;            /* cfg key "beta"; MSLDA:136                                         */
  uint64_t seed;          /* key of the Philox streams                                         */
  int32_t flags;          /* GGS_FLAG_*                                                        */
  int32_t phi_burn_in;    /* iterations: (phi_mean_burnin/100)*iterations, UPLDA:206-207       */
  int32_t phi_mean_thin;  /* cfg key phi_mean_thin, UPLDA:208                                  */
  int32_t reserved;
} ggs_config;

typedef struct ggs_timings {
  double theta_ms;  /* cumulative: per-document theta draw (GGS:57-72)                         */
  double z_ms;      /* cumulative: token loop (GGS:79-130)                                     */
  double merge_ms;  /* cumulative: count rebuild, the device form of updateCounts (UPLDA:1107-1221) */
  double phi_ms;    /* cumulative: samplePhi (GGS:139-198); with an exchange: this rank's topic slice + the repack */
  int64_t sweeps;
  int64_t tokens_sampled;
  double exchange_ms; /* cumulative: the collectives of an attached exchange (count reduce-scatter + Phi all-gather);
                         0 without one.  ABI version 2. */
  double exchange_rs_ms; /* ... of which the count reduce-scatter.  ABI version 3. */
  double exchange_ag_ms; /* ... of which what the sweep WAITS for of the Phi all-gathers: the first half of the gammas
                            travels under the draw of the second half, so this is the time from the end of the slice's
                            draw to both halves being in.  exchange_ms = exchange_rs_ms + exchange_ag_ms.  ABI version 3. */
} ggs_timings;

/* ---- lifecycle ------------------------------------------------------------ */
/* replaces: new LDAGroupedGibbsSampler(config) (GGS:25-27, UPLDA ctor :134-215) */
int ggs_create(const ggs_config *cfg, ggs_handle **out);
void ggs_destroy(ggs_handle *h);
const char *ggs_last_error(const ggs_handle *h);
int ggs_abi_version(void);
/* Run every kernel on this hipStream_t (NULL = the legacy default stream). */
int ggs_set_stream(ggs_handle *h, void *hip_stream);

/* ---- corpus and initial state --------------------------------------------- */
/* replaces: the data part of addInstances (UPLDA:357-456, GGS:33-37): CSR of
 * FeatureSequence.getFeatures() in instance order.  doc_base / tok_base are the
 * global indices of this shard's first document / token (0 for one GPU); they
 * only enter the RNG element ids, so a doc-sharded run draws exactly what a
 * one-GPU run draws. */
int ggs_set_corpus(ggs_handle *h, int64_t num_docs, const int64_t *doc_ptr /*D+1*/,
                   const int32_t *tokens /*N*/, int64_t doc_base, int64_t tok_base);
/* replaces: initialDrawTopicIndicator loop (UPLDA:398-406,458-460): z0 =
 * java.util.Random(seed).nextInt(K) in (doc, position) order, counts rebuilt.
 * Sequential stream => only valid on an unsharded corpus (tok_base == 0). */
int ggs_init_z_java_lcg(ggs_handle *h, int32_t seed);
/* Host utility: the first n values of java.util.Random(seed).nextInt(bound) -- what a
 * doc-sharded start-up slices per rank before ggs_set_z (the stream is sequential). */
int ggs_java_lcg_next_ints(int32_t seed, int32_t bound, int64_t n, int32_t *out);
/* replaces: setZIndicators (UPLDA:1797-1843): rebuild counts from z, zero the
 * deltas and (redraw_phi != 0) re-draw Phi as initialSamplePhi does. */
int ggs_set_z(ggs_handle *h, const int32_t *z /*N*/, int32_t redraw_phi);
/* replaces: initialSamplePhi (UPLDA:1287-1294 -> MarsagliaSparseDirichlet.java:31-55) */
int ggs_init_phi(ggs_handle *h);
/* currentIteration (UPLDA:646); enters the RNG counters */
int ggs_set_iteration(ggs_handle *h, int32_t iteration);
int ggs_get_iteration(const ggs_handle *h, int32_t *iteration);

/* ---- the sweep ------------------------------------------------------------- */
/* replaces one loop body of sample() (UPLDA:645-687) n_sweeps times:
 * loopOverBatches (GGS:47-132 for every document), updateCounts, samplePhi.
 * Returns after the device is idle.  n_sweeps = 1 keeps the Java per-iteration
 * abort / exec_time / diagnostics loop in the caller. */
int ggs_sweep(ggs_handle *h, int32_t n_sweeps);
/* The same, split where a doc-sharded run exchanges counts:
 *   begin = ++iteration, theta draw + z draw + this shard's counts (UPLDA:660, and the
 *           local half of updateCounts)
 *   [caller sum-all-reduces the counts buffer across shards -- see ggs_counts_device_ptr]
 *   end   = samplePhi on the corpus-wide counts (UPLDA:664-687) */
int ggs_sweep_begin(ggs_handle *h);
int ggs_sweep_end(ggs_handle *h);
/* ggs_sweep_end without the wait: the Phi draw is enqueued and the call returns; errors the sweep flags on the
 * device (they are sticky) and its phase times surface at the next ggs_sweep_end / ggs_sweep / ggs_synchronize-
 * then-getter.  For a doc-sharded loop that exchanges counts every sweep but looks at the result every n. */
int ggs_sweep_end_async(ggs_handle *h);
/* replaces: SerialCollapsedLDA.sample's loop over documents (SerialCollapsedLDA.java:159-172 -> sampleTopicsForOneDoc,
 * MSLDA:158-226) n_sweeps times, in the reference's own SERIAL schedule: one chain over all tokens, counts moved in
 * place, uniforms from the sampler's ONE java.util.Random: after ggs_init_z_java_lcg(h, seed) the stream simply
 * continues where the initial topics left it -- SerialCollapsedLDA draws both from the same Randoms(seed) object
 * (SerialCollapsedLDA.java:60-65,789; MSLDA:206) -- and java_seed is not looked at; after ggs_set_z a new
 * Random(java_seed) is created at the first call and carried on.  Needs GGS_FLAG_COLLAPSED, an unsharded corpus and no exchange.  One wave does it all: for parity on small
 * corpora (BASELINE config 1), not for throughput. */
int ggs_collapsed_serial_sweep(ggs_handle *h, int32_t java_seed, int32_t n_sweeps);
/* replaces: sampleZGivenPhi (UPLDA:975-1014): z step + updateCounts, Phi kept */
int ggs_sample_z_given_phi(ggs_handle *h, int32_t n_sweeps);
/* Device pointer / element count of the int32 [V][K] type-topic counts
 * (typeTopicCounts, MSLDA:73) -- the ONE buffer a doc-sharded run exchanges.  After
 * ggs_sweep_begin (and after ggs_set_z(h, z, 0) at start-up) it holds THIS shard's counts;
 * the caller sum-all-reduces it in place over the shards (RCCL) and then calls ggs_sweep_end
 * (resp. ggs_init_phi).  This replaces the Java merge of the thread-shared AtomicInteger
 * deltas (batchLocalTopicTypeUpdates, UPLDA:102,1107-1221; ADLDA's sumTypeTopicCounts,
 * ADLDA.java:302): n_wk(new) = n_wk(old) + sum of deltas = sum over shards of the local
 * (word, z) histograms -- the same integers. */
int ggs_counts_device_ptr(ggs_handle *h, void **dev_ptr, int64_t *num_elems);   /* GGS_ERR_STATE with an exchange attached */
/* Corpus-wide token count (all shards); what ggs_check_invariants expects the
 * counts to sum to.  Defaults to this handle's own token count. */
int ggs_set_global_token_count(ggs_handle *h, int64_t n_tokens);
/* Block until everything queued on the handle's stream has finished; surfaces
 * device-side error flags (what Java throws from the worker threads). */
int ggs_synchronize(ggs_handle *h);

/* ---- multi-GPU: the exchange --------------------------------------------------------------------------------------
 * Replaces the thread-shared merge of the reference -- updateCounts/updateTopics (UPLDA:1107-1221), the blocking queue
 * of topic batches in samplePhi (GGS:139-171), ADLDA's sumTypeTopicCounts + copy-back (ADLDA.java:302-332) -- by three
 * collectives over `nranks` handles, rank r owning the documents of shard r and the TOPICS of slice r
 * (sizes K/nranks + (K % nranks > r), EvenSplitTopicBatchBuilder.java:28-39; Ksm = the largest slice):
 *   reduce_scatter_i32  send int32 [nranks][V][Ksm] (this shard's (word, topic) histogram, slice-major, zero padded)
 *                       -> recv int32 [V][Ksm] = the corpus-wide counts of the rank's own topics.  Once per sweep, on the
 *                       handle's stream.  (Or the sparse form in its place: all_gather_i32 of nranks pair counts, then
 *                       all_to_all_v_i32 of (cell, count) pairs -- ggs_set_count_exchange.)
 *   all_gather_f64      TWICE per Phi phase when the vocabulary has 16 or more 64-row segments (V >= 961), with DIFFERENT
 *                       counts and on DIFFERENT streams of the one handle, never concurrently: what travels is the rank's
 *                       UNNORMALISED gamma draws [V][Ksm] in two halves of the vocabulary -- first the rows below v_split
 *                       (send_count = v_split * Ksm) on the handle's high-priority COMMUNICATION stream, under the draw of
 *                       the second half; then, behind it on the handle's own stream, the rows from v_split on followed by
 *                       the slice's Ksm column sums (send_count = (V - v_split) * Ksm + Ksm).  The receiver divides by the
 *                       owner's sum while repacking (the division of ParallelDirichlet.java:60-66 on the same operands).
 *                       A short vocabulary goes in one call of (V * Ksm + Ksm) elements.  A transport must therefore take
 *                       count and stream from each call (no cached stream, no staging sized for one fixed count).
 *   all_gather_i32      the same for the count slices ([V][Ksm] per rank) -- only when a getter / diagnostic needs corpus-wide
 *                       counts -- and, with the sparse count exchange, for the nranks pair counts of every sweep
 * Attach after ggs_create and before ggs_set_corpus.  From then on ggs_sweep / ggs_sweep_end / ggs_init_phi /
 * ggs_set_z(redraw) / ggs_sample_z_given_phi contain the collectives, and every call that reads corpus-wide counts
 * (ggs_get_type_topic_counts, ggs_get_topic_totals, ggs_check_invariants, ggs_model_log_likelihood,
 * ggs_heldout_log_likelihood) gathers them first: all of these are COLLECTIVE calls -- every rank makes them in the
 * same order.  The handle then runs on a stream of its own (not the legacy default stream) and orders the collectives
 * on it; nothing relies on another library's current-stream convention. */
typedef struct ggs_exchange_ops {
  int32_t struct_size;   /* = sizeof(ggs_exchange_ops) */
  int32_t reserved;
  void *ctx;
  /* Each callback enqueues (or performs) the collective in order behind the work already queued on hip_stream and
   * returns 0 on success; counts are ELEMENTS per rank. */
  int (*reduce_scatter_i32)(void *ctx, const void *send, void *recv, int64_t recv_count, void *hip_stream);
  int (*all_gather_f64)(void *ctx, const void *send, void *recv, int64_t send_count, void *hip_stream);
  int (*all_gather_i32)(void *ctx, const void *send, void *recv, int64_t send_count, void *hip_stream);
  /* Optional (ABI version 4; may be NULL, and a table of the version-3 size -- without this member -- is accepted): the
   * SPARSE count exchange.  Rank r receives the recv_counts[s] int32 elements rank s addressed to it, at recv + recv_offsets[s];
   * it sends send_counts[d] elements from send + send_offsets[d] to every rank d (its own block included: a local copy).
   * The four arrays have nranks entries, live in HOST memory and are only read during the call; the counts agree pairwise
   * (the library exchanges them beforehand with all_gather_i32).  Used when a rank's (word, topic) histogram is sparse --
   * BASELINE config 5: 3 % of the cells of a shard are non-zero -- instead of reduce_scatter_i32 over the dense
   * [nranks][V][Ksm] buffer: what travels are (cell, count) pairs of the non-zero cells.  See ggs_set_count_exchange. */
  int (*all_to_all_v_i32)(void *ctx, const void *send, const int64_t *send_offsets, const int64_t *send_counts, void *recv,
                          const int64_t *recv_offsets, const int64_t *recv_counts, void *hip_stream);
} ggs_exchange_ops;
#define GGS_EXCHANGE_OPS_V3_SIZE ((int32_t)(sizeof(ggs_exchange_ops) - sizeof(void *)))
/* Caller-supplied transport (tests: gloo through host staging; a JVM with its own collectives). */
int ggs_attach_exchange(ggs_handle *h, int32_t rank, int32_t nranks, const ggs_exchange_ops *ops);
/* RCCL, one process per GPU: rank 0 calls ggs_rccl_unique_id and hands the 128 bytes to every rank (any channel);
 * every rank then calls ggs_attach_rccl, which is ncclCommInitRank on the handle's device (collective, blocking).
 * librccl is dlopen'ed on first use (librccl.so.1: the copy already in the process if there is one).
 * ONE RCCL per process: a host that later maps a SECOND copy of librccl (measured: PyTorch's wheel brings its own; loaded
 * after this library had dlopen'ed ROCm's, the process aborted at exit with "double free or corruption" in the two
 * copies' teardown) must load its copy first, so that this dlopen resolves to it -- ldagroupedgibbssampler_amd/_lib.py
 * imports torch before the first exchange for exactly that reason; a JVM host has only the one copy. */
#define GGS_RCCL_UNIQUE_ID_BYTES 128
int ggs_rccl_unique_id(void *out_id /* GGS_RCCL_UNIQUE_ID_BYTES */);
int ggs_attach_rccl(ggs_handle *h, int32_t rank, int32_t nranks, const void *unique_id);
/* The same with a communicator the caller created (ncclComm_t; not destroyed by ggs_destroy). */
int ggs_attach_rccl_comm(ggs_handle *h, int32_t rank, int32_t nranks, void *nccl_comm);
/* RCCL, ONE process driving n GPUs (the JVM of the reference is one process): creates n handles on device_ids[0..n)
 * from cfg (cfg->device_id ignored), joined by ncclCommInitAll.  The group entry points take the handles in rank
 * order and issue every phase for all devices from the calling thread, the collectives inside ncclGroupStart/End. */
int ggs_group_create(const ggs_config *cfg, int32_t n, const int32_t *device_ids, ggs_handle **out_handles /* n */);
void ggs_group_destroy(ggs_handle **handles, int32_t n);
int ggs_group_set_z(ggs_handle **handles, int32_t n, const int32_t *const *z /* n pointers, each shard's N */, int32_t redraw_phi);
int ggs_group_sweep(ggs_handle **handles, int32_t n, int32_t n_sweeps);
/* The same group entry points over a CALLER-SUPPLIED transport (a JVM with collectives of its own): handles 0..n-1, each
 * created by ggs_create and joined with ggs_attach_exchange(h_i, i, n, ops_i), are adopted as a one-process group.  The
 * library then issues every collective step for handle 0, 1, .. n-1 in turn before any handle goes on to the next step
 * -- the order ncclGroupStart/End gives the RCCL group -- so a transport that completes a collective only when all n
 * handles have called it is never kept waiting by the calling thread (tests/test_native_exchange_gpu.py drives two
 * handles from ONE thread this way). */
int ggs_group_adopt(ggs_handle **handles, int32_t n);
/* Corpus-wide counts onto every handle of the group.  The per-handle getters that need them (ggs_get_type_topic_counts,
 * ggs_get_topic_totals, ggs_check_invariants, the log likelihoods) would each start a collective of their own --
 * from one thread, for one device at a time, that cannot complete -- so a one-process driver calls this first; the
 * getters then find the counts in place.  (GGS_FLAG_PARANOID's per-sweep check is not run by ggs_group_sweep.) */
int ggs_group_gather_counts(ggs_handle **handles, int32_t n);
/* Timing aid, NOT a sampler: behaves as rank `rank` of `nranks` with the peers' contributions missing (the collectives
 * become local copies), so that one GPU can time the per-rank compute phases of an N-GPU split.  Counts and Phi are
 * then wrong by construction (ggs_check_invariants fails). */
int ggs_attach_null_exchange(ggs_handle *h, int32_t rank, int32_t nranks);
/* How the counts travel (typeTopicCounts of the reference; BASELINE config 5 names its rows sparse; the reference itself
 * tracks the touched cells per topic, UPLDA:1166-1176): mode 0 = by rule (sparse where the dense buffer is large and mostly
 * zero: V*K >= 2^26 cells and fewer tokens per rank than half of them -- both known alike on every rank, so all ranks
 * agree), 1 = always the dense reduce-scatter, 2 = always the (cell, count) pairs.  Every rank must make the same call
 * (before the first sweep).  Sparse needs all_to_all_v_i32 and a handle of its own process or thread (not the one-process
 * group entry points, whose collectives are collected step by step): otherwise the dense form runs.  Results are the
 * same integers either way. */
int ggs_set_count_exchange(ggs_handle *h, int32_t mode);
/* *sparse = 1 if the next count exchange of this handle ships (cell, count) pairs; *pairs_last (or NULL) = pairs this rank
 * sent in the last one, *cells (or NULL) = V * Ksm * nranks, the dense buffer's cells. */
int ggs_get_count_exchange(const ggs_handle *h, int32_t *sparse, int64_t *pairs_last, int64_t *cells);
/* rank, nranks and the rank's topic slice [k_begin, k_end) (0, 1, 0, K without an exchange) */
int ggs_get_exchange_info(const ggs_handle *h, int32_t *rank, int32_t *nranks, int32_t *k_begin, int32_t *k_end);
/* Who carries the collectives: *provider = 0 none, 1 RCCL, 2 the caller's callbacks, 3 the null timing aid; for RCCL
 * *comm_nranks / *comm_rank are read back from the communicator itself (ncclCommCount / ncclCommUserRank: what RCCL
 * saw, not what the caller passed to the attach call; -1 if the query fails), otherwise the attach call's values.
 * A benchmark line quotes these beside its number.  ABI version 4. */
int ggs_get_exchange_provider(const ggs_handle *h, int32_t *provider, int32_t *comm_nranks, int32_t *comm_rank);

/* ---- state copy-back (the Java getters) ------------------------------------ */
int ggs_get_z(ggs_handle *h, int32_t *z /*N*/);                         /* getZIndicators, MSLDA:464-477 */
int ggs_get_type_topic_counts(ggs_handle *h, int32_t *n_wk /*[V][K]*/); /* getTypeTopicMatrix, UPLDA:226-234 */
int ggs_get_topic_totals(ggs_handle *h, int32_t *n_k /*K*/);            /* getTopicTotals, MSLDA:976 */
int ggs_get_phi(ggs_handle *h, double *phi /*[K][V]*/);                 /* getPhi, UPLDA:1946-1948 */
int ggs_set_phi(ggs_handle *h, const double *phi /*[K][V]*/);           /* setPhi, UPLDA:1897-1926 */
int ggs_get_phi_mean(ggs_handle *h, double *phi_mean /*[K][V]*/, int32_t *n_sampled); /* getPhiMeans, UPLDA:1954-1966 */
int ggs_get_theta(ggs_handle *h, int64_t doc_begin, int64_t doc_end, double *theta /*[(end-begin)][K]*/); /* thetaMatrix, GGS:72, UPLDA:716-720 */
int ggs_get_doc_topic_counts(ggs_handle *h, int64_t doc_begin, int64_t doc_end, int32_t *n_dk); /* getDocumentTopicMatrix, MSLDA:536-547 */
int ggs_get_timings(ggs_handle *h, ggs_timings *out);                   /* zSamplingTimeCum / phiSamplingTimeCum, UPLDA:642-693 */
int ggs_reset_timings(ggs_handle *h);
/* replaces: ensureConsistentTopicTypeCounts (UPLDA:299-338): counts >= 0, they sum to the
 * corpus size, column sums == tokensPerTopic.  (The reference's other paranoid assert, "all
 * deltas zero at postSample", ParanoidUncollapsedParallelLDA.java:42-55, has no device
 * counterpart: there is no delta matrix, the counts are rebuilt from z every sweep.) */
int ggs_check_invariants(ggs_handle *h);
/* Launch geometry of the z kernel, for bench.py's roofline accounting. */
int ggs_get_launch_info(ggs_handle *h, int64_t *num_chunks, int32_t *lds_bytes_z, int32_t *docs_per_block_theta);
/* Words whose phiT rows the z kernels keep in LDS tables for the current corpus -- the hot-word table plus the warm
 * tiers' tables (0 for K > 192 and scheme pcgs): tokens of these, the most frequent words of the handle's corpus, read
 * no phiT row from memory -- bench.py's cold-row byte accounting. */
int ggs_get_num_hot_words(ggs_handle *h, int32_t *num_hot);
/* The warm tiers of the current corpus (z_warm_kernel: the words next in frequency after the hot table's, one LDS table
 * per tier, their chunks drawn from up to *docs_per_chunk documents): tiers kept, their words in all (part of what
 * ggs_get_num_hot_words reports).  All 0 where the score-register kernels do not run.  ABI version 5. */
int ggs_get_warm_tiers(ggs_handle *h, int32_t *tiers, int32_t *warm_words, int32_t *docs_per_chunk);
/* Launches of the z kernel per sweep for the current corpus: 1, or the number of document parts the streaming kernel's
 * step is cut into (K > 192; the next theta of a part is drawn beside the following parts) -- bench.py scales the
 * per-launch PMC counters of a profile by it. */
int ggs_get_z_parts(ggs_handle *h, int32_t *parts);
/* Which z kernel(s) the sweeps of the current corpus run, so that a benchmark line can NAME what it timed instead of
 * assuming it: *kernel = 0 whole-row tile kernel, 1 score-register kernels (K <= 160), 2 one-pass streaming kernel,
 * 3 its two-pass cross-check, 4 pcgs lane-per-document, 5 pcgs wave-per-document; *form (kernel 1 only, else 0) =
 * 1 split (cold chunks and hot chunks as two kernels side by side), 2 fused (one kernel takes both in turn);
 * *calibrated = 1 once the first z step of the corpus has timed both forms and kept the faster (0 before that, and
 * when a form is forced or there is nothing to split).  ABI version 4. */
int ggs_get_z_form(ggs_handle *h, int32_t *kernel, int32_t *form, int32_t *calibrated);
/* ---- primitives, exported so the parity tests can pin each layer ----------- */
int ggs_debug_philox(int32_t device_id, int64_t n, const uint32_t *ctr /*n*4*/, const uint32_t *key /*n*2*/, uint32_t *out /*n*4*/);
int ggs_debug_math(int32_t device_id, int32_t op /*0 log,1 pow,2 sqrt,3 div*/, int64_t n, const double *x, const double *y, double *out);
int ggs_debug_draw(int32_t device_id, int32_t kind /*0 uniform,1 gaussian,2 gamma (general loops),3 gamma as the kernels draw it: first try then general,4 as 3 with first-try results negated*/, uint64_t seed, uint32_t iteration,
                   uint32_t purpose, uint64_t elem0, int64_t n, const double *shape, double *out, int32_t *status);
/* replaces: modelLogLikelihood (UPLDA:1644-1758), the Dirichlet-multinomial log likelihood of the current topic
 * assignments, split where a doc-sharded run splits it: doc_side covers THIS handle's documents (sum_d [...] +
 * D*lgS(alphaSum), UPLDA:1674-1694) and topic_side the (replicated) type-topic counts (UPLDA:1701-1747); the model's
 * value is the sum of every shard's doc_side plus one topic_side.  lgS = MALLET Dirichlet.logGammaStirling.  A
 * diagnostic: partial sums are reduced in a fixed tree, so the value is run-to-run identical and within ~1e-12 (measured: 8e-13 at K=100, 5e-11 at K=1024, 18 M tokens)
 * relative of the Java loop's running sum, not bit-equal to it.  Needs tokensPerTopic up to date (any completed sweep,
 * ggs_init_phi or ggs_set_z with redraw). */
int ggs_model_log_likelihood(ggs_handle *h, double *doc_side, double *topic_side);
/* replaces: computeLogPosterior (UPLDA:1573-1634, the LDA log posterior of Doss and George 2025, logged every
 * diagnostic iteration, UPLDA:820-821) for scheme ggs: doc_side = sum over THIS handle's tokens of
 * log(phi[z][w] + 1e-12) + sum_d sum_k (n_dk + alpha_k - 1) log(theta[d][k] + 1e-12) with theta = the rows the last z
 * step used (thetaMatrix); topic_side = (beta - 1) sum_{k,v} log(phi[k][v] + 1e-12); the value is the sum over the
 * shards' doc_side plus one topic_side.  (The Java loop fills a dense K x V matrix per document -- D*K*V operations
 * -- to compute what is a sum over tokens.)  Same accuracy contract as ggs_model_log_likelihood.
 * Scheme pcgs keeps no thetaMatrix: the reference draws theta_d ~ Dirichlet(n_d. + alpha) afresh for this diagnostic
 * (UPLDA:710-714, LDAUtils.drawDirichlets) from a clock-seeded MALLET generator; here that draw is the one of GGS:57-72
 * under the Philox stream GGS_PURPOSE_THETA at the current iteration (reproducible; ggs_get_theta returns it afterwards).
 * GGS_ERR_UNSUPPORTED for scheme collapsed (no Phi). */
int ggs_log_posterior(ggs_handle *h, double *doc_side, double *topic_side);
/* replaces: addTestInstances (UPLDA:340-343; the test set of the held-out estimator).  CSR like ggs_set_corpus; token
 * ids are indices of the TRAINING alphabet, ids >= num_types are out of vocabulary and skipped as at MPE:341-345.
 * doc_base = global index of the first test document (RNG element ids only; for a sharded test set).  Documents longer
 * than 2^25 tokens (the 24-bit block field of a particle's Philox stream, two uniforms per block): GGS_ERR_UNSUPPORTED. */
int ggs_set_test_corpus(ggs_handle *h, int64_t D, const int64_t *doc_ptr /*D+1*/, const int32_t *tokens, int64_t doc_base);
/* replaces: new MarginalProbEstimatorPlain(numTopics, alpha, alphaSum, beta, typeTopicCounts, tokensPerTopic)
 * .evaluateLeftToRight(testSet, numParticles, null) (MPE:51-121,123-519; call sites UPLDA:604-622,677-682,840-844 with
 * numParticles = 100, UPLDA:615) on the handle's CURRENT counts (corpus-wide after the sweep's exchange): the
 * left-to-right estimate of the test set's log likelihood, without the resampling pass (MPE:125).  doc_ll (D or NULL)
 * = the per-document values (what Java prints to docProbabilityStream); *total = their sum in document order.
 * Bit-identical to the oracle's restatement under the Philox stream GGS_PURPOSE_HELDOUT (the reference's Randoms is
 * clock-seeded).  The estimator's IllegalStateException ("Sampled invalid topic") -> GGS_ERR_INVALID_TOPIC.
 * The per-particle topic counts (1, 2 or 4 bytes each by the length of the test document) live in LDS where K * 64 of
 * them fit beside the tables -- up to 1704 topics with test documents of at most 255 tokens, 1024 with longer ones --
 * and in global memory otherwise (same arithmetic, slower).  GGS_ERR_UNSUPPORTED only when the estimator's tables
 * themselves do not fit LDS (about 6000 topics). */
int ggs_heldout_log_likelihood(ggs_handle *h, int32_t num_particles, double *doc_ll, double *total);
/* out[k] = x[0][k] + x[1][k] + ... in index order (exactly one of x / counts given; with counts the addends are
 * beta + counts[v][k]): the Phi normalisers' exact parallel column sum on its own, for adversarial inputs */
int ggs_debug_column_sum(int32_t device_id, int32_t V, int32_t K, const double *x /*V*K or NULL*/, const int32_t *counts /*V*K or NULL*/,
                         double beta, double *out /*K*/);
/* The same with the caller's GUESS of the running sums at every 64-row segment start (guess [V/64 rounded up + 1][K],
 * or NULL: the kernels make their own) -- what a sweep feeds from the previous sweep's exact values.  The guess must
 * never decide a result: the tests pass wrong, zero, NaN and exact ones.  pref_out (same shape, or NULL) = the exact
 * running sums the walk leaves behind; n_k_out (K, or NULL; counts only) = the integer column sums (tokensPerTopic). */
int ggs_debug_column_sum_guided(int32_t device_id, int32_t V, int32_t K, const double *x, const int32_t *counts, double beta,
                                const double *guess, double *out, double *pref_out, int32_t *n_k_out);

#ifdef __cplusplus
}
#endif
#endif /* GGS_HIP_H */
