// ggs_api.hip -- host side of libggs_hip.so: handle, launches, C-ABI (include/ggs_hip.h).
//
// Build (see build.py): hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ggs_kernels.hpp"
#include "ggs_z_sliced.hpp"
#include "ggs_z_stream.hpp"
#include "ggs_exact_sum.hpp"
#include "ggs_z_pcgs.hpp"
#include "ggs_z_pcgs_wave.hpp"
#include "ggs_z_collapsed.hpp"
#include "ggs_loglik.hpp"
#include "ggs_heldout.hpp"
#include "ggs_exchange.hpp"

using namespace ggs;

namespace {

// The GGS_DEBUG_* variables select kernels, scale proof margins and switch overlaps off: experiments and tests.  A
// production process must not pick one up by accident, so they are only read when GGS_DEBUG=1 is set as well.
const char *debug_env(const char *name) {
  static const bool enabled = [] { const char *e = std::getenv("GGS_DEBUG"); return e && std::atoi(e) == 1; }();
  return enabled ? std::getenv(name) : nullptr;
}

constexpr int kThetaBlock = 256;
constexpr int kMaxLdsBytes = 160 * 1024;

// The phase events of one sweep.  Sweeps are settled (waited for, checked, timed) in batches, so a ring of them:
// sweep i records into slot i % kEvRing; the theta drawn ahead for sweep i + 1 records th0/th1 of THAT slot.
constexpr int kEvRing = 8;
struct Events {
  hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t x[3] = {nullptr, nullptr, nullptr};   // with an exchange: after the count reduce-scatter, after the slice's Phi draw, after the last all-gather (both halves in)
  hipEvent_t th0 = nullptr, th1 = nullptr;   // side-stream theta draw consumed by this sweep
  bool used_ahead = false, exchanged = false;
  bool e4_is_e3 = false;        // a whole sweep (ggs_sweep): nothing happens between the end of the count rebuild and the start of the Phi phase
  bool theta_on_main = false;   // the theta this sweep consumes was drawn on the handle's own stream, right behind the previous z step: no th0 (= that sweep's e[2])
  bool light = false;           // with an exchange: only e[1], e[2], e[5] were recorded; the span e[2]..e[5] is split as the last fully timed sweep's was
};

}  // namespace

struct ggs_handle {
  int32_t K = 0, V = 0, Kp = 0, pitch16 = 0, device = 0;
  double beta = 0;
  std::vector<double> alpha;
  uint64_t seed = 0;
  int32_t flags = 0, phi_burn_in = 0, phi_thin = 1;
  int32_t iteration = 0;
  int32_t n_sampled_phi = 0;
  int32_t ablate = 0;   // GGS_DEBUG_ABLATE: timing-only experiments, results are wrong on purpose

  int64_t D = 0, N = 0, C = 0, S = 0, doc_base = 0, tok_base = 0, global_tokens = -1;
  bool have_corpus = false, have_phi = false, in_sweep = false;
  int32_t theta_lds_main = 0, theta_b_main = 0;       // the theta draw as the critical leg (theta_main): workgroup size and LDS request
  int32_t theta_docs_per_block = 0, theta_lds = 0, z_lds = 0, z_tile_tokens = 0, z_waves_per_cu = 0, num_cus = 0;
  bool z_sliced = false;   // scores-in-registers kernel (K <= kSlicedMaxTopics)
  bool z_stream = false;   // streaming kernel (K > kSlicedMaxTopics): rows once (z_stream1_kernel) ...
  bool z_two_pass = false; // ... or twice (z_stream_kernel, GGS_DEBUG_ZKERNEL=3: the cross-check)
  bool z_regck = false;    // z_stream1_kernel keeps its checkpoints in registers (K <= 1024)
  int32_t z_group = 4;     // ... one per z_group slices
  bool z_two_rows = false; // z_stream1_kernel with two theta rows per wave: chunks may run across one document boundary (K <= 512)
  int32_t *d_chunk_doc1 = nullptr;
  double margin_scale = 1.0;   // GGS_DEBUG_MARGIN: scales z_stream1_kernel's certainty margin (tests force its exact replay)

  hipStream_t stream = nullptr;
  // device buffers
  int64_t *d_doc_ptr = nullptr, *d_chunk_start = nullptr;
  int32_t *d_tok = nullptr, *d_z = nullptr, *d_chunk_doc = nullptr, *d_chunk_len = nullptr;
  // sliced z kernel (ggs_z_sliced.hpp): its chunk lists (cold chunks first), the hot-word table's word ids, LDS layout
  int32_t *d_ct_tok = nullptr, *d_ct_idx = nullptr, *d_ct_ip = nullptr, *d_c_docs = nullptr, *d_hot_words = nullptr;
  int64_t Cs = 0, Cc = 0;                              // sliced chunks in all, cold ones
  int32_t *d_order = nullptr;                          // scheme=pcgs: local documents, longest first
  int32_t pcgs_lds = 0, pcgs_waves_per_cu = 0, max_doc_len = 0;
  int64_t pcgs_order_len = 0;                          // entries of d_order (documents, or the padded two-round list)
  bool pcgs_sliced = false;                            // K <= 192: scores in registers, one pass over the rows per step
  bool pcgs_wave = false;                              // wide rows or long documents: one wave per document (ggs_z_pcgs_wave.hpp)
  bool pcgs_wave_forced = false;                       // ... because of K; otherwise decided per corpus (a document of 32 768 tokens or more)
  int32_t pcgs_wave_nb = 0, pcgs_wave_lds = 0, pcgs_wave_waves_per_cu = 0;
  bool collapsed = false;                              // scheme=collapsed: the pcgs machinery over psi = (beta + n_wk)/(betaSum + n_k)
  uint64_t *d_lcg = nullptr;                           // ggs_collapsed_serial_sweep: the java.util.Random state
  bool lcg_ready = false;
  int32_t hot_cap = 0, num_hot = 0, hot_pitch = 0, wave_lds = 0, ring_base = 0;
  int32_t *d_perm = nullptr, *d_inv_perm = nullptr, *d_zw = nullptr, *d_seg_word = nullptr, *d_seg_begin = nullptr;
  // theta of the current / last z step, and the buffer the next iteration's theta is drawn into
  // on the side stream while this iteration's counts and Phi are computed (theta_{t+1} depends
  // on z_t only, GGS:57-72)
  double *d_theta_next = nullptr;
  hipStream_t side = nullptr;
  hipStream_t side_hot = nullptr;                      // z_hot_kernel runs here, beside z_sliced_kernel on the main stream
  hipEvent_t ev_hot_fork = nullptr, ev_hot_join = nullptr;
  hipEvent_t ev_theta_tail = nullptr;                  // GGS_DEBUG_THETA_TAIL_PCT (timing experiment), created on first use
  // Whole sweeps on one GPU (ggs_sweep, K <= 160): the next theta is the LONGER of the two legs behind the z step (0.61 ms
  // against 0.50 for the counts and the Phi chain), and a dependency across streams takes 10-25 us to resolve -- so the
  // long leg stays on the handle's stream, directly between two z steps, and the short one (count rebuild + Phi chain)
  // goes to the high-priority stream the hot chunks used during the z step.  GGS_DEBUG_THETA_MAIN=0: the other way round.
  bool theta_main = true, chain_on_side = false, theta_main_always = false;   // GGS_DEBUG_THETA_MAIN=2: whatever D and V are (tests)
  bool whole_sweep = false;                            // inside ggs_sweep: z phase and Phi phase are enqueued back to back
  hipEvent_t hot_fork_from = nullptr;                  // an event already on the handle's stream that the hot kernel's stream may wait for instead of a fork event of its own
  hipEvent_t ev_chain_done = nullptr;
  bool z_split = true;                                 // GGS_DEBUG_SPLIT=0: one kernel takes cold and hot chunks in turn
  bool z_split_allowed = true, z_split_forced = false, z_split_tried = false;  // the first z step of a corpus times both forms and keeps the faster
  int32_t hot_wave_lds = 0;
  // the warm tiers (z_warm_kernel): tables of the words next in frequency after the hot table's, chunks of up to warm_docs documents
  int32_t warm_docs = 0, warm_wave_lds = 0, warm_cap = 0;   // LDS layout: theta rows per wave; rows a tier's table may have
  // What a tier must bring (measured on the benchmark corpus and its halves, ggs_set_corpus): chunks at least 40 % full, at
  // least 3 of them per resident wave, three tiers at most -- GGS_DEBUG_WARM / GGS_DEBUG_WARM_FILL / GGS_DEBUG_WARM_CPW
  int32_t warm_tiers_max = 3, warm_min_fill_pct = 40, warm_min_chunks_per_wave = 3;
  int32_t warm_tiers = 0, num_warm = 0, warm_rows_max = 0;  // of the current corpus: tiers kept, their words in all, the largest table
  int64_t Cw = 0, warm_chunks_max = 0;                      // warm chunks in all, of the largest tier
  int32_t *d_wt_pack = nullptr, *d_w_docs = nullptr, *d_warm_words = nullptr;   // d_wt_pack: four int32 per lane (ZParams::wt_pack)
  int32_t *d_ht_pack = nullptr, *d_h_docs = nullptr;        // the hot chunks in the same packed form (z_hot_kernel)
  int64_t *d_warm_meta = nullptr;
  bool overlap_theta = true;
  // K > 192: the z step is cut into parts of consecutive documents and the NEXT iteration's theta of a part is drawn
  // (side stream) while the following parts are still being sampled: the streaming z kernel waits on memory, the theta
  // draw on the VALU.  The last part's theta runs beside the Phi phase as before.
  int32_t z_parts = 1, theta_lds_beside_z = 0;
  // the two launch configurations of the K > 192 path, chosen per corpus (ggs_set_corpus): with the z step cut into parts
  // (theta workgroups beside the z waves) or in one piece
  struct ZCfg { int32_t parts = 1, z_waves_per_cu = 0, theta_docs_per_block = 0, theta_lds = 0, theta_lds_beside_z = 0; } cfg_plain, cfg_parts;
  bool z_parts_forced = false;
  int32_t gamma_queue_cap = 1 << 20;   // GGS_DEBUG_GAMMA_QUEUE: a tiny queue sends the leftovers of the first try down the on-the-spot path
  std::vector<int64_t> part_doc, part_chunk;           // [z_parts + 1] boundaries
  hipEvent_t ev_part[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int64_t theta_ahead_iter = INT64_MIN;     // iteration the side-stream theta was drawn for, or none
  double *d_alpha = nullptr, *d_theta = nullptr, *d_phiT = nullptr, *d_mag = nullptr, *d_tot = nullptr, *d_phi_mean = nullptr;
  int32_t *d_n_wk = nullptr, *d_n_k = nullptr;
  double *d_sum_pref = nullptr, *d_sum_fn = nullptr;   // work space of the exact parallel column sums (ggs_exact_sum.hpp)
  int32_t sum_nseg = 0;
  bool exact_sum = true;                               // GGS_DEBUG_CHAIN=1: the element-by-element column_chain_kernel instead
  bool sum_guided = true;                              // GGS_DEBUG_GUIDED=0: every magnitude sum makes its own guess (seg + prefix launches)
  const void *guess_src = nullptr;                     // d_sum_pref = the exact running sums of the last magnitude sum over this buffer ...
  int32_t guess_cols = 0;                              // ... and this many columns: a good guess for the next one
  uint32_t *d_status = nullptr;
  // test set of the held-out estimator (ggs_heldout.hpp)
  int64_t *d_test_ptr = nullptr;
  int32_t *d_test_tok = nullptr, *d_test_docs = nullptr;       // d_test_docs: ids of the documents of <= 255 tokens, then of those up to 65 535, then of the longer ones
  std::vector<int32_t> test_short, test_long, test_huge;
  void *d_heldout_spill = nullptr;
  size_t heldout_spill_bytes = 0;
  double *d_test_ll = nullptr;
  std::vector<int64_t> test_ptr;
  int64_t test_doc_base = 0;
  bool have_test = false;
  void *d_scratch = nullptr;
  size_t scratch_bytes = 0;

  // multi-GPU exchange (ggs_exchange.hpp).  Without one: Ks = Ksm = K, k0 = 0 and the counts live in d_n_wk.
  Exchange *xg = nullptr;
  hipStream_t own_stream = nullptr;                    // with an exchange the handle leaves the legacy default stream
  int32_t Ks = 0, Ksm = 0, k0 = 0;                     // this rank's topic slice [k0, k0 + Ks), widest slice Ksm
  int64_t *d_koff = nullptr;                           // [K] column of topic k in the slice-major arrays
  int32_t *d_cnt_send = nullptr, *d_cnt_own = nullptr, *d_cnt_all = nullptr, *d_n_k_own = nullptr;
  // the rank's unnormalised gammas [V][Ksm] and, behind them, their Ksm column sums; gathered in two halves (rows below /
  // from v_split), the first half's all-gather on `comm_stream` under the second half's draw, the sums riding with the second
  double *d_phi_own = nullptr, *d_phi_all0 = nullptr, *d_phi_all1 = nullptr, *d_mag_own = nullptr;
  int32_t *d_krank = nullptr, *d_kcol = nullptr;      // [K] owner rank and column in its slice, for the repack
  int32_t seg_split = 0, v_split = 0;                  // first segment / row of the second half (0: one all-gather)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_half_drawn = nullptr, ev_half_gathered = nullptr;
  int32_t *d_hseg_word = nullptr, *d_hseg_begin = nullptr, *d_hseg_end = nullptr;   // the hot words' count segments (the z kernels count the cold tokens themselves)
  int64_t HS = 0;
  // With an exchange every event packet on the handle's stream sits on the sweep's critical path (2.8 us each, 7.5 for two
  // back to back: scripts/probes/event_cost_probe.hip), and a fully timed sweep records five that only the phase split
  // needs (e[3], e[4], x[0..2]).  One sweep in kDetailEvery records them all; the others record the ends of the z step and
  // of the sweep and split the rest in the proportions of the last fully timed sweep (GGS_DEBUG_TIMING_EVERY=1: all).
  int32_t detail_every = 4;
  bool force_detail = false;                           // a z step that no Phi phase follows (ggs_sample_z_given_phi) is timed in full
  int64_t sweeps_enqueued = 0;
  bool have_frac = false;
  double frac[5] = {0, 0, 0, 0, 0};                    // of e[2]..e[5]: merge | reduce-scatter | slice draw | all-gather wait | repack
  // the SPARSE count exchange (ggs_set_count_exchange): (cell, count) pairs of the non-zero cells instead of the dense buffer
  int32_t count_mode = 0;                              // 0 by rule, 1 dense, 2 sparse
  bool counted_sparse = false;                         // the last count of this shard was pass A of the sparse form (the dense send buffer does not hold it)
  bool in_group = false;                               // a handle of ggs_group_create / ggs_group_adopt: its collectives are collected step by step, no host round trip inside
  int64_t *d_sp_count = nullptr;                       // [nranks] pairs per destination (pass A), then the write cursors of pass B
  int32_t *d_sp_cnt32 = nullptr, *d_sp_all = nullptr;  // [nranks] the same as int32 for the all-gather; [nranks][nranks] everybody's
  int32_t *d_sp_wg_count = nullptr;                    // [workgroups][nranks] pairs per workgroup and destination
  int64_t *d_sp_wg_off = nullptr;                      // ... and their exclusive prefix over the workgroups
  int64_t sp_wgs = 0;
  int32_t *d_sp_send = nullptr, *d_sp_recv = nullptr;
  size_t sp_send_cap = 0, sp_recv_cap = 0;             // elements
  int64_t sp_pairs_last = 0;
  bool hot_join_pending = false, hot_counted = false;   // launch_z(defer_join): the handle's stream has not yet waited for the hot chunks' stream; the hot words' count ran there
  bool z_counted = false;                              // this sweep's z step has already added its cells into d_cnt_send
  bool cnt_send_zeroed = false;                        // d_cnt_send is all zero (cleared behind the last reduce-scatter): what the z kernels' own count updates start from
  bool z_counts_forced = false;
  bool z_counts = true;                                // GGS_DEBUG_ZCOUNTS=0: the count kernel also with an exchange (the cross-check)
  SliceMap smap{};                                     // topic -> cell of the slice-major send buffer
  bool counts_global = true;                           // d_n_wk holds the corpus-wide counts
  bool cnt_own_valid = false;                          // d_cnt_own = reduce-scatter of the current d_cnt_send
  bool n_k_valid = false;                              // d_n_k follows d_n_wk
  std::vector<ggs_handle *> group;                     // ggs_group_create: the handles of the group, in rank order (rank 0 only)

  Events evs[kEvRing];
  int ev_head = 0, ev_pending = 0;                     // slot of the sweep in progress; sweeps enqueued but not settled
  ggs_timings tm{};
  std::string err;
};

namespace {

int set_err(ggs_handle *h, int code, const std::string &msg) {
  if (h) h->err = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                        \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return set_err((h), GGS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
  } while (0)

int bind_device(ggs_handle *h) {
  HIP_TRY(h, hipSetDevice(h->device));
  return GGS_OK;
}

template <typename T>
int dev_alloc(ggs_handle *h, T **p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T)));
  return GGS_OK;
}

int ensure_scratch(ggs_handle *h, size_t bytes) {
  if (h->scratch_bytes >= bytes) return GGS_OK;
  if (h->d_scratch) (void)hipFree(h->d_scratch);
  h->d_scratch = nullptr; h->scratch_bytes = 0;
  HIP_TRY(h, hipMalloc(&h->d_scratch, bytes));
  h->scratch_bytes = bytes;
  return GGS_OK;
}

int grid_for(int64_t n, int block, int per_thread = 1) {
  const int64_t want = (n + (int64_t)block * per_thread - 1) / ((int64_t)block * per_thread);
  return (int)std::max<int64_t>(1, std::min<int64_t>(want, 256 * 8));  // grid-stride above ~8 blocks/CU
}

// Surfaces what the Java code throws from its worker threads.
int check_status(ggs_handle *h) {
  uint32_t st = 0;
  HIP_TRY(h, hipMemcpyAsync(&st, h->d_status, sizeof st, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (!st) return GGS_OK;
  HIP_TRY(h, hipMemsetAsync(h->d_status, 0, sizeof(uint32_t), h->stream));
  if (st & ST_INVALID_TOPIC)
    return set_err(h, GGS_ERR_INVALID_TOPIC, h->collapsed ? "SimpleLDA: New topic not sampled." /* MSLDA:216-218 */ : "LDAGroupedGibbsSampler: Topic sampled is invalid!");
  if (st & ST_NEGATIVE_COUNT) return set_err(h, GGS_ERR_NEGATIVE_COUNT, "Negative count for topic (Invalid count!)");
  if (st & ST_BAD_SHAPE) return set_err(h, GGS_ERR_BAD_ARG, "alpha and beta must be strictly positive (gamma shape <= 0)");
  return set_err(h, GGS_ERR_RNG_EXHAUSTED, "a gamma rejection loop exceeded GGS_MAX_BLOCKS Philox blocks");
}

// after a host upload of z (document order): refresh the word-sorted copy the count kernel reads
int launch_permute_z(ggs_handle *h) {
  if (h->N > 0)
    hipLaunchKernelGGL(permute_z_kernel, dim3(grid_for(h->N, 256)), dim3(256), 0, h->stream, h->d_perm, h->d_z, h->d_zw, h->N);
  HIP_TRY(h, hipGetLastError());
  return GGS_OK;
}

// Does the next count exchange of this handle ship (cell, count) pairs?  Every input of the rule is the same on every rank.
bool use_sparse(const ggs_handle *h) {
  if (!h->xg || h->in_group || !h->xg->ops.all_to_all_v_i32 || h->xg->nranks > kSparseMaxRanks) return false;
  if ((int64_t)h->V * h->Ksm >= ((int64_t)1 << 31)) return false;       // a cell index is an int32
  if (h->count_mode == 1) return false;
  if (h->count_mode == 2) return true;
  if (h->global_tokens < 0) return false;              // the rule needs a figure all ranks share
  const int64_t cells = (int64_t)h->V * h->K;
  return cells >= ((int64_t)1 << 26) && h->global_tokens / h->xg->nranks < cells / 2;
}
SparseCountParams sparse_params(const ggs_handle *h) {
  SparseCountParams sp{};
  sp.zw = h->d_zw; sp.seg_word = h->d_seg_word; sp.seg_begin = h->d_seg_begin; sp.K = h->K; sp.num_segs = (int32_t)h->S; sp.nranks = h->xg->nranks;
  sp.ksm = h->Ksm; sp.rem = h->smap.rem; sp.size = h->smap.size; sp.m_size = h->smap.m_size; sp.m_size1 = h->smap.m_size1;
  return sp;
}
// pass A of the sparse form: how many pairs this rank has for every destination
int launch_sparse_count(ggs_handle *h) {
  const int n = h->xg->nranks;
  int rc;
  if (!h->d_sp_count && ((rc = dev_alloc(h, &h->d_sp_count, (size_t)n)) || (rc = dev_alloc(h, &h->d_sp_cnt32, (size_t)n)) || (rc = dev_alloc(h, &h->d_sp_all, (size_t)n * n)))) return rc;
  const int64_t wgs = (h->S + kSparseSegsPerBlock - 1) / kSparseSegsPerBlock;
  if (wgs != h->sp_wgs || !h->d_sp_wg_count) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if ((rc = dev_alloc(h, &h->d_sp_wg_count, (size_t)wgs * n)) || (rc = dev_alloc(h, &h->d_sp_wg_off, (size_t)wgs * n))) return rc;
    h->sp_wgs = wgs;
  }
  if (wgs > 0) {
    SparseCountParams sp = sparse_params(h);
    sp.wg_count = h->d_sp_wg_count;
    hipLaunchKernelGGL(sparse_count_kernel<false>, dim3((unsigned)wgs), dim3(256), (size_t)h->K * sizeof(int32_t), h->stream, sp);
    hipLaunchKernelGGL(sparse_scan_kernel, dim3((unsigned)n), dim3(256), 0, h->stream, h->d_sp_wg_count, (int32_t)wgs, n, h->d_sp_wg_off, h->d_sp_count);
    HIP_TRY(h, hipGetLastError());
  } else {
    HIP_TRY(h, hipMemsetAsync(h->d_sp_count, 0, sizeof(int64_t) * (size_t)n, h->stream));
  }
  h->n_k_valid = false; h->counts_global = false; h->cnt_own_valid = false;
  h->counted_sparse = true;
  return GGS_OK;
}

// n_wk = histogram of (word, z) over this handle's tokens (UPLDA:471-474 summed over the corpus).  With an exchange the
// histogram goes straight into the slice-major send buffer of the count reduce-scatter.
int launch_count_rebuild(ggs_handle *h) {
  if (use_sparse(h)) return launch_sparse_count(h);     // no dense buffer: the pairs are emitted inside the exchange step, once their number is known
  h->counted_sparse = false;
  const size_t cells = h->xg ? (size_t)h->xg->nranks * h->V * h->Ksm : (size_t)h->K * h->V;
  int32_t *dst = h->xg ? h->d_cnt_send : h->d_n_wk;
  if (!(h->xg && h->cnt_send_zeroed)) HIP_TRY(h, hipMemsetAsync(dst, 0, cells * sizeof(int32_t), h->stream));
  h->cnt_send_zeroed = false;
  if (h->S > 0) {
    CountParams cp{};
    cp.zw = h->d_zw; cp.seg_word = h->d_seg_word; cp.seg_begin = h->d_seg_begin; cp.n_wk = dst; cp.K = h->K; cp.num_segs = (int32_t)h->S;
    cp.koff = h->xg ? h->d_koff : nullptr; cp.row_stride = h->xg ? h->Ksm : h->K;
    cp.seg_end = nullptr; cp.segs_per_block = kCountSegsPerBlock;
    hipLaunchKernelGGL(count_sorted_kernel, dim3((unsigned)((h->S + kCountSegsPerBlock - 1) / kCountSegsPerBlock)), dim3(256), (size_t)h->K * sizeof(int32_t),
                       h->stream, cp);
  }
  HIP_TRY(h, hipGetLastError());
  h->n_k_valid = false;
  if (h->xg) { h->counts_global = false; h->cnt_own_valid = false; }
  return GGS_OK;
}
// The hot words' share of the send buffer, behind a z step whose kernels counted the cold tokens themselves.
int launch_count_hot(ggs_handle *h) {
  if (h->HS > 0) {
    CountParams cp{};
    cp.zw = h->d_zw; cp.seg_word = h->d_hseg_word; cp.seg_begin = h->d_hseg_begin; cp.seg_end = h->d_hseg_end; cp.n_wk = h->d_cnt_send; cp.K = h->K;
    cp.num_segs = (int32_t)h->HS; cp.koff = h->d_koff; cp.row_stride = h->Ksm; cp.segs_per_block = 1;
    hipLaunchKernelGGL(count_sorted_kernel, dim3((unsigned)h->HS), dim3(256), (size_t)h->K * sizeof(int32_t), h->stream, cp);
    HIP_TRY(h, hipGetLastError());
  }
  h->n_k_valid = false; h->counts_global = false; h->cnt_own_valid = false;
  return GGS_OK;
}

// out[k] = sum over v, in index order, of src[v][k] (MAGNITUDE: of beta + src[v][k]) for the Ks columns of a topic
// slice -- the exact parallel formulation of ggs_exact_sum.hpp, or the element-by-element chain it replaces.
// `guided`: h->d_sum_pref already holds a guess of the running sums for these columns (the exact values of the last
// magnitude sum over the same buffer); otherwise two extra launches make one.  The walk leaves the exact running
// sums there (write_pref): the next sweep's guess, and this sweep's guess for the sum of the gammas.
template <typename T, bool MAGNITUDE>
void launch_column_sum(ggs_handle *h, const T *src, int32_t pitch, int32_t Ks, double *out, int32_t *n_k, bool guided, bool write_pref) {
  if (Ks <= 0) return;
  if (!h->exact_sum) {
    hipLaunchKernelGGL((column_chain_kernel<T, MAGNITUDE>), dim3((Ks + 7) / 8), dim3(256), 0, h->stream, src, pitch, Ks, h->V, h->beta, out);
    return;
  }
  SumParams sp{};
  sp.src = src; sp.guess = h->d_sum_pref; sp.fn = h->d_sum_fn; sp.out = out; sp.beta = h->beta; sp.n_k = n_k;
  sp.pitch = pitch; sp.K = Ks; sp.V = h->V; sp.nseg = h->sum_nseg; sp.write_pref = write_pref ? 1 : 0;
  const dim3 rows((unsigned)h->sum_nseg, (unsigned)((Ks + kSumBlock - 1) / kSumBlock));
  if (!guided) {
    hipLaunchKernelGGL((sum_seg_kernel<T, MAGNITUDE>), rows, dim3(kSumBlock), 0, h->stream, sp);
    hipLaunchKernelGGL(sum_prefix_kernel, dim3((unsigned)Ks), dim3(64), 0, h->stream, sp);
  }
  hipLaunchKernelGGL((sum_segfn_kernel<T, MAGNITUDE>), dim3((unsigned)h->sum_nseg, (unsigned)((Ks + kSegFnCols - 1) / kSegFnCols)), dim3(256), 0, h->stream, sp);
  hipLaunchKernelGGL((sum_walk_kernel<T, MAGNITUDE>), dim3((unsigned)Ks), dim3(64), 0, h->stream, sp);
}

// magnitude_k = sum_v (beta + n_kv) and tokensPerTopic for the Ks columns of `cnt`
int launch_magnitude_on(ggs_handle *h, const int32_t *cnt, int32_t pitch, int32_t Ks, double *mag, int32_t *n_k) {
  if (Ks <= 0) return GGS_OK;
  if (h->exact_sum) {
    // the guess of the guided form is only a guess for the SAME columns of the SAME buffer
    const bool guided = h->sum_guided && h->guess_src == cnt && h->guess_cols == Ks;
    launch_column_sum<int32_t, true>(h, cnt, pitch, Ks, mag, n_k, guided, true);   // tokensPerTopic falls out of the segment pass
    h->guess_src = cnt; h->guess_cols = Ks;
  } else {
    HIP_TRY(h, hipMemsetAsync(n_k, 0, sizeof(int32_t) * (size_t)Ks, h->stream));
    launch_column_sum<int32_t, true>(h, cnt, pitch, Ks, mag, nullptr, false, false);
    hipLaunchKernelGGL(topic_totals_kernel, dim3(grid_for((int64_t)Ks * h->V, 256, 16)), dim3(256), (size_t)Ks * sizeof(int32_t), h->stream, cnt,
                       Ks, pitch, h->V, n_k);
  }
  HIP_TRY(h, hipGetLastError());
  return GGS_OK;
}

// One collective of the attached exchange, on the handle's stream.
int xcall(ggs_handle *h, int rc, const char *what) {
  if (!rc) return GGS_OK;
  return set_err(h, GGS_ERR_HIP, std::string("exchange ") + what + " failed" + (h->xg->err.empty() ? "" : ": " + h->xg->err));
}
// Behind the reduce-scatter the send buffer is dead until the next z step fills it again (the z kernels add their cells
// into it, or count_sorted_kernel does): it is cleared right here, off the z step's critical path -- on the communication
// stream when the Phi phase is about to use it anyway (`defer_clear`: phi_step_b2 enqueues the fill behind the first
// all-gather, and the event the main stream waits for before the second one covers it), otherwise on the handle's stream.
int clear_send_buffer(ggs_handle *h, hipStream_t on) {
  HIP_TRY(h, hipMemsetAsync(h->d_cnt_send, 0, (size_t)h->xg->nranks * h->V * h->Ksm * sizeof(int32_t), on));
  h->cnt_send_zeroed = true;
  return GGS_OK;
}
// NOT cleared in here: inside ncclGroupStart/End a collective is only collected, and a fill enqueued beside it would land
// on the stream BEFORE it.  The callers clear once the collective is really enqueued (clear_send_buffer_if_dead); a
// buffer nobody cleared is cleared by the next z step itself.
// The sparse form of the count exchange: pair counts to the host, all-gathered; the pairs emitted at their offsets;
// all-to-all of the variable blocks; scatter-add into the zeroed slice.  Two host round trips (the sizes of what travels
// are only known on the device) -- this form is for exchanges whose dense buffer is hundreds of megabytes.
int sparse_exchange(ggs_handle *h) {
  const int n = h->xg->nranks, me = h->xg->rank;
  std::vector<int64_t> mine((size_t)n), soff((size_t)n), scnt((size_t)n), roff((size_t)n), rcnt((size_t)n);
  std::vector<int32_t> mine32((size_t)n), all((size_t)n * n);
  HIP_TRY(h, hipMemcpyAsync(mine.data(), h->d_sp_count, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  int64_t total = 0;
  for (int r = 0; r < n; ++r) {
    if (mine[(size_t)r] > INT32_MAX / 2) return set_err(h, GGS_ERR_UNSUPPORTED, "sparse count exchange: more than 2^30 pairs for one destination");
    mine32[(size_t)r] = (int32_t)mine[(size_t)r];
    soff[(size_t)r] = 2 * total; scnt[(size_t)r] = 2 * mine[(size_t)r];
    total += mine[(size_t)r];
  }
  h->sp_pairs_last = total;
  HIP_TRY(h, hipMemcpyAsync(h->d_sp_cnt32, mine32.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  int rc = xcall(h, h->xg->ops.all_gather_i32(h->xg->ops.ctx, h->d_sp_cnt32, h->d_sp_all, n, h->stream), "all_gather_i32 (pair counts)");
  if (rc) return rc;
  HIP_TRY(h, hipMemcpyAsync(all.data(), h->d_sp_all, sizeof(int32_t) * (size_t)n * n, hipMemcpyDeviceToHost, h->stream));
  // the pairs, at their destinations' offsets
  if ((size_t)(2 * total) > h->sp_send_cap) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if ((rc = dev_alloc(h, &h->d_sp_send, (size_t)(2 * total) + (size_t)(2 * total) / 8 + 64))) return rc;
    h->sp_send_cap = (size_t)(2 * total) + (size_t)(2 * total) / 8 + 64;
  }
  HIP_TRY(h, hipMemcpyAsync(h->d_sp_count, soff.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, h->stream));   // the destinations' blocks
  if (h->sp_wgs > 0) {
    SparseCountParams sp = sparse_params(h);
    sp.wg_off = h->d_sp_wg_off; sp.dest_base = h->d_sp_count; sp.pairs = h->d_sp_send;
    hipLaunchKernelGGL(sparse_count_kernel<true>, dim3((unsigned)h->sp_wgs), dim3(256), (size_t)h->K * sizeof(int32_t), h->stream, sp);
    HIP_TRY(h, hipGetLastError());
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));        // `all` has arrived (and soff / mine32 may go out of scope)
  int64_t rtotal = 0;
  for (int s2 = 0; s2 < n; ++s2) {
    rcnt[(size_t)s2] = 2 * (int64_t)all[(size_t)s2 * n + me];
    roff[(size_t)s2] = rtotal;
    rtotal += rcnt[(size_t)s2];
  }
  if ((size_t)rtotal > h->sp_recv_cap) {
    if ((rc = dev_alloc(h, &h->d_sp_recv, (size_t)rtotal + (size_t)rtotal / 8 + 64))) return rc;
    h->sp_recv_cap = (size_t)rtotal + (size_t)rtotal / 8 + 64;
  }
  if ((rc = xcall(h, h->xg->ops.all_to_all_v_i32(h->xg->ops.ctx, h->d_sp_send, soff.data(), scnt.data(), h->d_sp_recv, roff.data(), rcnt.data(), h->stream), "all_to_all_v_i32")))
    return rc;
  HIP_TRY(h, hipMemsetAsync(h->d_cnt_own, 0, sizeof(int32_t) * (size_t)h->V * h->Ksm, h->stream));
  if (rtotal > 0) {
    hipLaunchKernelGGL(scatter_add_pairs_kernel, dim3(grid_for(rtotal / 2, 256)), dim3(256), 0, h->stream, h->d_sp_recv, rtotal / 2, h->d_cnt_own);
    HIP_TRY(h, hipGetLastError());
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));        // the host arrays handed to the transport stay alive until it has used them
  h->cnt_own_valid = true;
  return GGS_OK;
}
int launch_count_rebuild(ggs_handle *h);
int exchange_reduce_scatter(ggs_handle *h) {
  if (h->cnt_own_valid) return GGS_OK;
  if (use_sparse(h) != h->counted_sparse) {            // the form was switched between the count and its exchange: count again in the form that travels
    int rc = launch_count_rebuild(h);
    if (rc) return rc;
  }
  if (use_sparse(h)) return sparse_exchange(h);
  int rc = xcall(h, h->xg->ops.reduce_scatter_i32(h->xg->ops.ctx, h->d_cnt_send, h->d_cnt_own, (int64_t)h->V * h->Ksm, h->stream), "reduce_scatter_i32");
  if (rc) return rc;
  h->cnt_own_valid = true;
  h->cnt_send_zeroed = false;
  return GGS_OK;
}
// the send buffer's contents have been through the reduce-scatter (cnt_own is their sum): dead, clear them on `on`
int clear_send_buffer_if_dead(ggs_handle *h, hipStream_t on) {
  if (!h->xg || h->cnt_send_zeroed || !h->cnt_own_valid || use_sparse(h)) return GGS_OK;
  return clear_send_buffer(h, on);
}

// Corpus-wide counts in d_n_wk: with an exchange, gathered from the ranks' slices on demand (a COLLECTIVE call).  In
// steps, so that a one-process group can issue each collective for all of its devices inside ncclGroupStart/End.
int gather_counts_step_gather(ggs_handle *h) {
  const size_t cells = (size_t)h->xg->nranks * h->V * h->Ksm;
  int rc;
  if (!h->d_cnt_all && (rc = dev_alloc(h, &h->d_cnt_all, cells))) return rc;
  return xcall(h, h->xg->ops.all_gather_i32(h->xg->ops.ctx, h->d_cnt_own, h->d_cnt_all, (int64_t)h->V * h->Ksm, h->stream), "all_gather_i32");
}
int gather_counts_step_unslice(ggs_handle *h) {
  hipLaunchKernelGGL(counts_unslice_kernel, dim3(grid_for((int64_t)h->K * h->V, 256)), dim3(256), 0, h->stream, h->d_cnt_all, h->d_koff, h->Ksm, h->d_n_wk,
                     h->K, h->V);
  HIP_TRY(h, hipGetLastError());
  h->counts_global = true; h->n_k_valid = false;
  return GGS_OK;
}
int ensure_global_counts(ggs_handle *h) {
  if (!h->xg || h->counts_global) return GGS_OK;
  int rc;
  if ((rc = exchange_reduce_scatter(h)) || (rc = clear_send_buffer_if_dead(h, h->stream)) || (rc = gather_counts_step_gather(h))) return rc;
  return gather_counts_step_unslice(h);
}

// tokensPerTopic (and the Dirichlet magnitudes) of the corpus-wide counts
int launch_magnitude(ggs_handle *h) {
  int rc = ensure_global_counts(h);
  if (rc) return rc;
  if (h->n_k_valid) return GGS_OK;
  if ((rc = launch_magnitude_on(h, h->d_n_wk, h->K, h->K, h->d_mag, h->d_n_k))) return rc;
  h->n_k_valid = true;
  return GGS_OK;
}

// Phi draw: initial (K8) or per sweep (K6) for the topics [k0, k0 + Ks) from their corpus-wide counts
// cnt [V][cnt_pitch]; rows into out [V][out_pitch].  In steps, so that the exchange can slot its collectives in:
//   phi_slice_magnitude   the Dirichlet magnitudes (and tokensPerTopic)
//   phi_slice_gamma       the gamma draws of the segments [seg0, seg1) + the segment functions of their column sums
//   phi_slice_total       the walk of those: tot[k] = sum_v gamma, in index order
//   phi_slice_normalise   one GPU: divide in place (with an exchange the division is done by the repack kernel)
int phi_slice_gamma(ggs_handle *h, bool initial, const int32_t *cnt, int32_t cnt_pitch, int32_t Ks, int32_t k0, double *out, int32_t out_pitch,
                    const double *mag, int32_t seg0, int32_t seg1) {
  if (Ks <= 0 || seg1 <= seg0) return GGS_OK;
  PhiGammaParams gp{};
  gp.n_wk = cnt; gp.mag = mag; gp.phiT = out; gp.status = h->d_status;
  gp.seed = h->seed; gp.iteration = (uint32_t)h->iteration;
  gp.purpose = initial ? GGS_PURPOSE_INIT_PHI : GGS_PURPOSE_PHI;
  gp.K = Ks; gp.Kp = out_pitch; gp.V = h->V; gp.cnt_pitch = cnt_pitch; gp.k0 = k0; gp.beta = h->beta;
  // Dirichlet(int size, double beta): magnitude = V*beta, partition = 1.0/V
  gp.prior_pm = (1.0 / (double)h->V) * ((double)h->V * h->beta);
  gp.initial = initial ? 1 : 0;
  // tiles of one 64-row segment x kc <= kPhiCols topics, the column groups as equal as they come.  A single wave per
  // tile issues an instruction every ~8 cycles, so what makes the draw fast is waves per SIMD: a narrow slice (one rank
  // in eight draws 13 topics) is cut into narrower tiles until there are ~4 waves for each of the chip's SIMDs; the price
  // is a thinner leftover queue per tile (the general rejection loops then run with fewer lanes in use).
  const int64_t want_tiles = (int64_t)h->num_cus * 4 * 4;
  int kc_cap = (int)std::min<int64_t>(kPhiCols, std::max<int64_t>(2, (int64_t)Ks * (seg1 - seg0) / want_tiles));
  if (const char *e = debug_env("GGS_DEBUG_PHICOLS")) kc_cap = std::max(1, std::min(kPhiCols, std::atoi(e)));
  gp.ncg = (Ks + kc_cap - 1) / kc_cap;
  gp.kc = (Ks + gp.ncg - 1) / gp.ncg;
  gp.ncg = (Ks + gp.kc - 1) / gp.kc;
  gp.seg_begin = seg0; gp.seg_end = seg1;
  gp.queue_cap = std::min(kPhiQueue, h->gamma_queue_cap);
  gp.prio = h->xg ? 1 : 0;
  gp.guess = h->exact_sum ? h->d_sum_pref : nullptr; gp.fn = h->d_sum_fn;
  const int64_t tiles = (int64_t)(seg1 - seg0) * gp.ncg;
  // single-wave workgroups; a grid of at most 32 per CU, the rest by striding
  hipLaunchKernelGGL(phi_gamma_kernel, dim3((unsigned)std::min<int64_t>(tiles, (int64_t)h->num_cus * 32)), dim3(64), 0, h->stream, gp);
  HIP_TRY(h, hipGetLastError());
  return GGS_OK;
}
int phi_slice_total(ggs_handle *h, const double *out, int32_t out_pitch, int32_t Ks, double *tot) {
  if (Ks <= 0) return GGS_OK;
  if (h->exact_sum) {
    SumParams sp{};
    sp.src = out; sp.guess = h->d_sum_pref; sp.fn = h->d_sum_fn; sp.out = tot; sp.beta = h->beta; sp.n_k = nullptr;
    sp.pitch = out_pitch; sp.K = Ks; sp.V = h->V; sp.nseg = h->sum_nseg; sp.write_pref = 0;   // the guess stays the magnitudes' running sums
    hipLaunchKernelGGL((sum_walk_kernel<double, false>), dim3((unsigned)Ks), dim3(64), 0, h->stream, sp);
  } else {
    launch_column_sum<double, false>(h, out, out_pitch, Ks, tot, nullptr, false, false);
  }
  HIP_TRY(h, hipGetLastError());
  return GGS_OK;
}
int launch_phi_slice(ggs_handle *h, bool initial, const int32_t *cnt, int32_t cnt_pitch, int32_t Ks, int32_t k0, double *out, int32_t out_pitch,
                     double *mag, double *tot, int32_t *n_k, double *phi_mean) {
  if (Ks <= 0) return GGS_OK;
  int rc;
  if ((rc = launch_magnitude_on(h, cnt, cnt_pitch, Ks, mag, n_k)) || (rc = phi_slice_gamma(h, initial, cnt, cnt_pitch, Ks, k0, out, out_pitch, mag, 0, h->sum_nseg)) ||
      (rc = phi_slice_total(h, out, out_pitch, Ks, tot)))
    return rc;
  hipLaunchKernelGGL(phi_normalise_kernel, dim3(grid_for((int64_t)Ks * h->V, 256, 2)), dim3(256), 0, h->stream, out, tot, Ks, out_pitch, h->V, phi_mean);
  HIP_TRY(h, hipGetLastError());
  return GGS_OK;
}

// The steps of the Phi phase with an exchange attached (ggs_group_sweep issues each for all handles of a one-process
// group, the collective ones inside ncclGroupStart/End; one handle per process runs them back to back):
//   A     reduce-scatter of the counts by topic slice
//   B1    this rank's slice: magnitudes, the gammas of the first half of the vocabulary
//   G0    all-gather of that half on the communication stream -- it runs UNDER step B2
//   B2    the gammas of the second half, then the walk of the column sums (it needs every gamma of the slice)
//   G1    all-gather of the second half with the Ksm column sums behind it, on the main stream, issued behind G0
//   C     repack [nranks][..] -> phiT [V][Kp], dividing by the owner's column sum on the way (the same IEEE division
//         the one-GPU normalise kernel does: bit-identical) (+ the running phi mean)
// What travels is therefore the UNNORMALISED gammas: the division does not have to wait for the slice to be complete
// before the first bytes go out.  Events are recorded by the callers AFTER a grouped step (inside
// ncclGroupStart/End the collectives are only collected, not enqueued).
size_t half0_elems(const ggs_handle *h) { return (size_t)h->v_split * h->Ksm; }
size_t half1_elems(const ggs_handle *h) { return (size_t)(h->V - h->v_split) * h->Ksm + (size_t)h->Ksm; }
int phi_step_a(ggs_handle *h) { return exchange_reduce_scatter(h); }
// behind step A, outside any ncclGroupStart/End: a short vocabulary has no communication-stream step to hide the fill under
int phi_step_a_clear(ggs_handle *h) { return h->seg_split > 0 ? GGS_OK : clear_send_buffer_if_dead(h, h->stream); }
int phi_step_b1(ggs_handle *h, bool initial) {
  int rc;
  if ((rc = launch_magnitude_on(h, h->d_cnt_own, h->Ksm, h->Ks, h->d_mag_own, h->d_n_k_own))) return rc;
  if (h->seg_split > 0) {
    if ((rc = phi_slice_gamma(h, initial, h->d_cnt_own, h->Ksm, h->Ks, h->k0, h->d_phi_own, h->Ksm, h->d_mag_own, 0, h->seg_split))) return rc;
    HIP_TRY(h, hipEventRecord(h->ev_half_drawn, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->comm_stream, h->ev_half_drawn, 0));
  }
  return GGS_OK;
}
int phi_step_g0(ggs_handle *h) {
  if (h->seg_split <= 0) return GGS_OK;
  return xcall(h, h->xg->ops.all_gather_f64(h->xg->ops.ctx, h->d_phi_own, h->d_phi_all0, (int64_t)half0_elems(h), h->comm_stream), "all_gather_f64");
}
int phi_step_b2(ggs_handle *h, bool initial) {
  int rc;
  if (h->seg_split > 0) {
    // behind the first all-gather on the communication stream (which waited for the slice's first half, hence for the
    // reduce-scatter before it): the send buffer's zero fill, deferred by phi_step_a
    if ((rc = clear_send_buffer_if_dead(h, h->comm_stream))) return rc;
    HIP_TRY(h, hipEventRecord(h->ev_half_gathered, h->comm_stream));
  }
  if ((rc = phi_slice_gamma(h, initial, h->d_cnt_own, h->Ksm, h->Ks, h->k0, h->d_phi_own, h->Ksm, h->d_mag_own, h->seg_split, h->sum_nseg))) return rc;
  return phi_slice_total(h, h->d_phi_own, h->Ksm, h->Ks, h->d_phi_own + (size_t)h->V * h->Ksm);
}
int phi_step_g1(ggs_handle *h) {
  return xcall(h, h->xg->ops.all_gather_f64(h->xg->ops.ctx, h->d_phi_own + half0_elems(h), h->d_phi_all1, (int64_t)half1_elems(h), h->stream), "all_gather_f64");
}
int phi_join_halves(ggs_handle *h) {                   // the main stream goes on only when the first half has arrived too
  if (h->seg_split > 0) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_half_gathered, 0));
  return GGS_OK;
}
int phi_step_c(ggs_handle *h, bool accumulate_mean) {
  PhiRepackParams rp{};
  rp.all0 = h->d_phi_all0; rp.all1 = h->d_phi_all1; rp.krank = h->d_krank; rp.kcol = h->d_kcol; rp.phiT = h->d_phiT;
  rp.phi_mean = accumulate_mean ? h->d_phi_mean : nullptr;
  rp.c0 = (int64_t)half0_elems(h); rp.c1 = (int64_t)half1_elems(h); rp.K = h->K; rp.Kp = h->Kp; rp.V = h->V; rp.Ksm = h->Ksm; rp.v_split = h->v_split;
  hipLaunchKernelGGL(phi_repack_kernel, dim3(grid_for((int64_t)h->K * h->V, 256, 2)), dim3(256), 0, h->stream, rp);
  HIP_TRY(h, hipGetLastError());
  h->have_phi = true;
  return GGS_OK;
}

// Phi draw: initial (K8) or per sweep (K6).  One GPU: the whole matrix in place (and tokensPerTopic refreshed);
// with an exchange: steps A, B, C above.
int launch_phi(ggs_handle *h, bool initial, bool accumulate_mean, Events *E = nullptr) {
  int rc;
  if (h->xg) {
    if ((rc = phi_step_a(h)) || (rc = phi_step_a_clear(h))) return rc;
    if (E) HIP_TRY(h, hipEventRecord(E->x[0], h->stream));
    if ((rc = phi_step_b1(h, initial)) || (rc = phi_step_g0(h)) || (rc = phi_step_b2(h, initial))) return rc;
    if (E) HIP_TRY(h, hipEventRecord(E->x[1], h->stream));
    // the second all-gather is issued only behind the first: no two collectives of one communicator are ever in flight
    // on different streams at once (they could not share the links anyway), whatever the transport does about that itself
    if ((rc = phi_join_halves(h)) || (rc = phi_step_g1(h))) return rc;
    if (E) HIP_TRY(h, hipEventRecord(E->x[2], h->stream));
    return phi_step_c(h, accumulate_mean);
  }
  if ((rc = launch_phi_slice(h, initial, h->d_n_wk, h->K, h->K, 0, h->d_phiT, h->Kp, h->d_mag, h->d_tot, h->d_n_k, accumulate_mean ? h->d_phi_mean : nullptr)))
    return rc;
  h->n_k_valid = true;
  h->have_phi = true;
  return GGS_OK;
}

int launch_theta(ggs_handle *h, hipStream_t stream, double *dst, int32_t iteration, int64_t d0 = 0, int64_t d1 = -1, int32_t lds = 0, int32_t docs_per_block = 0) {
  if (d1 < 0) d1 = h->D;
  if (d1 <= d0) return GGS_OK;
  if (docs_per_block <= 0) docs_per_block = h->theta_docs_per_block;
  ThetaParams tp{};
  tp.doc_ptr = h->d_doc_ptr + d0; tp.z = h->d_z; tp.alpha = h->d_alpha; tp.theta = dst + (size_t)d0 * h->K; tp.status = h->d_status;
  tp.num_docs = d1 - d0; tp.doc_base = h->doc_base + d0; tp.seed = h->seed; tp.iteration = (uint32_t)iteration;
  tp.K = h->K; tp.docs_per_block = docs_per_block; tp.queue_cap = std::min(kGammaQueue, h->gamma_queue_cap);
  const int64_t grid = (d1 - d0 + docs_per_block - 1) / docs_per_block;
  hipLaunchKernelGGL(theta_kernel<kThetaBlock>, dim3((unsigned)grid), dim3(kThetaBlock), lds ? lds : h->theta_lds, stream, tp);
  HIP_TRY(h, hipGetLastError());
  return GGS_OK;
}

// z_sliced_kernel<KMAX> / z_hot_kernel<KMAX> for KMAX = K rounded up to a multiple of 8
#define GGS_KMAX_SWITCH(KERNEL)                                                                                                     \
  switch ((K + 7) / 8) {                                                                                                            \
    case 1: return reinterpret_cast<const void *>(KERNEL<8>);      case 2: return reinterpret_cast<const void *>(KERNEL<16>);       \
    case 3: return reinterpret_cast<const void *>(KERNEL<24>);     case 4: return reinterpret_cast<const void *>(KERNEL<32>);       \
    case 5: return reinterpret_cast<const void *>(KERNEL<40>);     case 6: return reinterpret_cast<const void *>(KERNEL<48>);       \
    case 7: return reinterpret_cast<const void *>(KERNEL<56>);     case 8: return reinterpret_cast<const void *>(KERNEL<64>);       \
    case 9: return reinterpret_cast<const void *>(KERNEL<72>);     case 10: return reinterpret_cast<const void *>(KERNEL<80>);      \
    case 11: return reinterpret_cast<const void *>(KERNEL<88>);    case 12: return reinterpret_cast<const void *>(KERNEL<96>);      \
    case 13: return reinterpret_cast<const void *>(KERNEL<104>);   case 14: return reinterpret_cast<const void *>(KERNEL<112>);     \
    case 15: return reinterpret_cast<const void *>(KERNEL<120>);   case 16: return reinterpret_cast<const void *>(KERNEL<128>);     \
    case 17: return reinterpret_cast<const void *>(KERNEL<136>);   case 18: return reinterpret_cast<const void *>(KERNEL<144>);     \
    case 19: return reinterpret_cast<const void *>(KERNEL<152>);   case 20: return reinterpret_cast<const void *>(KERNEL<160>);     \
    case 21: return reinterpret_cast<const void *>(KERNEL<168>);   case 22: return reinterpret_cast<const void *>(KERNEL<176>);     \
    case 23: return reinterpret_cast<const void *>(KERNEL<184>);   default: return reinterpret_cast<const void *>(KERNEL<192>);     \
  }
const void *sliced_kernel_for(int K) { GGS_KMAX_SWITCH(z_sliced_kernel) }
const void *hot_kernel_for(int K) { GGS_KMAX_SWITCH(z_hot_kernel) }
const void *warm_kernel_for(int K) { GGS_KMAX_SWITCH(z_warm_kernel) }
const void *pcgs_kernel_for(int K) { GGS_KMAX_SWITCH(pcgs_sliced_kernel) }
const void *collapsed_kernel_for(int K) {
  switch ((K + 7) / 8) {
#define GGS_CK(N) case N: return reinterpret_cast<const void *>(pcgs_sliced_kernel<8 * N, true>);
    GGS_CK(1) GGS_CK(2) GGS_CK(3) GGS_CK(4) GGS_CK(5) GGS_CK(6) GGS_CK(7) GGS_CK(8) GGS_CK(9) GGS_CK(10) GGS_CK(11) GGS_CK(12)
    GGS_CK(13) GGS_CK(14) GGS_CK(15) GGS_CK(16) GGS_CK(17) GGS_CK(18) GGS_CK(19) GGS_CK(20) GGS_CK(21) GGS_CK(22) GGS_CK(23)
#undef GGS_CK
    default: return reinterpret_cast<const void *>(pcgs_sliced_kernel<192, true>);
  }
}

// pcgs_wave_kernel<NB, COLLAPSED> for NB = blocks of 128 topics, rounded up to a power of two
const void *pcgs_wave_kernel_for(int nb, bool collapsed) {
#define GGS_WK(N) if (nb <= N) return collapsed ? reinterpret_cast<const void *>(pcgs_wave_kernel<N, true>) : reinterpret_cast<const void *>(pcgs_wave_kernel<N, false>);
  GGS_WK(1) GGS_WK(2) GGS_WK(4) GGS_WK(8) GGS_WK(16)
#undef GGS_WK
  return collapsed ? reinterpret_cast<const void *>(pcgs_wave_kernel<32, true>) : reinterpret_cast<const void *>(pcgs_wave_kernel<32, false>);
}
constexpr int kPcgsWaveMaxTopics = 32 * 128;          // 4096: two rows of K/64 doubles per lane in registers
// Where the wave-per-document kernel takes over from the lane-per-document score-register kernels (which exist up to 192
// topics; their 176..192 variants spill 12-164 bytes per lane to scratch).  Measured on the benchmark corpus, z step in ms,
// lane-per-document / wave-per-document (round 4, after the wave kernel's rework; the wave kernel's time steps with the
// number of 128-topic blocks: flat up to 128 topics, flat from 129 to 256):
//   pcgs       K = 100: 2.51 / 3.33, 128: 3.69 / 3.3, 144: 4.02 / 4.9, 160: 4.28 / 4.9, 176: 4.69 / 4.9, 192: 6.04 / 5.0
//   collapsed  K = 100: 3.60 / 3.74, 128: 6.31 / 3.71, 144: 6.96 / 5.6, 160: 7.63 / 5.6, 176: 8.62 / 5.6, 192: 11.4 / 5.6
constexpr int kPcgsWaveFromTopics = 176, kCollapsedWaveFromTopics = 96;

int launch_pcgs_z(ggs_handle *h) {
  if (h->N == 0) return GGS_OK;
  PcgsParams pp{};
  pp.tok = h->d_tok; pp.inv_perm = h->d_inv_perm; pp.z = h->d_z; pp.zw = h->d_zw; pp.doc_ptr = h->d_doc_ptr; pp.order = h->d_order;
  pp.alpha = h->d_alpha; pp.phiT = h->d_phiT; pp.status = h->d_status;
  pp.num_docs = h->pcgs_order_len; pp.tok_base = h->tok_base; pp.seed = h->seed; pp.iteration = (uint32_t)h->iteration;   // the length of the (padded) order list
  pp.K = h->K; pp.Kp = h->Kp;
  const int64_t groups = (h->pcgs_order_len + 63) / 64;
  const dim3 grid((unsigned)std::min<int64_t>(groups, (int64_t)h->num_cus * h->pcgs_waves_per_cu)), block(64);
  if (h->pcgs_wave) {
    // one wave per document: wide topic rows, or a document the lane-per-document kernels' int16 counts cannot hold
    if (h->collapsed) {
      int rc = launch_magnitude(h);
      if (rc) return rc;
      pp.n_wk = h->d_n_wk; pp.n_k = h->d_n_k; pp.beta = h->beta; pp.beta_sum = h->beta * (double)h->V;
      hipLaunchKernelGGL(psi_kernel, dim3(grid_for((int64_t)h->K * h->V, 256, 2)), dim3(256), 0, h->stream, h->d_n_wk, h->d_n_k, pp.beta, pp.beta_sum, h->d_phiT,
                         h->K, h->Kp, h->V);
    }
    const dim3 wgrid((unsigned)std::min<int64_t>(h->pcgs_order_len, (int64_t)h->num_cus * h->pcgs_wave_waves_per_cu));
    void *args[] = {&pp, &h->margin_scale};
    HIP_TRY(h, hipLaunchKernel(pcgs_wave_kernel_for(h->pcgs_wave_nb, h->collapsed), wgrid, block, args, (size_t)h->pcgs_wave_lds, h->stream));
    return GGS_OK;
  }
  if (h->collapsed) {
    // the sweep-start ratios (beta + n_wk)/(betaSum + n_k) of the corpus-wide counts, then the pcgs loop over them
    int rc = launch_magnitude(h);
    if (rc) return rc;
    pp.n_wk = h->d_n_wk; pp.n_k = h->d_n_k; pp.beta = h->beta; pp.beta_sum = h->beta * (double)h->V;   // betaSum = beta * numTypes, MSLDA:136
    hipLaunchKernelGGL(psi_kernel, dim3(grid_for((int64_t)h->K * h->V, 256, 2)), dim3(256), 0, h->stream, h->d_n_wk, h->d_n_k, pp.beta, pp.beta_sum, h->d_phiT,
                       h->K, h->Kp, h->V);
    if (h->pcgs_sliced) {
      void *args[] = {&pp};
      HIP_TRY(h, hipLaunchKernel(collapsed_kernel_for(h->K), grid, block, args, (size_t)h->pcgs_lds, h->stream));
    } else {
      hipLaunchKernelGGL(pcgs_z_kernel<true>, grid, block, (size_t)h->pcgs_lds, h->stream, pp);
    }
  } else if (h->pcgs_sliced) {
    void *args[] = {&pp};
    HIP_TRY(h, hipLaunchKernel(pcgs_kernel_for(h->K), grid, block, args, (size_t)h->pcgs_lds, h->stream));
  } else {
    hipLaunchKernelGGL(pcgs_z_kernel<false>, grid, block, (size_t)h->pcgs_lds, h->stream, pp);
  }
  HIP_TRY(h, hipGetLastError());
  return GGS_OK;
}

// `count`: with an exchange the sliced kernels add every token's new (word, topic) cell into the (zeroed) send buffer of the
// count reduce-scatter themselves; the timed comparison of the two z forms launches without it
// ... where few tokens share a cell: the atomics of different XCDs on one line serialise at the memory side.  Measured on the
// benchmark corpus split N ways (z step with / without them, and the count kernel they replace): N = 8 (50 tokens per word
// on the rank) 0.139 / 0.132 ms against 0.040 of count kernel; N = 4 (100) 0.290 / 0.26 against 0.05; N = 2 (200) 0.627 / 0.52
// against 0.07 -- they pay up to about 64 tokens per word (GGS_DEBUG_ZCOUNTS=2 forces them).
bool z_counts_itself(const ggs_handle *h) {
  return h->xg && h->z_counts && h->z_sliced && !(h->flags & GGS_FLAG_PCGS) && (h->z_counts_forced || h->N <= (int64_t)64 * h->V) && !use_sparse(h);
}
// `defer_join` (with `count`, split form): the hot words' count launch follows the hot kernel on ITS stream and the
// handle's stream is not made to wait for that stream here -- the caller does (join_hot_stream), behind the z step's end
// event, so that the hot chunks' count sits beside the tail of the cold kernel instead of behind it.
int launch_count_hot(ggs_handle *h);
int launch_z(ggs_handle *h, bool force_fused = false, int64_t c0 = 0, int64_t c1 = -1, bool count = false, bool defer_join = false) {
  h->hot_join_pending = false;
  if (h->C == 0) return GGS_OK;
  if (c1 < 0) c1 = h->C;
  if (c1 <= c0) return GGS_OK;
  ZParams zp{};
  zp.tok = h->d_tok; zp.inv_perm = h->d_inv_perm; zp.z = h->d_z; zp.zw = h->d_zw; zp.chunk_start = h->d_chunk_start; zp.chunk_doc = h->d_chunk_doc; zp.chunk_len = h->d_chunk_len;
  zp.theta = h->d_theta; zp.phiT = h->d_phiT; zp.status = h->d_status;
  zp.tok_base = h->tok_base; zp.seed = h->seed; zp.iteration = (uint32_t)h->iteration;
  zp.K = h->K; zp.Kp = h->Kp; zp.pitch16 = h->pitch16; zp.tile_tokens = h->z_tile_tokens;
  zp.num_chunks = h->C;
  zp.ablate = h->ablate;
  zp.margin_scale = h->margin_scale;
  zp.chunk_doc1 = h->d_chunk_doc1; zp.two_rows = (h->z_stream && h->z_two_rows && h->d_chunk_doc1) ? 1 : 0;
  zp.ct_tok = h->d_ct_tok; zp.ct_idx = h->d_ct_idx; zp.ct_ip = h->d_ct_ip; zp.c_docs = h->d_c_docs; zp.num_cold = h->Cc;
  zp.hot_words = h->d_hot_words; zp.num_hot = h->num_hot; zp.hot_pitch = h->hot_pitch;
  zp.wave_lds = h->wave_lds; zp.hot_off = kSlicedWaves * h->wave_lds; zp.ring_base = h->ring_base;
  zp.cnt_send = (count && z_counts_itself(h)) ? h->d_cnt_send : nullptr;
  zp.smap = h->smap;
  zp.ht_pack = reinterpret_cast<const int4 *>(h->d_ht_pack); zp.h_docs = h->d_h_docs;
  zp.wt_pack = reinterpret_cast<const int4 *>(h->d_wt_pack); zp.w_docs = h->d_w_docs; zp.warm_words = h->d_warm_words;
  zp.warm_meta = h->d_warm_meta; zp.warm_tiers = h->warm_tiers; zp.warm_rows = h->warm_cap;
  if (!h->z_sliced) {                                  // a range of the chunk table (the one-document chunks are in document order)
    zp.chunk_start += c0; zp.chunk_doc += c0; zp.chunk_len += c0; zp.num_chunks = c1 - c0;
    if (zp.chunk_doc1) zp.chunk_doc1 += c0;
  }
  // persistent waves: as many single-wave workgroups as stay resident, each strides the chunk table
  const dim3 grid((unsigned)std::min<int64_t>(zp.num_chunks, (int64_t)h->num_cus * h->z_waves_per_cu)), block(64);
  const int nt = (h->K + 63) / 64;
  if (h->z_sliced) {
    // one 4-wave workgroup per CU (a wave per SIMD), persistent; the hot-word table fills the LDS the rings leave
    zp.num_chunks = h->Cs;
    void *args[] = {&zp};
    const dim3 sblock(kSlicedWaves * 64);
    auto grid_of = [&](int64_t chunks) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((chunks + kSlicedWaves - 1) / kSlicedWaves, (int64_t)h->num_cus))); };
    // the warm tiers (z_warm_kernel): behind the hot chunks on their stream, or behind the one kernel of the fused form
    ZParams wp = zp;
    wp.wave_lds = h->warm_wave_lds; wp.hot_off = kSlicedWaves * h->warm_wave_lds;
    void *wargs[] = {&wp};
    auto launch_warm = [&](hipStream_t st) -> int {
      if (h->Cw == 0) return GGS_OK;
#ifdef GGS_WARM_TRACE
      static long long *dbg = nullptr;
      static int launches = 0;
      if (!dbg) HIP_TRY(h, hipMalloc(&dbg, sizeof(long long) * 8 * 1024 * 4));
      wp.dbg = dbg;
#endif
      HIP_TRY(h, hipLaunchKernel(warm_kernel_for(h->K), grid_of(h->warm_chunks_max), sblock, wargs, (size_t)(wp.hot_off + h->warm_rows_max * h->hot_pitch + kHotTailBytes), st));
#ifdef GGS_WARM_TRACE
      if (++launches == 12) {                                        // a steady-state sweep: print the phase averages once
        HIP_TRY(h, hipDeviceSynchronize());
        std::vector<long long> v(8 * 1024);
        HIP_TRY(h, hipMemcpy(v.data(), dbg, sizeof(long long) * v.size(), hipMemcpyDeviceToHost));
        double a[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int w = 0; w < 1024; ++w) for (int i = 0; i < 7; ++i) a[i] += (double)v[(size_t)w * 8 + i] / 1024;
        fprintf(stderr, "[warm trace] cycles per wave: wait %.0f stage %.0f issue %.0f arithmetic %.0f stores %.0f rest %.0f | kernel %.0f\n", a[0], a[1], a[2], a[3], a[4], a[5], a[6]);
      }
#endif
      return GGS_OK;
    };
    if (h->z_split && !force_fused && h->Cs > h->Cc && h->Cc > 0) {
      // cold chunks on the main stream, hot chunks beside them (z_hot_kernel): two waves per SIMD
      ZParams hp = zp;
      hp.wave_lds = h->hot_wave_lds; hp.hot_off = kSlicedWaves * h->hot_wave_lds;
      void *hargs[] = {&hp};
      if (h->hot_fork_from) {
        HIP_TRY(h, hipStreamWaitEvent(h->side_hot, h->hot_fork_from, 0));
      } else {
        HIP_TRY(h, hipEventRecord(h->ev_hot_fork, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->side_hot, h->ev_hot_fork, 0));
      }
      zp.num_chunks = h->Cc; zp.num_hot = 0;
      static const int only = debug_env("GGS_DEBUG_ONLY") ? std::atoi(debug_env("GGS_DEBUG_ONLY")) : 0;   // timing experiments: 1 cold only, 2 hot only
      if (only != 2) HIP_TRY(h, hipLaunchKernel(sliced_kernel_for(h->K), grid_of(h->Cc), sblock, args, (size_t)(kSlicedWaves * h->wave_lds), h->stream));
      if (only != 1) HIP_TRY(h, hipLaunchKernel(hot_kernel_for(h->K), grid_of(h->Cs - h->Cc), sblock, hargs, (size_t)(hp.hot_off + h->num_hot * h->hot_pitch + kHotTailBytes), h->side_hot));
      if (only != 1) { const int rc = launch_warm(h->side_hot); if (rc) return rc; }
      if (defer_join && zp.cnt_send && only == 0) {
        hipStream_t main_stream = h->stream;
        h->stream = h->side_hot;
        const int rc = launch_count_hot(h);
        h->stream = main_stream;
        if (rc) return rc;
        h->hot_counted = true;
      }
      HIP_TRY(h, hipEventRecord(h->ev_hot_join, h->side_hot));
      if (defer_join) h->hot_join_pending = true;
      else HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_hot_join, 0));
    } else {
      HIP_TRY(h, hipLaunchKernel(sliced_kernel_for(h->K), grid_of(std::max(h->Cc, h->Cs - h->Cc)), sblock, args,
                                 (size_t)(kSlicedWaves * h->wave_lds + h->num_hot * h->hot_pitch), h->stream));
      const int rc = launch_warm(h->stream);
      if (rc) return rc;
    }
  } else if (h->z_stream && h->z_two_pass) hipLaunchKernelGGL(z_stream_kernel, grid, block, h->z_lds, h->stream, zp);
  else if (h->z_stream && h->z_regck && h->z_group == 1) hipLaunchKernelGGL((z_stream1_kernel<true, 1>), grid, block, h->z_lds, h->stream, zp);
  else if (h->z_stream && h->z_regck && h->z_group == 2) hipLaunchKernelGGL((z_stream1_kernel<true, 2>), grid, block, h->z_lds, h->stream, zp);
  else if (h->z_stream && h->z_regck) hipLaunchKernelGGL((z_stream1_kernel<true, 4>), grid, block, h->z_lds, h->stream, zp);
  else if (h->z_stream) hipLaunchKernelGGL((z_stream1_kernel<false, 4>), grid, block, h->z_lds, h->stream, zp);
  else if (nt <= 1) hipLaunchKernelGGL(z_kernel<1>, grid, block, h->z_lds, h->stream, zp);
  else if (nt <= 2) hipLaunchKernelGGL(z_kernel<2>, grid, block, h->z_lds, h->stream, zp);
  else if (nt <= 4) hipLaunchKernelGGL(z_kernel<4>, grid, block, h->z_lds, h->stream, zp);
  else if (nt <= 8) hipLaunchKernelGGL(z_kernel<8>, grid, block, h->z_lds, h->stream, zp);
  else if (nt <= 16) hipLaunchKernelGGL(z_kernel<16>, grid, block, h->z_lds, h->stream, zp);
  else hipLaunchKernelGGL(z_kernel<20>, grid, block, h->z_lds, h->stream, zp);
  HIP_TRY(h, hipGetLastError());
  return GGS_OK;
}

bool sample_phi_this_iteration(const ggs_handle *h) {  // UPLDA:1350-1352
  return h->phi_burn_in > 0 && h->iteration > h->phi_burn_in && (h->iteration % h->phi_thin) == 0;
}

// java.util.Random(seed).nextInt(bound), n times (JDK 8 javadoc algorithm: 48-bit LCG,
// next(31), power-of-two shortcut, modulo rejection loop otherwise).
uint64_t java_lcg_next_ints(int32_t seed, int32_t bound, int64_t n, int32_t *out) {   // returns the generator's state after the n draws
  constexpr uint64_t kMult = 0x5DEECE66DULL, kMask = (1ULL << 48) - 1;
  uint64_t s = ((uint64_t)(int64_t)seed ^ kMult) & kMask;
  auto next31 = [&]() { s = (s * kMult + 0xBULL) & kMask; return (int32_t)(s >> 17); };
  const int32_t m = bound - 1;
  for (int64_t i = 0; i < n; ++i) {
    int32_t r = next31();
    if ((bound & m) == 0) r = (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
    else
      for (int32_t u = r;; u = next31()) {
        r = u % bound;
        if ((int32_t)((uint32_t)u - (uint32_t)r + (uint32_t)m) >= 0) break;
      }
    out[i] = r;
  }
  return s;
}

int require_ready(ggs_handle *h, bool need_phi) {
  if (!h) return GGS_ERR_BAD_ARG;
  if (!h->have_corpus) return set_err(h, GGS_ERR_STATE, "no corpus: call ggs_set_corpus first");
  if (need_phi && !h->have_phi) return set_err(h, GGS_ERR_STATE, "no Phi: call ggs_init_phi / ggs_set_z / ggs_set_phi first");
  return bind_device(h);
}

// The side-stream theta must not outlive the z it was drawn from.
int drop_theta_ahead(ggs_handle *h) {
  if (h->side) HIP_TRY(h, hipStreamSynchronize(h->side));
  h->theta_ahead_iter = INT64_MIN;
  return GGS_OK;
}

// waits for every enqueued sweep and adds its phase times to the timings
int settle_sweeps(ggs_handle *h) {
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int j = h->ev_pending - 1; j >= 0; --j) {
    const Events &E = h->evs[((h->ev_head - j) % kEvRing + kEvRing) % kEvRing];
    float ms = 0;
    // a consumed ahead-draw: the stream waited on th1 before the z kernel, so both side-stream events are complete;
    // their span is the duration of a kernel that ran beside the previous sweep's Phi phase
    // (every event or barrier packet on the critical stream costs ~6 us of device time: a theta drawn on it starts at the
    // previous sweep's e[2])
    if (E.used_ahead && E.theta_on_main) { HIP_TRY(h, hipEventElapsedTime(&ms, h->evs[((h->ev_head - j - 1) % kEvRing + kEvRing) % kEvRing].e[2], E.th1)); }
    else if (E.used_ahead) { HIP_TRY(h, hipEventElapsedTime(&ms, E.th0, E.th1)); }
    else { HIP_TRY(h, hipEventElapsedTime(&ms, E.e[0], E.e[1])); }
    h->tm.theta_ms += ms;
    HIP_TRY(h, hipEventElapsedTime(&ms, E.e[1], E.e[2])); h->tm.z_ms += ms;
    if (E.light) {                                       // the ends only: the phases in the last fully timed sweep's proportions
      HIP_TRY(h, hipEventElapsedTime(&ms, E.e[2], E.e[5]));
      h->tm.merge_ms += ms * h->frac[0];
      h->tm.exchange_rs_ms += ms * h->frac[1]; h->tm.exchange_ag_ms += ms * h->frac[3]; h->tm.exchange_ms += ms * (h->frac[1] + h->frac[3]);
      h->tm.phi_ms += ms * (h->frac[2] + h->frac[4]);
      h->tm.sweeps += 1;
      h->tm.tokens_sampled += h->N;
      continue;
    }
    HIP_TRY(h, hipEventElapsedTime(&ms, E.e[2], E.e[3])); h->tm.merge_ms += ms;
    if (E.exchanged) {
      float span = 0, part[5] = {ms, 0, 0, 0, 0};
      (void)hipEventElapsedTime(&span, E.e[2], E.e[5]);
      (void)hipEventElapsedTime(&part[1], E.e[E.e4_is_e3 ? 3 : 4], E.x[0]); (void)hipEventElapsedTime(&part[2], E.x[0], E.x[1]);
      (void)hipEventElapsedTime(&part[3], E.x[1], E.x[2]); (void)hipEventElapsedTime(&part[4], E.x[2], E.e[5]);
      if (span > 0) {
        // what lies between e[3] and e[4] in a split sweep (the caller's own work) is nobody's phase: renormalise
        const float sum = part[0] + part[1] + part[2] + part[3] + part[4];
        for (int q = 0; q < 5; ++q) h->frac[q] = sum > 0 ? part[q] / sum : 0.0;
        h->have_frac = true;
      }   // reduce-scatter | slice draw (the first half's all-gather beneath it) | what is left of the all-gathers | repack
      HIP_TRY(h, hipEventElapsedTime(&ms, E.e[E.e4_is_e3 ? 3 : 4], E.x[0])); h->tm.exchange_ms += ms; h->tm.exchange_rs_ms += ms;
      HIP_TRY(h, hipEventElapsedTime(&ms, E.x[0], E.x[1])); h->tm.phi_ms += ms;
      HIP_TRY(h, hipEventElapsedTime(&ms, E.x[1], E.x[2])); h->tm.exchange_ms += ms; h->tm.exchange_ag_ms += ms;
      HIP_TRY(h, hipEventElapsedTime(&ms, E.x[2], E.e[5])); h->tm.phi_ms += ms;
    } else {
      HIP_TRY(h, hipEventElapsedTime(&ms, E.e[E.e4_is_e3 ? 3 : 4], E.e[5])); h->tm.phi_ms += ms;
    }
    h->tm.sweeps += 1;
    h->tm.tokens_sampled += h->N;
  }
  h->ev_pending = 0;
  return GGS_OK;
}

// the handle's launches go to another stream for a scope (the count rebuild and the Phi chain of a whole sweep)
struct StreamSwap {
  ggs_handle *h;
  hipStream_t user;
  StreamSwap(ggs_handle *h_, hipStream_t s) : h(h_), user(h_->stream) { h->stream = s; }
  ~StreamSwap() { h->stream = user; }
};

int z_phase(ggs_handle *h) {
  int rc;
  if (h->ev_pending >= kEvRing - 2 && (rc = settle_sweeps(h))) return rc;   // keep this slot and the next one free
  h->ev_head = (h->ev_head + 1) % kEvRing;
  Events &E = h->evs[h->ev_head];
  E.light = h->xg && h->have_frac && h->detail_every > 1 && !h->force_detail && (h->sweeps_enqueued % h->detail_every) != 0 && !h->collapsed;
  h->sweeps_enqueued += 1;
  if (h->flags & GGS_FLAG_PCGS) {                       // no theta: it is integrated out (UPLDA:1509-1513)
    E.used_ahead = false;
    HIP_TRY(h, hipEventRecord(E.e[0], h->stream));
    HIP_TRY(h, hipEventRecord(E.e[1], h->stream));
    if ((rc = launch_pcgs_z(h))) return rc;
    HIP_TRY(h, hipEventRecord(E.e[2], h->stream));
    if ((rc = launch_count_rebuild(h))) return rc;
    if (!E.light) HIP_TRY(h, hipEventRecord(E.e[3], h->stream));
    return GGS_OK;
  }
  E.used_ahead = h->theta_ahead_iter == (int64_t)h->iteration;
  h->hot_fork_from = nullptr;
  if (E.used_ahead) {
    if (E.theta_on_main) h->hot_fork_from = E.th1;          // drawn on this very stream: nothing to wait for, and the hot kernel's stream (which ran the Phi chain) forks from th1
    else HIP_TRY(h, hipStreamWaitEvent(h->stream, E.th1, 0));   // drawn during the previous iteration's Phi phase
    std::swap(h->d_theta, h->d_theta_next);
  } else {
    E.theta_on_main = false;
    if (h->side) HIP_TRY(h, hipStreamSynchronize(h->side));
    HIP_TRY(h, hipEventRecord(E.e[0], h->stream));
    if ((rc = launch_theta(h, h->stream, h->d_theta, h->iteration))) return rc;
  }
  HIP_TRY(h, hipEventRecord(E.e[1], h->stream));
  if (h->z_sliced && h->z_split && !h->z_split_tried && h->Cs > h->Cc && h->Cc > 0) {
    // Whether the guest kernel really runs beside the cold one depends on how the runtime maps streams to hardware
    // queues (measured: beside each other in a plain process, one after the other under torchrun + RCCL).  Both forms
    // write the same z, so the first z step of a corpus simply runs twice, timed, and the slower form is dropped.
    h->z_split_tried = true;
    float t_split = 0, t_fused = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr, t2 = nullptr;
    if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess || hipEventCreate(&t2) != hipSuccess) return set_err(h, GGS_ERR_HIP, "hipEventCreate");
    if ((rc = launch_z(h, false)) || (rc = launch_z(h, true))) return rc;   // untimed: code upload, cold caches
    HIP_TRY(h, hipEventRecord(t0, h->stream));
    if ((rc = launch_z(h, false))) return rc;
    HIP_TRY(h, hipEventRecord(t1, h->stream));
    if ((rc = launch_z(h, true))) return rc;
    HIP_TRY(h, hipEventRecord(t2, h->stream));
    HIP_TRY(h, hipEventSynchronize(t2));
    HIP_TRY(h, hipEventElapsedTime(&t_split, t0, t1));
    HIP_TRY(h, hipEventElapsedTime(&t_fused, t1, t2));
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1); (void)hipEventDestroy(t2);
    if (t_fused < t_split) h->z_split = false;
    HIP_TRY(h, hipEventRecord(E.e[1], h->stream));          // this sweep's z time: one launch of the form kept
  }
  if (!h->hot_fork_from) h->hot_fork_from = E.e[1];        // the event just recorded serves as the hot chunks' fork: one packet less
  h->theta_ahead_iter = INT64_MIN;
  const bool ahead = h->overlap_theta && h->side && h->D > 0;
  const int32_t P = (int32_t)h->part_doc.size() - 1;   // 1 for a small corpus
  if (ahead && P > 1 && h->z_stream) {
    // part by part: z of part p on the main stream, then -- beside z of part p + 1 -- theta of iteration t+1 for the
    // documents of part p on the side stream; the last part's theta runs beside the counts and the Phi draw
    Events &N = h->evs[(h->ev_head + 1) % kEvRing];
    for (int32_t p = 0; p < P; ++p) {
      if ((rc = launch_z(h, false, h->part_chunk[(size_t)p], h->part_chunk[(size_t)p + 1]))) return rc;
      HIP_TRY(h, hipEventRecord(h->ev_part[p], h->stream));
      HIP_TRY(h, hipStreamWaitEvent(h->side, h->ev_part[p], 0));
      if (p == 0) { N.theta_on_main = false; HIP_TRY(h, hipEventRecord(N.th0, h->side)); }
      if ((rc = launch_theta(h, h->side, h->d_theta_next, h->iteration + 1, h->part_doc[(size_t)p], h->part_doc[(size_t)p + 1],
                             p + 1 < P ? h->theta_lds_beside_z : 0)))
        return rc;
    }
    HIP_TRY(h, hipEventRecord(N.th1, h->side));
    HIP_TRY(h, hipEventRecord(E.e[2], h->stream));
    h->theta_ahead_iter = (int64_t)h->iteration + 1;
  } else {
    // TIMING EXPERIMENT (GGS_DEBUG_THETA_EARLY=1; results are WRONG on purpose): the next theta is launched BESIDE the z
    // step instead of behind it -- drawn from the z the step is still writing.  What it measures is the best case of
    // "theta_{t+1} of document part p beside the z step of part p + 1"  without the parts' own costs
    // (an event packet, a launch and the drain of the persistent z waves each): the z step with a theta workgroup as the
    // SIMDs' guest, and the counts + Phi chain alone behind it.  Needs LDS beside the z kernels: GGS_DEBUG_HOT shrinks
    // the hot-word table.  DESIGN.md section 5 has the numbers.
    static const int theta_early = debug_env("GGS_DEBUG_THETA_EARLY") ? std::atoi(debug_env("GGS_DEBUG_THETA_EARLY")) : 0;
    bool early = false;
    if (theta_early && ahead && !h->xg && h->z_sliced) {
      Events &N = h->evs[(h->ev_head + 1) % kEvRing];
      h->chain_on_side = false;
      N.theta_on_main = false;
      HIP_TRY(h, hipStreamWaitEvent(h->side, E.e[1], 0));
      HIP_TRY(h, hipEventRecord(N.th0, h->side));
      const int b = std::max(1, std::min(h->theta_docs_per_block, theta_early > 1 ? theta_early : 16));
      const int bp = b | 1;
      if ((rc = launch_theta(h, h->side, h->d_theta_next, h->iteration + 1, 0, -1, (int32_t)((size_t)h->K * bp * 8 + (size_t)b * 20 + kThetaQueueBytes), b))) return rc;
      HIP_TRY(h, hipEventRecord(N.th1, h->side));
      h->theta_ahead_iter = (int64_t)h->iteration + 1;
      early = true;
    }
    const bool counting = z_counts_itself(h) && h->C > 0;
    if (counting && !h->cnt_send_zeroed && (rc = clear_send_buffer(h, h->stream))) return rc;   // nobody cleared it behind the last reduce-scatter (or none came): counts no z step asked for are overwritten, as a count rebuild overwrites them
    h->hot_counted = false;
    if ((rc = launch_z(h, false, 0, -1, counting, /*defer_join=*/counting))) return rc;
    if (counting) { h->z_counted = true; h->cnt_send_zeroed = false; h->counted_sparse = false; }
    HIP_TRY(h, hipEventRecord(E.e[2], h->stream));      // the cold kernel's end; the hot chunks' stream is joined below, and by the theta draw's stream
    h->chain_on_side = h->chain_on_side && ahead;
    // TIMING EXPERIMENT (GGS_DEBUG_THETA_TAIL_PCT=p; results are WRONG on purpose): the table kernels' stream ends ~0.2 ms
    // before the cold kernel does, and their LDS is free from then on.  What would the next theta of the first p % of the
    // documents cost and save if it ran THERE (behind z_warm_kernel, beside the cold kernel's tail), the rest behind the z
    // step as always?  Drawn from the z the cold kernel is still writing -- the real thing needs the z step cut into two
    // document parts whose first is complete by then.  DESIGN.md section 5 has the numbers.
    static const int theta_tail_pct = debug_env("GGS_DEBUG_THETA_TAIL_PCT") ? std::max(0, std::min(100, std::atoi(debug_env("GGS_DEBUG_THETA_TAIL_PCT")))) : 0;
    int64_t d_tail = 0;
    if (theta_tail_pct && ahead && !early && !h->xg && h->z_sliced && h->z_split && h->chain_on_side && h->side_hot) {
      if (!h->ev_theta_tail) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_theta_tail, hipEventDisableTiming));
      d_tail = h->D * theta_tail_pct / 100;
      // GGS_DEBUG_THETA_TAIL_STREAM=1: on a stream of its own behind the table kernels (the count + Phi chain, which runs on
      // their stream, then starts when the cold kernel ends and not behind this draw)
      static const bool own = debug_env("GGS_DEBUG_THETA_TAIL_STREAM") && std::atoi(debug_env("GGS_DEBUG_THETA_TAIL_STREAM")) == 1;
      hipStream_t tail_on = own ? h->side : h->side_hot;
      if (own) HIP_TRY(h, hipStreamWaitEvent(h->side, h->ev_hot_join, 0));
      if ((rc = launch_theta(h, tail_on, h->d_theta_next, h->iteration + 1, 0, d_tail, h->theta_lds_main, h->theta_b_main))) return rc;
      HIP_TRY(h, hipEventRecord(h->ev_theta_tail, tail_on));
    }
    if (ahead && !early) {
      // theta of iteration t+1 from the z just drawn, concurrent with the counts and the Phi draw
      Events &N = h->evs[(h->ev_head + 1) % kEvRing];
      hipStream_t ts = h->chain_on_side ? h->stream : h->side;
      HIP_TRY(h, hipStreamWaitEvent(h->chain_on_side ? h->side_hot : h->side, E.e[2], 0));
      if (h->hot_join_pending) HIP_TRY(h, hipStreamWaitEvent(ts, h->ev_hot_join, 0));   // the theta draw reads the hot chunks' z as well
      N.theta_on_main = h->chain_on_side;
      if (!h->chain_on_side) HIP_TRY(h, hipEventRecord(N.th0, ts));
      if ((rc = launch_theta(h, ts, h->d_theta_next, h->iteration + 1, d_tail, -1, h->chain_on_side ? h->theta_lds_main : 0, h->chain_on_side ? h->theta_b_main : 0))) return rc;
      if (d_tail) HIP_TRY(h, hipStreamWaitEvent(ts, h->ev_theta_tail, 0));
      HIP_TRY(h, hipEventRecord(N.th1, ts));
      h->theta_ahead_iter = (int64_t)h->iteration + 1;
    }
  }
  {
    StreamSwap on_chain(h, h->chain_on_side ? h->side_hot : h->stream);
    if (h->hot_join_pending) { HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_hot_join, 0)); h->hot_join_pending = false; }
    if (h->z_counted) {                              // the z kernels added the cold tokens' cells: the hot words' segments are left
      h->z_counted = false;
      if (h->hot_counted) { h->n_k_valid = false; h->counts_global = false; h->cnt_own_valid = false; }   // ... and were counted on the hot chunks' stream
      else if ((rc = launch_count_hot(h))) return rc;
    } else if ((rc = launch_count_rebuild(h))) return rc;   // this shard's counts; summed across shards by the caller
    if (!E.light) HIP_TRY(h, hipEventRecord(E.e[3], h->stream));
    if (h->chain_on_side && !h->whole_sweep) HIP_TRY(h, hipEventRecord(h->ev_chain_done, h->stream));
  }
  // ggs_sweep_begin on its own: what the caller does on the handle's stream before ggs_sweep_end (a getter) sees the counts
  if (h->chain_on_side && !h->whole_sweep) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_chain_done, 0));
  return GGS_OK;
}

// `settle` = wait for the device, raise what the sweeps flagged and add their phase times to the timings.  A batch
// (ggs_sweep with n_sweeps > 1) settles once, after its last sweep: the device flags are sticky, and the ~40 us host
// round trip per sweep is 2 % of a 1.9 ms sweep.
int finish_sweep_enqueue_on(ggs_handle *h, bool with_phi);
int finish_sweep_enqueue(ggs_handle *h, bool with_phi) {
  if (!h->chain_on_side) return finish_sweep_enqueue_on(h, with_phi);
  int rc;
  {
    StreamSwap on_chain(h, h->side_hot);
    if ((rc = finish_sweep_enqueue_on(h, with_phi))) return rc;
    HIP_TRY(h, hipEventRecord(h->ev_chain_done, h->stream));
  }
  HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_chain_done, 0));   // behind the theta draw: the next z step, a getter, a synchronise see both legs
  return GGS_OK;
}
int finish_sweep_enqueue_on(ggs_handle *h, bool with_phi) {
  int rc;
  Events &E = h->evs[h->ev_head];
  E.e4_is_e3 = h->whole_sweep;                         // one event packet less on the stream (~6 us each)
  if (!E.e4_is_e3 && !E.light) HIP_TRY(h, hipEventRecord(E.e[4], h->stream));
  bool acc = false;
  E.exchanged = false;
  if (with_phi) {
    acc = (h->flags & GGS_FLAG_SAVE_PHI_MEAN) && sample_phi_this_iteration(h);
    if (h->collapsed) {                    // no Phi in the count form: the merge (with an exchange: the gather of the count slices) and tokensPerTopic
      acc = false;
      if ((rc = launch_magnitude(h))) return rc;
    } else {
      E.exchanged = h->xg != nullptr;
      if ((rc = launch_phi(h, false, acc, E.light ? nullptr : &E))) return rc;
    }
  }
  HIP_TRY(h, hipEventRecord(E.e[5], h->stream));
  if (acc) h->n_sampled_phi++;             // GGS:168-170
  h->ev_pending += 1;
  return GGS_OK;
}
int finish_sweep_settle(ggs_handle *h) {
  int rc;
  if ((rc = check_status(h))) return rc;   // synchronises the stream
  if ((rc = settle_sweeps(h))) return rc;
  if (h->flags & GGS_FLAG_PARANOID) return ggs_check_invariants(h);
  return GGS_OK;
}
int finish_sweep(ggs_handle *h, bool with_phi, bool settle = true) {
  int rc = finish_sweep_enqueue(h, with_phi);
  if (rc) return rc;
  if (!settle && !(h->flags & GGS_FLAG_PARANOID)) return GGS_OK;
  return finish_sweep_settle(h);
}

// An Exchange that never got attached: its communicator goes with it when the library created it.
void exchange_free(Exchange *x) {
  if (!x) return;
  if (x->own_comm && x->comm && x->api) (void)x->api->CommDestroy(x->comm);
  delete x;
}
// what every attach call checks BEFORE it creates anything (ncclCommInitRank is a blocking collective)
int exchange_precheck(ggs_handle *h, int32_t rank, int32_t nranks) {
  if (!h) return GGS_ERR_BAD_ARG;
  if (h->xg) return set_err(h, GGS_ERR_STATE, "an exchange is already attached");
  if (h->have_corpus) return set_err(h, GGS_ERR_STATE, "attach the exchange before ggs_set_corpus");
  if (nranks < 1 || rank < 0 || rank >= nranks) return set_err(h, GGS_ERR_BAD_ARG, "rank outside [0, nranks)");
  return bind_device(h);
}

// Buffers and the column map of an attached exchange; the handle moves to a stream of its own.  Takes ownership of x.
int setup_exchange(ggs_handle *h, Exchange *x) {
  int rc = exchange_precheck(h, x->rank, x->nranks);
  if (rc) { exchange_free(x); return rc; }
  const std::vector<int32_t> sl = topic_slices(h->K, x->nranks);
  h->xg = x;
  h->k0 = sl[(size_t)x->rank]; h->Ks = sl[(size_t)x->rank + 1] - h->k0; h->Ksm = (h->K + x->nranks - 1) / x->nranks;
  {
    auto magic = [](uint32_t d) { return d > 1 ? (uint32_t)((0x100000000ull + d - 1) / d) : 0u; };   // udiv_magic (ggs_kernels.hpp)
    h->smap.size = h->K / x->nranks; h->smap.rem = h->K % x->nranks; h->smap.ksm = h->Ksm;
    h->smap.m_size = magic((uint32_t)std::max(h->smap.size, 1)); h->smap.m_size1 = magic((uint32_t)h->smap.size + 1);
    h->smap.rank_stride = (int64_t)h->V * h->Ksm;
    if (const char *e = debug_env("GGS_DEBUG_ZCOUNTS")) { h->z_counts = std::atoi(e) != 0; h->z_counts_forced = std::atoi(e) == 2; }
    if (const char *e = debug_env("GGS_DEBUG_TIMING_EVERY")) h->detail_every = std::max(1, std::atoi(e));
    if (const char *e = debug_env("GGS_DEBUG_COUNT_EXCHANGE")) h->count_mode = std::max(0, std::min(2, std::atoi(e)));
  }
  std::vector<int64_t> koff((size_t)h->K);
  for (int32_t r = 0; r < x->nranks; ++r)
    for (int32_t k = sl[(size_t)r]; k < sl[(size_t)r + 1]; ++k) koff[(size_t)k] = (int64_t)r * h->V * h->Ksm + (k - sl[(size_t)r]);
  std::vector<int32_t> krank((size_t)h->K), kcol((size_t)h->K);
  for (int32_t r = 0; r < x->nranks; ++r)
    for (int32_t k = sl[(size_t)r]; k < sl[(size_t)r + 1]; ++k) { krank[(size_t)k] = r; kcol[(size_t)k] = k - sl[(size_t)r]; }
  // the all-gather of the gammas in two halves of the vocabulary (whole 64-row segments: the draw's tiles); a short
  // vocabulary goes in one
  h->seg_split = h->sum_nseg >= 16 ? h->sum_nseg / 2 : 0;
  if (const char *e = debug_env("GGS_DEBUG_AGSPLIT")) h->seg_split = std::max(0, std::min(h->sum_nseg - 1, std::atoi(e)));
  h->v_split = h->seg_split * kSumSegRows;
  const size_t slice = (size_t)h->V * h->Ksm, all = slice * (size_t)x->nranks;
  auto fail = [&](int code) {                        // a half-built exchange must not stay attached
    void *bufs[] = {h->d_koff, h->d_krank, h->d_kcol, h->d_cnt_send, h->d_cnt_own, h->d_phi_own, h->d_phi_all0, h->d_phi_all1, h->d_mag_own, h->d_n_k_own};
    for (void *b : bufs) if (b) (void)hipFree(b);
    h->d_koff = nullptr; h->d_krank = h->d_kcol = nullptr; h->d_cnt_send = h->d_cnt_own = nullptr; h->d_phi_own = h->d_phi_all0 = h->d_phi_all1 = nullptr;
    h->d_mag_own = nullptr; h->d_n_k_own = nullptr;
    h->xg = nullptr; h->Ks = h->Ksm = h->K; h->k0 = 0;
    exchange_free(x);
    return code;
  };
  if ((rc = dev_alloc(h, &h->d_koff, (size_t)h->K)) || (rc = dev_alloc(h, &h->d_krank, (size_t)h->K)) || (rc = dev_alloc(h, &h->d_kcol, (size_t)h->K)) ||
      (rc = dev_alloc(h, &h->d_cnt_send, all)) || (rc = dev_alloc(h, &h->d_cnt_own, slice)) || (rc = dev_alloc(h, &h->d_phi_own, slice + (size_t)h->Ksm)) ||
      (rc = dev_alloc(h, &h->d_phi_all0, half0_elems(h) * (size_t)x->nranks)) || (rc = dev_alloc(h, &h->d_phi_all1, half1_elems(h) * (size_t)x->nranks)) ||
      (rc = dev_alloc(h, &h->d_mag_own, (size_t)h->Ksm)) || (rc = dev_alloc(h, &h->d_n_k_own, (size_t)h->Ksm)))
    return fail(rc);
  if (hipMemcpy(h->d_koff, koff.data(), sizeof(int64_t) * koff.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->d_krank, krank.data(), sizeof(int32_t) * krank.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->d_kcol, kcol.data(), sizeof(int32_t) * kcol.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(h->d_cnt_send, 0, all * sizeof(int32_t)) != hipSuccess || hipMemset(h->d_cnt_own, 0, slice * sizeof(int32_t)) != hipSuccess ||
      hipMemset(h->d_phi_own, 0, (slice + (size_t)h->Ksm) * sizeof(double)) != hipSuccess ||
      hipMemset(h->d_phi_all0, 0, std::max<size_t>(half0_elems(h) * (size_t)x->nranks, 1) * sizeof(double)) != hipSuccess ||
      hipMemset(h->d_phi_all1, 0, half1_elems(h) * (size_t)x->nranks * sizeof(double)) != hipSuccess)
    return fail(set_err(h, GGS_ERR_HIP, "exchange buffers: copy / memset failed"));
  {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, hi) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_half_drawn, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_half_gathered, hipEventDisableTiming) != hipSuccess)
      return fail(set_err(h, GGS_ERR_HIP, "exchange: communication stream / events"));
  }
  if (!h->stream) {
    // Collectives are ordered by THIS stream alone: not the legacy default stream, whose implicit synchronisation with
    // other libraries' blocking streams is exactly the convention not to rely on.  HIGH priority: measured with RCCL
    // and a torch process group in the process, a normal-priority stream shares its hardware queue with theirs and the
    // sweep's phases stretch (one rank: 2.59 ms per sweep against 1.93 on a high-priority stream or the null stream).
    static const bool stay = debug_env("GGS_DEBUG_OWN_STREAM") && std::atoi(debug_env("GGS_DEBUG_OWN_STREAM")) == 0;   // experiments
    if (!stay) {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      HIP_TRY(h, hipStreamCreateWithPriority(&h->own_stream, hipStreamNonBlocking, hi));
      h->stream = h->own_stream;
    }
  }
  HIP_TRY(h, hipDeviceSynchronize());
  h->counts_global = false; h->cnt_own_valid = false; h->n_k_valid = false;
  h->cnt_send_zeroed = true;                           // the memset above
  return GGS_OK;
}

Exchange *new_rccl_exchange(ggs_handle *h, int32_t rank, int32_t nranks, int *rc) {
  std::string err;
  RcclApi *api = RcclApi::get(err);
  if (!api) { *rc = set_err(h, GGS_ERR_UNSUPPORTED, err); return nullptr; }
  auto *x = new (std::nothrow) Exchange();
  if (!x) { *rc = GGS_ERR_HIP; return nullptr; }
  x->rank = rank; x->nranks = nranks; x->api = api;
  x->ops.struct_size = (int32_t)sizeof(ggs_exchange_ops); x->ops.ctx = x;
  x->ops.reduce_scatter_i32 = xops::rccl_reduce_scatter_i32;
  x->ops.all_gather_f64 = xops::rccl_all_gather_f64;
  x->ops.all_gather_i32 = xops::rccl_all_gather_i32;
  x->ops.all_to_all_v_i32 = (api->Send && api->Recv) ? xops::rccl_all_to_all_v_i32 : nullptr;
  *rc = GGS_OK;
  return x;
}

// a collective step for every handle of the group: inside ncclGroupStart/End for RCCL; an adopted group (the caller's
// transport) simply gets the calls one after the other, rank 0 first
template <typename Step>
int group_collective(ggs_handle **hs, int32_t n, Step step) {
  RcclApi *api = hs[0]->xg->api;
  int r = GGS_OK;
  if (api) api->GroupStart();
  for (int32_t i = 0; i < n && !r; ++i)
    if (!(r = bind_device(hs[i]))) r = step(hs[i]);
  if (api && api->GroupEnd() != ncclSuccess && !r) r = set_err(hs[0], GGS_ERR_HIP, "ncclGroupEnd failed");
  return r;
}

}  // namespace

extern "C" {

int ggs_abi_version(void) { return GGS_ABI_VERSION; }

const char *ggs_last_error(const ggs_handle *h) { return h ? h->err.c_str() : "null handle"; }

int ggs_create(const ggs_config *cfg, ggs_handle **out) {
  if (!cfg || !out) return GGS_ERR_BAD_ARG;
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(ggs_config)) return GGS_ERR_BAD_ARG;
  if (cfg->num_topics <= 0 || cfg->num_types <= 0 || !(cfg->beta > 0)) return GGS_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device_id < 0 || cfg->device_id >= ndev) return GGS_ERR_HIP;
  ggs_handle *h = new (std::nothrow) ggs_handle();
  if (!h) return GGS_ERR_HIP;
  h->K = cfg->num_topics; h->V = cfg->num_types; h->device = cfg->device_id;
  h->Ks = h->Ksm = h->K; h->k0 = 0;
  h->Kp = (h->K + 1) & ~1;
  h->pitch16 = (h->Kp / 2) | 1;             // odd number of 16-byte units per LDS row
  h->beta = cfg->beta; h->seed = cfg->seed; h->flags = cfg->flags;
  if (h->flags & GGS_FLAG_COLLAPSED) { h->collapsed = true; h->flags |= GGS_FLAG_PCGS; }   // the lane-per-document z loop, a different matrix
  h->phi_burn_in = cfg->phi_burn_in; h->phi_thin = cfg->phi_mean_thin > 0 ? cfg->phi_mean_thin : 1;
  if (const char *ab = debug_env("GGS_DEBUG_ABLATE")) h->ablate = std::atoi(ab);
  h->alpha.assign(h->K, cfg->alpha_scalar);
  if (cfg->alpha) std::copy(cfg->alpha, cfg->alpha + h->K, h->alpha.begin());
  for (double a : h->alpha)
    if (!(a > 0)) { delete h; return GGS_ERR_BAD_ARG; }

  int rc = GGS_OK;
  auto bail = [&](int code) { ggs_destroy(h); return code; };
  if (hipSetDevice(h->device) != hipSuccess) return bail(GGS_ERR_HIP);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) return bail(GGS_ERR_HIP);
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  // LDS budget of the z kernel: a tile of T token rows + one theta row per wave.  The waves
  // are persistent and stride the chunk table statically, so the grid must not exceed what
  // is truly co-resident: measured on MI355X, LDS is handed out in granules (6 x 26,912 B
  // do not fit a CU although the occupancy API says they do), so residency is computed
  // with the request rounded up to 2 KiB.  T is the largest tile that still lets 6
  // single-wave workgroups share a CU -- the passes are latency chains, so waves in flight
  // matter more than lanes in use -- but at least 8 rows.
  {
    constexpr int kGranule = 2048;
    auto alloc_of = [&](int bytes) { return (bytes + kGranule - 1) / kGranule * kGranule; };
    const int pitch = h->pitch16 * 16, thbytes = ((h->Kp * 8 + 15) / 16) * 16;
    // the sliced kernel tags chunk tokens (word ids) in bit 30
    // score-register kernels up to K = 160: measured on the benchmark corpus (sweep, ms; round 3) K=136: 2.13 sliced / 2.40
    // streaming, 152: 2.33 / 2.56, 160: 2.40 / 2.50, 164: 3.06 / 2.83, 168: 3.06 / 2.75, 184: 3.24 / 2.94 (round 2: 184: 3.55 /
    // 3.64, 192: 4.22 / 3.59) -- from KMAX = 168 on the cold kernel fills the whole register file (256 + 256) and the one-pass
    // streaming kernel with the next theta drawn beside it wins (the sliced kernels can take K up to kSlicedMaxTopics = 192:
    // GGS_DEBUG_ZKERNEL=1)
    const bool sliced_ok = h->K <= kSlicedMaxTopics && h->V < (1 << kSlotShift);
    h->z_sliced = sliced_ok && h->K <= kSlicedDefaultTopics;
    h->z_stream = !h->z_sliced && h->K > 2 * kSliceTopics;
    if (const char *e = debug_env("GGS_DEBUG_ZKERNEL")) {          // 0: whole-row tile kernel, 1: sliced where possible, 2: streaming kernel where it applies, 3: its two-pass form
      const int mode = std::atoi(e);
      h->z_sliced = sliced_ok && mode == 1;
      h->z_stream = (mode == 2 || mode == 3) ? h->K > 2 * kSliceTopics : (h->z_stream && mode != 0);
      h->z_two_pass = mode == 3;
      if (h->z_sliced) h->z_stream = false;
    }
    // only values above 1 are meaningful (they force the exact replay in tests); anything below would void the proof
    if (const char *e = debug_env("GGS_DEBUG_MARGIN")) h->margin_scale = std::max(1.0, std::atof(e));
    if (h->z_stream) {
      // 64-token chunks, a 2-slot slice ring + the theta row zero-padded to whole slices (one-pass kernel: to whole
      // checkpoint groups, plus a checkpoint per group and lane); no score registers, so 8 waves per CU fit the
      // register file and LDS bounds the residency
      h->z_tile_tokens = 64;
      const int ns = (h->K + kSliceTopics - 1) / kSliceTopics;
      // checkpoint group: the smallest that keeps the checkpoints in registers (K <= 256: one slice, <= 512: two, <= 1024: four); beyond, four slices and LDS
      h->z_group = ns <= kRegCheckpoints ? 1 : ns <= 2 * kRegCheckpoints ? 2 : 4;
      if (const char *e = debug_env("GGS_DEBUG_GROUP")) { const int gq = std::atoi(e); if (gq == 1 || gq == 2 || gq == 4) h->z_group = gq; }
      const int ng = (ns + h->z_group - 1) / h->z_group;
      h->z_regck = !h->z_two_pass && ng <= kRegCheckpoints;          // checkpoints in registers or in LDS
      if (const char *e = debug_env("GGS_DEBUG_REGCK")) h->z_regck = h->z_regck && std::atoi(e) != 0;
      if (!h->z_regck) h->z_group = 4;                               // the LDS-checkpoint kernel is instantiated for groups of four
      const int ngl = (ns + h->z_group - 1) / h->z_group;
      // Chunks of 64 consecutive tokens across ONE document boundary (two theta rows per wave) instead of near-equal
      // cuts of single documents: 98 % of the lanes busy instead of 78 % at 200-token documents.  Where the second row
      // would cost resident waves (K > 512: 8 KiB at K = 1024) the single-document chunks stay.
      h->z_two_rows = !h->z_two_pass && h->K <= 512;
      if (const char *e = debug_env("GGS_DEBUG_TWOROWS")) h->z_two_rows = !h->z_two_pass && std::atoi(e) != 0;
      h->z_lds = h->z_two_pass ? kStreamRingSlots * kSliceBytes + ns * kSliceTopics * 8
                               : kStream1RingSlots * kSliceBytes + (h->z_two_rows ? 2 : 1) * ngl * h->z_group * kSliceTopics * 8 + (h->z_regck ? 0 : ngl * 64 * 8);
      if (h->z_lds > kMaxLdsBytes) return bail(GGS_ERR_UNSUPPORTED);   // K > ~16000: the theta row itself would need slicing
      // the waves are persistent, so the grid must be what is truly co-resident -- and a CU's 160 KiB cannot be filled
      // to the last granule: measured, 5 x 32 KiB and 4 x 40 KiB leave one workgroup waiting for a second round
      // (z at K = 1024: 18.6 ms with 5 waves of 32 KiB requested, 12.7 ms with 4)
      h->z_waves_per_cu = std::max(1, std::min(8, (kMaxLdsBytes - kGranule) / alloc_of(h->z_lds)));
      if (const char *e = debug_env("GGS_DEBUG_WPC")) h->z_waves_per_cu = std::max(1, std::atoi(e));
    } else if (h->z_sliced) {
      // 64-token chunks, per wave the two theta rows and the slice ring, per workgroup (4 waves, one per SIMD) the hot-word table
      // (the score registers take most of the 512-entry file)
      h->z_tile_tokens = 64;
      const int kmax = ((h->K + 7) / 8) * 8, ns = (kmax + kSliceTopics - 1) / kSliceTopics;
      // per wave: the chunk's kChunkDocs theta rows (zero-padded to KMAX), then the ring; a DMA's immediate slice offset
      // (< ns*128) is subtracted from its LDS destination, so the ring must not start below that
      h->ring_base = (std::max(kChunkDocs * kmax * 8, ns * 128) + 255) / 256 * 256;
      h->wave_lds = h->ring_base + kRingSlots * kSliceBytes;
      h->hot_pitch = ((h->K + 7) / 8) * 64 + 16;                     // KMAX doubles + one unit: an odd number of 16-byte units
      if (const char *e = debug_env("GGS_DEBUG_SPLIT")) { h->z_split = std::atoi(e) != 0; h->z_split_forced = std::atoi(e) == 2; }   // 2: the split form without the timed comparison
      h->z_split_allowed = h->z_split;
      h->hot_wave_lds = (kChunkDocs * ns * kSliceTopics * 8 + 255) / 256 * 256;   // z_hot_kernel: two theta rows per wave (whole slices), then the table and its zeroed tail
      // split: the two workgroups must fit one CU together, each request rounded up to the LDS allocation granule
      h->hot_cap = h->z_split ? ((kMaxLdsBytes - alloc_of(kSlicedWaves * h->wave_lds) - kSlicedWaves * h->hot_wave_lds) / kGranule * kGranule - kHotTailBytes) / h->hot_pitch
                              : (kMaxLdsBytes - kSlicedWaves * h->wave_lds) / h->hot_pitch;
      h->hot_cap = std::max(0, std::min(255, h->hot_cap));
      if (const char *e = debug_env("GGS_DEBUG_HOT")) h->hot_cap = std::max(0, std::min(h->hot_cap, std::atoi(e)));
      h->z_lds = kSlicedWaves * h->wave_lds + h->hot_cap * h->hot_pitch;
      h->z_waves_per_cu = kSlicedWaves;
      // the warm tiers: the same LDS beside the cold kernel's workgroup, more of it for theta rows (warm_docs per wave), the rest a table
      h->warm_docs = warm_docs_for(kmax);
      h->warm_wave_lds = (h->warm_docs * (ns * kSliceTopics * 8 + kWarmThetaPad) + 255) / 256 * 256;
      h->warm_cap = ((kMaxLdsBytes - alloc_of(kSlicedWaves * h->wave_lds) - kSlicedWaves * h->warm_wave_lds) / kGranule * kGranule - kHotTailBytes) / h->hot_pitch;
      h->warm_cap = std::max(0, std::min(255, h->warm_cap));
      if (h->warm_cap < 16) h->warm_cap = 0;                           // the table loads and barriers of a tier want tokens to pay them
      if (const char *e = debug_env("GGS_DEBUG_WARM")) h->warm_tiers_max = std::max(0, std::min(kWarmMaxTiers, std::atoi(e)));
      if (const char *e = debug_env("GGS_DEBUG_WARM_ROWS")) h->warm_cap = std::max(0, std::min(h->warm_cap, std::atoi(e)));
      if (const char *e = debug_env("GGS_DEBUG_WARM_FILL")) h->warm_min_fill_pct = std::max(1, std::min(100, std::atoi(e)));
      if (const char *e = debug_env("GGS_DEBUG_WARM_CPW")) h->warm_min_chunks_per_wave = std::max(0, std::atoi(e));
    } else {
    int T = (kMaxLdsBytes / 6 / kGranule * kGranule - thbytes) / pitch;
    if (const char *e = debug_env("GGS_DEBUG_TILE")) T = std::atoi(e);
    T = std::max(8, std::min(64, T));
    h->z_tile_tokens = T;
    h->z_lds = T * pitch + thbytes;
    if (h->z_lds > kMaxLdsBytes) return bail(GGS_ERR_UNSUPPORTED);   // K > ~2400 needs a K-sliced kernel (not in this round)
    h->z_waves_per_cu = std::max(1, std::min(8, (kMaxLdsBytes - kGranule) / alloc_of(h->z_lds)));
    if (const char *e = debug_env("GGS_DEBUG_WPC")) h->z_waves_per_cu = std::max(1, std::atoi(e));
    }
  }
  {
    int B = 64;
    auto lds_of = [&](int b) { const int bp = b | 1; return (int)((size_t)h->K * bp * 8 + (size_t)b * 20 + kThetaQueueBytes); };
    while (B > 1 && lds_of(B) > 32 * 1024) B >>= 1;
    if (lds_of(B) > kMaxLdsBytes) return bail(GGS_ERR_UNSUPPORTED);
    // The request is padded to a quarter of the CU's LDS: at most 4 workgroups (16 waves) of the theta draw per CU.
    // It runs on the side stream beside the Phi phase, which is the critical path; stream priority orders dispatch,
    // not running waves, and measured with 5-6 resident workgroups theta finishes early (0.53 ms instead of 0.75)
    // while the Phi phase it starves gets longer (0.67 -> 0.72 ms).
    // ... and 24 KiB of every CU stay free for the main stream's own LDS users (the walk of the exact column sums,
    // 14.5 KiB per workgroup: with the CU's LDS handed out to theta workgroups to the last granule it waited for the
    // theta draw to END -- 5 ms at K = 1024).
    h->theta_docs_per_block = B; h->theta_lds = std::max(lds_of(B), (kMaxLdsBytes - 24 * 1024) / 4);
    // ... unless the theta draw is the critical leg itself (theta_main, below): then five workgroups per CU, of 16 documents
    // each.  Measured at K = 100 on one box, ms per sweep (documents per workgroup x workgroups per CU): 32x4 1.60-1.61,
    // 32x5 1.598-1.62, 24x5 1.593, 16x4 1.614, 16x5 1.568-1.587, 16x6 1.58, 16x8 1.598, 12x5 1.609, 8x8 1.635 -- the smaller
    // workgroups leave the Phi chain beside them its pace (0.39 ms against 0.44), so that it ends before the theta draw does.
    h->theta_b_main = std::min(B, 16);
    if (const char *e = debug_env("GGS_DEBUG_THETA_B")) h->theta_b_main = std::max(1, std::min(B, std::atoi(e)));
    h->theta_lds_main = std::max(lds_of(h->theta_b_main), (kMaxLdsBytes - 24 * 1024) / (debug_env("GGS_DEBUG_THETA_WGS") ? std::max(1, std::atoi(debug_env("GGS_DEBUG_THETA_WGS"))) : 5));
    // K > 192 (one-pass streaming z kernel): theta workgroups small enough to sit BESIDE the z waves -- three of
    // them, on the LDS the z waves give up -- so that the next theta of a part of the documents is drawn while the
    // following parts are sampled (z_phase).  The padded request caps them at three per CU while z runs.
    h->cfg_plain.parts = 1; h->cfg_plain.z_waves_per_cu = h->z_waves_per_cu; h->cfg_plain.theta_docs_per_block = h->theta_docs_per_block; h->cfg_plain.theta_lds = h->theta_lds;
    if (const char *e = debug_env("GGS_DEBUG_ZPARTS")) { h->z_parts = std::max(1, std::min(8, std::atoi(e))); h->z_parts_forced = true; }
    else h->z_parts = (h->z_stream && !h->z_two_pass && !(h->flags & GGS_FLAG_PCGS)) ? 8 : 1;   // measured at K = 1024: 1 part 18.4 ms per sweep, 2: 18.1, 4: 16.5, 8: 15.9
    if (h->z_parts > 1 && h->z_stream && !h->z_two_pass) {
      constexpr int kGranule = 2048;
      const int kBeside = debug_env("GGS_DEBUG_BESIDE") ? std::max(1, std::atoi(debug_env("GGS_DEBUG_BESIDE"))) : 4;   // measured at K = 1024 (sweep): 2 -> 16.9 ms, 3 -> 16.1, 4 -> 15.0, 5 -> 15.0
      auto alloc_of = [&](int bytes) { return (bytes + kGranule - 1) / kGranule * kGranule; };
      int Bt = 64;
      while (Bt > 1 && lds_of(Bt) > 10 * 1024) Bt >>= 1;
      const int z_alloc = alloc_of(h->z_lds), zw = std::min(h->z_waves_per_cu, (kMaxLdsBytes - kGranule - kBeside * alloc_of(lds_of(Bt))) / z_alloc);
      if (lds_of(Bt) <= 12 * 1024 && zw >= 2) {
        h->z_waves_per_cu = zw;
        h->theta_docs_per_block = Bt;
        h->theta_lds_beside_z = std::max(lds_of(Bt), (kMaxLdsBytes - kGranule - zw * z_alloc) / kBeside / kGranule * kGranule);
        h->theta_lds = std::max(lds_of(Bt), (kMaxLdsBytes - 24 * 1024) / 4);
      } else {
        h->z_parts = 1;
      }
    }
    h->cfg_parts.parts = h->z_parts; h->cfg_parts.z_waves_per_cu = h->z_waves_per_cu; h->cfg_parts.theta_docs_per_block = h->theta_docs_per_block;
    h->cfg_parts.theta_lds = h->theta_lds; h->cfg_parts.theta_lds_beside_z = h->theta_lds_beside_z;
  }
  if (h->z_sliced && (hipFuncSetAttribute(sliced_kernel_for(h->K), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess ||
                      hipFuncSetAttribute(hot_kernel_for(h->K), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess ||
                      hipFuncSetAttribute(warm_kernel_for(h->K), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess))
    return bail(GGS_ERR_HIP);
  const void *zk[] = {reinterpret_cast<const void *>(z_kernel<1>), reinterpret_cast<const void *>(z_kernel<2>),
                      reinterpret_cast<const void *>(z_kernel<4>), reinterpret_cast<const void *>(z_kernel<8>),
                      reinterpret_cast<const void *>(z_kernel<16>), reinterpret_cast<const void *>(z_kernel<20>)};
  // The attribute is process-global per kernel, not per handle: always the hardware maximum, so that a later handle
  // with a smaller K never lowers the cap under a live one.
  if (h->z_stream && (hipFuncSetAttribute(reinterpret_cast<const void *>(z_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess ||
                      hipFuncSetAttribute(reinterpret_cast<const void *>(z_stream1_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess ||
                      hipFuncSetAttribute(reinterpret_cast<const void *>(z_stream1_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess ||
                      hipFuncSetAttribute(reinterpret_cast<const void *>(z_stream1_kernel<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess ||
                      hipFuncSetAttribute(reinterpret_cast<const void *>(z_stream1_kernel<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess))
    return bail(GGS_ERR_HIP);
  for (const void *f : zk)
    if (!h->z_sliced && !h->z_stream && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess) return bail(GGS_ERR_HIP);
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(theta_kernel<kThetaBlock>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess)
    return bail(GGS_ERR_HIP);
  const size_t kv = (size_t)h->K * h->V;
  if ((rc = dev_alloc(h, &h->d_alpha, h->K)) || (rc = dev_alloc(h, &h->d_phiT, (size_t)h->V * h->Kp + kPhiTailPadBytes / 8)) ||
      (rc = dev_alloc(h, &h->d_mag, h->K)) || (rc = dev_alloc(h, &h->d_tot, h->K)) || (rc = dev_alloc(h, &h->d_n_wk, kv)) ||
      (rc = dev_alloc(h, &h->d_n_k, h->K)) || (rc = dev_alloc(h, &h->d_status, 4)))
    return bail(rc);
  if ((h->flags & GGS_FLAG_SAVE_PHI_MEAN) && (rc = dev_alloc(h, &h->d_phi_mean, kv))) return bail(rc);
  if (const char *e = debug_env("GGS_DEBUG_CHAIN")) h->exact_sum = std::atoi(e) == 0;
  if (const char *e = debug_env("GGS_DEBUG_GUIDED")) h->sum_guided = std::atoi(e) != 0;
  h->sum_nseg = (h->V + kSumSegRows - 1) / kSumSegRows;
  if (h->exact_sum && ((rc = dev_alloc(h, &h->d_sum_pref, ((size_t)h->sum_nseg + 1) * h->K)) ||
                       (rc = dev_alloc(h, &h->d_sum_fn, (size_t)h->sum_nseg * h->K * 4))))
    return bail(rc);
  if (hipMemcpy(h->d_alpha, h->alpha.data(), sizeof(double) * h->K, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(h->d_phiT, 0, sizeof(double) * ((size_t)h->V * h->Kp + kPhiTailPadBytes / 8)) != hipSuccess ||
      hipMemset(h->d_n_wk, 0, sizeof(int32_t) * kv) != hipSuccess ||
      hipMemset(h->d_n_k, 0, sizeof(int32_t) * h->K) != hipSuccess || hipMemset(h->d_status, 0, 16) != hipSuccess ||
      (h->d_phi_mean && hipMemset(h->d_phi_mean, 0, sizeof(double) * kv) != hipSuccess))
    return bail(GGS_ERR_HIP);
  if (h->flags & GGS_FLAG_PCGS) {
    // the wave-per-document kernel: any K up to 4096, any document length
    if (h->K <= kPcgsWaveMaxTopics) {
      int nb = 1;
      while (nb * 128 < h->Kp) nb *= 2;
      h->pcgs_wave_nb = nb;
      h->pcgs_wave_lds = nb * 128 * 12;                        // counts int32 + alpha fp64
      // waves per CU (the grid is persistent: exactly what is resident): what the kernel's registers allow (asked of the
      // runtime) and what LDS allows (computed here: the runtime's answer ignores the 2 KiB allocation granule)
      int by_regs = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&by_regs, pcgs_wave_kernel_for(nb, h->collapsed), 64, (size_t)h->pcgs_wave_lds) != hipSuccess || by_regs < 1) by_regs = 1;
      h->pcgs_wave_waves_per_cu = std::max(1, std::min(std::min(by_regs, 32), (kMaxLdsBytes - 2048) / ((h->pcgs_wave_lds + 2047) / 2048 * 2048)));
    }
    h->pcgs_wave_forced = h->K > (h->collapsed ? kCollapsedWaveFromTopics : kPcgsWaveFromTopics);
    if (const char *e = debug_env("GGS_DEBUG_PCGS_WAVE")) h->pcgs_wave_forced = std::atoi(e) != 0;
    if (h->pcgs_wave_forced && !h->pcgs_wave_nb) return bail(GGS_ERR_UNSUPPORTED);   // more than 4096 topics
    if (h->pcgs_wave_nb && hipFuncSetAttribute(pcgs_wave_kernel_for(h->pcgs_wave_nb, h->collapsed), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess)
      return bail(GGS_ERR_HIP);
    // pcgs_z_kernel: slice ring + alpha row + int16 [KT][64] document counts per single-wave workgroup
    const int ns = std::max(kPcgsRingSlots - 1, (h->K + kSliceTopics - 1) / kSliceTopics), kt = ns * kSliceTopics;
    h->pcgs_sliced = h->K <= kSlicedMaxTopics;
    if (const char *e = debug_env("GGS_DEBUG_PCGS_STREAM")) h->pcgs_sliced = h->pcgs_sliced && std::atoi(e) == 0;
    if (h->pcgs_sliced) {
      const int kmax = ((h->K + 7) / 8) * 8;                       // alpha row + counts below the ring (pcgs_sliced_kernel's kHead)
      h->pcgs_lds = (kmax * 8 + kmax * 128 + 255) / 256 * 256 + kPcgsRingSlots * kSliceBytes;
    } else {
      h->pcgs_lds = kPcgsRingSlots * kSliceBytes + kt * 8 + kt * 128;
    }
    if (h->pcgs_lds > kMaxLdsBytes && !h->pcgs_wave_forced) return bail(GGS_ERR_UNSUPPORTED);
    h->pcgs_waves_per_cu = std::max(1, std::min(8, (kMaxLdsBytes - 2048) / ((h->pcgs_lds + 2047) / 2048 * 2048)));   // never a CU filled to the last granule (see z_waves_per_cu)
    if (!h->pcgs_wave_forced &&
        hipFuncSetAttribute(h->pcgs_sliced ? (h->collapsed ? collapsed_kernel_for(h->K) : pcgs_kernel_for(h->K))
                                           : h->collapsed ? reinterpret_cast<const void *>(pcgs_z_kernel<true>) : reinterpret_cast<const void *>(pcgs_z_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess)
      return bail(GGS_ERR_HIP);
    if (h->collapsed && ((rc = dev_alloc(h, &h->d_lcg, 2)) ||
                         hipFuncSetAttribute(reinterpret_cast<const void *>(collapsed_serial_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes) != hipSuccess))
      return bail(rc ? rc : GGS_ERR_HIP);
  }
  for (auto &e : h->ev_part)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail(GGS_ERR_HIP);
  for (auto &E : h->evs) {
    for (auto &e : E.e)
      if (hipEventCreate(&e) != hipSuccess) return bail(GGS_ERR_HIP);
    for (auto &e : E.x)
      if (hipEventCreate(&e) != hipSuccess) return bail(GGS_ERR_HIP);
    if (hipEventCreate(&E.th0) != hipSuccess || hipEventCreate(&E.th1) != hipSuccess) return bail(GGS_ERR_HIP);
  }
  if (const char *e = debug_env("GGS_DEBUG_NO_OVERLAP")) h->overlap_theta = std::atoi(e) == 0;
  if (const char *e = debug_env("GGS_DEBUG_THETA_MAIN")) { h->theta_main = std::atoi(e) != 0; h->theta_main_always = std::atoi(e) == 2; }
  if (const char *e = debug_env("GGS_DEBUG_GAMMA_QUEUE")) h->gamma_queue_cap = std::max(0, std::atoi(e));
  {
    // lowest priority: the theta draw fills whatever the Phi phase (on the caller's stream) leaves idle
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    // (a CU-masked side stream was tried: hipExtStreamCreateWithCUMask gives a blocking stream that
    // serialises with the legacy default stream, so the overlap is lost -- priority alone it is)
    if (hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo) != hipSuccess) return bail(GGS_ERR_HIP);
    if (hipStreamCreateWithPriority(&h->side_hot, hipStreamNonBlocking, hi) != hipSuccess || hipEventCreate(&h->ev_hot_fork) != hipSuccess ||
        hipEventCreate(&h->ev_hot_join) != hipSuccess || hipEventCreateWithFlags(&h->ev_chain_done, hipEventDisableTiming) != hipSuccess)
      return bail(GGS_ERR_HIP);
  }
  if (hipDeviceSynchronize() != hipSuccess) return bail(GGS_ERR_HIP);   // the memsets above ran on the null stream
  *out = h;
  return GGS_OK;
}

void ggs_destroy(ggs_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  void *bufs[] = {h->d_doc_ptr, h->d_chunk_start, h->d_tok, h->d_z, h->d_chunk_doc, h->d_chunk_len, h->d_alpha, h->d_theta, h->d_theta_next,
                  h->d_phiT, h->d_mag, h->d_tot, h->d_phi_mean, h->d_n_wk, h->d_n_k, h->d_perm, h->d_inv_perm, h->d_zw, h->d_seg_word, h->d_seg_begin,
                  h->d_status, h->d_scratch, h->d_sum_pref, h->d_sum_fn, h->d_ct_tok, h->d_ct_idx, h->d_ct_ip, h->d_c_docs, h->d_hot_words, h->d_order,
                  h->d_test_ptr, h->d_test_tok, h->d_test_ll, h->d_test_docs, h->d_koff, h->d_cnt_send, h->d_cnt_own, h->d_cnt_all, h->d_n_k_own,
                  h->d_heldout_spill, h->d_phi_own, h->d_phi_all0, h->d_phi_all1, h->d_mag_own, h->d_krank, h->d_kcol, h->d_lcg, h->d_chunk_doc1,
                  h->d_hseg_word, h->d_hseg_begin, h->d_hseg_end, h->d_sp_count, h->d_sp_cnt32, h->d_sp_all, h->d_sp_send, h->d_sp_recv, h->d_sp_wg_count,
                  h->d_sp_wg_off, h->d_ht_pack, h->d_h_docs, h->d_wt_pack, h->d_w_docs, h->d_warm_words, h->d_warm_meta};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  exchange_free(h->xg);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  if (h->ev_theta_tail) (void)hipEventDestroy(h->ev_theta_tail);
  if (h->ev_half_drawn) (void)hipEventDestroy(h->ev_half_drawn);
  if (h->ev_half_gathered) (void)hipEventDestroy(h->ev_half_gathered);
  for (auto &e : h->ev_part)
    if (e) (void)hipEventDestroy(e);
  for (auto &E : h->evs) {
    for (auto &e : E.e)
      if (e) (void)hipEventDestroy(e);
    for (auto &e : E.x)
      if (e) (void)hipEventDestroy(e);
    if (E.th0) (void)hipEventDestroy(E.th0);
    if (E.th1) (void)hipEventDestroy(E.th1);
  }
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->side_hot) (void)hipStreamDestroy(h->side_hot);
  if (h->ev_hot_fork) (void)hipEventDestroy(h->ev_hot_fork);
  if (h->ev_hot_join) (void)hipEventDestroy(h->ev_hot_join);
  if (h->ev_chain_done) (void)hipEventDestroy(h->ev_chain_done);
  delete h;
}

int ggs_set_stream(ggs_handle *h, void *hip_stream) {
  if (!h) return GGS_ERR_BAD_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if ((rc = drop_theta_ahead(h))) return rc;
  if (h->xg && !hip_stream) return set_err(h, GGS_ERR_BAD_ARG, "with an exchange attached the handle does not run on the legacy default stream");
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return GGS_OK;
}

int ggs_set_corpus(ggs_handle *h, int64_t D, const int64_t *doc_ptr, const int32_t *tokens, int64_t doc_base, int64_t tok_base) {
  if (!h || D < 0 || !doc_ptr || doc_base < 0 || tok_base < 0) return set_err(h, GGS_ERR_BAD_ARG, "bad corpus arguments");
  if (doc_ptr[0] != 0) return set_err(h, GGS_ERR_BAD_ARG, "doc_ptr[0] must be 0");
  for (int64_t d = 0; d < D; ++d)
    if (doc_ptr[d + 1] < doc_ptr[d]) return set_err(h, GGS_ERR_BAD_ARG, "doc_ptr must be non-decreasing");
  const int64_t N = doc_ptr[D];
  if (N > 0 && !tokens) return set_err(h, GGS_ERR_BAD_ARG, "tokens is null");
  if (N >= (int64_t)1 << 31 || D >= (int64_t)1 << 31) return set_err(h, GGS_ERR_UNSUPPORTED, "more than 2^31-1 tokens or documents per device");
  for (int64_t i = 0; i < N; ++i)
    if (tokens[i] < 0 || tokens[i] >= h->V) return set_err(h, GGS_ERR_BAD_ARG, "token id outside [0, num_types)");
  int rc = bind_device(h);
  if (rc) return rc;
  if ((rc = drop_theta_ahead(h))) return rc;

  // z-kernel work items: each document is cut into ceil(len/T) near-equal chunks of <= T tokens.
  std::vector<int64_t> cstart;
  std::vector<int32_t> cdoc, clen;
  cstart.reserve((size_t)(N / 48 + D)); cdoc.reserve(cstart.capacity()); clen.reserve(cstart.capacity());
  std::vector<int32_t> cdoc1;
  const bool two_rows = h->z_stream && h->z_two_rows;
  if (two_rows) {
    // 64 consecutive tokens per chunk, across at most one document boundary; clen = tokens | tokens of the first document << 8
    int64_t pos = 0;
    int64_t d = 0;
    while (pos < N) {
      while (doc_ptr[d + 1] <= pos) ++d;                            // the document of token `pos` (empty documents hold none)
      const int64_t take0 = std::min<int64_t>(64, doc_ptr[d + 1] - pos);
      int64_t len = take0, d1 = d;
      if (take0 < 64 && pos + take0 < N) {                          // room left: the next non-empty document joins
        d1 = d + 1;
        while (doc_ptr[d1 + 1] <= pos + take0) ++d1;
        len += std::min<int64_t>(64 - take0, doc_ptr[d1 + 1] - (pos + take0));
      }
      cstart.push_back(pos); cdoc.push_back((int32_t)d); cdoc1.push_back((int32_t)d1);
      clen.push_back((int32_t)(len | (take0 << 8)));
      pos += len;
    }
  } else
  for (int64_t d = 0; d < D; ++d) {
    const int64_t len = doc_ptr[d + 1] - doc_ptr[d];
    if (len == 0) continue;
    const int64_t T = h->z_tile_tokens, n = (len + T - 1) / T, base = len / n, rem = len % n;
    int64_t s = doc_ptr[d];
    for (int64_t j = 0; j < n; ++j) {
      const int64_t l = base + (j < rem ? 1 : 0);
      cstart.push_back(s); cdoc.push_back((int32_t)d); clen.push_back((int32_t)l);
      s += l;
    }
  }
  // count-kernel work items: tokens sorted by word (counting sort, stable), each word's run
  // cut into segments of at most kSegTokens entries.
  constexpr int64_t kSegTokens = 4096;
  std::vector<int32_t> perm((size_t)N), seg_word, seg_begin, hot_words, warm_cand, hseg_word, hseg_begin, hseg_end;
  {
    std::vector<int64_t> wptr((size_t)h->V + 1, 0);
    for (int64_t i = 0; i < N; ++i) wptr[(size_t)tokens[i] + 1]++;
    for (int32_t w = 0; w < h->V; ++w) wptr[(size_t)w + 1] += wptr[(size_t)w];
    for (int32_t w = 0; w < h->V; ++w)
      for (int64_t b = wptr[(size_t)w]; b < wptr[(size_t)w + 1]; b += kSegTokens) { seg_word.push_back(w); seg_begin.push_back((int32_t)b); }
    seg_begin.push_back((int32_t)N);
    std::vector<int64_t> cur(wptr.begin(), wptr.end() - 1);
    for (int64_t i = 0; i < N; ++i) perm[(size_t)cur[(size_t)tokens[i]]++] = (int32_t)i;
    // the hot-word table of the sliced z kernel: the hot_cap most frequent words of THIS handle's tokens
    if (h->z_sliced && h->hot_cap > 0) {
      std::vector<int32_t> order((size_t)h->V);
      for (int32_t w = 0; w < h->V; ++w) order[(size_t)w] = w;
      const size_t nh = (size_t)std::min<int32_t>(h->hot_cap, h->V);
      // ... and, behind them, the candidates of the warm tiers (z_warm_kernel): the next warm_tiers_max x warm_cap words
      const size_t nw = std::min<size_t>((size_t)h->V, nh + (size_t)h->warm_tiers_max * (size_t)h->warm_cap);
      auto freq = [&](int32_t w) { return wptr[(size_t)w + 1] - wptr[(size_t)w]; };
      std::partial_sort(order.begin(), order.begin() + nw, order.end(), [&](int32_t a, int32_t b) { return freq(a) != freq(b) ? freq(a) > freq(b) : a < b; });
      for (size_t r = 0; r < nh && freq(order[r]) > 0; ++r) hot_words.push_back(order[r]);
      for (size_t r = nh; r < nw && freq(order[r]) > 0; ++r) warm_cand.push_back(order[r]);
      for (int32_t w : hot_words)
        for (int64_t b = wptr[(size_t)w]; b < wptr[(size_t)w + 1]; b += kSegTokens) {
          hseg_word.push_back(w); hseg_begin.push_back((int32_t)b); hseg_end.push_back((int32_t)std::min(b + kSegTokens, wptr[(size_t)w + 1]));
        }
    }
    // a segment ends where the next begins, or at the end of its word's run
    // (seg_begin[s+1] is the next segment's start, which is exactly that)
  }
  if (h->flags & GGS_FLAG_PCGS) {
    // documents longest first: the 64 of a pcgs wave are then equally long
    std::vector<int32_t> order((size_t)D);
    int64_t longest = 0;
    for (int64_t d = 0; d < D; ++d) { order[(size_t)d] = (int32_t)d; longest = std::max(longest, doc_ptr[d + 1] - doc_ptr[d]); }
    // the lane-per-document kernels keep the counts as int16: a longer document sends the corpus to the wave-per-document kernel
    h->pcgs_wave = h->pcgs_wave_forced || longest > kPcgsMaxDocLen;
    if (h->pcgs_wave && !h->pcgs_wave_nb) return set_err(h, GGS_ERR_UNSUPPORTED, "scheme=pcgs: a document of 32768 tokens or more with more than 4096 topics");
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return doc_ptr[a + 1] - doc_ptr[a] > doc_ptr[b + 1] - doc_ptr[b]; });
    // A wave takes the groups w, w + W, ... of this list (W = the resident waves).  With between one and two rounds of
    // groups (the benchmark corpus: 1 563 groups for 1 024 waves) the plain order would give the waves of the 539
    // LONGEST groups a second one: 420 steps against 205 for the rest.  Instead the W - m longest groups run alone and
    // the 2m shortest are paired long-with-short on the last m waves (-1 = no document): 360 steps at most.
    const int64_t n_groups = (D + 63) / 64, W = (int64_t)h->num_cus * h->pcgs_waves_per_cu;
    h->pcgs_order_len = D;
    if (n_groups > W && n_groups <= 2 * W) {
      const int64_t m = n_groups - W;
      std::vector<int32_t> padded((size_t)(2 * W * 64), -1);
      auto put = [&](int64_t position, int64_t group) {
        for (int64_t j = 0; j < 64 && group * 64 + j < D; ++j) padded[(size_t)(position * 64 + j)] = order[(size_t)(group * 64 + j)];
      };
      for (int64_t g = 0; g < W; ++g) put(g, g);
      for (int64_t j = 0; j < m; ++j) put(W + (W - m + j), n_groups - 1 - j);
      order.swap(padded);
      h->pcgs_order_len = (int64_t)order.size();
    }
    if ((rc = dev_alloc(h, &h->d_order, order.size()))) return rc;
    if (!order.empty()) HIP_TRY(h, hipMemcpy(h->d_order, order.data(), sizeof(int32_t) * order.size(), hipMemcpyHostToDevice));
  }
  h->D = D; h->N = N; h->C = (int64_t)cstart.size(); h->S = (int64_t)seg_word.size(); h->doc_base = doc_base; h->tok_base = tok_base;
  {
    // the parts of the z step (z_phase): consecutive documents with about equal token counts, and their chunk ranges
    // Parts pay where the theta draw (D x K gammas) is longer than the Phi chain it otherwise hides beside (V x K gammas):
    // the excess is what is drawn beside the z parts.  Measured: D = 100 000, V = 50 000, K = 1 024: 1 part 18.4 ms per
    // sweep, 2: 18.1, 4: 16.5, 8: 15.9; D = 18 846, V = 60 000, K = 200: 8 parts 1.112, 4: 1.071, 2: 1.019, 1: 1.006 (every part
    // costs an event packet, a launch and the drain of the persistent z waves).
    {
      int32_t want = h->cfg_parts.parts;
      if (!h->z_parts_forced && want > 1) want = D <= (int64_t)h->V ? 1 : D < 2 * (int64_t)h->V ? std::min(want, 4) : want;
      const ggs_handle::ZCfg &c = want > 1 ? h->cfg_parts : h->cfg_plain;
      h->z_parts = want; h->z_waves_per_cu = c.z_waves_per_cu; h->theta_docs_per_block = c.theta_docs_per_block; h->theta_lds = c.theta_lds;
      h->theta_lds_beside_z = c.theta_lds_beside_z;
    }
    const int32_t P = (h->z_parts > 1 && D >= 64 * h->z_parts) ? h->z_parts : 1;
    h->part_doc.assign((size_t)P + 1, D); h->part_chunk.assign((size_t)P + 1, (int64_t)cstart.size());
    h->part_doc[0] = 0; h->part_chunk[0] = 0;
    int64_t d = 0;
    size_t c = 0;
    for (int32_t p = 1; p < P; ++p) {
      const int64_t want = N * p / P;
      while (d < D && doc_ptr[d] < want) ++d;
      while (c < cdoc.size() && cdoc[c] < d) ++c;
      h->part_doc[(size_t)p] = d; h->part_chunk[(size_t)p] = (int64_t)c;
    }
    if (P != h->z_parts) { h->part_doc.resize(2); h->part_chunk.resize(2); h->part_doc[1] = D; h->part_chunk[1] = (int64_t)cstart.size(); }
  }
  if ((rc = dev_alloc(h, &h->d_doc_ptr, (size_t)D + 1)) || (rc = dev_alloc(h, &h->d_tok, (size_t)N)) || (rc = dev_alloc(h, &h->d_z, (size_t)N)) ||
      (rc = dev_alloc(h, &h->d_theta, (size_t)D * h->K + 2)) || (rc = dev_alloc(h, &h->d_theta_next, (size_t)D * h->K + 2)) || (rc = dev_alloc(h, &h->d_chunk_start, (size_t)h->C)) ||
      (rc = dev_alloc(h, &h->d_chunk_doc, (size_t)h->C)) || (rc = dev_alloc(h, &h->d_chunk_len, (size_t)h->C)) ||
      (rc = dev_alloc(h, &h->d_perm, (size_t)N)) || (rc = dev_alloc(h, &h->d_inv_perm, (size_t)N)) || (rc = dev_alloc(h, &h->d_zw, (size_t)N)) || (rc = dev_alloc(h, &h->d_seg_word, (size_t)h->S)) ||
      (rc = dev_alloc(h, &h->d_seg_begin, (size_t)h->S + 1)))
    return rc;
  std::vector<int32_t> inv((size_t)N);
  if (N) {
    HIP_TRY(h, hipMemcpy(h->d_perm, perm.data(), sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice));
    for (int64_t i = 0; i < N; ++i) inv[(size_t)perm[(size_t)i]] = (int32_t)i;
    HIP_TRY(h, hipMemcpy(h->d_inv_perm, inv.data(), sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice));
  }
  HIP_TRY(h, hipMemset(h->d_zw, 0, sizeof(int32_t) * std::max<size_t>((size_t)N, 1)));
  if (h->S) HIP_TRY(h, hipMemcpy(h->d_seg_word, seg_word.data(), sizeof(int32_t) * seg_word.size(), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->d_seg_begin, seg_begin.data(), sizeof(int32_t) * seg_begin.size(), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->d_doc_ptr, doc_ptr, sizeof(int64_t) * ((size_t)D + 1), hipMemcpyHostToDevice));
  if (N) HIP_TRY(h, hipMemcpy(h->d_tok, tokens, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice));
  h->num_hot = (int32_t)hot_words.size();
  h->HS = (int64_t)hseg_word.size();
  if ((rc = dev_alloc(h, &h->d_hseg_word, hseg_word.size())) || (rc = dev_alloc(h, &h->d_hseg_begin, hseg_begin.size())) || (rc = dev_alloc(h, &h->d_hseg_end, hseg_end.size()))) return rc;
  if (h->HS) {
    HIP_TRY(h, hipMemcpy(h->d_hseg_word, hseg_word.data(), sizeof(int32_t) * hseg_word.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_hseg_begin, hseg_begin.data(), sizeof(int32_t) * hseg_begin.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_hseg_end, hseg_end.data(), sizeof(int32_t) * hseg_end.size(), hipMemcpyHostToDevice));
  }
  h->Cs = h->Cc = 0;
  h->warm_tiers = 0; h->num_warm = 0; h->Cw = 0;
  if (h->z_sliced) {
    // Chunk lists of the sliced kernel: walk the documents in order and deal every token to the open
    // cold chunk or the open hot chunk; a chunk closes at 64 tokens or when a third document would enter it.
    // (documents are visited in order, so a token's document is always the chunk's newest: slot = documents so far - 1)
    struct Builder {
      int maxdocs = kChunkDocs, docslots = kChunkDocs, shift = kSlotShift;   // documents a chunk may draw from; ids stored per chunk
      std::vector<int32_t> tok, idx, docs;
      int fill = 64, ndocs = 0, last = -1;
      int64_t tokens = 0;
      void add(int32_t value, int32_t token, int32_t doc) {
        if (fill == 64 || (doc != last && ndocs == maxdocs)) {
          tok.resize(tok.size() + 64, 0); idx.resize(idx.size() + 64, -1);
          docs.insert(docs.end(), (size_t)docslots, doc);
          fill = 0; ndocs = 1; last = doc;
        } else if (doc != last) {
          docs[docs.size() - (size_t)docslots + (size_t)ndocs] = doc; ++ndocs; last = doc;
        }
        const size_t at = tok.size() - 64 + (size_t)fill;
        tok[at] = value | ((ndocs - 1) << shift); idx[at] = token;
        ++fill; ++tokens;
      }
      int64_t chunks() const { return (int64_t)(tok.size() / 64); }
    } cold, hot;
    std::vector<int32_t> row_of((size_t)h->V, -1);
    for (size_t r = 0; r < hot_words.size(); ++r) row_of[(size_t)hot_words[r]] = (int32_t)r;
    // The warm tiers: tier t = candidates [t*warm_cap, (t+1)*warm_cap).  A tier is kept while its chunks (64 lanes, up to
    // warm_docs documents) are reasonably full -- a token in a half-empty chunk costs what two cost -- and numerous enough
    // to pay for the tier's table load, its two barriers and the ragged end of its chunk list; tiers are kept in order: the
    // first one that falls short ends the list, its words and all later ones stay cold.  Measured with the table kernels'
    // hand-counted loads (profiles/r04_warm_tier_sweep.txt; before them a tier wanted 10 chunks per wave), sweep in ms with
    // 0 / 1 / 2 / 3 tiers: the benchmark corpus (20 M tokens; 14.4, 11.7, 10.7 chunks per wave) 1.517 / 1.494 / 1.461 /
    // 1.471 (4, 5, 6 tiers: 1.464 / 1.471 / 1.472, 8: 1.513); half of it (rank 0 of 2: 7 chunks per wave in the first tier)
    // 0.887 / 0.873 / 0.870 / 0.864; a quarter 0.552 / 0.547 / 0.543 / 0.545; an eighth (under 2 chunks per wave) 0.376 /
    // 0.378 / 0.380 / 0.394 -- there a tier's table load and barriers cost what its tokens save.
    std::vector<Builder> warm;
    int32_t tiers = 0;
    if (h->warm_cap > 0 && hot_words.size() == (size_t)h->hot_cap && !warm_cand.empty()) {
      const int32_t cand_tiers = (int32_t)((warm_cand.size() + (size_t)h->warm_cap - 1) / (size_t)h->warm_cap);
      warm.resize((size_t)cand_tiers);
      for (Builder &b : warm) { b.maxdocs = h->warm_docs; b.docslots = kWarmDocSlots; b.shift = kWarmSlotShift; }
      std::vector<int32_t> warm_of((size_t)h->V, -1);
      for (size_t r = 0; r < warm_cand.size(); ++r) warm_of[(size_t)warm_cand[r]] = (int32_t)r;
      for (int64_t d = 0; d < D; ++d)
        for (int64_t i = doc_ptr[d]; i < doc_ptr[d + 1]; ++i) {
          const int32_t r = warm_of[(size_t)tokens[i]];
          if (r >= 0) warm[(size_t)(r / h->warm_cap)].add(r % h->warm_cap, (int32_t)i, (int32_t)d);
        }
      const int64_t min_chunks = (int64_t)h->warm_min_chunks_per_wave * h->num_cus * kSlicedWaves;
      while (tiers < cand_tiers && warm[(size_t)tiers].tokens > 0 && warm[(size_t)tiers].chunks() >= min_chunks &&
             warm[(size_t)tiers].tokens * 100 >= warm[(size_t)tiers].chunks() * 64 * h->warm_min_fill_pct)
        ++tiers;
      warm.resize((size_t)tiers);
      warm_cand.resize(std::min(warm_cand.size(), (size_t)tiers * (size_t)h->warm_cap));
      for (int32_t w : warm_cand) row_of[(size_t)w] = -2;              // in a kept tier: neither cold nor hot
    }
    for (int64_t d = 0; d < D; ++d)
      for (int64_t i = doc_ptr[d]; i < doc_ptr[d + 1]; ++i) {
        const int32_t r = row_of[(size_t)tokens[i]];
        if (r >= 0) hot.add(r, (int32_t)i, (int32_t)d);
        else if (r == -1) cold.add(tokens[i], (int32_t)i, (int32_t)d);
      }
    // Lanes of a chunk in (document, row) order: the 16 lanes one LDS pass serves then mostly read the same theta row
    // and, in hot chunks, few distinct table rows (tokens of one word share a row: a broadcast, not a bank conflict).
    auto sort_lanes = [](Builder &b) {
      std::vector<std::pair<uint32_t, int32_t>> tmp(64);
      for (size_t c0 = 0; c0 < b.tok.size(); c0 += 64) {
        int n = 0;
        while (n < 64 && b.idx[c0 + (size_t)n] >= 0) ++n;                 // active lanes are a prefix
        for (int j = 0; j < n; ++j) tmp[(size_t)j] = {(uint32_t)b.tok[c0 + (size_t)j], b.idx[c0 + (size_t)j]};
        std::sort(tmp.begin(), tmp.begin() + n);
        for (int j = 0; j < n; ++j) { b.tok[c0 + (size_t)j] = (int32_t)tmp[(size_t)j].first; b.idx[c0 + (size_t)j] = tmp[(size_t)j].second; }
      }
    };
    sort_lanes(cold);
    sort_lanes(hot);
    for (Builder &b : warm) sort_lanes(b);
    h->Cc = (int64_t)(cold.docs.size() / 2);
    h->Cs = h->Cc + (int64_t)(hot.docs.size() / 2);
    {
      // z_hot_kernel reads its chunks in the packed form of the warm tiers: one 16-byte entry per lane, the chunk's
      // documents in kWarmDocSlots slots, the document slot at kWarmSlotShift
      const size_t nh64 = hot.tok.size(), nhc = nh64 / 64;
      std::vector<int32_t> hpack(4 * nh64, 0), hdocs(nhc * (size_t)kWarmDocSlots, 0);
      for (size_t j = 0; j < nh64; ++j) {
        const uint32_t t = (uint32_t)hot.tok[j];
        hpack[4 * j] = (int32_t)((t & ((1u << kSlotShift) - 1)) | ((t >> kSlotShift) << kWarmSlotShift));
        hpack[4 * j + 1] = hot.idx[j];
        if (hot.idx[j] >= 0) hpack[4 * j + 2] = inv[(size_t)hot.idx[j]];
      }
      for (size_t c = 0; c < nhc; ++c)
        for (int r = 0; r < kWarmDocSlots; ++r) hdocs[c * (size_t)kWarmDocSlots + (size_t)r] = hot.docs[2 * c + (size_t)std::min(r, 1)];
      if ((rc = dev_alloc(h, &h->d_ht_pack, hpack.size())) || (rc = dev_alloc(h, &h->d_h_docs, hdocs.size()))) return rc;
      if (nh64) {
        HIP_TRY(h, hipMemcpy(h->d_ht_pack, hpack.data(), sizeof(int32_t) * hpack.size(), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_h_docs, hdocs.data(), sizeof(int32_t) * hdocs.size(), hipMemcpyHostToDevice));
      }
    }
    cold.tok.insert(cold.tok.end(), hot.tok.begin(), hot.tok.end());
    cold.idx.insert(cold.idx.end(), hot.idx.begin(), hot.idx.end());
    cold.docs.insert(cold.docs.end(), hot.docs.begin(), hot.docs.end());
    std::vector<int32_t> ip(cold.idx.size(), 0);
    for (size_t j = 0; j < ip.size(); ++j)
      if (cold.idx[j] >= 0) ip[j] = inv[(size_t)cold.idx[j]];
    const size_t n64 = cold.tok.size();
    if ((rc = dev_alloc(h, &h->d_ct_tok, n64)) || (rc = dev_alloc(h, &h->d_ct_idx, n64)) || (rc = dev_alloc(h, &h->d_ct_ip, n64)) ||
        (rc = dev_alloc(h, &h->d_c_docs, cold.docs.size())) || (rc = dev_alloc(h, &h->d_hot_words, hot_words.size())))
      return rc;
    if (n64) {
      HIP_TRY(h, hipMemcpy(h->d_ct_tok, cold.tok.data(), sizeof(int32_t) * n64, hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(h->d_ct_idx, cold.idx.data(), sizeof(int32_t) * n64, hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(h->d_ct_ip, ip.data(), sizeof(int32_t) * n64, hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(h->d_c_docs, cold.docs.data(), sizeof(int32_t) * cold.docs.size(), hipMemcpyHostToDevice));
    }
    if (h->num_hot) HIP_TRY(h, hipMemcpy(h->d_hot_words, hot_words.data(), sizeof(int32_t) * hot_words.size(), hipMemcpyHostToDevice));
    // the warm tiers' lists, tier after tier; meta = [tiers + 1] first chunk of a tier, then [tiers] rows of its table
    h->warm_tiers = tiers; h->num_warm = (int32_t)warm_cand.size(); h->Cw = 0; h->warm_chunks_max = 0; h->warm_rows_max = 0;
    if (tiers > 0) {
      std::vector<int32_t> wtok, widx, wdocs, wwords((size_t)tiers * (size_t)h->warm_cap, 0);
      std::vector<int64_t> meta((size_t)(2 * tiers + 1), 0);
      for (int32_t t = 0; t < tiers; ++t) {
        const Builder &b = warm[(size_t)t];
        meta[(size_t)t] = (int64_t)(wtok.size() / 64);
        const int32_t rows = (int32_t)std::min<size_t>((size_t)h->warm_cap, warm_cand.size() - (size_t)t * (size_t)h->warm_cap);
        meta[(size_t)(tiers + 1 + t)] = rows;
        h->warm_rows_max = std::max(h->warm_rows_max, rows);
        h->warm_chunks_max = std::max(h->warm_chunks_max, b.chunks());
        wtok.insert(wtok.end(), b.tok.begin(), b.tok.end());
        widx.insert(widx.end(), b.idx.begin(), b.idx.end());
        wdocs.insert(wdocs.end(), b.docs.begin(), b.docs.end());
        for (int32_t r = 0; r < rows; ++r) wwords[(size_t)t * (size_t)h->warm_cap + (size_t)r] = warm_cand[(size_t)t * (size_t)h->warm_cap + (size_t)r];
      }
      meta[(size_t)tiers] = (int64_t)(wtok.size() / 64);
      h->Cw = meta[(size_t)tiers];
      std::vector<int32_t> wpack(4 * widx.size(), 0);
      for (size_t j = 0; j < widx.size(); ++j) {
        wpack[4 * j] = wtok[j]; wpack[4 * j + 1] = widx[j];
        if (widx[j] >= 0) wpack[4 * j + 2] = inv[(size_t)widx[j]];
      }
      if ((rc = dev_alloc(h, &h->d_wt_pack, wpack.size())) ||
          (rc = dev_alloc(h, &h->d_w_docs, wdocs.size())) || (rc = dev_alloc(h, &h->d_warm_words, wwords.size())) || (rc = dev_alloc(h, &h->d_warm_meta, meta.size())))
        return rc;
      HIP_TRY(h, hipMemcpy(h->d_wt_pack, wpack.data(), sizeof(int32_t) * wpack.size(), hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(h->d_w_docs, wdocs.data(), sizeof(int32_t) * wdocs.size(), hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(h->d_warm_words, wwords.data(), sizeof(int32_t) * wwords.size(), hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(h->d_warm_meta, meta.data(), sizeof(int64_t) * meta.size(), hipMemcpyHostToDevice));
    }
  }
  HIP_TRY(h, hipMemset(h->d_z, 0, sizeof(int32_t) * std::max<size_t>((size_t)N, 1)));
  HIP_TRY(h, hipMemset(h->d_theta, 0, sizeof(double) * std::max<size_t>((size_t)D * h->K, 1)));
  HIP_TRY(h, hipMemset(h->d_theta_next, 0, sizeof(double) * std::max<size_t>((size_t)D * h->K, 1)));
  if (h->C) {
    HIP_TRY(h, hipMemcpy(h->d_chunk_start, cstart.data(), sizeof(int64_t) * cstart.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_chunk_doc, cdoc.data(), sizeof(int32_t) * cdoc.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_chunk_len, clen.data(), sizeof(int32_t) * clen.size(), hipMemcpyHostToDevice));
    if (two_rows) {
      if ((rc = dev_alloc(h, &h->d_chunk_doc1, cdoc1.size()))) return rc;
      HIP_TRY(h, hipMemcpy(h->d_chunk_doc1, cdoc1.data(), sizeof(int32_t) * cdoc1.size(), hipMemcpyHostToDevice));
    }
  }
  HIP_TRY(h, hipDeviceSynchronize());   // the uploads and memsets above ran on the null stream; the handle's stream may not synchronise with it
  h->have_corpus = true; h->have_phi = false; h->in_sweep = false; h->global_tokens = -1; h->lcg_ready = false;
  h->z_split = h->z_split_allowed; h->z_split_tried = h->z_split_forced;
  h->counts_global = h->xg == nullptr; h->cnt_own_valid = false; h->n_k_valid = false;
  return GGS_OK;
}

int ggs_init_z_java_lcg(ggs_handle *h, int32_t seed) {
  int rc = require_ready(h, false);
  if (rc) return rc;
  if (h->tok_base != 0) return set_err(h, GGS_ERR_STATE, "java-LCG init is one sequential stream: only valid with tok_base == 0");
  if ((rc = drop_theta_ahead(h))) return rc;
  // One sequential stream, so it runs on the host exactly once at start-up.
  std::vector<int32_t> z((size_t)h->N);
  const uint64_t lcg_state = java_lcg_next_ints(seed, h->K, h->N, z.data());
  if (h->collapsed) {
    // SerialCollapsedLDA owns ONE Randoms(seed) (SerialCollapsedLDA.java:60-65): addInstances draws the initial topics
    // from it (:789) and the sampling loop goes on with the same object (MSLDA:206): the serial sweep continues this stream
    HIP_TRY(h, hipMemcpyAsync(h->d_lcg, &lcg_state, sizeof lcg_state, hipMemcpyHostToDevice, h->stream));
    h->lcg_ready = true;
  }
  if (h->N) HIP_TRY(h, hipMemcpyAsync(h->d_z, z.data(), sizeof(int32_t) * (size_t)h->N, hipMemcpyHostToDevice, h->stream));
  if ((rc = launch_permute_z(h))) return rc;
  if ((rc = launch_count_rebuild(h))) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GGS_OK;
}

int ggs_set_z(ggs_handle *h, const int32_t *z, int32_t redraw_phi) {
  int rc = require_ready(h, false);
  if (rc) return rc;
  if (h->N > 0 && !z) return set_err(h, GGS_ERR_BAD_ARG, "z is null");
  for (int64_t i = 0; i < h->N; ++i)
    if (z[i] < 0 || z[i] >= h->K) return set_err(h, GGS_ERR_BAD_ARG, "topic indicator outside [0, num_topics)");
  if ((rc = drop_theta_ahead(h))) return rc;
  h->lcg_ready = false;                                // a restored z is not the state any java.util.Random stream left: the serial chain starts a new Random(java_seed)
  if (h->N) HIP_TRY(h, hipMemcpyAsync(h->d_z, z, sizeof(int32_t) * (size_t)h->N, hipMemcpyHostToDevice, h->stream));
  if ((rc = launch_permute_z(h))) return rc;
  if ((rc = launch_count_rebuild(h))) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (redraw_phi) return ggs_init_phi(h);
  return GGS_OK;
}

int ggs_init_phi(ggs_handle *h) {
  int rc = require_ready(h, false);
  if (rc) return rc;
  if (h->collapsed) {                                  // nothing to draw: corpus-wide counts and tokensPerTopic are the whole model
    if ((rc = launch_magnitude(h))) return rc;
    h->have_phi = true;
    return check_status(h);
  }
  if ((rc = launch_phi(h, true, false))) return rc;   // with an exchange: the start-up count reduce-scatter, the slice, the all-gather
  return check_status(h);
}

int ggs_set_iteration(ggs_handle *h, int32_t it) { if (!h) return GGS_ERR_BAD_ARG; h->iteration = it; return GGS_OK; }   // a theta drawn ahead for another iteration is simply not used
int ggs_get_iteration(const ggs_handle *h, int32_t *it) { if (!h || !it) return GGS_ERR_BAD_ARG; *it = h->iteration; return GGS_OK; }

// see theta_main: only where the z step is one launch pair (no parts), theta is drawn at all, no collective is in the chain, and
// the theta draw (D x K gammas) is the longer leg (the Phi chain draws V x K)
bool chain_on_side_ok(const ggs_handle *h) {
  return h->theta_main && h->z_sliced && h->side_hot && h->ev_chain_done && !h->xg && !h->collapsed && !(h->flags & GGS_FLAG_PCGS) &&
         (h->D >= (int64_t)h->V || h->theta_main_always);
}

int ggs_sweep_begin(ggs_handle *h) {
  int rc = require_ready(h, true);
  if (rc) return rc;
  if (h->in_sweep) return set_err(h, GGS_ERR_STATE, "ggs_sweep_begin called twice without ggs_sweep_end");
  h->iteration += 1;                                   // currentIteration = iteration, UPLDA:646
  h->chain_on_side = chain_on_side_ok(h);              // kept until the sweep's end
  if ((rc = z_phase(h))) { h->chain_on_side = false; return rc; }
  h->in_sweep = true;
  return GGS_OK;
}

int ggs_sweep_end(ggs_handle *h) {
  int rc = require_ready(h, true);
  if (rc) return rc;
  if (!h->in_sweep) return set_err(h, GGS_ERR_STATE, "ggs_sweep_end without ggs_sweep_begin");
  h->in_sweep = false;
  rc = finish_sweep(h, true);
  h->chain_on_side = false;
  return rc;
}

int ggs_sweep_end_async(ggs_handle *h) {
  int rc = require_ready(h, true);
  if (rc) return rc;
  if (!h->in_sweep) return set_err(h, GGS_ERR_STATE, "ggs_sweep_end_async without ggs_sweep_begin");
  h->in_sweep = false;
  rc = finish_sweep(h, true, false);
  h->chain_on_side = false;
  return rc;
}

int ggs_sweep(ggs_handle *h, int32_t n_sweeps) {
  for (int32_t i = 0; i < n_sweeps; ++i) {
    if (h) h->whole_sweep = true;
    int rc = ggs_sweep_begin(h);
    if (!rc) {
      h->in_sweep = false;
      rc = finish_sweep(h, true, i == n_sweeps - 1);
    }
    if (h) h->chain_on_side = h->whole_sweep = false;
    if (rc) return rc;
  }
  return GGS_OK;
}

int ggs_collapsed_serial_sweep(ggs_handle *h, int32_t java_seed, int32_t n_sweeps) {
  int rc = require_ready(h, false);
  if (rc) return rc;
  if (!h->collapsed) return set_err(h, GGS_ERR_STATE, "ggs_collapsed_serial_sweep needs GGS_FLAG_COLLAPSED");
  if (h->xg || h->tok_base != 0) return set_err(h, GGS_ERR_STATE, "the serial chain runs over ONE unsharded corpus");
  if (h->in_sweep) return set_err(h, GGS_ERR_STATE, "inside a split sweep");
  if ((size_t)h->K * 16 > (size_t)kMaxLdsBytes) return set_err(h, GGS_ERR_UNSUPPORTED, "num_topics too large for the serial kernel's LDS");
  if ((rc = launch_magnitude(h))) return rc;           // tokensPerTopic in step with the counts
  if (!h->lcg_ready) {                                 // new java.util.Random(seed): the scrambled initial state
    const uint64_t st = ((uint64_t)(int64_t)java_seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
    HIP_TRY(h, hipMemcpyAsync(h->d_lcg, &st, sizeof st, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->lcg_ready = true;
  }
  CollapsedSerialParams cp{};
  cp.doc_ptr = h->d_doc_ptr; cp.tok = h->d_tok; cp.inv_perm = h->d_inv_perm; cp.z = h->d_z; cp.zw = h->d_zw; cp.n_wk = h->d_n_wk; cp.n_k = h->d_n_k;
  cp.alpha = h->d_alpha; cp.lcg = h->d_lcg; cp.status = h->d_status; cp.num_docs = h->D; cp.beta = h->beta; cp.beta_sum = h->beta * (double)h->V; cp.K = h->K;
  for (int32_t i = 0; i < n_sweeps; ++i) {
    h->iteration += 1;
    hipLaunchKernelGGL(collapsed_serial_kernel, dim3(1), dim3(64), (size_t)h->K * 16, h->stream, cp);
  }
  HIP_TRY(h, hipGetLastError());
  h->have_phi = true;
  return check_status(h);
}

int ggs_sample_z_given_phi(ggs_handle *h, int32_t n_sweeps) {
  int rc = require_ready(h, true);
  if (rc) return rc;
  if (h->in_sweep) return set_err(h, GGS_ERR_STATE, "inside a split sweep");
  if (h->collapsed) return set_err(h, GGS_ERR_UNSUPPORTED, "scheme=collapsed has no Phi to condition on");
  for (int32_t i = 0; i < n_sweeps; ++i) {
    h->iteration += 1;                                 // UPLDA:980
    h->force_detail = true;
    rc = z_phase(h);
    h->force_detail = false;
    if (rc) return rc;
    if ((rc = finish_sweep(h, false))) return rc;
  }
  // tokensPerTopic follows the rebuilt counts (with an exchange: the counts are merged here, UPLDA:993)
  if ((rc = launch_magnitude(h))) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GGS_OK;
}

int ggs_counts_device_ptr(ggs_handle *h, void **dev_ptr, int64_t *num_elems) {
  if (!h || !dev_ptr || !num_elems) return GGS_ERR_BAD_ARG;
  if (h->xg) return set_err(h, GGS_ERR_STATE, "an exchange is attached: the library merges the counts itself");
  *dev_ptr = h->d_n_wk; *num_elems = (int64_t)h->K * h->V;
  h->n_k_valid = false;          // the caller may sum other shards' counts into the buffer
  return GGS_OK;
}

int ggs_set_global_token_count(ggs_handle *h, int64_t n_tokens) {
  if (!h || n_tokens < 0) return GGS_ERR_BAD_ARG;
  h->global_tokens = n_tokens;
  return GGS_OK;
}

// ---- multi-GPU: the exchange (include/ggs_hip.h) ------------------------------------------------------------------
int ggs_attach_exchange(ggs_handle *h, int32_t rank, int32_t nranks, const ggs_exchange_ops *ops) {
  int rc = exchange_precheck(h, rank, nranks);
  if (rc) return rc;
  if (!ops || (ops->struct_size != (int32_t)sizeof(ggs_exchange_ops) && ops->struct_size != GGS_EXCHANGE_OPS_V3_SIZE) || !ops->reduce_scatter_i32 ||
      !ops->all_gather_f64 || !ops->all_gather_i32)
    return set_err(h, GGS_ERR_BAD_ARG, "ggs_exchange_ops: wrong struct_size or a null callback");
  auto *x = new (std::nothrow) Exchange();
  if (!x) return GGS_ERR_HIP;
  x->rank = rank; x->nranks = nranks;
  std::memcpy(&x->ops, ops, (size_t)ops->struct_size);   // a version-3 table has no all_to_all_v_i32: it stays null, the count exchange dense
  x->ops.struct_size = (int32_t)sizeof(ggs_exchange_ops);
  return setup_exchange(h, x);
}

int ggs_attach_null_exchange(ggs_handle *h, int32_t rank, int32_t nranks) {
  int rc = exchange_precheck(h, rank, nranks);
  if (rc) return rc;
  auto *x = new (std::nothrow) Exchange();
  if (!x) return GGS_ERR_HIP;
  x->rank = rank; x->nranks = nranks; x->is_null = true;
  x->ops.struct_size = (int32_t)sizeof(ggs_exchange_ops); x->ops.ctx = x;
  x->ops.reduce_scatter_i32 = xops::null_reduce_scatter_i32;
  x->ops.all_gather_f64 = xops::null_all_gather<double>;
  x->ops.all_gather_i32 = xops::null_all_gather<int32_t>;
  x->ops.all_to_all_v_i32 = xops::null_all_to_all_v_i32;
  return setup_exchange(h, x);
}

int ggs_rccl_unique_id(void *out_id) {
  if (!out_id) return GGS_ERR_BAD_ARG;
  std::string err;
  RcclApi *api = RcclApi::get(err);
  if (!api) return GGS_ERR_UNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == GGS_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return GGS_ERR_HIP;
  std::memcpy(out_id, &id, sizeof id);
  return GGS_OK;
}

int ggs_attach_rccl(ggs_handle *h, int32_t rank, int32_t nranks, const void *unique_id) {
  if (!h) return GGS_ERR_BAD_ARG;
  if (!unique_id) return set_err(h, GGS_ERR_BAD_ARG, "null unique id");
  int rc = exchange_precheck(h, rank, nranks);       // before the blocking collective below: a rejected call must not have joined it
  if (rc) return rc;
  Exchange *x = new_rccl_exchange(h, rank, nranks, &rc);
  if (!x) return rc;
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  const ncclResult_t r = x->api->CommInitRank(&x->comm, nranks, id, rank);      // collective: every rank of the job is in here
  if (r != ncclSuccess) {
    const std::string msg = std::string("ncclCommInitRank: ") + x->api->GetErrorString(r);
    delete x;
    return set_err(h, GGS_ERR_HIP, msg);
  }
  x->own_comm = true;
  return setup_exchange(h, x);
}

int ggs_attach_rccl_comm(ggs_handle *h, int32_t rank, int32_t nranks, void *nccl_comm) {
  if (!h) return GGS_ERR_BAD_ARG;
  if (!nccl_comm) return set_err(h, GGS_ERR_BAD_ARG, "null communicator");
  int rc = exchange_precheck(h, rank, nranks);
  if (rc) return rc;
  Exchange *x = new_rccl_exchange(h, rank, nranks, &rc);
  if (!x) return rc;
  x->comm = static_cast<ncclComm_t>(nccl_comm);
  return setup_exchange(h, x);
}

int ggs_get_exchange_info(const ggs_handle *h, int32_t *rank, int32_t *nranks, int32_t *k_begin, int32_t *k_end) {
  if (!h) return GGS_ERR_BAD_ARG;
  if (rank) *rank = h->xg ? h->xg->rank : 0;
  if (nranks) *nranks = h->xg ? h->xg->nranks : 1;
  if (k_begin) *k_begin = h->k0;
  if (k_end) *k_end = h->k0 + h->Ks;
  return GGS_OK;
}

int ggs_set_count_exchange(ggs_handle *h, int32_t mode) {
  if (!h || mode < 0 || mode > 2) return GGS_ERR_BAD_ARG;
  if (h->in_sweep) return set_err(h, GGS_ERR_STATE, "inside a split sweep");
  h->count_mode = mode;
  return GGS_OK;
}
int ggs_get_count_exchange(const ggs_handle *h, int32_t *sparse, int64_t *pairs_last, int64_t *cells) {
  if (!h || !sparse) return GGS_ERR_BAD_ARG;
  *sparse = use_sparse(h) ? 1 : 0;
  if (pairs_last) *pairs_last = h->sp_pairs_last;
  if (cells) *cells = h->xg ? (int64_t)h->V * h->Ksm * h->xg->nranks : (int64_t)h->V * h->K;
  return GGS_OK;
}

int ggs_get_exchange_provider(const ggs_handle *h, int32_t *provider, int32_t *comm_nranks, int32_t *comm_rank) {
  if (!h) return GGS_ERR_BAD_ARG;
  const Exchange *x = h->xg;
  if (provider) *provider = !x ? 0 : x->api ? 1 : x->is_null ? 3 : 2;
  int n = x ? x->nranks : 1, r = x ? x->rank : 0;
  if (x && x->api && x->comm) {                         // what the communicator itself reports, not what the caller passed in
    if (x->api->CommCount && x->api->CommCount(x->comm, &n) != ncclSuccess) n = -1;
    if (x->api->CommUserRank && x->api->CommUserRank(x->comm, &r) != ncclSuccess) r = -1;
  }
  if (comm_nranks) *comm_nranks = n;
  if (comm_rank) *comm_rank = r;
  return GGS_OK;
}

// ---- one process, n GPUs ----
namespace {
bool is_group(ggs_handle **hs, int32_t n) {
  if (!hs || n < 1 || !hs[0] || (int32_t)hs[0]->group.size() != n) return false;
  for (int32_t i = 0; i < n; ++i)
    if (hs[i] != hs[0]->group[(size_t)i] || !hs[i]->xg || hs[i]->xg->api != hs[0]->xg->api) return false;   // all RCCL, or all the caller's transport
  return true;
}

// the Phi phase for every handle of the group: each collective step for all devices inside ncclGroupStart/End
int group_phi(ggs_handle **hs, int32_t n, bool initial, bool in_sweep) {
  std::vector<char> acc((size_t)n, 0);
  int rc = GGS_OK;
  for (int32_t i = 0; i < n && !rc; ++i) {
    ggs_handle *h = hs[i];
    if ((rc = bind_device(h))) break;
    if (in_sweep) {
      Events &E = h->evs[h->ev_head];
      if (!E.light && hipEventRecord(E.e[4], h->stream) != hipSuccess) { rc = set_err(h, GGS_ERR_HIP, "hipEventRecord"); break; }
      E.exchanged = true; E.e4_is_e3 = false;
      acc[(size_t)i] = (h->flags & GGS_FLAG_SAVE_PHI_MEAN) && sample_phi_this_iteration(h);
    }
  }
  if (rc) return rc;
  auto timed = [&](ggs_handle *h) -> Events * { return (in_sweep && !h->evs[h->ev_head].light) ? &h->evs[h->ev_head] : nullptr; };   // a light sweep records no phase events
  auto grouped = [&](auto step) { return group_collective(hs, n, [&](ggs_handle *h) { return step(h, timed(h)); }); };
  // a step for every handle; events are recorded AFTER a grouped step: inside ncclGroupStart/End the collectives are
  // only collected, and an event recorded there would land on the stream before them
  auto each = [&](auto step) {
    int r = GGS_OK;
    for (int32_t i = 0; i < n && !r; ++i)
      if (!(r = bind_device(hs[i]))) r = step(hs[i], timed(hs[i]));
    return r;
  };
  auto record = [](ggs_handle *h, hipEvent_t ev) { return hipEventRecord(ev, h->stream) == hipSuccess ? GGS_OK : set_err(h, GGS_ERR_HIP, "hipEventRecord"); };
  if ((rc = grouped([](ggs_handle *h, Events *) { return phi_step_a(h); })) ||
      (rc = each([&](ggs_handle *h, Events *E) { int r = phi_step_a_clear(h); if (!r && E) r = record(h, E->x[0]); return r ? r : phi_step_b1(h, initial); })) ||
      (rc = grouped([](ggs_handle *h, Events *) { return phi_step_g0(h); })) ||
      (rc = each([&](ggs_handle *h, Events *E) { int r = phi_step_b2(h, initial); if (!r && E) r = record(h, E->x[1]); return r ? r : phi_join_halves(h); })) ||
      (rc = grouped([](ggs_handle *h, Events *) { return phi_step_g1(h); })) ||
      (rc = each([&](ggs_handle *h, Events *E) { return E ? record(h, E->x[2]) : GGS_OK; })))
    return rc;
  for (int32_t i = 0; i < n; ++i) {
    ggs_handle *h = hs[i];
    if ((rc = bind_device(h)) || (rc = phi_step_c(h, acc[(size_t)i] != 0))) return rc;
    if (in_sweep) {
      HIP_TRY(h, hipEventRecord(h->evs[h->ev_head].e[5], h->stream));
      if (acc[(size_t)i]) h->n_sampled_phi++;
      h->ev_pending += 1;
    }
  }
  return GGS_OK;
}
}  // namespace

namespace {
// corpus-wide counts on every handle of the group: the two collectives of ensure_global_counts, each grouped
int group_gather_counts(ggs_handle **hs, int32_t n) {
  int rc;
  if ((rc = group_collective(hs, n, [](ggs_handle *h) { return h->counts_global ? GGS_OK : exchange_reduce_scatter(h); }))) return rc;
  for (int32_t i = 0; i < n; ++i)
    if ((rc = bind_device(hs[i])) || (rc = clear_send_buffer_if_dead(hs[i], hs[i]->stream))) return rc;
  if ((rc = group_collective(hs, n, [](ggs_handle *h) { return h->counts_global ? GGS_OK : gather_counts_step_gather(h); }))) return rc;
  for (int32_t i = 0; i < n; ++i)
    if (!hs[i]->counts_global && ((rc = bind_device(hs[i])) || (rc = gather_counts_step_unslice(hs[i])))) return rc;
  return GGS_OK;
}
}  // namespace

int ggs_group_gather_counts(ggs_handle **hs, int32_t n) {
  if (!is_group(hs, n)) return GGS_ERR_BAD_ARG;
  return group_gather_counts(hs, n);
}

int ggs_group_create(const ggs_config *cfg, int32_t n, const int32_t *device_ids, ggs_handle **out) {
  if (!cfg || n < 1 || !device_ids || !out) return GGS_ERR_BAD_ARG;
  for (int32_t i = 0; i < n; ++i) out[i] = nullptr;
  std::string err;
  RcclApi *api = RcclApi::get(err);
  if (!api) return GGS_ERR_UNSUPPORTED;
  int rc = GGS_OK;
  for (int32_t i = 0; i < n && !rc; ++i) {
    ggs_config c = *cfg;
    c.device_id = device_ids[i];
    rc = ggs_create(&c, &out[i]);
  }
  std::vector<ncclComm_t> comms((size_t)n, nullptr);
  if (!rc && api->CommInitAll(comms.data(), n, device_ids) != ncclSuccess) rc = GGS_ERR_HIP;
  for (int32_t i = 0; i < n && !rc; ++i) {
    Exchange *x = new_rccl_exchange(out[i], i, n, &rc);
    if (!x) break;
    x->comm = comms[(size_t)i]; x->own_comm = true; comms[(size_t)i] = nullptr;
    rc = setup_exchange(out[i], x);
  }
  if (rc) {
    for (ncclComm_t c : comms) if (c) (void)api->CommDestroy(c);
    for (int32_t i = 0; i < n; ++i) { ggs_destroy(out[i]); out[i] = nullptr; }
    return rc;
  }
  out[0]->group.assign(out, out + n);
  for (int32_t i = 0; i < n; ++i) out[i]->in_group = true;
  return GGS_OK;
}

int ggs_group_adopt(ggs_handle **hs, int32_t n) {
  if (!hs || n < 1) return GGS_ERR_BAD_ARG;
  for (int32_t i = 0; i < n; ++i) {
    if (!hs[i]) return GGS_ERR_BAD_ARG;
    if (!hs[i]->xg || hs[i]->xg->api) return set_err(hs[i], GGS_ERR_STATE, "ggs_group_adopt takes handles with a caller-supplied exchange (ggs_attach_exchange)");
    if (hs[i]->xg->rank != i || hs[i]->xg->nranks != n) return set_err(hs[i], GGS_ERR_BAD_ARG, "ggs_group_adopt: handle i must be rank i of n");
  }
  hs[0]->group.assign(hs, hs + n);
  for (int32_t i = 0; i < n; ++i) hs[i]->in_group = true;
  return GGS_OK;
}

void ggs_group_destroy(ggs_handle **handles, int32_t n) {
  if (!handles) return;
  for (int32_t i = 0; i < n; ++i) { ggs_destroy(handles[i]); handles[i] = nullptr; }
}

int ggs_group_set_z(ggs_handle **hs, int32_t n, const int32_t *const *z, int32_t redraw_phi) {
  if (!is_group(hs, n) || !z) return GGS_ERR_BAD_ARG;
  int rc;
  for (int32_t i = 0; i < n; ++i)
    if ((rc = ggs_set_z(hs[i], z[i], 0))) return rc;          // this shard's counts
  if (!redraw_phi) return GGS_OK;
  if (hs[0]->collapsed) {                                        // no Phi: the merged counts and tokensPerTopic are the model
    if ((rc = group_gather_counts(hs, n))) return rc;
    for (int32_t i = 0; i < n; ++i) {
      if ((rc = bind_device(hs[i])) || (rc = launch_magnitude(hs[i]))) return rc;
      hs[i]->have_phi = true;
    }
  } else if ((rc = group_phi(hs, n, true, false))) return rc;
  for (int32_t i = 0; i < n; ++i)
    if ((rc = bind_device(hs[i])) || (rc = check_status(hs[i]))) return rc;
  return GGS_OK;
}

int ggs_group_sweep(ggs_handle **hs, int32_t n, int32_t n_sweeps) {
  if (!is_group(hs, n)) return GGS_ERR_BAD_ARG;
  int rc;
  for (int32_t s = 0; s < n_sweeps; ++s) {
    if (hs[0]->collapsed && (rc = group_gather_counts(hs, n))) return rc;   // the z step conditions on the corpus-wide sweep-start counts
    for (int32_t i = 0; i < n; ++i) {
      ggs_handle *h = hs[i];
      if ((rc = require_ready(h, true))) return rc;
      if (h->in_sweep) return set_err(h, GGS_ERR_STATE, "inside a split sweep");
      h->iteration += 1;
      if ((rc = z_phase(h))) return rc;
    }
    if (hs[0]->collapsed) {                                      // the AD-LDA merge: gathered counts, then tokensPerTopic; no Phi
      if ((rc = group_gather_counts(hs, n))) return rc;
      for (int32_t i = 0; i < n; ++i) {
        ggs_handle *h = hs[i];
        if ((rc = bind_device(h))) return rc;
        Events &E = h->evs[h->ev_head];
        E.exchanged = false; E.e4_is_e3 = false;
        HIP_TRY(h, hipEventRecord(E.e[4], h->stream));
        if ((rc = launch_magnitude(h))) return rc;
        HIP_TRY(h, hipEventRecord(E.e[5], h->stream));
        h->ev_pending += 1;
      }
    } else if ((rc = group_phi(hs, n, false, true))) return rc;
    if (s == n_sweeps - 1 || (hs[0]->flags & GGS_FLAG_PARANOID))
      for (int32_t i = 0; i < n; ++i)
        if ((rc = bind_device(hs[i])) || (rc = check_status(hs[i])) || (rc = settle_sweeps(hs[i]))) return rc;
  }
  return GGS_OK;
}

static int copy_out(ggs_handle *h, void *dst, const void *src, size_t bytes) {
  if (!bytes) return GGS_OK;
  HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GGS_OK;
}

int ggs_synchronize(ggs_handle *h) {
  if (!h) return GGS_ERR_BAD_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  return check_status(h);
}

int ggs_get_z(ggs_handle *h, int32_t *z) {
  int rc = require_ready(h, false);
  if (rc) return rc;
  if (h->N > 0 && !z) return GGS_ERR_BAD_ARG;
  return copy_out(h, z, h->d_z, sizeof(int32_t) * (size_t)h->N);
}
int ggs_get_type_topic_counts(ggs_handle *h, int32_t *n_wk) {
  if (!h || !n_wk) return GGS_ERR_BAD_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  if ((rc = ensure_global_counts(h))) return rc;
  return copy_out(h, n_wk, h->d_n_wk, sizeof(int32_t) * (size_t)h->K * h->V);
}
int ggs_get_topic_totals(ggs_handle *h, int32_t *n_k) {
  if (!h || !n_k) return GGS_ERR_BAD_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  // valid whatever came last: a sweep, ggs_set_z(redraw=0), an external all-reduce on ggs_counts_device_ptr
  h->n_k_valid = h->n_k_valid && h->xg != nullptr;
  if ((rc = launch_magnitude(h))) return rc;
  return copy_out(h, n_k, h->d_n_k, sizeof(int32_t) * (size_t)h->K);
}

static int phi_out(ggs_handle *h, const double *src_T, int32_t pitch, double *dst, double scale) {
  const size_t kv = (size_t)h->K * h->V;
  int rc = ensure_scratch(h, kv * sizeof(double));
  if (rc) return rc;
  hipLaunchKernelGGL(phiT_to_phi_kernel, dim3(grid_for((int64_t)kv, 256)), dim3(256), 0, h->stream, src_T, static_cast<double *>(h->d_scratch),
                     h->K, pitch, h->V, scale);
  HIP_TRY(h, hipGetLastError());
  return copy_out(h, dst, h->d_scratch, kv * sizeof(double));
}
int ggs_get_phi(ggs_handle *h, double *phi) {
  if (!h || !phi) return GGS_ERR_BAD_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  if (h->collapsed) {                                  // the point estimate from the current counts
    if ((rc = launch_magnitude(h))) return rc;
    hipLaunchKernelGGL(psi_kernel, dim3(grid_for((int64_t)h->K * h->V, 256, 2)), dim3(256), 0, h->stream, h->d_n_wk, h->d_n_k, h->beta, h->beta * (double)h->V,
                       h->d_phiT, h->K, h->Kp, h->V);
    HIP_TRY(h, hipGetLastError());
  }
  return phi_out(h, h->d_phiT, h->Kp, phi, 1.0);
}
int ggs_set_phi(ggs_handle *h, const double *phi) {
  if (!h || !phi) return GGS_ERR_BAD_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  const size_t kv = (size_t)h->K * h->V;
  if ((rc = ensure_scratch(h, kv * sizeof(double)))) return rc;
  HIP_TRY(h, hipMemcpyAsync(h->d_scratch, phi, kv * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(phi_to_phiT_kernel, dim3(grid_for((int64_t)kv, 256)), dim3(256), 0, h->stream, static_cast<const double *>(h->d_scratch),
                     h->d_phiT, h->K, h->Kp, h->V);
  HIP_TRY(h, hipGetLastError());
  // UPLDA:1897-1902: `if (savePhiMeans()) phiMean = new double[numTopics][numTypes]` -- the running sum restarts, noSampledPhi keeps counting
  if (h->d_phi_mean) HIP_TRY(h, hipMemsetAsync(h->d_phi_mean, 0, kv * sizeof(double), h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->have_phi = true;
  return GGS_OK;
}
int ggs_get_phi_mean(ggs_handle *h, double *phi_mean, int32_t *n_sampled) {
  if (!h || !n_sampled) return GGS_ERR_BAD_ARG;
  *n_sampled = h->n_sampled_phi;
  if (h->n_sampled_phi == 0 || !h->d_phi_mean) return GGS_OK;   // Java returns null (UPLDA:1955-1958)
  if (!phi_mean) return GGS_ERR_BAD_ARG;
  int rc = bind_device(h);
  if (rc) return rc;
  return phi_out(h, h->d_phi_mean, h->K, phi_mean, (double)h->n_sampled_phi);
}
int ggs_get_theta(ggs_handle *h, int64_t doc_begin, int64_t doc_end, double *theta) {
  int rc = require_ready(h, false);
  if (rc) return rc;
  if (doc_begin < 0 || doc_end > h->D || doc_begin > doc_end || (!theta && doc_end > doc_begin)) return set_err(h, GGS_ERR_BAD_ARG, "bad document range");
  return copy_out(h, theta, h->d_theta + (size_t)doc_begin * h->K, sizeof(double) * (size_t)(doc_end - doc_begin) * h->K);
}
int ggs_get_doc_topic_counts(ggs_handle *h, int64_t doc_begin, int64_t doc_end, int32_t *n_dk) {
  int rc = require_ready(h, false);
  if (rc) return rc;
  if (doc_begin < 0 || doc_end > h->D || doc_begin > doc_end || (!n_dk && doc_end > doc_begin)) return set_err(h, GGS_ERR_BAD_ARG, "bad document range");
  const int64_t nd = doc_end - doc_begin;
  if (nd == 0) return GGS_OK;
  const size_t bytes = (size_t)nd * h->K * sizeof(int32_t);
  if ((rc = ensure_scratch(h, bytes))) return rc;
  HIP_TRY(h, hipMemsetAsync(h->d_scratch, 0, bytes, h->stream));
  hipLaunchKernelGGL(doc_topic_kernel, dim3((unsigned)nd), dim3(64), 0, h->stream, h->d_doc_ptr, h->d_z, doc_begin, h->K, static_cast<int32_t *>(h->d_scratch));
  HIP_TRY(h, hipGetLastError());
  return copy_out(h, n_dk, h->d_scratch, bytes);
}
int ggs_model_log_likelihood(ggs_handle *h, double *doc_side, double *topic_side) {
  int rc = require_ready(h, false);
  if (rc) return rc;
  if (!doc_side || !topic_side) return set_err(h, GGS_ERR_BAD_ARG, "null output");
  if ((rc = launch_magnitude(h))) return rc;       // corpus-wide counts gathered, tokensPerTopic in step with them
  const int K = h->K;
  const int64_t doc_blocks = (h->D + kLLBlock / 64 - 1) / (kLLBlock / 64), type_blocks = 1024;
  const size_t bytes = 32 + sizeof(double) * (size_t)(doc_blocks + type_blocks);
  if ((rc = ensure_scratch(h, bytes))) return rc;
  auto *d_out = static_cast<double *>(h->d_scratch);                       // [0] document side, [1] topic side
  auto *d_nz = reinterpret_cast<unsigned long long *>(d_out + 2);
  double *d_doc = d_out + 4, *d_type = d_doc + doc_blocks;
  HIP_TRY(h, hipMemsetAsync(h->d_scratch, 0, 32, h->stream));
  double alpha_sum = 0;
  for (int k = 0; k < K; ++k) alpha_sum += h->alpha[k];
  if (doc_blocks)
    hipLaunchKernelGGL(ll_docs_kernel, dim3((unsigned)doc_blocks), dim3(kLLBlock), (size_t)(kLLBlock / 64) * K * sizeof(int32_t), h->stream, h->d_doc_ptr,
                       h->d_z, h->d_alpha, alpha_sum, h->D, K, d_doc);
  hipLaunchKernelGGL(ll_types_kernel, dim3((unsigned)type_blocks), dim3(kLLBlock), 0, h->stream, h->d_n_wk, (int64_t)K * h->V, h->beta, d_type, d_nz);
  hipLaunchKernelGGL(ll_finish_kernel, dim3(1), dim3(kLLBlock), 0, h->stream, d_doc, doc_blocks, d_type, type_blocks, h->d_n_k, K, h->beta * h->V,
                     alpha_sum, h->beta, h->D, d_nz, d_out);
  HIP_TRY(h, hipGetLastError());
  double out[2];
  if ((rc = copy_out(h, out, d_out, sizeof out))) return rc;
  *doc_side = out[0]; *topic_side = out[1];
  return GGS_OK;
}
int ggs_log_posterior(ggs_handle *h, double *doc_side, double *topic_side) {
  int rc = require_ready(h, true);
  if (rc) return rc;
  if (!doc_side || !topic_side) return set_err(h, GGS_ERR_BAD_ARG, "null output");
  if (h->collapsed) return set_err(h, GGS_ERR_UNSUPPORTED, "scheme=collapsed has no Phi: the log posterior of UPLDA:1573-1634 does not apply");
  if (h->flags & GGS_FLAG_PCGS) {
    // UPLDA:710-714: every scheme but ggs draws theta_d ~ Dir(n_d. + alpha) afresh for the diagnostics
    // (LDAUtils.drawDirichlets); here: the theta draw of GGS:57-72 under the stream GGS_PURPOSE_THETA at the current iteration
    if ((rc = drop_theta_ahead(h)) || (rc = launch_theta(h, h->stream, h->d_theta, h->iteration))) return rc;
  }
  const int K = h->K;
  const int64_t doc_blocks = (h->D + kLLBlock / 64 - 1) / (kLLBlock / 64), phi_blocks = 1024;
  const size_t bytes = 32 + sizeof(double) * (size_t)(doc_blocks + phi_blocks);
  if ((rc = ensure_scratch(h, bytes))) return rc;
  auto *d_out = static_cast<double *>(h->d_scratch);
  double *d_doc = d_out + 4, *d_phi = d_doc + doc_blocks;
  if (doc_blocks)
    hipLaunchKernelGGL(lp_docs_kernel, dim3((unsigned)doc_blocks), dim3(kLLBlock), (size_t)(kLLBlock / 64) * K * sizeof(int32_t), h->stream, h->d_doc_ptr,
                       h->d_tok, h->d_z, h->d_alpha, h->d_theta, h->d_phiT, h->D, K, h->Kp, d_doc);
  hipLaunchKernelGGL(lp_phi_kernel, dim3((unsigned)phi_blocks), dim3(kLLBlock), 0, h->stream, h->d_phiT, (int64_t)h->V, K, h->Kp, d_phi);
  hipLaunchKernelGGL(lp_finish_kernel, dim3(1), dim3(kLLBlock), 0, h->stream, d_doc, doc_blocks, d_phi, phi_blocks, h->beta, d_out);
  HIP_TRY(h, hipGetLastError());
  double out[2];
  if ((rc = copy_out(h, out, d_out, sizeof out))) return rc;
  *doc_side = out[0]; *topic_side = out[1];
  return GGS_OK;
}
int ggs_set_test_corpus(ggs_handle *h, int64_t D, const int64_t *doc_ptr, const int32_t *tokens, int64_t doc_base) {
  if (!h || D < 0 || !doc_ptr || doc_base < 0) return set_err(h, GGS_ERR_BAD_ARG, "bad test corpus arguments");
  if (doc_ptr[0] != 0) return set_err(h, GGS_ERR_BAD_ARG, "doc_ptr[0] must be 0");
  for (int64_t d = 0; d < D; ++d) {
    if (doc_ptr[d + 1] < doc_ptr[d]) return set_err(h, GGS_ERR_BAD_ARG, "doc_ptr must be non-decreasing");
    if (doc_ptr[d + 1] - doc_ptr[d] > ((int64_t)1 << 25))
      return set_err(h, GGS_ERR_UNSUPPORTED, "a test document has more than 2^25 tokens (one Philox stream per particle: 2^24 blocks of two uniforms)");
  }
  const int64_t N = doc_ptr[D];
  if (N > 0 && !tokens) return set_err(h, GGS_ERR_BAD_ARG, "tokens is null");
  for (int64_t i = 0; i < N; ++i)
    if (tokens[i] < 0) return set_err(h, GGS_ERR_BAD_ARG, "negative token id");    // ids >= num_types are out of vocabulary (MPE:341-345)
  int rc = bind_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if ((rc = dev_alloc(h, &h->d_test_ptr, (size_t)D + 1)) || (rc = dev_alloc(h, &h->d_test_tok, (size_t)N)) || (rc = dev_alloc(h, &h->d_test_ll, (size_t)D))) return rc;
  HIP_TRY(h, hipMemcpy(h->d_test_ptr, doc_ptr, sizeof(int64_t) * (size_t)(D + 1), hipMemcpyHostToDevice));
  if (N) HIP_TRY(h, hipMemcpy(h->d_test_tok, tokens, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice));
  h->test_ptr.assign(doc_ptr, doc_ptr + D + 1);
  // by the width a particle's per-topic counts need: one byte (documents of at most 255 tokens), two (65 535), four
  h->test_short.clear(); h->test_long.clear(); h->test_huge.clear();
  for (int64_t d = 0; d < D; ++d) {
    const int64_t len = doc_ptr[d + 1] - doc_ptr[d];
    (len <= 255 ? h->test_short : len <= 65535 ? h->test_long : h->test_huge).push_back((int32_t)d);
  }
  if ((rc = dev_alloc(h, &h->d_test_docs, (size_t)D))) return rc;
  {
    size_t at = 0;
    for (const std::vector<int32_t> *ids : {&h->test_short, &h->test_long, &h->test_huge}) {
      if (!ids->empty()) HIP_TRY(h, hipMemcpy(h->d_test_docs + at, ids->data(), sizeof(int32_t) * ids->size(), hipMemcpyHostToDevice));
      at += ids->size();
    }
  }
  h->test_doc_base = doc_base;
  h->have_test = true;
  return GGS_OK;
}

int ggs_heldout_log_likelihood(ggs_handle *h, int32_t num_particles, double *doc_ll, double *total) {
  int rc = require_ready(h, false);
  if (rc) return rc;
  if (!h->have_test) return set_err(h, GGS_ERR_STATE, "no test set: call ggs_set_test_corpus first");
  if (num_particles < 1 || !total) return set_err(h, GGS_ERR_BAD_ARG, "num_particles < 1 or null output");
  if ((rc = launch_magnitude(h))) return rc;       // corpus-wide counts gathered, tokensPerTopic in step with them
  const int K = h->K;
  const int64_t D = (int64_t)h->test_ptr.size() - 1;
  // LDS: alpha, the denominators and the coefficient table once per block; per wave the word's cell list and 1 or 2 bytes
  // per (particle, topic).  The block shape that puts most waves on a CU wins (the kernel is issue-bound).
  const size_t Kpad = (size_t)(K + 63) / 64 * 64;
  auto lds_of = [&](int w, int cnt_bytes, int cap) { return (size_t)(16 + 8 * cap) * K + (size_t)w * (Kpad * 8 + (size_t)K * 64 * cnt_bytes); };
  struct Shape { int waves = 0, per_cu = 0, cap = 0; };
  auto shape_for = [&](int cnt_bytes, int cap) {
    Shape s; s.cap = cap;
    for (int w = kHeldoutMaxWaves; w >= 1; w >>= 1) {
      const size_t alloc = (lds_of(w, cnt_bytes, cap) + 2047) / 2048 * 2048;   // LDS is handed out in 2 KiB granules
      const int per_cu = alloc <= (size_t)160 * 1024 ? std::min((int)((size_t)160 * 1024 / alloc) * w, 32) : 0;
      if (per_cu && per_cu >= s.per_cu) { s.per_cu = per_cu; s.waves = w; }       // ties: the smaller block (documents differ in length)
    }
    return s;
  };
  // the deepest coefficient table that costs no wave (measured at K=100: 16 -> 31 ms, 32 -> 26, 48 -> 24.6 per evaluation)
  auto best_shape = [&](int cnt_bytes) {
    const int n_caps = (int)(sizeof(kHeldoutCoefCaps) / sizeof(kHeldoutCoefCaps[0]));
    const Shape floor = shape_for(cnt_bytes, kHeldoutCoefCaps[n_caps - 1]);
    for (int i = 0; i < n_caps; ++i) {
      const Shape s = shape_for(cnt_bytes, kHeldoutCoefCaps[i]);
      if (s.per_cu && s.per_cu >= floor.per_cu) return s;
    }
    return floor;
  };
  // a class of documents whose per-particle counts do not fit LDS (more than 1704 topics with one-byte counts, 1024 with
  // two; four-byte counts always) keeps them in global memory: the LDS then holds the tables and the cell lists only
  const Shape shape8 = best_shape(1), shape16 = best_shape(2), shape_spill = best_shape(0);
  const bool spill8 = !h->test_short.empty() && !shape8.waves, spill16 = !h->test_long.empty() && !shape16.waves, spill32 = !h->test_huge.empty();
  if ((spill8 || spill16 || spill32) && !shape_spill.waves)
    return set_err(h, GGS_ERR_UNSUPPORTED, "num_topics too large for the held-out estimator's tables in LDS (about 6000 topics)");
  const int64_t spill_blocks = (int64_t)h->num_cus * std::max(1, shape_spill.waves ? shape_spill.per_cu / shape_spill.waves : 1);   // a persistent grid: what is resident
  if (spill8 || spill16 || spill32) {
    const size_t need = (size_t)spill_blocks * shape_spill.waves * (size_t)K * 64 * (spill32 ? 4 : spill16 ? 2 : 1);
    if (h->heldout_spill_bytes < need) {
      if (h->d_heldout_spill) (void)hipFree(h->d_heldout_spill);
      h->d_heldout_spill = nullptr; h->heldout_spill_bytes = 0;
      HIP_TRY(h, hipMalloc(&h->d_heldout_spill, need));
      h->heldout_spill_bytes = need;
    }
  }
  const void *kernels[] = {reinterpret_cast<const void *>(heldout_particles_kernel<uint8_t, false>), reinterpret_cast<const void *>(heldout_particles_kernel<uint16_t, false>),
                           reinterpret_cast<const void *>(heldout_particles_kernel<uint8_t, true>), reinterpret_cast<const void *>(heldout_particles_kernel<uint16_t, true>),
                           reinterpret_cast<const void *>(heldout_particles_kernel<uint32_t, true>)};
  for (const void *f : kernels) HIP_TRY(h, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes));
  // wordProbabilities of a batch of documents: tokens x particles doubles, at most ~4 GiB at a time (one launch for
  // the 2 M-token test set of the benchmark: every extra launch has its own tail of half-empty CUs)
  int64_t want_cells = (int64_t)1 << 29;
  if (const char *e = debug_env("GGS_DEBUG_HELDOUT_CELLS")) want_cells = std::max<int64_t>(1, std::atoll(e));   // tests: force several batches
  int64_t longest = 1;
  for (int64_t d = 0; d < D; ++d) longest = std::max(longest, h->test_ptr[d + 1] - h->test_ptr[d]);
  const int64_t cap_cells = std::max<int64_t>(want_cells, longest * num_particles);                                 // a document is never split
  int64_t max_cells = 0;
  std::vector<int64_t> cuts{0};
  for (int64_t d = 0; d < D;) {
    int64_t e = d;
    while (e < D && (h->test_ptr[e + 1] - h->test_ptr[d]) * num_particles <= cap_cells) ++e;
    max_cells = std::max(max_cells, (h->test_ptr[e] - h->test_ptr[d]) * num_particles);
    cuts.push_back(e);
    d = e;
  }
  const size_t tab_bytes = sizeof(double) * (size_t)(2 * K + 2);
  if ((rc = ensure_scratch(h, tab_bytes + sizeof(double) * (size_t)std::max<int64_t>(max_cells, 1)))) return rc;
  HeldoutParams hp{};
  hp.doc_ptr = h->d_test_ptr; hp.tok = h->d_test_tok; hp.n_wk = h->d_n_wk;
  auto *tab = static_cast<double *>(h->d_scratch);
  hp.tab = tab; hp.probs = tab + 2 * K + 2; hp.doc_ll = h->d_test_ll; hp.status = h->d_status;
  hp.beta = h->beta;
  double alpha_sum = 0;
  for (int k = 0; k < K; ++k) alpha_sum += h->alpha[k];
  hp.alpha_sum = alpha_sum;
  hp.seed = h->seed; hp.iteration = (uint32_t)h->iteration; hp.doc_base = h->test_doc_base;
  hp.K = K; hp.V = h->V; hp.P = num_particles; hp.blocks_per_doc = (num_particles + 63) / 64;
  hipLaunchKernelGGL(heldout_setup_kernel, dim3(1), dim3(64), 0, h->stream, h->d_alpha, h->d_n_k, h->beta, h->beta * h->V, K, tab);
  for (size_t b = 0; b + 1 < cuts.size(); ++b) {
    hp.d0 = cuts[b]; hp.d1 = cuts[b + 1];
    if (hp.d1 == hp.d0) continue;
    // the batch's short and long documents: sub-ranges of the two ascending id lists
    auto range = [&](const std::vector<int32_t> &ids, size_t &lo, size_t &hi) {
      lo = std::lower_bound(ids.begin(), ids.end(), (int32_t)hp.d0) - ids.begin();
      hi = std::lower_bound(ids.begin(), ids.end(), (int32_t)hp.d1) - ids.begin();
    };
    // one launch per class of documents present in the batch: counts in LDS where they fit, in global memory otherwise
    struct Class { const std::vector<int32_t> *ids; size_t offset; int bytes; bool spill; Shape shape; };
    const Class classes[] = {{&h->test_short, 0, 1, spill8, spill8 ? shape_spill : shape8},
                             {&h->test_long, h->test_short.size(), 2, spill16, spill16 ? shape_spill : shape16},
                             {&h->test_huge, h->test_short.size() + h->test_long.size(), 4, true, shape_spill}};
    for (const Class &c : classes) {
      size_t lo, hi;
      range(*c.ids, lo, hi);
      if (hi <= lo) continue;
      hp.docs = h->d_test_docs + c.offset + lo; hp.n_docs = (int64_t)(hi - lo); hp.waves = c.shape.waves; hp.cap = c.shape.cap;
      hp.cnt_spill = c.spill ? h->d_heldout_spill : nullptr;
      const int64_t units = hp.n_docs * hp.blocks_per_doc, blocks = (units + hp.waves - 1) / hp.waves;
      const dim3 grid((unsigned)(c.spill ? std::min(blocks, spill_blocks) : blocks)), block((unsigned)(hp.waves * 64));
      const size_t lds = lds_of(hp.waves, c.spill ? 0 : c.bytes, hp.cap);
      const void *f = c.spill ? (c.bytes == 1 ? kernels[2] : c.bytes == 2 ? kernels[3] : kernels[4]) : (c.bytes == 1 ? kernels[0] : kernels[1]);
      void *args[] = {&hp};
      HIP_TRY(h, hipLaunchKernel(f, grid, block, args, lds, h->stream));
    }
    hipLaunchKernelGGL(heldout_reduce_kernel, dim3((unsigned)((hp.d1 - hp.d0 + 3) / 4)), dim3(256), 0, h->stream, hp);
  }
  HIP_TRY(h, hipGetLastError());
  std::vector<double> ll((size_t)D);
  if (D) HIP_TRY(h, hipMemcpyAsync(ll.data(), h->d_test_ll, sizeof(double) * (size_t)D, hipMemcpyDeviceToHost, h->stream));
  if ((rc = check_status(h))) return rc;                    // synchronises; MPE:416,447,455,464-469 surface here
  double t = 0;
  for (int64_t d = 0; d < D; ++d) t += ll[(size_t)d];       // MPE:116: in document order
  *total = t;
  if (doc_ll && D) std::memcpy(doc_ll, ll.data(), sizeof(double) * (size_t)D);
  return GGS_OK;
}

int ggs_get_timings(ggs_handle *h, ggs_timings *out) { if (!h || !out) return GGS_ERR_BAD_ARG; *out = h->tm; return GGS_OK; }
int ggs_reset_timings(ggs_handle *h) { if (!h) return GGS_ERR_BAD_ARG; h->tm = ggs_timings{}; return GGS_OK; }

int ggs_check_invariants(ggs_handle *h) {
  int rc = require_ready(h, false);
  if (rc) return rc;
  const int K = h->K;
  const size_t bytes = 16 + sizeof(int32_t) * (size_t)K;
  if ((rc = ensure_global_counts(h))) return rc;
  if ((rc = ensure_scratch(h, bytes))) return rc;
  auto *d_total = static_cast<unsigned long long *>(h->d_scratch);
  auto *d_flags = reinterpret_cast<uint32_t *>(static_cast<char *>(h->d_scratch) + 8);
  auto *d_col = reinterpret_cast<int32_t *>(static_cast<char *>(h->d_scratch) + 16);
  HIP_TRY(h, hipMemsetAsync(h->d_scratch, 0, bytes, h->stream));
  const int64_t kv = (int64_t)K * h->V;
  hipLaunchKernelGGL(invariants_kernel, dim3(grid_for(kv, 256)), dim3(256), 0, h->stream, h->d_n_wk, kv, K, d_total, d_col, d_flags);
  HIP_TRY(h, hipGetLastError());
  if ((rc = launch_magnitude(h))) return rc;
  std::vector<unsigned char> host(bytes);
  std::vector<int32_t> nk((size_t)K);
  if ((rc = copy_out(h, host.data(), h->d_scratch, bytes))) return rc;
  if ((rc = copy_out(h, nk.data(), h->d_n_k, sizeof(int32_t) * (size_t)K))) return rc;
  unsigned long long total; uint32_t fl;
  std::memcpy(&total, host.data(), 8); std::memcpy(&fl, host.data() + 8, 4);
  const int32_t *col = reinterpret_cast<const int32_t *>(host.data() + 16);
  if (fl & 1u) return set_err(h, GGS_ERR_INVARIANT, "negative type-topic count");
  if ((int64_t)total != (h->global_tokens >= 0 ? h->global_tokens : h->N)) return set_err(h, GGS_ERR_INVARIANT, "type-topic counts do not sum to the corpus size");
  for (int k = 0; k < K; ++k)
    if (col[k] != nk[(size_t)k]) return set_err(h, GGS_ERR_INVARIANT, "column sum differs from tokensPerTopic");
  return GGS_OK;
}

int ggs_get_launch_info(ggs_handle *h, int64_t *num_chunks, int32_t *lds_bytes_z, int32_t *docs_per_block_theta) {
  if (!h) return GGS_ERR_BAD_ARG;
  if (num_chunks) *num_chunks = h->z_sliced ? h->Cs + h->Cw : h->C;
  if (lds_bytes_z) *lds_bytes_z = h->z_lds;
  if (docs_per_block_theta) *docs_per_block_theta = h->theta_docs_per_block;
  return GGS_OK;
}

int ggs_get_num_hot_words(ggs_handle *h, int32_t *num_hot) {
  if (!h || !num_hot) return GGS_ERR_BAD_ARG;
  *num_hot = (h->z_sliced && !(h->flags & GGS_FLAG_PCGS)) ? h->num_hot + h->num_warm : 0;
  return GGS_OK;
}

int ggs_get_z_parts(ggs_handle *h, int32_t *parts) {
  if (!h || !parts) return GGS_ERR_BAD_ARG;
  const int32_t P = (int32_t)h->part_doc.size() - 1;
  *parts = (h->z_stream && h->overlap_theta && P > 1 && !(h->flags & GGS_FLAG_PCGS)) ? P : 1;
  return GGS_OK;
}

int ggs_get_warm_tiers(ggs_handle *h, int32_t *tiers, int32_t *warm_words, int32_t *docs_per_chunk) {
  if (!h) return GGS_ERR_BAD_ARG;
  const bool on = h->have_corpus && h->z_sliced && !(h->flags & GGS_FLAG_PCGS);
  if (tiers) *tiers = on ? h->warm_tiers : 0;
  if (warm_words) *warm_words = on ? h->num_warm : 0;
  if (docs_per_chunk) *docs_per_chunk = on && h->warm_tiers ? h->warm_docs : 0;
  return GGS_OK;
}

int ggs_get_z_form(ggs_handle *h, int32_t *kernel, int32_t *form, int32_t *calibrated) {
  if (!h) return GGS_ERR_BAD_ARG;
  const bool pcgs = (h->flags & GGS_FLAG_PCGS) != 0;
  const bool splittable = !pcgs && h->z_sliced && h->Cs > h->Cc && h->Cc > 0;
  if (kernel) *kernel = pcgs ? (h->pcgs_wave ? 5 : 4) : h->z_sliced ? 1 : h->z_stream ? (h->z_two_pass ? 3 : 2) : 0;
  if (form) *form = (!pcgs && h->z_sliced) ? (splittable && h->z_split ? 1 : 2) : 0;
  if (calibrated) *calibrated = (splittable && h->z_split_tried && !h->z_split_forced && h->z_split_allowed) ? 1 : 0;
  return GGS_OK;
}

int ggs_java_lcg_next_ints(int32_t seed, int32_t bound, int64_t n, int32_t *out) {
  if (bound <= 0 || n < 0 || (n > 0 && !out)) return GGS_ERR_BAD_ARG;
  java_lcg_next_ints(seed, bound, n, out);
  return GGS_OK;
}

// ---- primitives for the parity tests ---------------------------------------------
namespace {
struct TmpDev {
  std::vector<void *> p;
  ~TmpDev() { for (void *q : p) if (q) (void)hipFree(q); }
  void *get(size_t bytes) { void *q = nullptr; if (hipMalloc(&q, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr; p.push_back(q); return q; }
};
}  // namespace

int ggs_debug_philox(int32_t device_id, int64_t n, const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
  if (n <= 0 || !ctr || !key || !out || hipSetDevice(device_id) != hipSuccess) return GGS_ERR_BAD_ARG;
  TmpDev t;
  auto *dc = static_cast<uint32_t *>(t.get(n * 16)); auto *dk = static_cast<uint32_t *>(t.get(n * 8)); auto *dou = static_cast<uint32_t *>(t.get(n * 16));
  if (!dc || !dk || !dou) return GGS_ERR_HIP;
  if (hipMemcpy(dc, ctr, n * 16, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dk, key, n * 8, hipMemcpyHostToDevice) != hipSuccess) return GGS_ERR_HIP;
  hipLaunchKernelGGL(debug_philox_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, n, dc, dk, dou);
  if (hipMemcpy(out, dou, n * 16, hipMemcpyDeviceToHost) != hipSuccess) return GGS_ERR_HIP;
  return GGS_OK;
}

int ggs_debug_math(int32_t device_id, int32_t op, int64_t n, const double *x, const double *y, double *out) {
  if (n <= 0 || !x || !out || hipSetDevice(device_id) != hipSuccess) return GGS_ERR_BAD_ARG;
  TmpDev t;
  auto *dx = static_cast<double *>(t.get(n * 8)); auto *dy = static_cast<double *>(t.get(n * 8)); auto *dou = static_cast<double *>(t.get(n * 8));
  if (!dx || !dy || !dou) return GGS_ERR_HIP;
  if (hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice) != hipSuccess) return GGS_ERR_HIP;
  if (hipMemcpy(dy, y ? y : x, n * 8, hipMemcpyHostToDevice) != hipSuccess) return GGS_ERR_HIP;
  hipLaunchKernelGGL(debug_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, op, n, dx, dy, dou);
  if (hipMemcpy(out, dou, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return GGS_ERR_HIP;
  return GGS_OK;
}

int ggs_debug_draw(int32_t device_id, int32_t kind, uint64_t seed, uint32_t iteration, uint32_t purpose, uint64_t elem0, int64_t n,
                   const double *shape, double *out, int32_t *status) {
  if (n <= 0 || !out || (kind == 2 && !shape) || hipSetDevice(device_id) != hipSuccess) return GGS_ERR_BAD_ARG;
  TmpDev t;
  auto *ds = static_cast<double *>(t.get(n * 8)); auto *dou = static_cast<double *>(t.get(n * 8)); auto *dst = static_cast<uint32_t *>(t.get(16));
  if (!ds || !dou || !dst) return GGS_ERR_HIP;
  if (shape && hipMemcpy(ds, shape, n * 8, hipMemcpyHostToDevice) != hipSuccess) return GGS_ERR_HIP;
  if (hipMemset(dst, 0, 16) != hipSuccess) return GGS_ERR_HIP;
  hipLaunchKernelGGL(debug_draw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, kind, seed, iteration, purpose, elem0, n, ds, dou, dst);
  uint32_t st = 0;
  if (hipMemcpy(out, dou, n * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&st, dst, 4, hipMemcpyDeviceToHost) != hipSuccess) return GGS_ERR_HIP;
  if (status) *status = (int32_t)st;
  return GGS_OK;
}

int ggs_debug_column_sum_guided(int32_t device_id, int32_t V, int32_t K, const double *x, const int32_t *counts, double beta, const double *guess,
                                double *out, double *pref_out, int32_t *n_k_out) {
  if (V <= 0 || K <= 0 || (!x == !counts) || !out || hipSetDevice(device_id) != hipSuccess) return GGS_ERR_BAD_ARG;
  TmpDev t;
  const size_t kv = (size_t)V * K;
  ggs_handle tmp;                                     // only the fields launch_column_sum reads
  tmp.K = K; tmp.V = V; tmp.beta = beta; tmp.stream = nullptr; tmp.exact_sum = true;
  tmp.sum_nseg = (V + kSumSegRows - 1) / kSumSegRows;
  const size_t pref_bytes = ((size_t)tmp.sum_nseg + 1) * K * 8;
  tmp.d_sum_pref = static_cast<double *>(t.get(pref_bytes));
  tmp.d_sum_fn = static_cast<double *>(t.get((size_t)tmp.sum_nseg * K * 32));
  void *dsrc = t.get(kv * (x ? 8 : 4));
  auto *dou = static_cast<double *>(t.get((size_t)K * 8));
  auto *dnk = static_cast<int32_t *>(t.get((size_t)K * 4));
  int rc = GGS_OK;
  if (!tmp.d_sum_pref || !tmp.d_sum_fn || !dsrc || !dou || !dnk) rc = GGS_ERR_HIP;
  else if (hipMemcpy(dsrc, x ? (const void *)x : (const void *)counts, kv * (x ? 8 : 4), hipMemcpyHostToDevice) != hipSuccess) rc = GGS_ERR_HIP;
  else if (guess && hipMemcpy(tmp.d_sum_pref, guess, pref_bytes, hipMemcpyHostToDevice) != hipSuccess) rc = GGS_ERR_HIP;
  else {
    if (x) launch_column_sum<double, false>(&tmp, static_cast<const double *>(dsrc), K, K, dou, nullptr, guess != nullptr, true);
    else launch_column_sum<int32_t, true>(&tmp, static_cast<const int32_t *>(dsrc), K, K, dou, dnk, guess != nullptr, true);
    if (hipGetLastError() != hipSuccess || hipMemcpy(out, dou, (size_t)K * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = GGS_ERR_HIP;
    else if (pref_out && hipMemcpy(pref_out, tmp.d_sum_pref, pref_bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = GGS_ERR_HIP;
    else if (n_k_out && counts && hipMemcpy(n_k_out, dnk, (size_t)K * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = GGS_ERR_HIP;
  }
  tmp.d_sum_pref = nullptr; tmp.d_sum_fn = nullptr;   // owned by t
  return rc;
}

int ggs_debug_column_sum(int32_t device_id, int32_t V, int32_t K, const double *x, const int32_t *counts, double beta, double *out) {
  return ggs_debug_column_sum_guided(device_id, V, K, x, counts, beta, nullptr, out, nullptr, nullptr);
}

}  // extern "C"
