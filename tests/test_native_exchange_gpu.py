"""The NATIVE multi-GPU exchange of libggs_hip (include/ggs_hip.h, "multi-GPU"): reduce-scatter of the counts by topic
slice, Phi drawn for the rank's own topics, all-gather of the fp64 slices -- the device form of the reference's merge
(UPLDA:1107-1221) and topic-batched samplePhi (GGS:139-171, EvenSplitTopicBatchBuilder.java:28-39).

One GPU is all these tests have, and RCCL refuses two ranks on one device, so the pieces are covered separately:
  * one rank through the REAL RCCL provider (ncclCommInitRank / ncclReduceScatter / ncclAllGather on the handle's stream)
  * several ranks with the callback provider: handles in threads of one process over an in-memory transport, and two
    real processes over gloo -- the native sweep, its slice-major layout and its slice kernels are the product's
  * the one-process group API (ncclCommInitAll) with one device
Every case must reproduce the one-handle run / the oracle bit for bit."""
import os
import socket
import sys
import threading

import numpy as np
import pytest

from ldagroupedgibbssampler_amd.corpus import Corpus, even_split, random_corpus
from ldagroupedgibbssampler_amd.sharded import TopicSliceLayout, java_lcg_initial_z

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def assert_bit_equal(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    if a.dtype == np.float64:
        a, b = a.view(np.int64), b.view(np.int64)
    assert np.array_equal(a, b), "%s differs in %d of %d places" % (what, int((a != b).sum()), a.size)


def reference_run(oracle, corpus, K, alpha, beta, seed, zseed, sweeps, scheme="ggs", save_mean=False):
    o = oracle.OracleSampler(K, corpus.num_types, alpha, beta, seed, threads=4)
    o.set_scheme(scheme)
    o.set_phi_mean_gating(save_mean, 1, 1)
    o.set_corpus(corpus.doc_ptr, corpus.tokens)
    o.set_z(java_lcg_initial_z(corpus.num_tokens, K, zseed), redraw_phi=True)
    o.sweep(sweeps)
    return o


@pytest.mark.parametrize("scheme,K,V", [("ggs", 20, 700), ("ggs", 100, 700), ("ggs", 200, 700), ("pcgs", 24, 700),
                                        # V >= 1024 (16 segments of 64 rows): the vocabulary travels in TWO halves, the first on the
                                        # communication stream under the draw of the second (this path had no parity test)
                                        ("ggs", 20, 1300), ("ggs", 100, 2100), ("pcgs", 24, 1100), ("ggs", 200, 1029)])
def test_one_rank_through_rccl(native, oracle, scheme, K, V):
    """ncclCommInitRank with one rank: the whole exchange path (own stream, slice-major send buffer, reduce-scatter,
    slice draw, all-gather in one piece or in two halves, repack, lazily gathered counts) must not change a bit."""
    c = random_corpus(240, V, 140, seed=11 + K, empty_every=8)
    flags = (native.FLAG_PCGS if scheme == "pcgs" else 0) | native.FLAG_SAVE_PHI_MEAN | native.FLAG_PARANOID
    h = native.GGSHandle(K, c.num_types, 0.1, 0.01, 99, flags=flags, phi_burn_in=1, phi_mean_thin=1)
    h.attach_rccl(0, 1, native.rccl_unique_id())
    info = h.exchange_info()
    assert {k: info[k] for k in ("rank", "nranks", "k_begin", "k_end")} == {"rank": 0, "nranks": 1, "k_begin": 0, "k_end": K}
    assert info["provider"] == "rccl" and info["comm_nranks"] == 1 and info["comm_rank"] == 0   # read back from the communicator
    h.set_corpus(c.doc_ptr, c.tokens)
    h.set_z(java_lcg_initial_z(c.num_tokens, K, 3), redraw_phi=True)
    h.sweep(2)
    h.sweep_begin()
    h.sweep_end()
    o = reference_run(oracle, c, K, 0.1, 0.01, 99, 3, 3, scheme, save_mean=True)
    assert_bit_equal(h.get_z(), o.get_z(), "z")
    assert_bit_equal(h.get_type_topic_counts(), o.get_type_topic_counts(), "n_wk")
    assert_bit_equal(h.get_topic_totals(), o.get_topic_totals(), "n_k")
    assert_bit_equal(h.get_phi(), o.get_phi(), "phi")
    if scheme == "ggs":
        assert_bit_equal(h.get_theta(), o.get_theta(), "theta")
    gm, gn = h.get_phi_mean()
    om, on = o.get_phi_mean()
    assert gn == on == 2
    assert_bit_equal(gm, om, "phi mean")
    t = h.get_timings()
    assert t["sweeps"] == 3 and t["exchange_ms"] > 0
    with pytest.raises(native.GGSError):
        h.counts_device_ptr()                      # the library merges the counts itself now
    h.close()


def test_callers_own_communicator(native, oracle):
    """ggs_attach_rccl_comm: the communicator is the caller's (here made with ncclCommInitRank through ctypes on the
    process's librccl) and is not destroyed with the handle."""
    import ctypes as C
    from ldagroupedgibbssampler_amd import _lib
    _lib.share_rccl_with_torch()
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so.1")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    import torch
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    c = random_corpus(120, 300, 80, seed=21, empty_every=7)
    K = 30
    h = native.GGSHandle(K, c.num_types, 0.1, 0.01, 5)
    h._chk(h._L.ggs_attach_rccl_comm(h._h, 0, 1, comm))
    h.set_corpus(c.doc_ptr, c.tokens)
    h.set_z(java_lcg_initial_z(c.num_tokens, K, 4), redraw_phi=True)
    h.sweep(2)
    o = reference_run(oracle, c, K, 0.1, 0.01, 5, 4, 2)
    assert_bit_equal(h.get_z(), o.get_z(), "z")
    assert_bit_equal(h.get_phi(), o.get_phi(), "phi")
    h.close()
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclCommDestroy(comm) == 0           # still alive: the library did not destroy what it does not own


class ThreadTransport:
    """In-memory collectives between handles driven by threads of one process."""

    def __init__(self, n):
        self.n, self.bar, self.box = n, threading.Barrier(n, timeout=300), [None] * n

    def exchange(self, rank, payload):
        self.box[rank] = payload
        self.bar.wait()
        got = list(self.box)
        self.bar.wait()
        return got


def _thread_rank(native, tr, rank, world, whole, K, scheme, zseed, sweeps, out, errs):
    import torch
    from ldagroupedgibbssampler_amd.sharded import _DevPtr
    try:
        dev = torch.device("cuda", 0)

        def view(ptr, n, typestr):
            return torch.as_tensor(_DevPtr(ptr, n, typestr), device=dev)

        def reduce_scatter_i32(send, recv, count, stream):
            torch.cuda.synchronize()
            mine = view(send, count * world, "<i4").cpu().numpy().reshape(world, count)
            parts = tr.exchange(rank, mine)
            own = np.sum([p[rank] for p in parts], axis=0, dtype=np.int32)
            view(recv, count, "<i4").copy_(torch.from_numpy(own))
            torch.cuda.synchronize()
            return 0

        def all_gather(typestr):
            def cb(send, recv, count, stream):
                torch.cuda.synchronize()
                parts = tr.exchange(rank, view(send, count, typestr).cpu().numpy())
                view(recv, count * world, typestr).copy_(torch.from_numpy(np.concatenate(parts)))
                torch.cuda.synchronize()
                return 0
            return cb

        bounds = even_split(whole.num_docs, world)
        sub, doc_base, tok_base = whole.shard(bounds[rank], bounds[rank + 1])
        flags = (native.FLAG_PCGS if scheme == "pcgs" else 0) | native.FLAG_SAVE_PHI_MEAN
        h = native.GGSHandle(K, whole.num_types, 0.1, 0.01, 4242, flags=flags, phi_burn_in=1, phi_mean_thin=2)
        h.attach_exchange(rank, world, reduce_scatter_i32, all_gather("<f8"), all_gather("<i4"))
        h.set_corpus(sub.doc_ptr, sub.tokens, doc_base, tok_base)
        h.set_global_token_count(whole.num_tokens)
        z0 = java_lcg_initial_z(whole.num_tokens, K, zseed)
        h.set_z(z0[tok_base:tok_base + sub.num_tokens], redraw_phi=True)
        h.sweep(sweeps - 1)
        h.sweep_begin()
        h.sweep_end()
        h.check_invariants()                       # collective: gathers the corpus-wide counts
        doc_side, topic_side = h.model_log_likelihood()
        out[rank] = dict(z=h.get_z(), nwk=h.get_type_topic_counts(), nk=h.get_topic_totals(), phi=h.get_phi(),
                         theta=h.get_theta() if scheme == "ggs" else None, mean=h.get_phi_mean(), info=h.exchange_info(),
                         ll=(doc_side, topic_side))
        h.close()
    except BaseException as e:                      # noqa: BLE001 -- re-raised by the test body
        errs.append(e)
        tr.bar.abort()


@pytest.mark.parametrize("scheme,K,world,docs,V", [("ggs", 100, 3, 310, 900), ("ggs", 7, 4, 310, 900), ("pcgs", 24, 2, 310, 900), ("ggs", 200, 3, 310, 900),
                                                   ("ggs", 2, 3, 310, 900), ("ggs", 5, 4, 3, 40), ("pcgs", 5, 3, 2, 40),
                                                   # the two-half all-gather (V >= 1024): unequal slices, a rank without a topic, pcgs, a ragged last segment
                                                   ("ggs", 100, 3, 310, 2100), ("ggs", 2, 3, 200, 1100), ("pcgs", 24, 2, 310, 1500), ("ggs", 7, 4, 200, 1025)])
def test_topic_sliced_exchange_between_handles(native, oracle, scheme, K, world, docs, V):
    """`world` doc shards, each a handle with the callback exchange: z, theta per shard and counts, Phi, phi mean on
    every rank equal the unsharded oracle.  K = 100 over 3 ranks has unequal slices (34, 33, 33); K = 2 over 3 leaves
    rank 2 without a topic; 3 documents over 4 ranks (2 over 3) leave a rank without a document; V >= 1024 sends the
    gammas in two halves (the first on the communication stream under the draw of the second)."""
    whole = random_corpus(docs, V, 120 if docs > 10 else 25, seed=5 + K, empty_every=9 if docs > 10 else 0)
    sweeps = 4
    tr, out, errs = ThreadTransport(world), [None] * world, []
    ts = [threading.Thread(target=_thread_rank, args=(native, tr, r, world, whole, K, scheme, 17, sweeps, out, errs)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    o = reference_run(oracle, whole, K, 0.1, 0.01, 4242, 17, sweeps, scheme, save_mean=False)
    o2 = oracle.OracleSampler(K, whole.num_types, 0.1, 0.01, 4242, threads=4)
    o2.set_scheme(scheme)
    o2.set_phi_mean_gating(True, 1, 2)
    o2.set_corpus(whole.doc_ptr, whole.tokens)
    o2.set_z(java_lcg_initial_z(whole.num_tokens, K, 17), redraw_phi=True)
    o2.sweep(sweeps)
    om, on = o2.get_phi_mean()
    lay = TopicSliceLayout(K, whole.num_types, world)
    assert_bit_equal(np.concatenate([p["z"] for p in out]), o.get_z(), "z")
    if scheme == "ggs":
        assert_bit_equal(np.concatenate([p["theta"] for p in out]), o.get_theta(), "theta")
    ll_docs = 0.0
    for r, p in enumerate(out):
        a, b = lay.slice_of(r)
        assert {k: p["info"][k] for k in ("rank", "nranks", "k_begin", "k_end")} == {"rank": r, "nranks": world, "k_begin": a, "k_end": b}
        assert p["info"]["provider"] == "callbacks" and p["info"]["comm_nranks"] == world
        assert_bit_equal(p["nwk"], o.get_type_topic_counts(), "n_wk on rank %d" % r)
        assert_bit_equal(p["nk"], o.get_topic_totals(), "n_k on rank %d" % r)
        assert_bit_equal(p["phi"], o.get_phi(), "phi on rank %d" % r)
        assert p["mean"][1] == on
        assert_bit_equal(p["mean"][0], om, "phi mean on rank %d" % r)
        assert p["ll"][1] == out[0]["ll"][1]       # the topic side is computed from identical counts everywhere
        ll_docs += p["ll"][0]
    ref_ll = sum(o.model_log_likelihood())
    assert abs((ll_docs + out[0]["ll"][1]) - ref_ll) <= 1e-9 * abs(ref_ll)


@pytest.mark.parametrize("K,world,V,zcounts", [(100, 3, 2100, "2"), (37, 2, 900, "0")])
def test_topic_sliced_exchange_with_warm_tiers(native, oracle, monkeypatch, K, world, V, zcounts):
    """The same with the warm tiers forced onto the small shards (every handle of the process reads the knobs in
    ggs_set_corpus): what a rank of 2 or 4 of the benchmark corpus runs by itself."""
    for k, v in WARM_TIERS_ON_A_SMALL_CORPUS.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("GGS_DEBUG_ZCOUNTS", zcounts)
    test_topic_sliced_exchange_between_handles(native, oracle, "ggs", K, world, 310, V)


def _process_rank(rank, world, port, out_dir, scheme, K, count_exchange="auto"):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ldagroupedgibbssampler_amd import native
    from ldagroupedgibbssampler_amd.sharded import ShardedGGS, gloo_callback_exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    whole = random_corpus(200, 400, 110, seed=77, empty_every=9)
    h = native.GGSHandle(K, whole.num_types, 0.1, 0.01, 777, device_id=0, flags=native.FLAG_PCGS if scheme == "pcgs" else 0)
    if count_exchange != "auto":
        h.set_count_exchange(count_exchange)
    sh = ShardedGGS(h, gloo_callback_exchange(rank, world), whole, rank, world)
    assert h.count_exchange()["sparse"] == (count_exchange == "sparse")
    sh.set_z_global(java_lcg_initial_z(whole.num_tokens, K, 5))
    sh.sweep(2)
    sh.sweep(1)
    sh.set_test_corpus(random_corpus(30, 400, 60, seed=6, empty_every=5))
    ho_total, ho_docs = sh.heldout_log_likelihood(40)        # the estimator reads the (gathered) corpus-wide counts
    h.check_invariants()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), z=h.get_z(), nwk=h.get_type_topic_counts(), phi=h.get_phi(), ho_total=ho_total,
             ho_docs=ho_docs)
    h.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("scheme,count_exchange", [("ggs", "auto"), ("pcgs", "auto"), ("ggs", "sparse")])
def test_two_processes_native_exchange_over_gloo(oracle, tmp_path, scheme, count_exchange):
    """Two real processes, each with its own HIP handle, joined by the callback exchange over gloo: what bench.py runs
    at --gpus 2 with the RCCL provider swapped for host staging -- also with the counts travelling as (cell, count) pairs
    (gloo point-to-point sends behind all_to_all_v_i32)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world, K = 2, 33
    mp.spawn(_process_rank, args=(world, port, str(tmp_path), scheme, K, count_exchange), nprocs=world, join=True)
    whole = random_corpus(200, 400, 110, seed=77, empty_every=9)
    o = reference_run(oracle, whole, K, 0.1, 0.01, 777, 5, 3, scheme)
    t = random_corpus(30, 400, 60, seed=6, empty_every=5)
    ho_total, ho_docs = o.heldout_log_likelihood(t.doc_ptr, t.tokens, 40)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert_bit_equal(np.concatenate([p["z"] for p in parts]), o.get_z(), "z")
    for p in parts:
        assert_bit_equal(p["nwk"], o.get_type_topic_counts(), "n_wk")
        assert_bit_equal(p["phi"], o.get_phi(), "phi")
        assert float(p["ho_total"]) == ho_total
        assert_bit_equal(p["ho_docs"], ho_docs, "held-out per document")


@pytest.mark.parametrize("V", [300, 1200])
def test_one_process_group_api(native, oracle, V):
    """ggs_group_create / ggs_group_set_z / ggs_group_sweep with the one device there is (ncclCommInitAll, every
    collective inside ncclGroupStart/End); V = 1200: the all-gather in two halves, grouped step by step."""
    c = random_corpus(150, V, 90, seed=3, empty_every=6)
    K = 40
    g = native.GGSGroup(K, c.num_types, 0.1, 0.01, 31337, device_ids=[0])
    h = g.handles[0]
    h.set_corpus(c.doc_ptr, c.tokens)
    g.set_z([java_lcg_initial_z(c.num_tokens, K, 9)], redraw_phi=True)
    g.sweep(3)
    o = reference_run(oracle, c, K, 0.1, 0.01, 31337, 9, 3)
    assert_bit_equal(h.get_z(), o.get_z(), "z")
    assert_bit_equal(h.get_phi(), o.get_phi(), "phi")
    assert_bit_equal(h.get_type_topic_counts(), o.get_type_topic_counts(), "n_wk")
    assert h.get_timings()["sweeps"] == 3
    g.close()
    # scheme=collapsed through the group: the merge is the grouped gather of the count slices
    g = native.GGSGroup(K, c.num_types, 0.1, 0.01, 31337, device_ids=[0], flags=native.FLAG_COLLAPSED)
    h = g.handles[0]
    h.set_corpus(c.doc_ptr, c.tokens)
    g.set_z([java_lcg_initial_z(c.num_tokens, K, 9)], redraw_phi=True)
    g.sweep(2)
    g.gather_counts()
    o = oracle.OracleSampler(K, c.num_types, 0.1, 0.01, 31337, threads=2)
    o.set_corpus(c.doc_ptr, c.tokens)
    o.set_z(java_lcg_initial_z(c.num_tokens, K, 9), redraw_phi=True)
    o.collapsed_parallel_sweep(2)
    assert_bit_equal(h.get_z(), o.get_z(), "collapsed group z")
    assert_bit_equal(h.get_type_topic_counts(), o.get_type_topic_counts(), "collapsed group n_wk")
    g.close()


class OneThreadTransport:
    """A transport that completes a collective only when ALL n ranks have called it -- and is driven from ONE thread, the
    way a JVM drives ggs_group_*: a call of rank i returns at once (nothing moved yet), the n-th call moves the data for
    everybody.  A library that let one handle run ahead to a later step while another had not yet made this one would
    find its step mismatched here (asserted), or wait for ever in a real transport."""

    def __init__(self, n):
        import torch
        self.torch, self.n, self.dev = torch, n, torch.device("cuda", 0)
        self.pending, self.kind, self.log = {}, None, []

    def _view(self, ptr, count, typestr):
        from ldagroupedgibbssampler_amd.sharded import _DevPtr
        return self.torch.as_tensor(_DevPtr(ptr, count, typestr), device=self.dev)

    def _arrive(self, kind, rank, send, recv, count):
        assert self.kind in (None, kind), "rank %d calls %s while %s is still open" % (rank, kind, self.kind)
        assert rank not in self.pending, "rank %d calls %s twice before its peers called it once" % (rank, kind)
        assert rank == len(self.pending), "handles are issued in rank order"
        self.kind = kind
        self.pending[rank] = (send, recv, count)
        if len(self.pending) < self.n:
            return 0
        self.torch.cuda.synchronize()
        counts = {c for _, _, c in self.pending.values()}
        assert len(counts) == 1
        count = counts.pop()
        typestr = "<f8" if kind == "ag64" else "<i4"
        if kind == "rs":
            total = sum(self._view(s_, count * self.n, typestr).clone() for s_, _, _ in self.pending.values())
            for r, (_, recv_, _) in self.pending.items():
                self._view(recv_, count, typestr).copy_(total[r * count:(r + 1) * count])
        else:
            parts = self.torch.cat([self._view(s_, count, typestr).clone() for _, (s_, _, _) in sorted(self.pending.items())])
            for _, recv_, _ in self.pending.values():
                self._view(recv_, count * self.n, typestr).copy_(parts)
        self.torch.cuda.synchronize()
        self.log.append(kind)
        self.pending, self.kind = {}, None
        return 0

    def callbacks(self, rank):
        return (lambda s_, r_, c, st: self._arrive("rs", rank, s_, r_, c), lambda s_, r_, c, st: self._arrive("ag64", rank, s_, r_, c),
                lambda s_, r_, c, st: self._arrive("ag32", rank, s_, r_, c))


@pytest.mark.parametrize("scheme,V", [("ggs", 260), ("pcgs", 260), ("collapsed", 260), ("ggs", 1100), ("pcgs", 1290)])
def test_group_entry_points_from_one_thread_over_a_deferred_transport(native, oracle, scheme, V):
    """ggs_group_adopt + ggs_group_set_z / ggs_group_sweep / ggs_group_gather_counts with TWO handles driven from one
    thread -- the call order a JVM would use -- over a transport that moves data only when both handles have made the
    call: every collective step is issued for all handles before any handle goes on, so nothing waits on a call that
    the same thread has yet to make.  Results: the one-handle run's, bit for bit."""
    c = random_corpus(170, V, 70, seed=12, empty_every=7)
    K, n = 13, 2
    flags = {"ggs": 0, "pcgs": native.FLAG_PCGS, "collapsed": native.FLAG_COLLAPSED}[scheme]
    tr = OneThreadTransport(n)
    bounds = even_split(c.num_docs, n)
    z0 = java_lcg_initial_z(c.num_tokens, K, 5)
    hs, zs = [], []
    for r in range(n):
        h = native.GGSHandle(K, c.num_types, 0.1, 0.01, 606, flags=flags)
        h.attach_exchange(r, n, *tr.callbacks(r))
        sub, doc_base, tok_base = c.shard(bounds[r], bounds[r + 1])
        h.set_corpus(sub.doc_ptr, sub.tokens, doc_base, tok_base)
        h.set_global_token_count(c.num_tokens)
        zs.append(z0[tok_base:tok_base + sub.num_tokens])
        hs.append(h)
    g = native.GGSGroup.adopt(hs)
    g.set_z(zs, redraw_phi=True)
    g.sweep(3)
    g.gather_counts()
    if scheme == "collapsed":
        o = oracle.OracleSampler(K, c.num_types, 0.1, 0.01, 606, threads=2)
        o.set_corpus(c.doc_ptr, c.tokens)
        o.set_z(z0, redraw_phi=True)
        # the doc-sharded parallel schedule is AD-LDA with a merge per sweep: the single-handle restatement of the same
        # schedule sees the same sweep-start counts
        o.collapsed_parallel_sweep(3)
    else:
        o = reference_run(oracle, c, K, 0.1, 0.01, 606, 5, 3, scheme)
    assert_bit_equal(np.concatenate([h.get_z() for h in hs]), o.get_z(), scheme + " z")
    for h in hs:
        assert_bit_equal(h.get_type_topic_counts(), o.get_type_topic_counts(), scheme + " n_wk")
        if scheme != "collapsed":
            assert_bit_equal(h.get_phi(), o.get_phi(), scheme + " phi")
    assert tr.log.count("rs") >= 4 and not tr.pending
    g.close()
    for h in hs:
        h.close()


@pytest.mark.parametrize("split", [1, 3, 6])
def test_forced_split_point_of_the_all_gather(native, oracle, monkeypatch, split):
    """GGS_DEBUG_AGSPLIT=k: the first k 64-row segments travel as the first half whatever V is -- the split point must
    not matter (one segment, an uneven cut, all but one)."""
    monkeypatch.setenv("GGS_DEBUG_AGSPLIT", str(split))
    c = random_corpus(120, 440, 90, seed=31, empty_every=7)       # 7 segments, the last one ragged
    K = 17
    h = native.GGSHandle(K, c.num_types, 0.1, 0.01, 12, flags=native.FLAG_SAVE_PHI_MEAN, phi_burn_in=1, phi_mean_thin=1)
    h.attach_rccl(0, 1, native.rccl_unique_id())
    h.set_corpus(c.doc_ptr, c.tokens)
    h.set_z(java_lcg_initial_z(c.num_tokens, K, 8), redraw_phi=True)
    h.sweep(3)
    o = reference_run(oracle, c, K, 0.1, 0.01, 12, 8, 3, save_mean=True)
    assert_bit_equal(h.get_z(), o.get_z(), "z")
    assert_bit_equal(h.get_phi(), o.get_phi(), "phi")
    gm, gn = h.get_phi_mean()
    om, on = o.get_phi_mean()
    assert gn == on
    assert_bit_equal(gm, om, "phi mean")
    h.close()


def _sparse_rank(native, tr, rank, world, whole, K, scheme, mode, sweeps, out, errs):
    import torch
    from ldagroupedgibbssampler_amd.sharded import _DevPtr
    try:
        dev = torch.device("cuda", 0)

        def view(ptr, n, typestr):
            return torch.as_tensor(_DevPtr(ptr, n, typestr), device=dev)

        def reduce_scatter_i32(send, recv, count, stream):
            torch.cuda.synchronize()
            parts = tr.exchange(rank, view(send, count * world, "<i4").cpu().numpy().reshape(world, count))
            view(recv, count, "<i4").copy_(torch.from_numpy(np.sum([p[rank] for p in parts], axis=0, dtype=np.int32)))
            torch.cuda.synchronize()
            out[rank]["dense_calls"] += 1
            return 0

        def all_gather(typestr):
            def cb(send, recv, count, stream):
                torch.cuda.synchronize()
                parts = tr.exchange(rank, view(send, count, typestr).cpu().numpy())
                view(recv, count * world, typestr).copy_(torch.from_numpy(np.concatenate(parts)))
                torch.cuda.synchronize()
                return 0
            return cb

        def all_to_all_v(send, soff, scnt, recv, roff, rcnt, stream):
            torch.cuda.synchronize()
            total = max(soff[i] + scnt[i] for i in range(world))
            mine = view(send, max(total, 1), "<i4").cpu().numpy()
            blocks = [mine[soff[d]:soff[d] + scnt[d]].copy() for d in range(world)]
            everyone = tr.exchange(rank, blocks)                   # everyone[s][d] = what rank s addresses to rank d
            for s_ in range(world):
                got = everyone[s_][rank]
                assert got.size == rcnt[s_], "rank %d expected %d elements from rank %d, got %d" % (rank, rcnt[s_], s_, got.size)
                if got.size:
                    view(recv + 4 * roff[s_], got.size, "<i4").copy_(torch.from_numpy(got))
            torch.cuda.synchronize()
            out[rank]["sparse_calls"] += 1
            out[rank]["pairs_sent"] += sum(scnt) // 2
            return 0

        bounds = even_split(whole.num_docs, world)
        sub, doc_base, tok_base = whole.shard(bounds[rank], bounds[rank + 1])
        flags = native.FLAG_PCGS if scheme == "pcgs" else 0
        h = native.GGSHandle(K, whole.num_types, 0.1, 0.01, 515, flags=flags)
        h.attach_exchange(rank, world, reduce_scatter_i32, all_gather("<f8"), all_gather("<i4"), all_to_all_v)
        h.set_count_exchange(mode)
        h.set_corpus(sub.doc_ptr, sub.tokens, doc_base, tok_base)
        h.set_global_token_count(whole.num_tokens)
        z0 = java_lcg_initial_z(whole.num_tokens, K, 23)
        h.set_z(z0[tok_base:tok_base + sub.num_tokens], redraw_phi=True)
        h.sweep(sweeps - 1)
        h.sweep_begin()
        h.sweep_end()
        h.check_invariants()
        out[rank].update(z=h.get_z(), nwk=h.get_type_topic_counts(), nk=h.get_topic_totals(), phi=h.get_phi(), how=h.count_exchange())
        h.close()
    except BaseException as e:                      # noqa: BLE001
        errs.append(e)
        tr.bar.abort()


@pytest.mark.parametrize("scheme,K,world,V", [("ggs", 100, 3, 900), ("ggs", 200, 3, 1500), ("pcgs", 24, 2, 700), ("ggs", 2, 3, 300), ("ggs", 37, 4, 1100)])
def test_sparse_count_exchange_equals_the_dense_one(native, oracle, scheme, K, world, V):
    """ggs_set_count_exchange: the counts as (cell, count) pairs of the non-zero cells over all_to_all_v_i32 (what BASELINE
    config 5 asks for: a shard's histogram there is 3 % dense) against the dense reduce-scatter and against the unsharded
    oracle: the same integers, hence the same Phi and the same z.  Unequal slices, a rank without a topic, pcgs, the
    score-register and the streaming z kernels."""
    whole = random_corpus(240, V, 110, seed=3 + K, empty_every=9)
    sweeps = 3
    res = {}
    for mode in ("sparse", "dense"):
        tr, errs = ThreadTransport(world), []
        out = [dict(dense_calls=0, sparse_calls=0, pairs_sent=0) for _ in range(world)]
        ts = [threading.Thread(target=_sparse_rank, args=(native, tr, r, world, whole, K, scheme, mode, sweeps, out, errs)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]
        res[mode] = out
    o = reference_run(oracle, whole, K, 0.1, 0.01, 515, 23, sweeps, scheme)
    for mode, out in res.items():
        assert_bit_equal(np.concatenate([p["z"] for p in out]), o.get_z(), mode + " z")
        for r, p in enumerate(out):
            assert_bit_equal(p["nwk"], o.get_type_topic_counts(), "%s n_wk on rank %d" % (mode, r))
            assert_bit_equal(p["nk"], o.get_topic_totals(), "%s n_k on rank %d" % (mode, r))
            assert_bit_equal(p["phi"], o.get_phi(), "%s phi on rank %d" % (mode, r))
            assert p["how"]["sparse"] == (mode == "sparse")
            if mode == "sparse":
                assert p["dense_calls"] == 0 and p["sparse_calls"] >= sweeps + 1 and 0 < p["pairs_sent"]
                assert p["how"]["pairs_last"] <= p["how"]["dense_cells"]
            else:
                assert p["sparse_calls"] == 0 and p["dense_calls"] >= sweeps + 1


def test_sparse_count_exchange_through_rccl_with_one_rank(native, oracle):
    """The RCCL provider's all_to_all_v_i32 (ncclSend/ncclRecv in a group; one rank: the own block) and the host round
    trips of the sparse form, forced on a small corpus; also switching the form between two sweeps."""
    c = random_corpus(200, 1300, 120, seed=19, empty_every=8)
    K = 29
    h = native.GGSHandle(K, c.num_types, 0.1, 0.01, 4, flags=native.FLAG_PARANOID)
    h.attach_rccl(0, 1, native.rccl_unique_id())
    h.set_count_exchange("sparse")
    h.set_corpus(c.doc_ptr, c.tokens)
    h.set_z(java_lcg_initial_z(c.num_tokens, K, 2), redraw_phi=True)
    assert h.count_exchange()["sparse"]
    h.sweep(2)
    assert 0 < h.count_exchange()["pairs_last"] <= c.num_tokens
    h.set_count_exchange("dense")
    h.sweep(1)
    h.set_count_exchange("sparse")
    h.sweep(1)
    o = reference_run(oracle, c, K, 0.1, 0.01, 4, 2, 4)
    assert_bit_equal(h.get_z(), o.get_z(), "z")
    assert_bit_equal(h.get_type_topic_counts(), o.get_type_topic_counts(), "n_wk")
    assert_bit_equal(h.get_phi(), o.get_phi(), "phi")
    h.close()


WARM_TIERS_ON_A_SMALL_CORPUS = {"GGS_DEBUG": "1", "GGS_DEBUG_WARM": "8", "GGS_DEBUG_WARM_ROWS": "16", "GGS_DEBUG_WARM_FILL": "1", "GGS_DEBUG_WARM_CPW": "0",
                               "GGS_DEBUG_HOT": "8"}


@pytest.mark.parametrize("zcounts,every,warm", [("0", "1", False), ("2", "1", False), ("2", "3", False), ("0", "4", False), ("2", "1", True), ("0", "3", True)])
def test_who_counts_and_how_often_the_phases_are_timed(native, oracle, monkeypatch, zcounts, every, warm):
    """With an exchange the score-register z kernels add the cold tokens' (word, topic) cells into the send buffer themselves
    and only the hot words' segments go through count_sorted_kernel (GGS_DEBUG_ZCOUNTS=2: whatever the corpus; =0: the count
    kernel alone, the cross-check); and only one sweep in GGS_DEBUG_TIMING_EVERY records every phase event.  Same bits, and
    the timers still add up.  `warm`: with warm tiers (z_warm_kernel; a rank of two or four of the benchmark corpus keeps
    them) -- a warm token adds its own cell like a cold one."""
    monkeypatch.setenv("GGS_DEBUG_ZCOUNTS", zcounts)
    monkeypatch.setenv("GGS_DEBUG_TIMING_EVERY", every)
    if warm:
        for k, v in WARM_TIERS_ON_A_SMALL_CORPUS.items():
            monkeypatch.setenv(k, v)
    c = random_corpus(260, 1300, 150, seed=41, empty_every=8)
    K = 37
    h = native.GGSHandle(K, c.num_types, 0.1, 0.01, 77, flags=native.FLAG_PARANOID)
    h.attach_rccl(0, 1, native.rccl_unique_id())
    h.set_corpus(c.doc_ptr, c.tokens)
    h.set_z(java_lcg_initial_z(c.num_tokens, K, 6), redraw_phi=True)
    h.sweep(5)
    h.sweep_begin()
    h.sweep_end()
    h.sweep(3)
    o = reference_run(oracle, c, K, 0.1, 0.01, 77, 6, 9)
    assert_bit_equal(h.get_z(), o.get_z(), "z")
    assert_bit_equal(h.get_type_topic_counts(), o.get_type_topic_counts(), "n_wk")
    assert_bit_equal(h.get_phi(), o.get_phi(), "phi")
    assert_bit_equal(h.get_theta(), o.get_theta(), "theta")
    if warm:
        assert h.launch_info()["warm_tiers"] >= 2
    t = h.get_timings()
    assert t["sweeps"] == 9 and t["z_ms"] > 0 and t["phi_ms"] > 0 and t["exchange_ms"] > 0
    assert abs(t["exchange_ms"] - (t["exchange_rs_ms"] + t["exchange_ag_ms"])) <= 1e-6 * max(t["exchange_ms"], 1.0)
    h.close()


def test_attach_order_and_errors(native):
    c = random_corpus(20, 50, 30, seed=1)
    h = native.GGSHandle(5, c.num_types, 0.1, 0.01, 1)
    h.set_corpus(c.doc_ptr, c.tokens)
    with pytest.raises(native.GGSError) as e:
        h.attach_null_exchange(0, 2)               # after the corpus: refused
    assert e.value.code == native.ERR_STATE
    h.close()
    h = native.GGSHandle(5, c.num_types, 0.1, 0.01, 1)
    with pytest.raises(native.GGSError) as e:
        h.attach_null_exchange(2, 2)
    assert e.value.code == native.ERR_BAD_ARG
    h.close()
