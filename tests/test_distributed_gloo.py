"""world_size-2 test of the doc-sharded orchestration (ldagroupedgibbssampler_amd.sharded)
over gloo on CPU.

The product engine is the HIP handle, which needs a GPU; here the TEST injects an oracle-backed
engine so that exactly the product's sharding / start-up / per-sweep exchange logic runs in two
real processes with a real torch.distributed all-reduce, and must reproduce the unsharded
oracle bit for bit.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """GGSHandle's method names on top of the oracle, with the same split-sweep contract:
    after sweep_begin / set_z(redraw=False) the exchange buffer holds THIS shard's
    contribution; after the all-reduce, sweep_end / init_phi proceed on corpus-wide counts."""

    def __init__(self, oracle, K, V, alpha, beta, seed, scheme="ggs"):
        self.o = oracle.OracleSampler(K, V, alpha, beta, seed)
        self.o.set_scheme(scheme)
        self.K, self.V = K, V
        self.buf = np.zeros((V, K), np.int32)        # what the exchange all-reduces
        self._mode = None

    def set_corpus(self, doc_ptr, tokens, doc_base=0, tok_base=0):
        self.o.set_corpus(doc_ptr, tokens, doc_base, tok_base)

    def set_global_token_count(self, n):
        self.global_tokens = n

    def set_z(self, z, redraw_phi=True):
        self.o.set_z(z, redraw_phi)
        self.buf[...] = self.o.get_type_topic_counts()
        self._local_counts = self.buf.copy()
        self._mode = "startup"

    def init_phi(self):
        # counts := all-reduced counts (add what the other shards contributed)
        self.o.add_delta(self.buf - self._local_counts)
        self.o.update_counts()
        self.o.init_phi()

    def sweep_begin(self):
        self.o.set_iteration(self.o.iteration + 1)
        self.o.z_step()
        self.buf[...] = self.o.get_delta()
        self._local_delta = self.buf.copy()
        self._mode = "sweep"

    def sweep_end(self):
        self.o.add_delta(self.buf - self._local_delta)
        self.o.update_counts()
        self.o.sample_phi()

    def set_test_corpus(self, doc_ptr, tokens, doc_base=0):
        self._test = (doc_ptr, tokens, doc_base)

    def heldout_log_likelihood(self, num_particles=100):
        return self.o.heldout_log_likelihood(self._test[0], self._test[1], num_particles, self._test[2])


class GlooExchange:
    def __init__(self, engine):
        import torch
        import torch.distributed as dist
        self.dist, self.t = dist, torch.from_numpy(engine.buf)   # shares memory with engine.buf

    def allreduce_sweep(self):
        self.dist.all_reduce(self.t, op=self.dist.ReduceOp.SUM)

    allreduce_startup = allreduce_sweep


def _worker(rank, world, port, out_dir, scheme):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ldagroupedgibbssampler_amd.corpus import random_corpus
    from ldagroupedgibbssampler_amd.sharded import ShardedGGS, java_lcg_initial_z
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = random_corpus(157, 120, 60, seed=17, empty_every=10)
    K, alpha, beta, seed = 9, 0.1, 0.01, 4242
    eng = OracleEngine(O, K, c.num_types, alpha, beta, seed, scheme)
    sh = ShardedGGS(eng, GlooExchange, c, rank, world)
    sh.set_z_global(java_lcg_initial_z(c.num_tokens, K, 77))
    sh.sweep(3)
    sh.set_test_corpus(random_corpus(23, 120, 40, seed=5, empty_every=6))
    ho_total, ho_docs = sh.heldout_log_likelihood(30)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), z=eng.o.get_z(), nwk=eng.o.get_type_topic_counts(), phi=eng.o.get_phi(),
             theta=eng.o.get_theta(), tok_base=sh.tok_base, doc_base=sh.doc_base, ho_total=ho_total, ho_docs=ho_docs)
    dist.destroy_process_group()


@pytest.mark.parametrize("scheme", ["ggs", "pcgs"])
def test_two_rank_sharded_sweep_equals_unsharded(oracle, tmp_path, scheme):
    import torch.multiprocessing as mp
    from ldagroupedgibbssampler_amd.corpus import even_split, random_corpus
    from ldagroupedgibbssampler_amd.sharded import java_lcg_initial_z
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    mp.spawn(_worker, args=(world, port, str(tmp_path), scheme), nprocs=world, join=True)

    c = random_corpus(157, 120, 60, seed=17, empty_every=10)
    K = 9
    ref = oracle.OracleSampler(K, c.num_types, 0.1, 0.01, 4242)
    ref.set_scheme(scheme)
    ref.set_corpus(c.doc_ptr, c.tokens)
    ref.set_z(java_lcg_initial_z(c.num_tokens, K, 77), redraw_phi=True)
    ref.sweep(3)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    bounds = even_split(c.num_docs, world)
    assert [int(p["doc_base"]) for p in parts] == bounds[:-1]
    assert np.array_equal(np.concatenate([p["z"] for p in parts]), ref.get_z())
    if scheme == "ggs":
        assert np.array_equal(np.concatenate([p["theta"] for p in parts]).view(np.int64), ref.get_theta().view(np.int64))
    t = random_corpus(23, 120, 40, seed=5, empty_every=6)
    ho_total, ho_docs = ref.heldout_log_likelihood(t.doc_ptr, t.tokens, 30)
    for p in parts:
        assert np.array_equal(p["nwk"], ref.get_type_topic_counts())
        assert np.array_equal(p["phi"].view(np.int64), ref.get_phi().view(np.int64))
        assert float(p["ho_total"]) == ho_total and np.array_equal(p["ho_docs"], ho_docs)   # sharded test set, same estimate


def _local_shards(world):
    from ldagroupedgibbssampler_amd.corpus import random_corpus
    return [random_corpus(40 + 23 * r, 120, 50, seed=100 + r, empty_every=7) for r in range(world)]


def _worker_local(rank, world, port, out_dir):
    """Every rank brings its OWN documents (bench.py --scaling weak): ShardedGGS.from_local_shard."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ldagroupedgibbssampler_amd.sharded import ShardedGGS, gather_shard_sizes, java_lcg_initial_z_slice
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = _local_shards(world)[rank]
    sizes = gather_shard_sizes(c, rank, world)                         # the helpers bench.py --scaling weak uses
    K = 9
    eng = OracleEngine(O, K, c.num_types, 0.1, 0.01, 4242)
    sh = ShardedGGS.from_local_shard(eng, GlooExchange, c, sizes, rank, world)
    sh.set_z_local(java_lcg_initial_z_slice(sh.tok_base, c.num_tokens, K, 77))
    sh.sweep(2)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), z=eng.o.get_z(), nwk=eng.o.get_type_topic_counts(), phi=eng.o.get_phi(),
             theta=eng.o.get_theta(), global_tokens=sh.global_tokens)
    dist.destroy_process_group()


def test_two_ranks_with_own_shards_equal_the_concatenated_corpus(oracle, tmp_path):
    import torch.multiprocessing as mp
    from ldagroupedgibbssampler_amd.corpus import Corpus
    from ldagroupedgibbssampler_amd.sharded import java_lcg_initial_z
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    mp.spawn(_worker_local, args=(world, port, str(tmp_path)), nprocs=world, join=True)

    shards = _local_shards(world)
    ptr = [np.zeros(1, np.int64)]
    for c in shards:
        ptr.append(c.doc_ptr[1:] + ptr[-1][-1])
    whole = Corpus(np.concatenate(ptr), np.concatenate([c.tokens for c in shards]), 120)
    K = 9
    ref = oracle.OracleSampler(K, 120, 0.1, 0.01, 4242)
    ref.set_corpus(whole.doc_ptr, whole.tokens)
    ref.set_z(java_lcg_initial_z(whole.num_tokens, K, 77), redraw_phi=True)
    ref.sweep(2)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert all(int(p["global_tokens"]) == whole.num_tokens for p in parts)
    assert np.array_equal(np.concatenate([p["z"] for p in parts]), ref.get_z())
    assert np.array_equal(np.concatenate([p["theta"] for p in parts]).view(np.int64), ref.get_theta().view(np.int64))
    for p in parts:
        assert np.array_equal(p["nwk"], ref.get_type_topic_counts())
        assert np.array_equal(p["phi"].view(np.int64), ref.get_phi().view(np.int64))


def test_from_local_shard_rejects_wrong_sizes(oracle):
    from ldagroupedgibbssampler_amd.sharded import ShardedGGS
    c = _local_shards(1)[0]
    eng = OracleEngine(oracle, 4, c.num_types, 0.1, 0.01, 1)
    with pytest.raises(ValueError):
        ShardedGGS.from_local_shard(eng, lambda e: None, c, [(c.num_docs, c.num_tokens + 1)], 0, 1)


def test_even_split_rule():
    """randomscan/document/EvenSplitBatchBuilder.java:30-44: sizes n//p + (remainder > b)."""
    from ldagroupedgibbssampler_amd.corpus import even_split
    assert even_split(10, 3) == [0, 4, 7, 10]
    assert even_split(9, 3) == [0, 3, 6, 9]
    assert even_split(2, 4) == [0, 1, 2, 2, 2]
    assert even_split(100000, 8)[-1] == 100000
    b = even_split(157, 8)
    sizes = np.diff(b)
    assert sizes.max() - sizes.min() <= 1 and sizes.sum() == 157 and list(sizes) == sorted(sizes, reverse=True)


def test_topic_slice_layout_rule():
    """randomscan/topic/EvenSplitTopicBatchBuilder.java:28-39 with one batch per rank, and the slice-major pack/unpack
    the native exchange uses for its reduce-scatter / all-gather buffers."""
    from ldagroupedgibbssampler_amd.sharded import TopicSliceLayout
    lay = TopicSliceLayout(100, 7, 8)
    assert [lay.slice_of(r) for r in range(8)] == [(0, 13), (13, 26), (26, 39), (39, 52), (52, 64), (64, 76), (76, 88), (88, 100)]
    assert lay.Ksm == 13
    lay = TopicSliceLayout(2, 5, 3)
    assert [lay.slice_of(r) for r in range(3)] == [(0, 1), (1, 2), (2, 2)] and lay.Ksm == 1
    lay = TopicSliceLayout(10, 6, 3)
    m = np.arange(60, dtype=np.int32).reshape(6, 10)
    p = lay.pack(m)
    assert p.shape == (3, 6, 4) and np.array_equal(p[0, :, :4], m[:, 0:4]) and np.array_equal(p[1, :, :3], m[:, 4:7]) and (p[1, :, 3] == 0).all()
    assert np.array_equal(lay.unpack(p), m)


class OracleSlicedEngine:
    """The NATIVE exchange's protocol (include/ggs_hip.h, "multi-GPU") restated over the oracle: per sweep the ranks
    reduce-scatter their local (word, topic) histograms by topic slice, each draws Phi for its own topic batch only
    (loopOverTopics, GGS:182-198, on the batch EvenSplitTopicBatchBuilder would hand it) and the slices are
    all-gathered.  `transport` supplies reduce_scatter / all_gather of numpy arrays (gloo here)."""

    def __init__(self, oracle, K, V, alpha, beta, seed, rank, world, transport, scheme="ggs", sparse=False):
        from ldagroupedgibbssampler_amd.sharded import TopicSliceLayout
        self.sparse = sparse
        self.o = oracle.OracleSampler(K, V, alpha, beta, seed)
        self.o.set_scheme(scheme)
        self.K, self.V, self.rank = K, V, rank
        self.lay, self.tr = TopicSliceLayout(K, V, world), transport
        self.k0, self.k1 = self.lay.slice_of(rank)

    def set_corpus(self, doc_ptr, tokens, doc_base=0, tok_base=0):
        self.tokens = np.asarray(tokens)
        self.o.set_corpus(doc_ptr, tokens, doc_base, tok_base)

    def set_global_token_count(self, n):
        self.global_tokens = n

    def _local_histogram(self):
        h = np.zeros((self.V, self.K), np.int32)
        np.add.at(h, (self.tokens, self.o.get_z()), 1)
        return h

    def _exchange_counts(self):
        if self.sparse:       # ggs_set_count_exchange: the non-zero cells as (cell, count) pairs, all-to-all, added up on arrival
            own = self.lay.from_pairs(self.tr.all_to_all_v(self.lay.pairs(self._local_histogram())))
        else:
            own = self.tr.reduce_scatter(self.lay.pack(self._local_histogram()))      # [V][Ksm]: corpus-wide counts of my topics
        self.o.set_counts(self.lay.unpack(self.tr.all_gather(own)))                  # (the product gathers these lazily)

    def _exchange_phi(self, initial):
        """What travels since round 3: the rank's UNNORMALISED gammas [V][Ksm] in two halves of the vocabulary (whole 64-row
        segments), the Ksm column sums behind the second; the receiver divides (ParallelDirichlet.java:60-66: the same
        IEEE division, the clamp to Double.MIN_VALUE) -- csrc/ggs_api.hip phi_step_b1 .. phi_step_c, phi_repack_kernel."""
        n = self.k1 - self.k0
        gam, sums = self.o.phi_gammas_range(self.k0, self.k1, initial)
        mine = np.zeros((self.V, self.lay.Ksm))
        mine[:, :n] = gam.T
        own_sums = np.zeros(self.lay.Ksm)
        own_sums[:n] = sums
        nseg = (self.V + 63) // 64
        v_split = (nseg // 2) * 64 if nseg >= 16 else 0
        half0 = self.tr.all_gather(mine[:v_split].reshape(-1)).reshape(self.lay.n, v_split, self.lay.Ksm) if v_split else None
        half1 = self.tr.all_gather(np.concatenate([mine[v_split:].reshape(-1), own_sums]))
        body = half1[:, :-self.lay.Ksm].reshape(self.lay.n, self.V - v_split, self.lay.Ksm)
        tot = half1[:, -self.lay.Ksm:]                                               # [nranks][Ksm]
        g = body if half0 is None else np.concatenate([half0, body], axis=1)       # [nranks][V][Ksm]
        with np.errstate(divide="ignore", invalid="ignore"):
            x = np.where(tot[:, None, :] != 0, g / tot[:, None, :], g)
        x = np.where((tot[:, None, :] != 0) & (x <= 0), 4.9e-324, x)
        self.o.set_phi_rows(0, np.ascontiguousarray(self.lay.unpack(x).T))

    def set_z(self, z, redraw_phi=True):
        assert not redraw_phi
        self.o.set_z(z, False)

    def init_phi(self):
        self._exchange_counts()
        self._exchange_phi(initial=True)

    def sweep_begin(self):
        self.o.set_iteration(self.o.iteration + 1)
        self.o.z_step()

    def sweep_end(self):
        self._exchange_counts()
        self._exchange_phi(initial=False)

    def set_test_corpus(self, doc_ptr, tokens, doc_base=0):
        self._test = (doc_ptr, tokens, doc_base)

    def heldout_log_likelihood(self, num_particles=100):
        return self.o.heldout_log_likelihood(self._test[0], self._test[1], num_particles, self._test[2])


def _worker_sliced(rank, world, port, out_dir, scheme, K, sparse=False):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ldagroupedgibbssampler_amd.corpus import random_corpus
    from ldagroupedgibbssampler_amd.sharded import GlooSliceTransport, NativeExchange, ShardedGGS, java_lcg_initial_z
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = random_corpus(157, 1100, 60, seed=17, empty_every=10)          # 18 segments of 64 types: the gammas travel in two halves
    eng = OracleSlicedEngine(O, K, c.num_types, 0.1, 0.01, 4242, rank, world, GlooSliceTransport(rank, world), scheme, sparse)
    sh = ShardedGGS(eng, NativeExchange, c, rank, world)
    sh.set_z_global(java_lcg_initial_z(c.num_tokens, K, 77))
    sh.sweep(3)
    sh.set_test_corpus(random_corpus(23, 120, 40, seed=5, empty_every=6))
    ho_total, ho_docs = sh.heldout_log_likelihood(30)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), z=eng.o.get_z(), nwk=eng.o.get_type_topic_counts(), phi=eng.o.get_phi(),
             theta=eng.o.get_theta(), ho_total=ho_total, ho_docs=ho_docs)
    dist.destroy_process_group()


@pytest.mark.parametrize("scheme,K,sparse", [("ggs", 9, False), ("pcgs", 9, False), ("ggs", 1, False), ("ggs", 9, True), ("ggs", 1, True)])
def test_two_rank_topic_sliced_exchange_equals_unsharded(oracle, tmp_path, scheme, K, sparse):
    """The topic-sliced exchange (reduce-scatter of counts -- or, sparse, the non-zero cells as (cell, count) pairs
    all-to-all --, per-rank Phi batch, all-gather of the unnormalised gammas in two halves with the column sums behind the
    second, division on arrival) over gloo, two real processes: bit-identical to the unsharded oracle.  K = 9 over 2 ranks
    has unequal slices (5, 4); K = 1 leaves rank 1 without a topic."""
    import torch.multiprocessing as mp
    from ldagroupedgibbssampler_amd.corpus import random_corpus
    from ldagroupedgibbssampler_amd.sharded import java_lcg_initial_z
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    mp.spawn(_worker_sliced, args=(world, port, str(tmp_path), scheme, K, sparse), nprocs=world, join=True)
    c = random_corpus(157, 1100, 60, seed=17, empty_every=10)
    ref = oracle.OracleSampler(K, c.num_types, 0.1, 0.01, 4242)
    ref.set_scheme(scheme)
    ref.set_corpus(c.doc_ptr, c.tokens)
    ref.set_z(java_lcg_initial_z(c.num_tokens, K, 77), redraw_phi=True)
    ref.sweep(3)
    t = random_corpus(23, 120, 40, seed=5, empty_every=6)
    ho_total, ho_docs = ref.heldout_log_likelihood(t.doc_ptr, t.tokens, 30)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate([p["z"] for p in parts]), ref.get_z())
    if scheme == "ggs":
        assert np.array_equal(np.concatenate([p["theta"] for p in parts]).view(np.int64), ref.get_theta().view(np.int64))
    for p in parts:
        assert np.array_equal(p["nwk"], ref.get_type_topic_counts())
        assert np.array_equal(p["phi"].view(np.int64), ref.get_phi().view(np.int64))
        assert float(p["ho_total"]) == ho_total and np.array_equal(p["ho_docs"], ho_docs)
