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
