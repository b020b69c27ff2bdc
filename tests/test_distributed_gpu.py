"""Two real processes, each with its own HIP handle on the (one) GPU, exchanging device-resident counts through
torch.distributed -- the product's ShardedGGS + TorchHipExchange exactly as bench.py --gpus 2 runs them, except that the
process group is gloo (RCCL refuses two ranks on one device; the round's multi-GPU run is the driver's).  Both
constructions: one corpus split across the ranks, and every rank bringing its own shard (bench.py --scaling weak)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _corpora(world):
    from ldagroupedgibbssampler_amd.corpus import random_corpus
    return [random_corpus(70 + 31 * r, 260, 110, seed=300 + r, empty_every=9) for r in range(world)]


def _whole(shards):
    from ldagroupedgibbssampler_amd.corpus import Corpus
    ptr = [np.zeros(1, np.int64)]
    for c in shards:
        ptr.append(c.doc_ptr[1:] + ptr[-1][-1])
    return Corpus(np.concatenate(ptr), np.concatenate([c.tokens for c in shards]), shards[0].num_types)


def _worker(rank, world, port, out_dir, mode):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ldagroupedgibbssampler_amd import native
    from ldagroupedgibbssampler_amd.sharded import (ShardedGGS, TorchHipExchange, gather_shard_sizes, java_lcg_initial_z,
                                                    java_lcg_initial_z_slice)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K, alpha, beta, seed = 24, 0.1, 0.01, 777
    shards = _corpora(world)
    h = native.GGSHandle(K, shards[0].num_types, alpha, beta, seed, device_id=0)
    if mode == "split":
        whole = _whole(shards)
        sh = ShardedGGS(h, TorchHipExchange, whole, rank, world)
        sh.set_z_global(java_lcg_initial_z(whole.num_tokens, K, 5))
    else:
        sizes = gather_shard_sizes(shards[rank], rank, world, device="cuda")
        sh = ShardedGGS.from_local_shard(h, TorchHipExchange, shards[rank], sizes, rank, world)
        sh.set_z_local(java_lcg_initial_z_slice(sh.tok_base, shards[rank].num_tokens, K, 5))
    sh.sweep(2)
    sh.sweep(1)
    sh.set_test_corpus(shards[0])
    ho_total, ho_docs = sh.heldout_log_likelihood(40)
    h.check_invariants()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), z=h.get_z(), nwk=h.get_type_topic_counts(), phi=h.get_phi(), theta=h.get_theta(),
             ho_total=ho_total, ho_docs=ho_docs, doc_base=sh.doc_base)
    h.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["split", "own_shards"])
def test_two_gpu_processes_equal_the_oracle(oracle, tmp_path, mode):
    import torch.multiprocessing as mp
    from ldagroupedgibbssampler_amd.sharded import java_lcg_initial_z
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    mp.spawn(_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    shards = _corpora(world)
    whole = _whole(shards)
    K = 24
    ref = oracle.OracleSampler(K, whole.num_types, 0.1, 0.01, 777, threads=4)
    ref.set_corpus(whole.doc_ptr, whole.tokens)
    ref.set_z(java_lcg_initial_z(whole.num_tokens, K, 5), redraw_phi=True)
    ref.sweep(3)
    ho_total, ho_docs = ref.heldout_log_likelihood(shards[0].doc_ptr, shards[0].tokens, 40)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate([p["z"] for p in parts]), ref.get_z())
    assert np.array_equal(np.concatenate([p["theta"] for p in parts]).view(np.int64), ref.get_theta().view(np.int64))
    for p in parts:
        assert np.array_equal(p["nwk"], ref.get_type_topic_counts())
        assert np.array_equal(p["phi"].view(np.int64), ref.get_phi().view(np.int64))
        assert float(p["ho_total"]) == ho_total and np.array_equal(p["ho_docs"], ho_docs)
