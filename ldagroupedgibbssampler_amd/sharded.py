"""Doc-sharded Grouped Gibbs sweep: one process per GPU, torch.distributed for the exchange.

Given (theta, Phi) every z is conditionally independent and theta_d depends on document d
alone, so contiguous document shards (the even-split rule of
randomscan/document/EvenSplitBatchBuilder.java:30-44) sample exactly what one GPU would:
every shard keys its Philox streams by GLOBAL token / document index.  Per sweep there is
ONE exchange, where the Java code merges its thread-shared AtomicInteger deltas
(UPLDA:1107-1221; the ADLDA analogue is sumTypeTopicCounts, ADLDA.java:302): a sum
all-reduce of the int32 [V][K] count buffer -- each shard's local (word, z) histogram --
(RCCL on GPUs).  Phi is then re-drawn on every
rank from identical counts and identical Philox keys, hence bit-identical without a
broadcast.

The engine is injected: the product passes ``native.GGSHandle`` (HIP); the CPU tests pass
an oracle-backed engine to exercise exactly this orchestration over gloo.
"""
import numpy as np

from .corpus import even_split


class _DevPtr:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can view it."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def wrap_device_int32(ptr, n, device=None):
    import torch
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.as_tensor(_DevPtr(ptr, n, "<i4"), device=dev)


class TorchHipExchange:
    """Sum all-reduce of device-resident int32 buffers owned by libggs_hip (RCCL when the
    process group backend is nccl)."""

    def __init__(self, handle, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.counts = wrap_device_int32(*handle.counts_device_ptr())

    def allreduce_sweep(self):
        self.dist.all_reduce(self.counts, op=self.dist.ReduceOp.SUM, group=self.group)

    allreduce_startup = allreduce_sweep


class ShardedGGS:
    """One rank's view of a doc-sharded sampler.

    engine      object with the GGSHandle method names (set_corpus, set_z, init_phi,
                sweep_begin, sweep_end, set_global_token_count, get_z, ...)
    exchange    object with allreduce_startup() / allreduce_sweep() acting on whatever the
                engine exchanges (the HIP engine: its count buffer, both times)
    """

    def __init__(self, engine, exchange_factory, corpus, rank, world_size):
        self.engine, self.rank, self.world = engine, int(rank), int(world_size)
        self.bounds = even_split(corpus.num_docs, self.world)
        sub, doc_base, tok_base = corpus.shard(self.bounds[self.rank], self.bounds[self.rank + 1])
        self._attach(exchange_factory, sub, doc_base, tok_base, corpus.num_tokens)

    @classmethod
    def from_local_shard(cls, engine, exchange_factory, shard, shard_sizes, rank, world_size):
        """The rank already holds its documents (a corpus too large to build on every rank: each rank
        loads or generates its own part).  ``shard_sizes`` = [(num_docs, num_tokens)] of every rank in
        rank order, e.g. from an all-gather; the global corpus is the concatenation of the shards."""
        self = cls.__new__(cls)
        self.engine, self.rank, self.world = engine, int(rank), int(world_size)
        if len(shard_sizes) != self.world or tuple(shard_sizes[self.rank]) != (shard.num_docs, shard.num_tokens):
            raise ValueError("shard_sizes must list (num_docs, num_tokens) of every rank, this rank's own included")
        docs = np.concatenate([[0], np.cumsum([int(d) for d, _ in shard_sizes])])
        toks = np.concatenate([[0], np.cumsum([int(t) for _, t in shard_sizes])])
        self.bounds = [int(x) for x in docs]
        self._attach(exchange_factory, shard, int(docs[self.rank]), int(toks[self.rank]), int(toks[-1]))
        return self

    def _attach(self, exchange_factory, sub, doc_base, tok_base, global_tokens):
        self.local, self.doc_base, self.tok_base, self.global_tokens = sub, doc_base, tok_base, global_tokens
        self.engine.set_corpus(sub.doc_ptr, sub.tokens, doc_base, tok_base)
        self.engine.set_global_token_count(global_tokens)
        self.exchange = exchange_factory(self.engine)

    def set_z_global(self, z_global):
        """Start-up: every rank takes its slice of the corpus-wide z (e.g. the seeded
        java.util.Random initialisation, which is one sequential stream and therefore
        computed once), builds local counts, sum-all-reduces them, draws the initial Phi."""
        self.set_z_local(z_global[self.tok_base:self.tok_base + self.local.num_tokens])

    def set_z_local(self, z_local):
        """The same start-up from this shard's own slice of z."""
        self.engine.set_z(np.ascontiguousarray(z_local, np.int32), redraw_phi=False)
        self.exchange.allreduce_startup()
        self.engine.init_phi()

    def set_test_corpus(self, test_corpus):
        """addTestInstances for a sharded run: the test documents are split by the same even rule, every rank keeps its
        part (the counts the estimator reads are corpus-wide on every rank after the sweep's exchange)."""
        self.test_bounds = even_split(test_corpus.num_docs, self.world)
        sub, doc_base, _ = test_corpus.shard(self.test_bounds[self.rank], self.test_bounds[self.rank + 1])
        self.engine.set_test_corpus(sub.doc_ptr, sub.tokens, doc_base)

    def heldout_log_likelihood(self, num_particles=100, gather=None):
        """MarginalProbEstimatorPlain.evaluateLeftToRight over the sharded test set: (total, per-document values of the
        whole test set) on every rank, bit-identical to the one-handle run -- the streams are keyed by the global
        document index and the total is added in document order (MPE:116).  ``gather(local_array) -> list of every
        rank's array`` defaults to torch.distributed.all_gather_object."""
        _, local = self.engine.heldout_log_likelihood(num_particles)
        if gather is None:
            import torch.distributed as dist

            def gather(a):
                out = [None] * self.world
                dist.all_gather_object(out, a)
                return out
        doc_ll = np.concatenate([np.asarray(a, np.float64) for a in gather(np.asarray(local, np.float64))])
        total = 0.0
        for v in doc_ll.tolist():
            total += v
        return total, doc_ll

    def sweep(self, n=1):
        """n sweeps, one count exchange each; only the last one is waited for (device-side error flags are sticky)."""
        end_async = getattr(self.engine, "sweep_end_async", self.engine.sweep_end)
        for i in range(n):
            self.engine.sweep_begin()
            self.exchange.allreduce_sweep()
            (self.engine.sweep_end if i == n - 1 else end_async)()


def gather_shard_sizes(shard, rank, world_size, device=None, group=None):
    """[(num_docs, num_tokens)] of every rank's own shard, in rank order (one small all-reduce): what
    ShardedGGS.from_local_shard needs when every rank loads or generates its own documents."""
    import torch
    import torch.distributed as dist
    sizes = torch.zeros(world_size, 2, dtype=torch.int64, device=device)
    sizes[rank, 0], sizes[rank, 1] = shard.num_docs, shard.num_tokens
    dist.all_reduce(sizes, group=group)
    return [(int(d), int(t)) for d, t in sizes.cpu().tolist()]


def java_lcg_initial_z_slice(tok_base, num_tokens, num_topics, seed):
    """This shard's part of the corpus-wide seeded z0: java.util.Random is one sequential stream, so the draws before
    tok_base are generated and dropped."""
    return java_lcg_initial_z(tok_base + num_tokens, num_topics, seed)[tok_base:]


def java_lcg_initial_z(num_tokens, num_topics, seed):
    """z0 = java.util.Random(seed).nextInt(K) per token in (doc, position) order
    (UPLDA:398-406,458-460), computed on the host once for a sharded start-up."""
    import ctypes as C

    from . import _lib
    z = np.empty(int(num_tokens), np.int32)
    rc = _lib.load().ggs_java_lcg_next_ints(int(seed), int(num_topics), z.size, z.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc:
        raise ValueError("ggs_java_lcg_next_ints rc=%d" % rc)
    return z
