"""Doc-sharded Grouped Gibbs sweep: one process per GPU.

The exchange of the product path is NATIVE: libggs_hip joins the ranks' handles itself (ggs_attach_rccl: RCCL over xGMI)
and a sweep contains its collectives -- reduce-scatter of the int32 counts by topic slice, Phi drawn for the rank's own
topics, all-gather of the fp64 Phi slices (include/ggs_hip.h, "multi-GPU"; `rccl_exchange` below only hands the
unique id around).  `TopicSliceLayout` restates that protocol's layout rules in numpy: the CPU tests run it over gloo
with an oracle-backed engine, and `gloo_callback_exchange` plugs the same transport into the HIP handle's callback
exchange for the two-processes-on-one-GPU test.  The older form, where the caller all-reduces the whole [V][K] count
buffer and every rank re-draws all of Phi (`TorchHipExchange`), is kept as a cross-check.

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
        self.exchange = exchange_factory(self.engine)        # a native exchange must be attached before the corpus
        self.engine.set_corpus(sub.doc_ptr, sub.tokens, doc_base, tok_base)
        self.engine.set_global_token_count(global_tokens)

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


class NativeExchange:
    """The exchange lives inside the engine (libggs_hip with ggs_attach_*): sweep_end / init_phi contain the
    collectives, nothing is left to do between the two halves of a sweep."""

    def __init__(self, engine=None):
        pass

    def allreduce_startup(self):
        pass

    allreduce_sweep = allreduce_startup


def rccl_exchange(rank, world_size, group=None):
    """exchange_factory for ShardedGGS: rank 0 draws the RCCL unique id, torch.distributed (any backend) hands it to
    the other ranks, every rank joins with ggs_attach_rccl.  From then on the library orders its collectives itself,
    on the handle's own stream."""
    def factory(engine):
        import torch.distributed as dist
        from . import native
        # ncclCommInitRank is a blocking collective: a rank that cannot even load librccl must say so BEFORE the others
        # enter it, or they wait for ever.  Every rank therefore makes the id call (rank 0's is the one used) and the
        # ranks agree on the outcome first.
        uid, err = None, None
        try:
            uid = native.rccl_unique_id()
        except Exception as e:      # noqa: BLE001 -- reported to every rank below
            err = "rank %d: %s: %s" % (rank, type(e).__name__, e)
        box = [uid if rank == 0 else None]
        if world_size > 1:
            errs = [None] * world_size
            dist.all_gather_object(errs, err, group=group)
            errs = [e for e in errs if e]
            if errs:
                raise RuntimeError("RCCL is not usable on every rank: " + "; ".join(errs))
            dist.broadcast_object_list(box, src=0, group=group)
        elif err:
            raise RuntimeError(err)
        engine.attach_rccl(rank, world_size, box[0])
        return NativeExchange()
    return factory


class TopicSliceLayout:
    """The layout rules of the native exchange, restated: rank r owns the topics of slice r (sizes K/n + (K % n > r),
    randomscan/topic/EvenSplitTopicBatchBuilder.java:28-39); counts and Phi slices travel slice-major,
    [nranks][V][Ksm] with Ksm the widest slice, unused columns zero."""

    def __init__(self, num_topics, num_types, nranks):
        self.K, self.V, self.n = int(num_topics), int(num_types), int(nranks)
        self.bounds = even_split(self.K, self.n)
        self.Ksm = -(-self.K // self.n)

    def slice_of(self, rank):
        return self.bounds[rank], self.bounds[rank + 1]

    def pack(self, m_vk):
        """[V][K] -> [nranks][V][Ksm]"""
        out = np.zeros((self.n, self.V, self.Ksm), m_vk.dtype)
        for r in range(self.n):
            a, b = self.slice_of(r)
            out[r, :, :b - a] = m_vk[:, a:b]
        return out

    def pairs(self, m_vk):
        """The sparse form of the count exchange (ggs_set_count_exchange): for every destination rank the non-zero cells of
        its topic slice as an int32 array of (cell, count) pairs, cell = v * Ksm + column in that rank's [V][Ksm] slice."""
        out = []
        for r in range(self.n):
            a, b = self.slice_of(r)
            v, c = np.nonzero(m_vk[:, a:b])
            blk = np.empty(2 * v.size, np.int32)
            blk[0::2] = v * self.Ksm + c
            blk[1::2] = m_vk[v, a + c]
            out.append(blk)
        return out

    def from_pairs(self, blocks):
        """[V][Ksm]: the received (cell, count) blocks of every rank added up (a cell may come several times)."""
        own = np.zeros(self.V * self.Ksm, np.int32)
        for blk in blocks:
            np.add.at(own, blk[0::2], blk[1::2])
        return own.reshape(self.V, self.Ksm)

    def unpack(self, m_nvk):
        """[nranks][V][Ksm] -> [V][K]"""
        out = np.empty((self.V, self.K), m_nvk.dtype)
        for r in range(self.n):
            a, b = self.slice_of(r)
            out[:, a:b] = m_nvk[r, :, :b - a]
        return out


class GlooSliceTransport:
    """reduce-scatter / all-gather of host arrays over a torch.distributed group (gloo has no reduce-scatter: an
    all-reduce of the whole send buffer, of which the rank keeps its own chunk -- the same integers)."""

    def __init__(self, rank, world_size, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.rank, self.world, self.group = torch, dist, int(rank), int(world_size), group

    def reduce_scatter(self, send_n_x):
        t = self.torch.from_numpy(np.ascontiguousarray(send_n_x).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.numpy()[self.rank].copy()

    def all_gather(self, send_x):
        t = self.torch.from_numpy(np.ascontiguousarray(send_x))
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return np.stack([o.numpy() for o in out])

    def all_to_all_v(self, blocks):
        """blocks[d] (int32, any length) goes to rank d; returns what every rank addressed to this one, in rank order.
        The lengths travel first (an all-gather of the length matrix's row, as the library all-gathers its pair counts),
        then point-to-point sends in a fixed order."""
        lens = self.all_gather(np.asarray([b.size for b in blocks], np.int64))       # [src][dst]
        got = [None] * self.world
        got[self.rank] = np.ascontiguousarray(blocks[self.rank]).copy()
        reqs, keep = [], []
        for p in range(self.world):
            if p == self.rank:
                continue
            if blocks[p].size:
                t = self.torch.from_numpy(np.ascontiguousarray(blocks[p]).copy())
                keep.append(t)
                reqs.append(self.dist.isend(t, dst=p, group=self.group))
            n = int(lens[p][self.rank])
            got[p] = np.empty(n, np.int32)
            if n:
                reqs.append(self.dist.irecv(self.torch.from_numpy(got[p]), src=p, group=self.group))
        for r in reqs:
            r.wait()
        return got


def gloo_callback_exchange(rank, world_size, group=None):
    """exchange_factory that plugs a gloo transport into the HIP handle's CALLBACK exchange (ggs_attach_exchange):
    device buffers are staged through the host.  For tests that run several ranks on one GPU, where RCCL refuses to
    form a communicator; the native sweep, its slice layout and its Phi-slice kernels are exactly the product's."""
    def factory(engine):
        import torch
        tr = GlooSliceTransport(rank, world_size, group)
        dev = torch.device("cuda", torch.cuda.current_device())

        def view(ptr, n, typestr):
            return torch.as_tensor(_DevPtr(ptr, n, typestr), device=dev)

        def reduce_scatter_i32(send, recv, count, stream):
            torch.cuda.synchronize()
            own = tr.reduce_scatter(view(send, count * world_size, "<i4").cpu().numpy().reshape(world_size, count))
            view(recv, count, "<i4").copy_(torch.from_numpy(own))
            torch.cuda.synchronize()
            return 0

        def all_gather(typestr):
            def cb(send, recv, count, stream):
                torch.cuda.synchronize()
                allv = tr.all_gather(view(send, count, typestr).cpu().numpy())
                view(recv, count * world_size, typestr).copy_(torch.from_numpy(allv.reshape(-1)))
                torch.cuda.synchronize()
                return 0
            return cb

        def all_to_all_v_i32(send, soff, scnt, recv, roff, rcnt, stream):
            torch.cuda.synchronize()
            total = max(int(soff[i] + scnt[i]) for i in range(world_size))
            mine = view(send, max(total, 1), "<i4").cpu().numpy()
            got = tr.all_to_all_v([mine[soff[d]:soff[d] + scnt[d]] for d in range(world_size)])
            for s_ in range(world_size):
                if got[s_].size != rcnt[s_]:
                    raise RuntimeError("all_to_all_v: expected %d elements from rank %d, got %d" % (rcnt[s_], s_, got[s_].size))
                if got[s_].size:
                    view(recv + 4 * int(roff[s_]), got[s_].size, "<i4").copy_(torch.from_numpy(got[s_]))
            torch.cuda.synchronize()
            return 0

        engine.attach_exchange(rank, world_size, reduce_scatter_i32, all_gather("<f8"), all_gather("<i4"), all_to_all_v_i32)
        return NativeExchange()
    return factory


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
