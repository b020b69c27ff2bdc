"""Thin numpy wrapper over the C-ABI (include/ggs_hip.h); one object = one ggs_handle."""
import ctypes as C

import numpy as np

from . import _lib

OK = 0
ERR_NEGATIVE_COUNT, ERR_INVALID_TOPIC, ERR_RNG_EXHAUSTED, ERR_BAD_ARG = 1, 2, 3, 4
ERR_HIP, ERR_STATE, ERR_UNSUPPORTED, ERR_INVARIANT = 5, 6, 7, 8
FLAG_PARANOID, FLAG_SAVE_PHI_MEAN, FLAG_PCGS, FLAG_COLLAPSED = 1, 2, 4, 8
PURPOSE_Z, PURPOSE_THETA, PURPOSE_PHI, PURPOSE_INIT_PHI = 1, 2, 3, 4


class GGSError(RuntimeError):
    """A non-zero return of the C-ABI.  ERR_INVALID_TOPIC / ERR_NEGATIVE_COUNT are the
    IllegalStateExceptions of LDAGroupedGibbsSampler.java:84-85,116-118; ERR_BAD_ARG the
    IllegalArgumentExceptions."""

    def __init__(self, code, msg):
        super().__init__("ggs error %d: %s" % (code, msg))
        self.code = code


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _make_config(num_topics, num_types, alpha, beta, seed, device_id, flags, phi_burn_in, phi_mean_thin):
    """(ggs_config, the alpha array it points into -- keep it alive for the duration of the call)"""
    cfg = _lib.GGSConfig()
    cfg.struct_size = C.sizeof(_lib.GGSConfig)
    cfg.num_topics, cfg.num_types, cfg.device_id = int(num_topics), int(num_types), int(device_id)
    alpha = np.asarray(alpha, np.float64)
    keep = None
    if alpha.ndim == 0:
        cfg.alpha = None
        cfg.alpha_scalar = float(alpha)
    else:
        keep = np.ascontiguousarray(alpha)
        if keep.size != int(num_topics):
            raise ValueError("alpha must have num_topics entries")
        cfg.alpha = _dp(keep)
    cfg.beta = float(beta)
    cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    cfg.flags, cfg.phi_burn_in, cfg.phi_mean_thin = int(flags), int(phi_burn_in), int(phi_mean_thin)
    return cfg, keep


def rccl_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 creates it; every rank passes it to GGSHandle.attach_rccl)."""
    _lib.share_rccl_with_torch()
    buf = C.create_string_buffer(128)
    rc = _lib.load().ggs_rccl_unique_id(C.cast(buf, C.c_void_p))
    if rc:
        raise GGSError(rc, "ggs_rccl_unique_id failed (librccl not loadable?)")
    return buf.raw


class GGSHandle:
    def __init__(self, num_topics, num_types, alpha, beta, seed, device_id=0, flags=0, phi_burn_in=0, phi_mean_thin=1, _adopt=None):
        self._L = _lib.load()
        self.K, self.V = int(num_topics), int(num_types)
        self.D = self.N = 0
        self._owned = _adopt is None
        self._keep = []
        if _adopt is not None:             # a handle created by ggs_group_create
            self._h = _adopt
            return
        cfg, self._alpha = _make_config(num_topics, num_types, alpha, beta, seed, device_id, flags, phi_burn_in, phi_mean_thin)
        h = C.c_void_p()
        rc = self._L.ggs_create(C.byref(cfg), C.byref(h))
        if rc:
            raise GGSError(rc, "ggs_create failed")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._L.ggs_destroy(self._h)
            self._h = None

    # ---- multi-GPU exchange (attach before set_corpus) ----
    def attach_rccl(self, rank, nranks, unique_id):
        """ncclCommInitRank on this handle's device: every rank of the job calls this together."""
        _lib.share_rccl_with_torch()
        uid = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._L.ggs_attach_rccl(self._h, int(rank), int(nranks), C.cast(uid, C.c_void_p)))

    def attach_null_exchange(self, rank, nranks):
        """Timing aid only (results are wrong by construction): rank `rank` of `nranks` with the peers missing."""
        self._chk(self._L.ggs_attach_null_exchange(self._h, int(rank), int(nranks)))

    def attach_exchange(self, rank, nranks, reduce_scatter_i32, all_gather_f64, all_gather_i32, all_to_all_v_i32=None):
        """Caller-supplied transport: three callables (send_ptr, recv_ptr, count, hip_stream_ptr) -> 0 on success, and
        optionally the one of the sparse count exchange: (send_ptr, send_offsets, send_counts, recv_ptr, recv_offsets,
        recv_counts, hip_stream_ptr) with the four lists in int32 elements, one entry per rank."""
        def wrap(fn):
            def cb(_ctx, send, recv, count, stream):
                try:
                    return int(fn(send, recv, count, stream) or 0)
                except Exception:          # an exception must not unwind through the C frame
                    import traceback
                    traceback.print_exc()
                    return 1
            return _lib.EXCHANGE_CB(cb)
        ops = _lib.GGSExchangeOps()
        ops.struct_size = C.sizeof(_lib.GGSExchangeOps)
        ops.reduce_scatter_i32, ops.all_gather_f64, ops.all_gather_i32 = wrap(reduce_scatter_i32), wrap(all_gather_f64), wrap(all_gather_i32)
        if all_to_all_v_i32 is not None:
            def a2a(_ctx, send, soff, scnt, recv, roff, rcnt, stream):
                try:
                    n = int(nranks)
                    return int(all_to_all_v_i32(send, [soff[i] for i in range(n)], [scnt[i] for i in range(n)], recv, [roff[i] for i in range(n)],
                                                [rcnt[i] for i in range(n)], stream) or 0)
                except Exception:
                    import traceback
                    traceback.print_exc()
                    return 1
            ops.all_to_all_v_i32 = _lib.A2AV_CB(a2a)
        self._keep.append(ops)             # the library copies the table, the thunks must outlive the handle
        self._chk(self._L.ggs_attach_exchange(self._h, int(rank), int(nranks), C.byref(ops)))

    def set_count_exchange(self, mode):
        """How the counts travel: "auto" (sparse where the dense buffer is large and mostly zero), "dense", "sparse"."""
        self._chk(self._L.ggs_set_count_exchange(self._h, {"auto": 0, "dense": 1, "sparse": 2}[mode]))

    def count_exchange(self):
        sp, pairs, cells = C.c_int32(), C.c_int64(), C.c_int64()
        self._chk(self._L.ggs_get_count_exchange(self._h, C.byref(sp), C.byref(pairs), C.byref(cells)))
        return {"sparse": bool(sp.value), "pairs_last": pairs.value, "dense_cells": cells.value}

    def exchange_info(self):
        r, n, a, b = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._L.ggs_get_exchange_info(self._h, C.byref(r), C.byref(n), C.byref(a), C.byref(b)))
        p, cn, cr = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._L.ggs_get_exchange_provider(self._h, C.byref(p), C.byref(cn), C.byref(cr)))
        return {"rank": r.value, "nranks": n.value, "k_begin": a.value, "k_end": b.value,
                "provider": {0: "none", 1: "rccl", 2: "callbacks", 3: "null (timing aid)"}.get(p.value, str(p.value)),
                "comm_nranks": cn.value, "comm_rank": cr.value}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise GGSError(rc, self._L.ggs_last_error(self._h).decode())

    # ---- corpus / state ----
    def set_stream(self, stream_ptr):
        self._chk(self._L.ggs_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_corpus(self, doc_ptr, tokens, doc_base=0, tok_base=0):
        doc_ptr = np.ascontiguousarray(doc_ptr, np.int64)
        tokens = np.ascontiguousarray(tokens, np.int32)
        self._chk(self._L.ggs_set_corpus(self._h, doc_ptr.size - 1, _lp(doc_ptr), _ip(tokens), int(doc_base), int(tok_base)))
        self.D, self.N = doc_ptr.size - 1, int(doc_ptr[-1])

    def init_z_java_lcg(self, seed):
        self._chk(self._L.ggs_init_z_java_lcg(self._h, int(seed)))

    def set_z(self, z, redraw_phi=True):
        z = np.ascontiguousarray(z, np.int32)
        if z.size != self.N:
            raise ValueError("z must have one entry per token")
        self._chk(self._L.ggs_set_z(self._h, _ip(z), int(bool(redraw_phi))))

    def init_phi(self):
        self._chk(self._L.ggs_init_phi(self._h))

    def set_iteration(self, it):
        self._chk(self._L.ggs_set_iteration(self._h, int(it)))

    @property
    def iteration(self):
        it = C.c_int32()
        self._chk(self._L.ggs_get_iteration(self._h, C.byref(it)))
        return it.value

    # ---- sweeps ----
    def sweep(self, n=1):
        self._chk(self._L.ggs_sweep(self._h, int(n)))

    def collapsed_serial_sweep(self, java_seed, n=1):
        """SerialCollapsedLDA's own schedule (MSLDA:158-226, one chain, java.util.Random(java_seed)); needs FLAG_COLLAPSED."""
        self._chk(self._L.ggs_collapsed_serial_sweep(self._h, int(java_seed), int(n)))

    def sweep_begin(self):
        self._chk(self._L.ggs_sweep_begin(self._h))

    def sweep_end_async(self):
        self._chk(self._L.ggs_sweep_end_async(self._h))

    def sweep_end(self):
        self._chk(self._L.ggs_sweep_end(self._h))

    def sample_z_given_phi(self, n=1):
        self._chk(self._L.ggs_sample_z_given_phi(self._h, int(n)))

    def synchronize(self):
        self._chk(self._L.ggs_synchronize(self._h))

    def counts_device_ptr(self):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self._L.ggs_counts_device_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_global_token_count(self, n):
        self._chk(self._L.ggs_set_global_token_count(self._h, int(n)))

    # ---- getters ----
    def get_z(self):
        out = np.empty(self.N, np.int32)
        self._chk(self._L.ggs_get_z(self._h, _ip(out)))
        return out

    def get_type_topic_counts(self):
        out = np.empty((self.V, self.K), np.int32)
        self._chk(self._L.ggs_get_type_topic_counts(self._h, _ip(out)))
        return out

    def get_topic_totals(self):
        out = np.empty(self.K, np.int32)
        self._chk(self._L.ggs_get_topic_totals(self._h, _ip(out)))
        return out

    def get_phi(self):
        out = np.empty((self.K, self.V), np.float64)
        self._chk(self._L.ggs_get_phi(self._h, _dp(out)))
        return out

    def set_phi(self, phi):
        phi = np.ascontiguousarray(phi, np.float64)
        if phi.shape != (self.K, self.V):
            raise ValueError("phi must be [K][V]")
        self._chk(self._L.ggs_set_phi(self._h, _dp(phi)))

    def get_phi_mean(self):
        out = np.empty((self.K, self.V), np.float64)
        n = C.c_int32()
        self._chk(self._L.ggs_get_phi_mean(self._h, _dp(out), C.byref(n)))
        return (out, n.value) if n.value else (None, 0)

    def get_theta(self, doc_begin=0, doc_end=None):
        doc_end = self.D if doc_end is None else doc_end
        out = np.empty((doc_end - doc_begin, self.K), np.float64)
        self._chk(self._L.ggs_get_theta(self._h, doc_begin, doc_end, _dp(out)))
        return out

    def get_doc_topic_counts(self, doc_begin=0, doc_end=None):
        doc_end = self.D if doc_end is None else doc_end
        out = np.empty((doc_end - doc_begin, self.K), np.int32)
        self._chk(self._L.ggs_get_doc_topic_counts(self._h, doc_begin, doc_end, _ip(out)))
        return out

    def get_timings(self):
        t = _lib.GGSTimings()
        self._chk(self._L.ggs_get_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def reset_timings(self):
        self._chk(self._L.ggs_reset_timings(self._h))

    def check_invariants(self):
        self._chk(self._L.ggs_check_invariants(self._h))

    def model_log_likelihood(self):
        """(document side, topic side) of modelLogLikelihood (UPLDA:1644-1758), computed on the device."""
        a, b = C.c_double(), C.c_double()
        self._chk(self._L.ggs_model_log_likelihood(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def log_posterior(self):
        """(document side, topic side) of computeLogPosterior (UPLDA:1573-1634), computed on the device."""
        a, b = C.c_double(), C.c_double()
        self._chk(self._L.ggs_log_posterior(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_test_corpus(self, doc_ptr, tokens, doc_base=0):
        """addTestInstances (MSLDA:918-923): the held-out estimator's test set; ids >= num_types are out of vocabulary."""
        doc_ptr = np.ascontiguousarray(doc_ptr, np.int64)
        tokens = np.ascontiguousarray(tokens, np.int32)
        self._test_docs = doc_ptr.size - 1
        self._chk(self._L.ggs_set_test_corpus(self._h, self._test_docs, doc_ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                              tokens.ctypes.data_as(C.POINTER(C.c_int32)), int(doc_base)))

    def heldout_log_likelihood(self, num_particles=100):
        """MarginalProbEstimatorPlain.evaluateLeftToRight (MPE:85-121) on the current counts, on the device:
        (total, per-document values)."""
        doc_ll = np.zeros(getattr(self, "_test_docs", 0), np.float64)
        tot = C.c_double()
        self._chk(self._L.ggs_heldout_log_likelihood(self._h, int(num_particles), doc_ll.ctypes.data_as(C.POINTER(C.c_double)), C.byref(tot)))
        return tot.value, doc_ll

    def launch_info(self):
        c, l, b, nh, zp = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._L.ggs_get_launch_info(self._h, C.byref(c), C.byref(l), C.byref(b)))
        self._chk(self._L.ggs_get_num_hot_words(self._h, C.byref(nh)))
        self._chk(self._L.ggs_get_z_parts(self._h, C.byref(zp)))
        zk, zf, zc = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._L.ggs_get_z_form(self._h, C.byref(zk), C.byref(zf), C.byref(zc)))
        wt, ww, wd = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._L.ggs_get_warm_tiers(self._h, C.byref(wt), C.byref(ww), C.byref(wd)))
        return {"num_chunks": c.value, "lds_bytes_z": l.value, "docs_per_block_theta": b.value, "num_hot": nh.value, "z_parts": zp.value,
                "warm_tiers": wt.value, "num_warm": ww.value, "warm_docs_per_chunk": wd.value,
                "z_kernel": Z_KERNEL_NAMES.get(zk.value, str(zk.value)) + (" + z_warm_kernel (%d tiers)" % wt.value if wt.value else ""), "z_form": {0: "n/a", 1: "split", 2: "fused"}.get(zf.value, str(zf.value)),
                "z_form_calibrated": bool(zc.value)}


Z_KERNEL_NAMES = {0: "z_kernel (whole-row tiles)", 1: "z_sliced_kernel + z_hot_kernel (score registers)", 2: "z_stream1_kernel (one pass)",
                  3: "z_stream_kernel (two passes)", 4: "pcgs_sliced_kernel (lane per document)", 5: "pcgs_wave_kernel (wave per document)"}


class GGSGroup:
    """ONE process driving several GPUs (ggs_group_create): the handles of the group in rank order, joined by
    ncclCommInitAll; sweeps are issued for all devices from the calling thread."""

    @classmethod
    def adopt(cls, handles):
        """ggs_group_adopt: handles 0..n-1, each already joined by attach_exchange(i, n, ...), driven as one group over
        the caller's transport.  The group does not own the handles (close() leaves them alone)."""
        self = cls.__new__(cls)
        self._L = _lib.load()
        n = len(handles)
        self._arr = (C.c_void_p * n)(*[h._h for h in handles])
        self.handles, self._owns = list(handles), False
        self._chk(self._L.ggs_group_adopt(self._arr, n))
        return self

    def __init__(self, num_topics, num_types, alpha, beta, seed, device_ids, flags=0, phi_burn_in=0, phi_mean_thin=1):
        self._L = _lib.load()
        self._owns = True
        _lib.share_rccl_with_torch()
        cfg, keep = _make_config(num_topics, num_types, alpha, beta, seed, 0, flags, phi_burn_in, phi_mean_thin)
        n = len(device_ids)
        devs = (C.c_int32 * n)(*[int(d) for d in device_ids])
        self._arr = (C.c_void_p * n)()
        rc = self._L.ggs_group_create(C.byref(cfg), n, devs, self._arr)
        if rc:
            raise GGSError(rc, "ggs_group_create failed")
        self.handles = [GGSHandle(num_topics, num_types, alpha, beta, seed, _adopt=C.c_void_p(self._arr[i])) for i in range(n)]

    def _chk(self, rc):
        if rc:
            msgs = [self._L.ggs_last_error(h._h).decode() for h in self.handles]
            raise GGSError(rc, next((m for m in msgs if m), ""))

    def set_z(self, z_list, redraw_phi=True):
        zs = [np.ascontiguousarray(z, np.int32) for z in z_list]
        ptrs = (C.POINTER(C.c_int32) * len(zs))(*[_ip(z) for z in zs])
        self._chk(self._L.ggs_group_set_z(self._arr, len(zs), ptrs, int(bool(redraw_phi))))

    def sweep(self, n=1):
        self._chk(self._L.ggs_group_sweep(self._arr, len(self.handles), int(n)))

    def gather_counts(self):
        """Corpus-wide counts onto every handle (grouped collectives); call before the per-handle getters that read them."""
        self._chk(self._L.ggs_group_gather_counts(self._arr, len(self.handles)))

    def close(self):
        if getattr(self, "handles", None):
            if getattr(self, "_owns", True):
                for h in self.handles:
                    h._h = None
                self._L.ggs_group_destroy(self._arr, len(self.handles))
            self.handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- primitives (parity tests) ----
def debug_philox(ctr, key, device_id=0):
    L = _lib.load()
    ctr = np.ascontiguousarray(ctr, np.uint32).reshape(-1, 4)
    key = np.ascontiguousarray(key, np.uint32).reshape(-1, 2)
    out = np.empty_like(ctr)
    up = C.POINTER(C.c_uint32)
    rc = L.ggs_debug_philox(device_id, ctr.shape[0], ctr.ctypes.data_as(up), key.ctypes.data_as(up), out.ctypes.data_as(up))
    if rc:
        raise GGSError(rc, "ggs_debug_philox")
    return out


def debug_math(op, x, y=None, device_id=0):
    L = _lib.load()
    x = np.ascontiguousarray(x, np.float64)
    y = x if y is None else np.ascontiguousarray(y, np.float64)
    out = np.empty_like(x)
    rc = L.ggs_debug_math(device_id, {"log": 0, "pow": 1, "sqrt": 2, "div": 3}[op], x.size, _dp(x), _dp(y), _dp(out))
    if rc:
        raise GGSError(rc, "ggs_debug_math")
    return out


def debug_draw(kind, seed, iteration, purpose, elem0, n=None, shape=None, device_id=0):
    L = _lib.load()
    if shape is not None:
        shape = np.ascontiguousarray(shape, np.float64)
        n = shape.size
    out = np.empty(n, np.float64)
    st = C.c_int32()
    rc = L.ggs_debug_draw(device_id, {"uniform": 0, "gaussian": 1, "gamma": 2, "gamma_first_try": 3, "gamma_first_try_marked": 4}[kind], seed, iteration, purpose, elem0, n,
                          _dp(shape) if shape is not None else None, _dp(out), C.byref(st))
    if rc:
        raise GGSError(rc, "ggs_debug_draw")
    return out, st.value


def debug_column_sum(x=None, counts=None, beta=0.0, device_id=0):
    """Column sums in index order (Java's sequential rounding) of x [V][K], or of beta + counts [V][K]."""
    L = _lib.load()
    if (x is None) == (counts is None):
        raise ValueError("exactly one of x / counts")
    if x is not None:
        x = np.ascontiguousarray(x, np.float64)
        V, K = x.shape
        xp, cp = _dp(x), None
    else:
        counts = np.ascontiguousarray(counts, np.int32)
        V, K = counts.shape
        xp, cp = None, _ip(counts)
    out = np.empty(K, np.float64)
    rc = L.ggs_debug_column_sum(device_id, V, K, xp, cp, float(beta), _dp(out))
    if rc:
        raise GGSError(rc, "ggs_debug_column_sum")
    return out


def debug_column_sum_guided(x=None, counts=None, beta=0.0, guess=None, device_id=0):
    """The same with the caller's guess [ceil(V/64) + 1][K] of the running sums at the segment starts (None: the kernels
    make their own).  Returns (sums [K], the exact running sums the walk left [ceil(V/64) + 1][K], tokensPerTopic [K] or None)."""
    L = _lib.load()
    if (x is None) == (counts is None):
        raise ValueError("exactly one of x / counts")
    if x is not None:
        x = np.ascontiguousarray(x, np.float64)
        V, K = x.shape
        xp, cp = _dp(x), None
    else:
        counts = np.ascontiguousarray(counts, np.int32)
        V, K = counts.shape
        xp, cp = None, _ip(counts)
    nseg = (V + 63) // 64
    gp = None
    if guess is not None:
        guess = np.ascontiguousarray(guess, np.float64)
        if guess.shape != (nseg + 1, K):
            raise ValueError("guess must be [ceil(V/64) + 1][K]")
        gp = _dp(guess)
    out = np.empty(K, np.float64)
    pref = np.empty((nseg + 1, K), np.float64)
    n_k = np.empty(K, np.int32)
    rc = L.ggs_debug_column_sum_guided(device_id, V, K, xp, cp, float(beta), gp, _dp(out), _dp(pref), _ip(n_k) if counts is not None else None)
    if rc:
        raise GGSError(rc, "ggs_debug_column_sum_guided")
    return out, pref, (n_k if counts is not None else None)
