"""ctypes binding of the CPU oracle (oracle/ggs_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, from
``__graft_entry__.smoke()`` and from ``bench.py``'s ``cpu_baseline`` leg; the
product package ``ldagroupedgibbssampler_amd`` never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libggs_oracle.so")

PURPOSE_Z, PURPOSE_THETA, PURPOSE_PHI, PURPOSE_INIT_PHI = 1, 2, 3, 4
OK, ERR_NEGATIVE_COUNT, ERR_INVALID_TOPIC, ERR_RNG_EXHAUSTED, ERR_BAD_ARG = range(5)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("ggs_oracle.c", "ggs_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "CC=gcc"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    lp = C.POINTER(C.c_int64)
    up = C.POINTER(C.c_uint32)
    vp = C.c_void_p
    sig = {
        "orc_philox4x32_10": (None, [up, up, up]),
        "orc_jrandom_next_ints": (None, [C.c_int64, C.c_int32, C.c_int64, ip]),
        "orc_jrandom_next_int_raw": (None, [C.c_int64, C.c_int64, ip]),
        "orc_jrandom_next_doubles": (None, [C.c_int64, C.c_int64, dp]),
        "orc_log": (C.c_double, [C.c_double]),
        "orc_pow": (C.c_double, [C.c_double, C.c_double]),
        "orc_log_array": (None, [C.c_int64, dp, dp]),
        "orc_pow_array": (None, [C.c_int64, dp, dp, dp]),
        "orc_uniform_array": (None, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int64, dp]),
        "orc_gaussian_array": (None, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int64, dp]),
        "orc_gamma_array": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int64, dp, dp]),
        "orc_dirichlet": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int64, dp, dp]),
        "orc_create": (vp, [C.c_int32, C.c_int32, dp, C.c_double, C.c_uint64]),
        "orc_destroy": (None, [vp]),
        "orc_set_corpus": (C.c_int, [vp, C.c_int64, lp, ip, C.c_int64, C.c_int64]),
        "orc_init_z_java_lcg": (C.c_int, [vp, C.c_int32]),
        "orc_set_z": (C.c_int, [vp, ip, C.c_int]),
        "orc_init_phi": (C.c_int, [vp]),
        "orc_set_phi_mean_gating": (None, [vp, C.c_int, C.c_int, C.c_int]),
        "orc_set_threads": (None, [vp, C.c_int]),
        "orc_set_scheme": (None, [vp, C.c_int]),
        "orc_model_log_likelihood": (None, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "orc_log_posterior": (None, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "orc_draw_diagnostic_theta": (C.c_int, [vp]),
        "orc_heldout_log_likelihood": (C.c_int, [vp, C.c_int64, lp, ip, C.c_int64, C.c_int32, dp, C.POINTER(C.c_double)]),
        "orc_set_iteration": (None, [vp, C.c_int32]),
        "orc_get_iteration": (C.c_int32, [vp]),
        "orc_sweep": (C.c_int, [vp, C.c_int32]),
        "orc_sweep_tuned": (C.c_int, [vp, C.c_int32]),
        "orc_sample_phi_range": (C.c_int, [vp, C.c_int32, C.c_int32]),
        "orc_init_phi_range": (C.c_int, [vp, C.c_int32, C.c_int32]),
        "orc_set_phi_rows": (None, [vp, C.c_int32, C.c_int32, dp]),
        "orc_phi_gammas_range": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, dp, dp]),
        "orc_set_counts": (None, [vp, ip]),
        "orc_z_step": (C.c_int, [vp]),
        "orc_update_counts": (C.c_int, [vp]),
        "orc_sample_phi": (C.c_int, [vp]),
        "orc_collapsed_sweep": (C.c_int, [vp, C.c_int32, C.c_int32]),
        "orc_collapsed_parallel_sweep": (C.c_int, [vp, C.c_int32]),
        "orc_num_tokens": (C.c_int64, [vp]),
        "orc_get_z": (None, [vp, ip]),
        "orc_get_type_topic_counts": (None, [vp, ip]),
        "orc_get_topic_type_counts": (None, [vp, ip]),
        "orc_get_topic_totals": (None, [vp, ip]),
        "orc_get_delta": (None, [vp, ip]),
        "orc_add_delta": (None, [vp, ip]),
        "orc_get_phi": (None, [vp, dp]),
        "orc_set_phi": (None, [vp, dp]),
        "orc_get_phi_mean": (C.c_int, [vp, dp]),
        "orc_get_theta": (None, [vp, dp]),
        "orc_get_doc_topic_counts": (None, [vp, ip]),
        "orc_last_error": (C.c_char_p, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def jrandom_ints(seed, bound, n):
    out = np.empty(n, np.int32)
    lib().orc_jrandom_next_ints(seed, bound, n, _ip(out))
    return out


def jrandom_raw(seed, n):
    out = np.empty(n, np.int32)
    lib().orc_jrandom_next_int_raw(seed, n, _ip(out))
    return out


def jrandom_doubles(seed, n):
    out = np.empty(n, np.float64)
    lib().orc_jrandom_next_doubles(seed, n, _dp(out))
    return out


def log(x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    lib().orc_log_array(x.size, _dp(x), _dp(out))
    return out


def pow(x, y):  # noqa: A001 - mirrors Math.pow
    x = np.ascontiguousarray(x, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    out = np.empty_like(x)
    lib().orc_pow_array(x.size, _dp(x), _dp(y), _dp(out))
    return out


def uniforms(seed, it, purpose, elem0, n):
    out = np.empty(n, np.float64)
    lib().orc_uniform_array(seed, it, purpose, elem0, n, _dp(out))
    return out


def gaussians(seed, it, purpose, elem0, n):
    out = np.empty(n, np.float64)
    lib().orc_gaussian_array(seed, it, purpose, elem0, n, _dp(out))
    return out


def gammas(seed, it, purpose, elem0, shape):
    shape = np.ascontiguousarray(shape, np.float64)
    out = np.empty_like(shape)
    rc = lib().orc_gamma_array(seed, it, purpose, elem0, shape.size, _dp(shape), _dp(out))
    if rc:
        raise RuntimeError("orc_gamma_array rc=%d" % rc)
    return out


def dirichlet(seed, it, purpose, elem0, p):
    p = np.ascontiguousarray(p, np.float64)
    out = np.empty_like(p)
    rc = lib().orc_dirichlet(seed, it, purpose, elem0, p.size, _dp(p), _dp(out))
    if rc:
        raise RuntimeError("orc_dirichlet rc=%d" % rc)
    return out


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("oracle error %d: %s" % (code, msg))
        self.code = code


class OracleSampler:
    """Handle-based wrapper, shaped like the product C-ABI so parity tests read alike."""

    def __init__(self, K, V, alpha, beta, seed, threads=1):
        self.K, self.V = int(K), int(V)
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, np.float64), (self.K,)))
        self._h = lib().orc_create(self.K, self.V, _dp(a), float(beta), int(seed))
        if not self._h:
            raise ValueError("orc_create failed")
        self.D = 0
        self.N = 0
        lib().orc_set_threads(self._h, threads)

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise OracleError(rc, lib().orc_last_error(self._h).decode())

    def set_corpus(self, doc_ptr, tokens, doc_base=0, tok_base=0):
        doc_ptr = np.ascontiguousarray(doc_ptr, np.int64)
        tokens = np.ascontiguousarray(tokens, np.int32)
        self.D = doc_ptr.size - 1
        self.N = int(doc_ptr[-1])
        self._chk(lib().orc_set_corpus(self._h, self.D, _lp(doc_ptr), _ip(tokens), doc_base, tok_base))

    def init_z_java_lcg(self, seed):
        self._chk(lib().orc_init_z_java_lcg(self._h, seed))

    def set_z(self, z, redraw_phi=True):
        z = np.ascontiguousarray(z, np.int32)
        assert z.size == self.N
        self._chk(lib().orc_set_z(self._h, _ip(z), int(redraw_phi)))

    def init_phi(self):
        self._chk(lib().orc_init_phi(self._h))

    def set_phi_mean_gating(self, save, burn_in, thin):
        lib().orc_set_phi_mean_gating(self._h, int(save), int(burn_in), int(thin))

    def set_threads(self, n):
        lib().orc_set_threads(self._h, n)

    def model_log_likelihood(self):
        """(document side, topic side) of UPLDA:1644-1758; their sum is the model log likelihood."""
        a, b = C.c_double(), C.c_double()
        lib().orc_model_log_likelihood(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def draw_diagnostic_theta(self):
        """UPLDA:710-714: the non-ggs schemes draw theta ~ Dir(n_d + alpha) afresh for the diagnostics (fills get_theta())."""
        self._chk(lib().orc_draw_diagnostic_theta(self._h))

    def log_posterior(self):
        """(document side, topic side) of UPLDA:1573-1634; their sum is the log posterior."""
        a, b = C.c_double(), C.c_double()
        lib().orc_log_posterior(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def heldout_log_likelihood(self, doc_ptr, tokens, num_particles=100, doc_base=0):
        """MarginalProbEstimatorPlain.evaluateLeftToRight (MPE:85-121) on the current counts: (total, per-document)."""
        doc_ptr = np.ascontiguousarray(doc_ptr, np.int64)
        tokens = np.ascontiguousarray(tokens, np.int32)
        doc_ll = np.zeros(doc_ptr.size - 1, np.float64)
        tot = C.c_double()
        self._chk(lib().orc_heldout_log_likelihood(self._h, doc_ptr.size - 1, _lp(doc_ptr), _ip(tokens), doc_base, num_particles,
                                                   _dp(doc_ll), C.byref(tot)))
        return tot.value, doc_ll

    def set_scheme(self, scheme):
        """'ggs' (default) or 'pcgs' (UPLDA:1466-1544 z loop, same Phi draw)."""
        lib().orc_set_scheme(self._h, {"ggs": 0, "pcgs": 1}[scheme])

    def set_iteration(self, it):
        lib().orc_set_iteration(self._h, it)

    @property
    def iteration(self):
        return lib().orc_get_iteration(self._h)

    def sweep(self, n=1):
        self._chk(lib().orc_sweep(self._h, n))

    def sample_phi_range(self, k0, k1):
        """loopOverTopics (GGS:182-198) for the topic batch [k0, k1)"""
        self._chk(lib().orc_sample_phi_range(self._h, int(k0), int(k1)))

    def init_phi_range(self, k0, k1):
        self._chk(lib().orc_init_phi_range(self._h, int(k0), int(k1)))

    def phi_gammas_range(self, k0, k1, initial=False):
        """The topic batch's gamma draws before the normalisation and their index-order sums: (gam [k1-k0][V], sums [k1-k0])."""
        gam = np.empty((int(k1) - int(k0), self.V), np.float64)
        sums = np.empty(int(k1) - int(k0), np.float64)
        self._chk(lib().orc_phi_gammas_range(self._h, int(bool(initial)), int(k0), int(k1), _dp(gam), _dp(sums)))
        return gam, sums

    def set_phi_rows(self, k0, rows):
        rows = np.ascontiguousarray(rows, np.float64)
        assert rows.ndim == 2 and rows.shape[1] == self.V and k0 + rows.shape[0] <= self.K
        lib().orc_set_phi_rows(self._h, int(k0), int(k0 + rows.shape[0]), _dp(rows))

    def set_counts(self, n_wk):
        n_wk = np.ascontiguousarray(n_wk, np.int32)
        assert n_wk.shape == (self.V, self.K)
        lib().orc_set_counts(self._h, _ip(n_wk))

    def sweep_tuned(self, n=1):
        """orc_sweep with CPU-friendly memory behaviour; identical results (bench.py's cpu_tuned_mt)."""
        self._chk(lib().orc_sweep_tuned(self._h, n))

    def z_step(self):
        self._chk(lib().orc_z_step(self._h))

    def update_counts(self):
        self._chk(lib().orc_update_counts(self._h))

    def sample_phi(self):
        self._chk(lib().orc_sample_phi(self._h))

    def collapsed_sweep(self, seed, n=1):
        self._chk(lib().orc_collapsed_sweep(self._h, seed, n))

    def collapsed_parallel_sweep(self, n=1):
        """MSLDA:196-203 with documents side by side on the sweep-start counts (the device's scheme=collapsed schedule)."""
        self._chk(lib().orc_collapsed_parallel_sweep(self._h, n))

    def get_z(self):
        out = np.empty(self.N, np.int32)
        lib().orc_get_z(self._h, _ip(out))
        return out

    def get_type_topic_counts(self):
        out = np.empty((self.V, self.K), np.int32)
        lib().orc_get_type_topic_counts(self._h, _ip(out))
        return out

    def get_topic_type_counts(self):
        out = np.empty((self.K, self.V), np.int32)
        lib().orc_get_topic_type_counts(self._h, _ip(out))
        return out

    def get_topic_totals(self):
        out = np.empty(self.K, np.int32)
        lib().orc_get_topic_totals(self._h, _ip(out))
        return out

    def get_delta(self):
        out = np.empty((self.V, self.K), np.int32)
        lib().orc_get_delta(self._h, _ip(out))
        return out

    def add_delta(self, d):
        d = np.ascontiguousarray(d, np.int32)
        lib().orc_add_delta(self._h, _ip(d))

    def get_phi(self):
        out = np.empty((self.K, self.V), np.float64)
        lib().orc_get_phi(self._h, _dp(out))
        return out

    def set_phi(self, phi):
        phi = np.ascontiguousarray(phi, np.float64)
        assert phi.shape == (self.K, self.V)
        lib().orc_set_phi(self._h, _dp(phi))

    def get_phi_mean(self):
        out = np.empty((self.K, self.V), np.float64)
        n = lib().orc_get_phi_mean(self._h, _dp(out))
        return (out, n) if n else (None, 0)

    def get_theta(self):
        out = np.empty((self.D, self.K), np.float64)
        lib().orc_get_theta(self._h, _dp(out))
        return out

    def get_doc_topic_counts(self):
        out = np.empty((self.D, self.K), np.int32)
        lib().orc_get_doc_topic_counts(self._h, _ip(out))
        return out
