"""A pure-Python model of csrc/ggs_exact_sum.hpp (segment functions under a binade guess, verified walk) against
the plain sequential sum.  Runs without a GPU: it pins the ARGUMENT the kernels rely on -- inside one binade a
sequential sum of non-negative doubles is integer arithmetic on addends quantised to the ulp, ties to even -- on the
inputs that break naive versions of it (exact ties, binade crossings, huge dynamic range, zeros).  The kernels
themselves are checked against the same sequential sum in tests/test_parity_gpu.py."""
import math
import struct

import numpy as np
import pytest

SEG = 64


def seq_sum(x):
    s = np.float64(0.0)
    for v in x:
        s = s + v
    return s


def binade(x):
    b = struct.unpack("<q", struct.pack("<d", float(x)))[0]
    return ((b >> 52) & 0x7FF) - 1023


def seg_func(x, e):
    """(D1, H, D2) of sum_segfn_kernel for the guessed binade e, or None (no usable guess)."""
    if e < -900 or e > 1000:
        return None
    u, scale = math.ldexp(1.0, e - 52), math.ldexp(1.0, 52 - e)
    pre = post = 0.0
    tie = False
    for v in x:
        if v < 0:
            return None
        y = float(v) * scale
        f = math.floor(y) if y < 2.0 ** 60 else y
        r = y - f
        if r == 0.5:
            if not tie:
                tie = True
                pre += f
            else:
                z = post + f
                post = z + (1.0 if (z * 0.5) != math.floor(z * 0.5) else 0.0)
        else:
            q = f + 1.0 if r > 0.5 else f
            if tie:
                post += q
            else:
                pre += q
    return pre * u, (0.5 * u if tie else 0.0), post * u


def exact_parallel_sum(x, dirty_count=None):
    x = np.asarray(x, np.float64)
    segs = [x[i:i + SEG] for i in range(0, len(x), SEG)]
    start_hat = np.concatenate([[0.0], np.cumsum([float(np.sum(s)) for s in segs])])      # any order: a guess only
    s = np.float64(0.0)
    dirty = 0
    for i, sg in enumerate(segs):
        ok = False
        s0, s1 = start_hat[i], start_hat[i + 1]
        if s0 > 0 and math.isfinite(s1):
            e = binade(s0)
            if binade(s0 * (1 - 1e-9)) == e and binade(s1 * (1 + 1e-9)) == e:
                fn = seg_func(sg, e)
                if fn is not None and binade(s) == e:
                    t = ((s + np.float64(fn[0])) + np.float64(fn[1])) + np.float64(fn[2])
                    if binade(t) == e:                      # s is monotone: both ends in the binade => all of it was
                        s, ok = t, True
        if not ok:
            dirty += 1
            for v in sg:
                s = s + v
    if dirty_count is not None:
        dirty_count.append(dirty)
    return s


@pytest.mark.parametrize("kind", ["tiny gammas", "sparse gammas", "counts + beta", "ties", "forty binades", "half zeros"])
def test_model_equals_sequential_sum(kind):
    rng = np.random.default_rng(hash(kind) % 2 ** 32)
    for trial in range(25):
        n = int(rng.integers(1, 3000))
        if kind == "tiny gammas":
            x = rng.gamma(0.01, size=n)
        elif kind == "sparse gammas":
            x = rng.gamma(0.01 + rng.integers(0, 3, n))
        elif kind == "counts + beta":
            x = (rng.integers(0, 5, n) + 0.01).astype(np.float64)
        elif kind == "ties":
            x = rng.integers(0, 8, n) * 2.0 ** -40 + (rng.random(n) < 0.2) * rng.integers(0, 3, n) * 2.0 ** -39
            x[0] = 2.0 ** 13 + float(rng.integers(0, 1000)) * 2.0 ** -39     # s in [2^13, 2^14): odd multiples of 2^-40 tie
            if trial % 3 == 0:
                x[int(rng.integers(0, n))] = 2.0 ** 13                          # a crossing somewhere
        elif kind == "forty binades":
            x = np.exp(rng.normal(0, 30, n))
        else:
            x = np.where(rng.random(n) < 0.5, 0.0, rng.gamma(0.1, size=n))
        a, b = seq_sum(x), exact_parallel_sum(x)
        assert a.tobytes() == b.tobytes(), (kind, trial, n, a, b)


def test_few_segments_need_the_sequential_path():
    rng = np.random.default_rng(5)
    n = 20000
    x = rng.gamma(0.01 + (rng.random(n) < 0.1) * rng.integers(1, 30, n))
    d = []
    assert seq_sum(x).tobytes() == exact_parallel_sum(x, d).tobytes()
    assert d[0] <= 20 and d[0] < (n // SEG) // 10          # the start at 0 and a handful of binade crossings
