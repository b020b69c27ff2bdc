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


def seg_func(x, g0, g1):
    """segment_function of ggs_exact_sum.hpp: (e_lo, D1(e_lo), D1(e_lo + 1)) under the guess (g0, g1) of the running sum at
    the segment's ends -- a D1 of -1 means unusable (an addend hits a tie in that binade) -- or None (no usable guess)."""
    if not (g0 > 0) or not math.isfinite(g1):
        return None
    e0, e1 = binade(g0), binade(g1)
    if e0 < -890 or e1 > 990 or e1 > e0 + 1 or e1 < e0:
        return None
    mant = struct.unpack("<q", struct.pack("<d", float(g0)))[0] >> 32 & 0xFFFFF
    e_lo = e0 - 1 if (e1 == e0 and mant < 0x6A09E) else e0
    out = []
    for e in (e_lo, e_lo + 1):
        u, scale = math.ldexp(1.0, e - 52), math.ldexp(1.0, 52 - e)
        acc, tie = 0.0, False
        for v in x:
            if v < 0 or v != v:
                return None if v < 0 else (e_lo, float("nan"), float("nan"))
            y = float(v) * scale
            f = math.floor(y) if y < 2.0 ** 60 else y
            r = y - f
            tie = tie or r == 0.5
            acc += f + 1.0 if r > 0.5 else f
        out.append(-1.0 if tie else acc * u)
    return e_lo, out[0], out[1]


def exact_parallel_sum(x, dirty_count=None, guess=None):
    """The walk: a step is accepted only when the running sum and the result lie in the binade the segment function was
    computed for; everything else goes element by element.  `guess(i)` = guessed running sum at the start of segment i
    (default: the order-free prefix sums, what the cold path computes)."""
    x = np.asarray(x, np.float64)
    segs = [x[i:i + SEG] for i in range(0, len(x), SEG)]
    if guess is None:
        with np.errstate(all="ignore"):
            start_hat = np.concatenate([[0.0], np.cumsum([float(np.sum(s)) for s in segs])])      # any order: a guess only
        guess = lambda i: start_hat[i]                                                         # noqa: E731
    s = np.float64(0.0)
    dirty = 0
    for i, sg in enumerate(segs):
        ok = False
        fn = seg_func(sg, guess(i), guess(i + 1))
        if fn is not None:
            e = binade(s)
            d = fn[1] if e == fn[0] else fn[2] if e == fn[0] + 1 else -1.0
            if d >= 0:
                t = s + np.float64(d)
                if binade(t) == e:                          # s is monotone: both ends in the binade => all of it was
                    s, ok = t, True
        if not ok:
            dirty += 1
            with np.errstate(all="ignore"):
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


@pytest.mark.parametrize("quality", ["exact", "30 % low", "35 % high", "a thousand times off", "zero", "nan", "negative", "decreasing"])
def test_the_guess_never_decides_the_result(quality):
    """Sweep t's magnitude sum is guided by sweep t-1's running sums, the gammas' sum by the magnitudes': whatever the
    guess, the result is the sequential sum -- a bad guess only sends more segments down the element-by-element path."""
    rng = np.random.default_rng(len(quality))
    for trial in range(12):
        n = int(rng.integers(65, 4000))
        x = rng.gamma(0.01 + (rng.random(n) < 0.15) * rng.integers(1, 40, n)) if trial % 2 else (rng.integers(0, 6, n) + 0.01).astype(np.float64)
        true = np.concatenate([[0.0], np.cumsum(x)])[::SEG]
        true = np.append(true, x.sum()) if len(true) < (n + SEG - 1) // SEG + 1 else true
        f = {"exact": lambda i: true[i], "30 % low": lambda i: 0.7 * true[i], "35 % high": lambda i: 1.35 * true[i],
             "a thousand times off": lambda i: 1000.0 * true[i] + 1.0, "zero": lambda i: 0.0, "nan": lambda i: float("nan"),
             "negative": lambda i: -true[i], "decreasing": lambda i: true[len(true) - 1 - i]}[quality]
        d = []
        assert seq_sum(x).tobytes() == exact_parallel_sum(x, d, guess=f).tobytes(), (quality, trial)
        nseg = (n + SEG - 1) // SEG
        if quality in ("exact", "30 % low", "35 % high") and nseg > 20:
            assert d[0] < nseg // 2, (quality, d[0], nseg)      # two candidate binades absorb a guess that is off by up to 41 %
