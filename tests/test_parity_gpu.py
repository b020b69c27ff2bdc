"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle, bit for bit.

Bar: integer state (z, n_wk, n_k) identical; fp64 state (theta, phi, phi mean) identical
to the last bit -- the kernels keep the Java operation order and are built with
-ffp-contract=off, so no tolerance is needed or allowed.
"""
import numpy as np
import pytest

from ldagroupedgibbssampler_amd.corpus import even_split, random_corpus, synthetic_lda_corpus

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.int64)


def assert_bit_equal(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    if a.dtype.kind == "f":
        bad = bits(a) != bits(b)
        # NaN payloads aside, every bit must match
        assert not bad.any(), "%s: %d of %d differ, first at %s: %r vs %r" % (
            what, bad.sum(), bad.size, np.argwhere(bad)[0], a[bad][0], b[bad][0])
    else:
        assert np.array_equal(a, b), "%s differs in %d places" % (what, (a != b).sum())


# ---------------------------------------------------------------- primitives
def test_philox_device_matches_kat_and_oracle(native, oracle):
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_vectors.json")))
    for v in kat["philox4x32_10"]:
        out = native.debug_philox([v["ctr"]], [v["key"]])
        assert [int(x) for x in out[0]] == v["out"]
    rng = np.random.default_rng(0)
    ctr = rng.integers(0, 2**32, (4096, 4), dtype=np.uint64).astype(np.uint32)
    key = rng.integers(0, 2**32, (4096, 2), dtype=np.uint64).astype(np.uint32)
    dev = native.debug_philox(ctr, key)
    for i in range(0, 4096, 257):
        assert [int(x) for x in dev[i]] == oracle.philox(ctr[i], key[i])


def test_strict_math_device_bit_exact(native, oracle):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.random(200000), rng.random(50000) * 1e-9, np.exp(rng.uniform(-700, 700, 100000)),
                        1 + rng.uniform(-1e-6, 1e-6, 20000), [1.0, 0.5, 2.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308]])
    assert_bit_equal(native.debug_math("log", x), oracle.log(x), "StrictMath.log")
    u = np.concatenate([rng.random(300000), [0.0, 1.0 - 2**-53, 2**-53, 0.5]])
    y = 1.0 / np.concatenate([rng.uniform(1e-6, 1, 300000), [0.3, 0.3, 1e-9, 1e-3]])
    assert_bit_equal(native.debug_math("pow", u, y), oracle.pow(u, y), "StrictMath.pow")
    # sqrt and / must be the IEEE correctly rounded results the JVM gets
    a = np.exp(rng.uniform(-300, 300, 200000))
    b = np.exp(rng.uniform(-300, 300, 200000))
    assert_bit_equal(native.debug_math("sqrt", a), np.sqrt(a), "sqrt")
    assert_bit_equal(native.debug_math("div", a, b), a / b, "division")
    sub = rng.random(10000) * 1e-310     # subnormal operands and results
    assert_bit_equal(native.debug_math("div", sub, np.full(10000, 3.0)), sub / 3.0, "subnormal division")


def test_draw_streams_bit_exact(native, oracle):
    seed, it = 0xDEADBEEFCAFEF00D, 7
    n = 100000
    dev, st = native.debug_draw("uniform", seed, it, native.PURPOSE_Z, 2**40 + 5, n=n)
    assert st == 0
    assert_bit_equal(dev, oracle.uniforms(seed, it, oracle.PURPOSE_Z, 2**40 + 5, n), "uniform stream")
    assert dev.min() >= 0 and dev.max() < 1
    dev, st = native.debug_draw("gaussian", seed, it, native.PURPOSE_THETA, 123, n=n)
    assert st == 0
    assert_bit_equal(dev, oracle.gaussians(seed, it, oracle.PURPOSE_THETA, 123, n), "gaussian stream")
    rng = np.random.default_rng(2)
    shape = np.concatenate([rng.uniform(1e-3, 1, 60000), rng.uniform(1, 50, 60000), np.exp(rng.uniform(-14, 10, 60000)),
                            [1.0, 1e-6, 0.01, 0.1, 5.0, 1e4]])
    dev, st = native.debug_draw("gamma", seed, it, native.PURPOSE_PHI, 99, shape=shape)
    assert st == 0
    want = oracle.gammas(seed, it, oracle.PURPOSE_PHI, 99, shape)
    assert_bit_equal(dev, want, "Marsaglia-Tsang gamma")
    # the way the theta and Phi kernels draw: a straight-line first try, the general loops for what it leaves over
    dev, st = native.debug_draw("gamma_first_try", seed, it, native.PURPOSE_PHI, 99, shape=shape)
    assert st == 0
    assert_bit_equal(dev, want, "gamma, first try + general")
    marked, _ = native.debug_draw("gamma_first_try_marked", seed, it, native.PURPOSE_PHI, 99, shape=shape)
    settled = np.signbit(marked) & (want > 0)
    assert_bit_equal(np.abs(marked), want, "gamma, marked")
    assert 0.70 < settled.mean() < 0.92, settled.mean()       # measured 79 %: both polar pairs rejected 4.6 %, the squeeze failed, v <= 0


# ---------------------------------------------------------------- full path
def make_pair(native, oracle, corpus, K, alpha, beta, seed, flags=0, burn_in=0, thin=1, zseed=None, doc_base=0, tok_base=0):
    g = native.GGSHandle(K, corpus.num_types, alpha, beta, seed, flags=flags, phi_burn_in=burn_in, phi_mean_thin=thin)
    o = oracle.OracleSampler(K, corpus.num_types, alpha, beta, seed, threads=4)
    o.set_phi_mean_gating(bool(flags & native.FLAG_SAVE_PHI_MEAN), burn_in, thin)
    g.set_corpus(corpus.doc_ptr, corpus.tokens, doc_base, tok_base)
    o.set_corpus(corpus.doc_ptr, corpus.tokens, doc_base, tok_base)
    if zseed is not None:
        g.init_z_java_lcg(zseed)
        o.init_z_java_lcg(zseed)
        g.init_phi()
        o.init_phi()
    return g, o


def compare_state(g, o, tag, theta=True):
    assert_bit_equal(g.get_z(), o.get_z(), tag + " z")
    assert_bit_equal(g.get_type_topic_counts(), o.get_type_topic_counts(), tag + " n_wk")
    assert_bit_equal(g.get_topic_totals(), o.get_topic_totals(), tag + " n_k")
    assert_bit_equal(g.get_phi(), o.get_phi(), tag + " phi")
    if theta:
        assert_bit_equal(g.get_theta(), o.get_theta(), tag + " theta")
    assert_bit_equal(g.get_doc_topic_counts(), o.get_doc_topic_counts(), tag + " n_dk")


@pytest.mark.parametrize("K", [3, 20])
def test_cats_matches_oracle_and_golden(native, oracle, cats, K):
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "cats_ggs_golden.npz"))
    g, o = make_pair(native, oracle, cats, K, 5.0, 7.0, 2019, flags=native.FLAG_PARANOID, zseed=2019)
    assert_bit_equal(g.get_z(), gold["K%d_z0" % K], "z0 vs golden")
    compare_state(g, o, "cats K=%d init" % K, theta=False)
    for it in range(3):
        g.sweep(1)
        o.sweep(1)
        compare_state(g, o, "cats K=%d sweep %d" % (K, it + 1))
    assert_bit_equal(g.get_z(), gold["K%d_z3" % K], "z3 vs golden")
    assert_bit_equal(g.get_topic_totals(), gold["K%d_nk3" % K], "n_k vs golden")
    assert_bit_equal(g.get_type_topic_counts(), gold["K%d_nwk3" % K], "n_wk vs golden")
    assert_bit_equal(g.get_theta()[0], gold["K%d_theta3_doc0" % K], "theta doc0 vs golden")
    assert_bit_equal(g.get_phi()[0, :8], gold["K%d_phi3_row0_head" % K], "phi head vs golden")
    assert g.iteration == 3
    g.check_invariants()


@pytest.mark.parametrize("K,alpha,beta", [(1, 0.5, 0.1), (2, 0.1, 0.01), (7, 0.1, 0.01), (64, 0.05, 0.01), (65, 1.5, 0.5), (100, 0.1, 0.01),
                                          (129, 0.1, 0.01)])
def test_ragged_corpus_with_empty_documents(native, oracle, K, alpha, beta):
    c = random_corpus(301, 500, 150, seed=K, empty_every=7)   # lengths 0..150: 1-, 2- and 3-chunk documents
    g, o = make_pair(native, oracle, c, K, alpha, beta, 42 + K, flags=native.FLAG_PARANOID, zseed=K)
    compare_state(g, o, "ragged K=%d init" % K, theta=False)
    g.sweep(4)
    o.sweep(4)
    compare_state(g, o, "ragged K=%d" % K)


@pytest.mark.parametrize("K", [193, 200, 257, 1024, 1100, 2049])
def test_wide_topic_rows_use_the_streaming_kernel(native, oracle, K):
    """K > 192: scores no longer fit the register file; z_stream1_kernel streams the phiT rows through the slice ring
    once, checkpointing the sum chain every 64 topics, plus the 64-topic group the draw falls into (BASELINE config 3
    is K = 1024; > 1024 topics take a second round of theta staging)."""
    c = random_corpus(120, 300, 150, seed=K, empty_every=9)
    g, o = make_pair(native, oracle, c, K, 0.1, 0.01, 7 + K, flags=native.FLAG_PARANOID, zseed=K)
    ns = (K + 15) // 16
    gs = 1 if ns <= 16 else 2 if ns <= 32 else 4                   # slices per checkpoint group: the smallest that keeps 16 checkpoints in registers
    groups = (ns + gs - 1) // gs
    rows = 2 if K <= 512 else 1                                     # up to K = 512 a chunk may run across a document boundary: two theta rows
    assert g.launch_info()["lds_bytes_z"] == 2 * 8192 + rows * groups * gs * 128 + (groups * 512 if groups > 16 else 0)   # ring, theta row(s), checkpoints (LDS beyond 16 groups)
    g.sweep(2)
    o.sweep(2)
    compare_state(g, o, "wide K=%d" % K)


@pytest.mark.parametrize("mode,K", [(2, 33), (2, 48), (2, 100), (2, 184), (1, 192), (1, 185), (3, 48), (3, 257), (3, 1024), ("margin9", 257), ("margin9", 1024), ("margin13", 64),
                                    ("margin13", 300), ("ldsck", 300), ("ldsck", 1024), ("group2", 200), ("group4", 200), ("group4", 500), ("onerow", 200), ("onerow", 40), ("tworows", 1024), ("tworows", 600), (0, 5), (0, 100), (0, 257), ("fused", 20), ("fused", 100), ("chain", 9), ("nohot", 20), ("hot3", 100), ("queue0", 100), ("queue5", 24), ("queue5", 300),
                                    ("split", 100), ("split", 8), ("split", 97), ("split", 184), ("hotmargin9", 100), ("hotmargin9", 184), ("hotmargin13", 20), ("hotmargin13", 104), ("hotmargin13", 113),
                                    ("warm0", 100), ("warm1", 100), ("warmsparse", 100), ("warmsparse", 20), ("warmsparse", 160), ("warmsparse", 113), ("warmfused", 100), ("warmmargin13", 100)])
def test_alternate_z_kernels_agree(native, oracle, K, mode, monkeypatch):
    """The z kernels are interchangeable: GGS_DEBUG_ZKERNEL=2 forces the (one-pass) streaming kernel below 193 topics,
    =3 its two-pass form (every row streamed twice, the walk replayed in full: the cross-check of the one-pass kernel's
    certainty argument), GGS_DEBUG_MARGIN scales that kernel's margin by 1e9 / 1e13 so that a good share of / nearly all
    tokens take its exact-replay path,
    =0 the whole-row LDS tile kernel (first-generation, kept as a cross-check); GGS_DEBUG_SPLIT=0 ("fused")
    lets one sliced kernel take cold and hot chunks in turn instead of running z_hot_kernel beside it;
    GGS_DEBUG_CHAIN=1 ("chain") walks the Phi normalisers element by element instead of the exact parallel sums;
    GGS_DEBUG_SPLIT=2 ("split", "hotmargin*") keeps z_hot_kernel beside the cold kernel whatever the first step's timing says -- its walk is decided by
    the same margin argument, so GGS_DEBUG_MARGIN drives its tokens through its own exact replay;
    GGS_DEBUG_HOT caps the hot-word table (0: every chunk is a cold chunk; 3: three hot words);
    GGS_DEBUG_GAMMA_QUEUE caps the LDS queue of the gamma draws' leftovers (0 / 5 entries: the elements the straight-line
    first try does not settle are drawn by the general loops on the spot instead of gathered into full waves);
    GGS_DEBUG_WARM caps the warm tiers of z_warm_kernel (0: none, the words behind the hot table stay cold; 1: one tier), and
    "warmsparse" cuts the tables to 16 rows (hot: 8) and accepts tiers however empty their chunks are (GGS_DEBUG_WARM_ROWS,
    GGS_DEBUG_WARM_FILL=1): eight tiers, most tokens of the corpus warm, chunks drawn from the full number of documents."""
    warm_env = {"warm0": {"GGS_DEBUG_WARM": "0"}, "warm1": {"GGS_DEBUG_WARM": "1"},
                "warmsparse": {"GGS_DEBUG_WARM": "8", "GGS_DEBUG_WARM_ROWS": "16", "GGS_DEBUG_WARM_FILL": "1", "GGS_DEBUG_HOT": "8"},
                "warmfused": {"GGS_DEBUG_WARM_ROWS": "24", "GGS_DEBUG_WARM_FILL": "1", "GGS_DEBUG_SPLIT": "0"},
                "warmmargin13": {"GGS_DEBUG_WARM_ROWS": "32", "GGS_DEBUG_MARGIN": "1e13", "GGS_DEBUG_SPLIT": "2", "GGS_DEBUG_HOT": "8"}}.get(mode)
    if warm_env is not None:
        for k, v in warm_env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("GGS_DEBUG_WARM_CPW", "0")               # unasked, a tier wants 3 chunks per resident wave (a corpus of millions of tokens)
        monkeypatch.setenv("GGS_DEBUG_ZKERNEL", "1")
        c = random_corpus(150, 400, 140, seed=K + 7, empty_every=11)
        g, o = make_pair(native, oracle, c, K, 0.1, 0.01, 70 + K, flags=native.FLAG_PARANOID, zseed=K)
        for k in list(warm_env) + ["GGS_DEBUG_ZKERNEL", "GGS_DEBUG_WARM_CPW"]:
            monkeypatch.delenv(k)
        info = g.launch_info()
        if mode == "warm0":
            assert info["num_warm"] == 0
        elif mode == "warmsparse":
            assert info["warm_tiers"] == 8 and info["num_warm"] == 128
        else:
            assert info["warm_tiers"] >= 1 and info["num_warm"] > 0
        g.sweep(3)
        o.sweep(3)
        compare_state(g, o, "z kernel mode %s K=%d" % (mode, K))
        return
    env = {"fused": ("GGS_DEBUG_SPLIT", "0"), "chain": ("GGS_DEBUG_CHAIN", "1"), "nohot": ("GGS_DEBUG_HOT", "0"),
           "hot3": ("GGS_DEBUG_HOT", "3"), "margin9": ("GGS_DEBUG_MARGIN", "1e9"), "margin13": ("GGS_DEBUG_MARGIN", "1e13"), "ldsck": ("GGS_DEBUG_REGCK", "0"), "group2": ("GGS_DEBUG_GROUP", "2"), "group4": ("GGS_DEBUG_GROUP", "4"),
           "onerow": ("GGS_DEBUG_TWOROWS", "0"), "tworows": ("GGS_DEBUG_TWOROWS", "1"),
           "queue0": ("GGS_DEBUG_GAMMA_QUEUE", "0"), "queue5": ("GGS_DEBUG_GAMMA_QUEUE", "5"), "split": ("GGS_DEBUG_SPLIT", "2"),
           "hotmargin9": ("GGS_DEBUG_MARGIN", "1e9"), "hotmargin13": ("GGS_DEBUG_MARGIN", "1e13")}.get(mode, ("GGS_DEBUG_ZKERNEL", str(mode)))
    monkeypatch.setenv(*env)
    if str(mode).startswith("hotmargin"):
        monkeypatch.setenv("GGS_DEBUG_SPLIT", "2")
    if str(mode).startswith("hotmargin") or mode in ("split", "fused", "nohot", "hot3"):
        monkeypatch.setenv("GGS_DEBUG_ZKERNEL", "1")                # the score-register kernels also above their default range (K <= 160)
    if str(mode).startswith("margin") or mode == "ldsck" or str(mode).startswith("group") or mode in ("onerow", "tworows"):
        monkeypatch.setenv("GGS_DEBUG_ZKERNEL", "2")
    c = random_corpus(150, 400, 140, seed=K + (mode if isinstance(mode, int) else 7), empty_every=11)
    g, o = make_pair(native, oracle, c, K, 0.1, 0.01, 70 + K, flags=native.FLAG_PARANOID, zseed=K)
    monkeypatch.delenv(env[0])
    monkeypatch.delenv("GGS_DEBUG_ZKERNEL", raising=False)
    monkeypatch.delenv("GGS_DEBUG_SPLIT", raising=False)
    g.sweep(3)
    o.sweep(3)
    compare_state(g, o, "z kernel mode %s K=%d" % (mode, K))


@pytest.mark.parametrize("K,legs", [(20, "theta_main"), (100, "theta_main"), (100, "chain_main"), (150, "theta_main"), (100, "forced"), (64, "forced")])
def test_more_documents_than_types_and_split_sweeps(native, oracle, monkeypatch, K, legs):
    """With at least as many documents as word types (and K <= 160, no exchange) the theta draw is the longer leg behind the z step
    and keeps the handle's stream, while the count rebuild and the Phi chain run on the side stream (GGS_DEBUG_THETA_MAIN=0: the
    other way round) -- through ggs_sweep and through the split ggs_sweep_begin / ggs_sweep_end(_async) of the Java binding,
    with getters between the two halves: after ggs_sweep_begin alone the counts of the z just drawn are there (UPLDA:1107-1221)."""
    if legs == "chain_main":
        monkeypatch.setenv("GGS_DEBUG_THETA_MAIN", "0")
    if legs == "forced":                                           # =2: the theta draw on the handle's stream whatever D and V are
        monkeypatch.setenv("GGS_DEBUG_THETA_MAIN", "2")
        c = random_corpus(140, 420, 150, seed=K + 5, empty_every=13)
    else:
        c = random_corpus(500, 180, 120, seed=K + 5, empty_every=13)
        assert c.num_docs >= c.num_types
    g, o = make_pair(native, oracle, c, K, 0.1, 0.01, 31 + K, flags=native.FLAG_PARANOID, zseed=K + 1)
    monkeypatch.delenv("GGS_DEBUG_THETA_MAIN", raising=False)
    g.sweep(3)
    o.sweep(3)
    compare_state(g, o, "D >= V, K=%d %s: whole sweeps" % (K, legs))
    for it in range(3):
        g.sweep_begin()
        o.set_iteration(o.iteration + 1)
        o.z_step()
        o.update_counts()
        assert_bit_equal(g.get_z(), o.get_z(), "z between the halves")
        assert_bit_equal(g.get_type_topic_counts(), o.get_type_topic_counts(), "n_wk between the halves")
        assert_bit_equal(g.get_doc_topic_counts(), o.get_doc_topic_counts(), "n_dk between the halves")
        if it == 1:
            g.sweep_end_async()
        else:
            g.sweep_end()
        o.sample_phi()
        compare_state(g, o, "D >= V, K=%d %s: split sweep %d" % (K, legs, it + 1))
    g.sweep(2)                                                      # and back: the theta drawn ahead by a split sweep is consumed by a whole one
    o.sweep(2)
    compare_state(g, o, "D >= V, K=%d %s: whole sweeps after split ones" % (K, legs))
    t = g.get_timings()
    assert t["sweeps"] == 8 and t["theta_ms"] > 0 and t["z_ms"] > 0 and t["merge_ms"] > 0 and t["phi_ms"] > 0


def test_asymmetric_alpha_and_tiny_alpha(native, oracle):
    K = 5
    alphas = np.array([0.01, 5.0, 0.04, 0.1, 0.000001])    # ModifiedSimpleLDATest.java:12
    c = random_corpus(64, 80, 40, seed=3)
    g, o = make_pair(native, oracle, c, K, alphas, 0.01, 5, zseed=11)
    g.sweep(5)
    o.sweep(5)
    compare_state(g, o, "asymmetric alpha")
    th = g.get_theta()[np.diff(c.doc_ptr) > 0]     # empty documents draw no theta (GGS:52-53)
    assert (th > 0).all()          # zero draws are clamped to Double.MIN_VALUE, never 0
    assert (th == 4.9e-324).any()  # ... and alpha = 1e-6 does underflow in this corpus


def test_edge_corpora(native, oracle):
    # one document of one token; all-empty documents; a single long document (many chunks)
    from ldagroupedgibbssampler_amd.corpus import Corpus
    for doc_ptr, toks in [([0, 1], [0]), ([0, 0, 0, 0], []), ([0, 1000], list(np.arange(1000) % 11)), ([0, 64, 128, 129], list(np.arange(129) % 11))]:
        c = Corpus(np.asarray(doc_ptr, np.int64), np.asarray(toks, np.int32), 11)
        g, o = make_pair(native, oracle, c, 4, 0.3, 0.2, 1, zseed=5)
        g.sweep(2)
        o.sweep(2)
        compare_state(g, o, "edge %r" % (doc_ptr,))


def test_synthetic_k100_slice(native, oracle):
    c = synthetic_lda_corpus(1000, 5000, 200, true_topics=20, seed=2019)
    g, o = make_pair(native, oracle, c, 100, 0.1, 0.01, 2019, zseed=2019)
    g.sweep(2)
    o.set_threads(8)
    o.sweep(2)
    compare_state(g, o, "synthetic K=100")
    t = g.get_timings()
    assert t["sweeps"] == 2 and t["tokens_sampled"] == 2 * c.num_tokens and t["z_ms"] > 0


def test_synthetic_k100_medium(native, oracle):
    """30 000 documents of the benchmark's corpus family (6 M tokens, V = 50 000, Zipfian words), a third of the benchmark:
    every persistent wave walks dozens of cold and hot chunks, the ring wraps across chunk boundaries, the guest kernel
    runs beside the cold one, the normalisers see the full vocabulary."""
    c = synthetic_lda_corpus(30000, 50000, 200, true_topics=100, seed=7)
    g, o = make_pair(native, oracle, c, 100, 0.1, 0.01, 31, zseed=32)
    o.set_threads(16)
    g.sweep(2)
    o.sweep(2)
    compare_state(g, o, "synthetic K=100, 30000 documents")
    assert g.launch_info()["num_chunks"] > 40 * 1024


def test_set_z_and_sample_z_given_phi(native, oracle):
    c = random_corpus(100, 200, 60, seed=9)
    K = 10
    g, o = make_pair(native, oracle, c, K, 0.2, 0.05, 77)
    z = np.random.default_rng(0).integers(0, K, c.num_tokens).astype(np.int32)
    g.set_z(z, True)
    o.set_z(z, True)
    compare_state(g, o, "after set_z", theta=False)
    phi = g.get_phi()
    g.sample_z_given_phi(2)                       # UPLDA:975-1014
    for _ in range(2):
        o.set_iteration(o.iteration + 1)
        o.z_step()
        o.update_counts()
    assert_bit_equal(g.get_z(), o.get_z(), "z given phi")
    assert_bit_equal(g.get_type_topic_counts(), o.get_type_topic_counts(), "counts given phi")
    assert_bit_equal(g.get_phi(), phi, "phi untouched")
    assert_bit_equal(g.get_topic_totals(), o.get_topic_totals(), "n_k given phi")
    # setPhi round trip (UPLDA:1897-1926)
    p2 = np.random.default_rng(1).dirichlet(np.ones(c.num_types), K)
    g.set_phi(p2)
    assert_bit_equal(g.get_phi(), p2, "set_phi/get_phi")
    o.set_phi(p2)
    g.sweep(1)
    o.sweep(1)
    compare_state(g, o, "after set_phi + sweep")


def test_phi_mean_gating(native, oracle):
    c = random_corpus(80, 120, 50, seed=4)
    g, o = make_pair(native, oracle, c, 6, 0.3, 0.1, 5, flags=native.FLAG_SAVE_PHI_MEAN, burn_in=2, thin=2, zseed=1)
    assert g.get_phi_mean() == (None, 0)          # getPhiMeans returns null before any sample, UPLDA:1955-1958
    g.sweep(8)
    o.sweep(8)
    (gm, gn), (om, on) = g.get_phi_mean(), o.get_phi_mean()
    assert gn == on == 3                          # iterations 4, 6, 8 (> burn_in, % thin == 0)
    assert_bit_equal(gm, om, "phi mean")
    # setPhi restarts the running sum while noSampledPhi keeps counting (UPLDA:1897-1902)
    p = g.get_phi()
    g.set_phi(p)
    o.set_phi(p)
    g.sweep(2)
    o.sweep(2)
    (gm, gn), (om, on) = g.get_phi_mean(), o.get_phi_mean()
    assert gn == on == 4
    assert_bit_equal(gm, om, "phi mean after setPhi")
    assert_bit_equal(gm, g.get_phi() / 4, "phi mean after setPhi = the one Phi accumulated since, over noSampledPhi")


@pytest.mark.parametrize("scheme", ["ggs", "pcgs"])
def test_sharded_via_torch(native, oracle, scheme):
    """Three doc shards on one GPU, the count/delta exchange done on the device buffers
    through torch (what ldagroupedgibbssampler_amd.sharded does over RCCL) == one handle ==
    the oracle: the doc-sharded decomposition is exact, not AD-LDA."""
    torch = pytest.importorskip("torch")
    from ldagroupedgibbssampler_amd.sharded import wrap_device_int32
    c = random_corpus(203, 300, 120, seed=21, empty_every=11)
    K, alpha, beta, seed = 12, 0.1, 0.01, 99
    flags = native.FLAG_PCGS if scheme == "pcgs" else 0
    ref, o = make_pair(native, oracle, c, K, alpha, beta, seed, flags=flags)
    o.set_scheme(scheme)
    for s in (ref, o):
        s.init_z_java_lcg(3)
        s.init_phi()
    z0 = ref.get_z()
    bounds = even_split(c.num_docs, 3)
    shards = []
    for r in range(3):
        sub, db, tb = c.shard(bounds[r], bounds[r + 1])
        h = native.GGSHandle(K, c.num_types, alpha, beta, seed, flags=flags)
        h.set_corpus(sub.doc_ptr, sub.tokens, db, tb)
        h.set_global_token_count(c.num_tokens)
        h.set_z(z0[tb:tb + sub.num_tokens], redraw_phi=False)
        shards.append((h, tb, sub.num_tokens))
    counts = [wrap_device_int32(*h.counts_device_ptr()) for h, _, _ in shards]
    tot = torch.stack(counts).sum(0, dtype=torch.int32)
    for t in counts:
        t.copy_(tot)
    torch.cuda.synchronize()
    for h, _, _ in shards:
        h.init_phi()
    for it in range(3):
        for h, _, _ in shards:
            h.sweep_begin()          # leaves THIS shard's (word, z) histogram in its count buffer
        for h, _, _ in shards:
            h.synchronize()
        tot = torch.stack(counts).sum(0, dtype=torch.int32)
        for t in counts:
            t.copy_(tot)
        torch.cuda.synchronize()
        for h, _, _ in shards:
            h.sweep_end()
        ref.sweep(1)
        o.sweep(1)
    z = np.concatenate([h.get_z() for h, _, _ in shards])
    assert_bit_equal(z, ref.get_z(), "sharded z vs one handle")
    assert_bit_equal(z, o.get_z(), "sharded z vs oracle")
    for h, _, _ in shards:
        assert_bit_equal(h.get_type_topic_counts(), o.get_type_topic_counts(), "sharded n_wk")
        assert_bit_equal(h.get_phi(), o.get_phi(), "sharded phi")
        h.check_invariants()
    if scheme == "ggs":
        th = np.concatenate([h.get_theta() for h, _, _ in shards])
        assert_bit_equal(th, o.get_theta(), "sharded theta")


def test_error_behaviour(native):
    c = random_corpus(10, 20, 10, seed=1)
    with pytest.raises(native.GGSError) as e:
        native.GGSHandle(0, 10, 0.1, 0.1, 1)
    assert e.value.code == native.ERR_BAD_ARG
    with pytest.raises(native.GGSError):
        native.GGSHandle(4, 10, -0.1, 0.1, 1)          # alpha must be strictly positive (ParallelRandoms.java:61-63)
    with pytest.raises(native.GGSError):
        native.GGSHandle(4, 10, 0.1, 0.0, 1)
    g = native.GGSHandle(4, 20, 0.1, 0.1, 1)
    with pytest.raises(native.GGSError) as e:
        g.sweep(1)
    assert e.value.code == native.ERR_STATE
    bad = c.tokens.copy()
    bad[0] = 20
    with pytest.raises(native.GGSError) as e:
        g.set_corpus(c.doc_ptr, bad)
    assert e.value.code == native.ERR_BAD_ARG
    g.set_corpus(c.doc_ptr, c.tokens)
    with pytest.raises(native.GGSError) as e:
        g.sweep(1)                                     # no Phi yet
    assert e.value.code == native.ERR_STATE
    with pytest.raises(native.GGSError) as e:
        g.set_z(np.full(c.num_tokens, 4, np.int32))
    assert e.value.code == native.ERR_BAD_ARG
    g.init_z_java_lcg(1)
    g.init_phi()
    with pytest.raises(native.GGSError):
        g.sweep_end()
    g.sweep_begin()
    with pytest.raises(native.GGSError):
        g.sweep_begin()
    g.sweep_end()
    # an all-zero Phi row makes sum == 0 -> newTopic stays -1 -> the Java throw of GGS:116-118
    g.set_phi(np.zeros((4, 20)))
    with pytest.raises(native.GGSError) as e:
        g.sweep(1)
    assert e.value.code == native.ERR_INVALID_TOPIC
    assert "Topic sampled is invalid" in str(e.value)
    # scheme=pcgs limits: 4096 topics (two phiT rows of K/64 doubles per lane live in registers); no limit on the document length
    with pytest.raises(native.GGSError) as e:
        native.GGSHandle(4097, 10, 0.1, 0.1, 1, flags=native.FLAG_PCGS)
    assert e.value.code == native.ERR_UNSUPPORTED
    native.GGSHandle(1100, 10, 0.1, 0.1, 1, flags=native.FLAG_PCGS).close()
    p = native.GGSHandle(4, 20, 0.1, 0.1, 1, flags=native.FLAG_PCGS)
    p.set_corpus(np.array([0, 40000], np.int64), np.zeros(40000, np.int32))      # was refused (int16 counts): now the wave-per-document kernel
    p.set_corpus(c.doc_ptr, c.tokens)
    p.init_z_java_lcg(1)
    p.init_phi()
    p.set_phi(np.zeros((4, 20)))
    with pytest.raises(native.GGSError) as e:
        p.sweep(1)                                     # UPLDA:1529-1531
    assert e.value.code == native.ERR_INVALID_TOPIC


# ---------------------------------------------------------------- full size (properties only)
def test_full_size_properties(native):
    """BASELINE config 2 shape (D=100k, V=50k, ~20M tokens, K=100): what does not depend on size -- count invariants
    after every sweep, z range, theta/phi rows summing to 1, run-to-run determinism of the whole state -- and two sweeps
    of the oracle itself at full size (the GPU box has the host cores for it): z, n_k, phi, theta bit for bit."""
    import hashlib
    c = synthetic_lda_corpus(100000, 50000, 200, true_topics=100, seed=2019)

    def run():
        g = native.GGSHandle(100, c.num_types, 0.1, 0.01, 2019, flags=native.FLAG_PARANOID)
        g.set_corpus(c.doc_ptr, c.tokens)
        g.init_z_java_lcg(2019)
        g.init_phi()
        g.sweep(2)
        z = g.get_z()
        nk = g.get_topic_totals()
        th = g.get_theta(0, 2000)
        phi = g.get_phi()
        h = hashlib.sha256(z.tobytes() + nk.tobytes() + phi.tobytes()).hexdigest()
        g.close()
        return z, nk, th, phi, h

    z, nk, th, phi, h1 = run()
    assert z.min() >= 0 and z.max() < 100
    assert nk.sum() == c.num_tokens
    assert np.array_equal(np.bincount(z, minlength=100), nk)
    assert np.allclose(th.sum(1), 1.0, atol=1e-12) and np.allclose(phi.sum(1), 1.0, atol=1e-9)
    assert (phi > 0).all()
    *_, h2 = run()
    assert h1 == h2
    # and, since the host has cores to spare for ONE sweep of the oracle at this size: the full benchmark state, bit for bit
    import os
    from oracle import oracle as O
    o = O.OracleSampler(100, c.num_types, 0.1, 0.01, 2019, threads=min(64, os.cpu_count() or 4))
    o.set_corpus(c.doc_ptr, c.tokens)
    o.init_z_java_lcg(2019)
    o.init_phi()
    o.sweep(2)
    assert_bit_equal(z, o.get_z(), "full size z")
    assert_bit_equal(nk, o.get_topic_totals(), "full size n_k")
    assert_bit_equal(phi, o.get_phi(), "full size phi")
    assert_bit_equal(th, o.get_theta()[:2000], "full size theta (first 2000 documents)")


def test_sharded_orchestration_over_rccl_single_rank(native, oracle):
    """The product's sharded path end to end -- ShardedGGS + TorchHipExchange (torch.distributed,
    backend nccl == RCCL) acting in place on the library's own count buffer -- with the one rank a
    one-GPU box allows.  (Two-rank logic: tests/test_distributed_gloo.py; three shards on one GPU:
    test_sharded_via_torch above.)"""
    torch = pytest.importorskip("torch")
    import os
    import torch.distributed as dist
    from ldagroupedgibbssampler_amd.sharded import ShardedGGS, TorchHipExchange, java_lcg_initial_z
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c = random_corpus(150, 220, 90, seed=31, empty_every=12)
        K, alpha, beta, seed = 16, 0.1, 0.01, 5150
        h = native.GGSHandle(K, c.num_types, alpha, beta, seed, flags=native.FLAG_PARANOID)
        sh = ShardedGGS(h, TorchHipExchange, c, 0, 1)
        z0 = java_lcg_initial_z(c.num_tokens, K, 9)
        sh.set_z_global(z0)
        sh.sweep(3)
        o = oracle.OracleSampler(K, c.num_types, alpha, beta, seed)
        o.set_corpus(c.doc_ptr, c.tokens)
        o.set_z(z0, redraw_phi=True)
        o.sweep(3)
        compare_state(h, o, "rccl single rank")
    finally:
        if created:
            dist.destroy_process_group()


# ---------------------------------------------------------------- the exact parallel column sums
def _sequential_column_sum(x):
    s = np.zeros(x.shape[1], np.float64)
    for row in x:                       # one IEEE add per element, rows in index order == the Java loop
        s = s + row
    return s


@pytest.mark.gpu
def test_exact_parallel_column_sum_adversarial(native):
    """ggs_exact_sum.hpp against the plain sequential sum: sparse gamma columns, exact rounding ties,
    binade crossings anywhere, huge dynamic range, zeros, non-finite values, ragged V."""
    rng = np.random.default_rng(11)
    cases = []
    cases.append(("sparse gammas", rng.gamma(0.01 + (rng.random((50000, 7)) < 0.1) * rng.integers(1, 30, (50000, 7)))))
    cases.append(("tiny gammas only", rng.gamma(0.01, size=(20000, 5))))
    x = rng.integers(0, 8, (5000, 64)) * 2.0 ** -40 + (rng.random((5000, 64)) < 0.2) * rng.integers(0, 3, (5000, 64)) * 2.0 ** -39
    x[0] = 2.0 ** 13 + rng.integers(0, 1000, 64) * 2.0 ** -39           # s in [2^13, 2^14): odd multiples of 2^-40 are exact ties
    x[rng.integers(1, 5000, 20), rng.integers(0, 64, 20)] = 2.0 ** 13   # and a few forced crossings
    cases.append(("ties", x))
    cases.append(("forty binades", np.exp(rng.normal(0, 30, (9000, 9)))))
    x = rng.gamma(0.5, size=(3000, 6))
    x[:, 0] = 0.0
    x[1234, 1] = np.nan
    x[77, 2] = np.inf
    x[:, 3] = 4.9e-324
    cases.append(("zeros, NaN, inf, denormals", x))
    for V in (1, 63, 64, 65, 129, 4097):
        cases.append(("V=%d" % V, rng.gamma(0.3, size=(V, 3))))
    cases.append(("K=200", rng.gamma(0.05, size=(3000, 200))))
    cases.append(("V=150001: three super-groups of 1024 segments", rng.gamma(0.01 + (rng.random((150001, 3)) < 0.02) * 5.0)))
    x = rng.gamma(2.0, size=(70000, 2))
    x[65536:, 0] *= 1e6                                                  # binade jumps right after the first super-group
    cases.append(("jump at a super-group boundary", x))
    for tag, x in cases:
        assert_bit_equal(native.debug_column_sum(x=x), _sequential_column_sum(x), "column sum: " + tag)
    for beta in (0.01, 0.5, 7.0):
        n = ((rng.random((50000, 5)) < 0.05) * rng.integers(1, 400, (50000, 5))).astype(np.int32)
        want = _sequential_column_sum(beta + n.astype(np.float64))
        assert_bit_equal(native.debug_column_sum(counts=n, beta=beta), want, "magnitude beta=%g" % beta)


@pytest.mark.gpu
def test_column_sum_guess_never_decides_the_result(native):
    """The guided form of ggs_exact_sum.hpp: the magnitude sum of sweep t is guided by sweep t-1's exact running sums, the
    gammas' sum by the magnitudes'.  Whatever the guess -- exact, stale by 30 %, a thousand times off, zero, NaN, negative,
    decreasing -- the sums are the sequential ones; the walk leaves the exact running sums behind, and tokensPerTopic."""
    rng = np.random.default_rng(23)
    V, K = 20000, 11
    n = ((rng.random((V, K)) < 0.2) * rng.integers(1, 60, (V, K))).astype(np.int32)
    n[:, 3] = 0                                                          # a topic without tokens: s grows by beta alone
    beta = 0.01
    mag_addends = beta + n.astype(np.float64)
    gam = rng.gamma(mag_addends)
    gam[rng.integers(0, V, 40), rng.integers(0, K, 40)] = 0.0
    nseg = (V + 63) // 64

    def running(x):                                                      # sequential running sums at the segment starts
        out = np.zeros((nseg + 1, x.shape[1]))
        s = np.zeros(x.shape[1])
        for v in range(x.shape[0]):
            if v % 64 == 0:
                out[v // 64] = s
            s = s + x[v]
        out[nseg] = s
        return out

    run_mag, run_gam = running(mag_addends), running(gam)
    guesses = {"none (cold path)": None, "exact": run_mag, "30 % low": 0.7 * run_mag, "35 % high": 1.35 * run_mag, "x1000": 1000.0 * run_mag + 1.0,
               "zero": np.zeros_like(run_mag), "nan": np.full_like(run_mag, np.nan), "negative": -run_mag, "decreasing": run_mag[::-1].copy(),
               "inf": np.full_like(run_mag, np.inf), "another topic's": np.roll(run_mag, 3, axis=1)}
    for tag, g in guesses.items():
        out, pref, n_k = native.debug_column_sum_guided(counts=n, beta=beta, guess=g)
        assert_bit_equal(out, run_mag[nseg], "magnitude, guess " + tag)
        assert_bit_equal(pref, run_mag, "running magnitude sums, guess " + tag)
        assert np.array_equal(n_k, n.sum(axis=0)), tag
        out, pref, _ = native.debug_column_sum_guided(x=gam, guess=g)     # the magnitudes' sums guide the gammas'
        assert_bit_equal(out, run_gam[nseg], "gamma sum, guess " + tag)
        assert_bit_equal(pref, run_gam, "running gamma sums, guess " + tag)
    # ties and forced crossings under an exact and a stale guess
    x = rng.integers(0, 8, (5000, 16)) * 2.0 ** -40 + (rng.random((5000, 16)) < 0.2) * rng.integers(0, 3, (5000, 16)) * 2.0 ** -39
    x[0] = 2.0 ** 13 + rng.integers(0, 1000, 16) * 2.0 ** -39
    x[rng.integers(1, 5000, 20), rng.integers(0, 16, 20)] = 2.0 ** 13
    want = _sequential_column_sum(x)
    nsx = (5000 + 63) // 64
    exact = np.zeros((nsx + 1, 16))
    s = np.zeros(16)
    for v in range(5000):
        if v % 64 == 0:
            exact[v // 64] = s
        s = s + x[v]
    exact[nsx] = s
    for tag, g in (("exact", exact), ("stale", exact * 0.9), ("none", None)):
        assert_bit_equal(native.debug_column_sum_guided(x=x, guess=g)[0], want, "ties, guess " + tag)


@pytest.mark.gpu
@pytest.mark.parametrize("beta", [0.01, 0.5])
def test_long_vocabulary(native, oracle, beta):
    """V = 30000 (469 segments, 8 groups per topic in the normalisers) with few tokens: sparse n_wk, mostly tiny gammas."""
    c = random_corpus(400, 30000, 120, seed=5)
    g, o = make_pair(native, oracle, c, 6, 0.1, beta, 1234, flags=native.FLAG_PARANOID, zseed=9)
    compare_state(g, o, "long vocabulary init", theta=False)
    g.sweep(3)
    o.sweep(3)
    compare_state(g, o, "long vocabulary")


# ---------------------------------------------------------------- scheme=pcgs
@pytest.mark.parametrize("K,alpha,beta", [(3, 5.0, 7.0), (7, 0.1, 0.01), (20, 0.5, 0.1), (40, 0.1, 0.01), (100, 0.1, 0.01), (200, 0.05, 0.01)])
def test_pcgs_matches_oracle(native, oracle, K, alpha, beta):
    """UPLDA:1466-1544 z loop (theta integrated out, sequential inside a document) + the shared count
    rebuild and Phi draw, against the oracle's restatement: integer and fp64 state bit for bit."""
    c = random_corpus(333, 400, 170, seed=K, empty_every=9)      # 333 documents: five full lane groups and a ragged one
    g = native.GGSHandle(K, c.num_types, alpha, beta, 77 + K, flags=native.FLAG_PARANOID | native.FLAG_PCGS)
    o = oracle.OracleSampler(K, c.num_types, alpha, beta, 77 + K, threads=4)
    o.set_scheme("pcgs")
    for s in (g, o):
        s.set_corpus(c.doc_ptr, c.tokens)
        s.init_z_java_lcg(K)
        s.init_phi()
    compare_state(g, o, "pcgs K=%d init" % K, theta=False)
    for it in range(3):
        g.sweep(1)
        o.sweep(1)
        compare_state(g, o, "pcgs K=%d sweep %d" % (K, it + 1), theta=False)


def _pcgs_pair(native, oracle, c, K, alpha, beta, seed, zseed, flags=0):
    g = native.GGSHandle(K, c.num_types, alpha, beta, seed, flags=native.FLAG_PARANOID | native.FLAG_PCGS | flags)
    o = oracle.OracleSampler(K, c.num_types, alpha, beta, seed, threads=8)
    o.set_scheme("pcgs")
    for s in (g, o):
        s.set_corpus(c.doc_ptr, c.tokens)
        s.init_z_java_lcg(zseed)
        s.init_phi()
    return g, o


@pytest.mark.parametrize("K,alpha,beta", [(321, 0.1, 0.01), (500, 0.1, 0.01), (1024, 0.05, 0.01), (2049, 0.1, 0.01), (4096, 0.02, 0.05)])
def test_pcgs_wide_topic_rows(native, oracle, K, alpha, beta):
    """BASELINE's K = 500 and K = 1024 under scheme=pcgs: above 320 topics one WAVE owns a document (ggs_z_pcgs_wave.hpp;
    the lane-per-document kernels' int16 [K][64] counts no longer fit LDS).  The wave's reduction / scan only proposes a
    topic; the margin argument proves it is the Java walk's -- the oracle's bits either way.  UPLDA:1466-1545."""
    c = random_corpus(140, 600, 90, seed=K, empty_every=11)
    g, o = _pcgs_pair(native, oracle, c, K, alpha, beta, 5 + K, K)
    compare_state(g, o, "pcgs K=%d init" % K, theta=False)
    for it in range(2):
        g.sweep(1)
        o.sweep(1)
        compare_state(g, o, "pcgs K=%d sweep %d" % (K, it + 1), theta=False)


@pytest.mark.parametrize("K", [3, 64, 129, 200, 1024])
@pytest.mark.parametrize("margin", ["1", "1e9", "1e300"])
def test_pcgs_wave_kernel_forced_and_its_exact_replay(native, oracle, monkeypatch, K, margin):
    """The wave-per-document kernel at any K (GGS_DEBUG_PCGS_WAVE=1), with its certainty margin scaled up so that most
    (1e9) or all (1e300) tokens take the element-by-element replay: the same bits every way."""
    monkeypatch.setenv("GGS_DEBUG_PCGS_WAVE", "1")
    monkeypatch.setenv("GGS_DEBUG_MARGIN", margin)
    c = random_corpus(70, 300, 60, seed=K + 1, empty_every=9)
    g, o = _pcgs_pair(native, oracle, c, K, 0.1, 0.01, 11, K)
    monkeypatch.delenv("GGS_DEBUG_PCGS_WAVE")
    monkeypatch.delenv("GGS_DEBUG_MARGIN")
    g.sweep(2)
    o.sweep(2)
    compare_state(g, o, "pcgs wave kernel K=%d margin x%s" % (K, margin), theta=False)


@pytest.mark.parametrize("K,wave", [(176, None), (177, None), (192, "0"), (184, "0"), (150, "1")])
def test_pcgs_either_kernel_around_the_switch_point(native, oracle, monkeypatch, K, wave):
    """The lane-per-document score-register kernels serve up to 176 topics by default and the wave-per-document kernel
    from 177 (ggs_api.hip: the measured break-even); both stay reachable on either side (GGS_DEBUG_PCGS_WAVE) and give the
    oracle's bits -- including the 176..192 lane variants that spill."""
    if wave is not None:
        monkeypatch.setenv("GGS_DEBUG_PCGS_WAVE", wave)
    c = random_corpus(150, 350, 110, seed=K + 3, empty_every=8)
    g, o = _pcgs_pair(native, oracle, c, K, 0.1, 0.01, 21, K)
    monkeypatch.delenv("GGS_DEBUG_PCGS_WAVE", raising=False)
    want = "wave" if (wave == "1" or (wave is None and K > 176)) else "lane"
    assert want in g.launch_info()["z_kernel"]
    g.sweep(2)
    o.sweep(2)
    compare_state(g, o, "pcgs K=%d (%s per document)" % (K, want), theta=False)


def test_pcgs_document_of_forty_thousand_tokens(native, oracle):
    """The lane-per-document kernels count a document's topics in int16: a document of 32 768 tokens or more used to be
    refused.  Such a corpus now goes to the wave-per-document kernel (int32 counts), whatever K."""
    rng = np.random.default_rng(3)
    lens = np.array([40000, 0, 7, 33000, 120, 1], np.int64)
    doc_ptr = np.concatenate([[0], np.cumsum(lens)])
    V = 500
    tokens = rng.zipf(1.3, int(doc_ptr[-1])).astype(np.int64) % V
    from ldagroupedgibbssampler_amd.corpus import Corpus
    c = Corpus(doc_ptr, tokens.astype(np.int32), V)
    for K in (20, 400):
        g, o = _pcgs_pair(native, oracle, c, K, 0.1, 0.01, 9, 4)
        g.sweep(2)
        o.sweep(2)
        compare_state(g, o, "pcgs long documents K=%d" % K, theta=False)
        assert g.get_doc_topic_counts(0, 1).sum() == 40000


def test_pcgs_wave_kernel_raises_what_java_raises(native):
    """UPLDA:1529-1531 through the wave kernel's replay path: an all-zero Phi makes every score 0."""
    c = random_corpus(10, 20, 30, seed=1)
    p = native.GGSHandle(400, 20, 0.1, 0.01, 1, flags=native.FLAG_PCGS)
    p.set_corpus(c.doc_ptr, c.tokens)
    p.init_z_java_lcg(1)
    p.init_phi()
    p.set_phi(np.zeros((400, 20)))
    with pytest.raises(native.GGSError) as e:
        p.sweep(1)
    assert e.value.code == native.ERR_INVALID_TOPIC


def test_pcgs_two_pass_kernel_below_193_topics(native, oracle, monkeypatch):
    """K <= 192 normally takes pcgs_sliced_kernel (scores in registers); GGS_DEBUG_PCGS_STREAM=1 keeps the two-pass
    pcgs_z_kernel, which larger K always use."""
    monkeypatch.setenv("GGS_DEBUG_PCGS_STREAM", "1")
    c = random_corpus(200, 300, 130, seed=3, empty_every=10)
    g = native.GGSHandle(40, c.num_types, 0.1, 0.01, 5, flags=native.FLAG_PARANOID | native.FLAG_PCGS)
    monkeypatch.delenv("GGS_DEBUG_PCGS_STREAM")
    o = oracle.OracleSampler(40, c.num_types, 0.1, 0.01, 5, threads=4)
    o.set_scheme("pcgs")
    for s in (g, o):
        s.set_corpus(c.doc_ptr, c.tokens)
        s.init_z_java_lcg(2)
        s.init_phi()
        s.sweep(2)
    compare_state(g, o, "pcgs two-pass K=40", theta=False)


def test_pcgs_on_cats(native, oracle, cats):
    """cats: 23 documents of very different lengths, so the lanes of the one group finish at different steps."""
    K = 20
    o = oracle.OracleSampler(K, cats.num_types, 5.0, 7.0, 2019, threads=2)
    o.set_scheme("pcgs")
    o.set_corpus(cats.doc_ptr, cats.tokens)
    o.init_z_java_lcg(2019)
    o.init_phi()
    g = native.GGSHandle(K, cats.num_types, 5.0, 7.0, 2019, flags=native.FLAG_PCGS)
    g.set_corpus(cats.doc_ptr, cats.tokens)
    g.init_z_java_lcg(2019)
    g.init_phi()
    o.sweep(3)
    g.sweep(3)
    compare_state(g, o, "pcgs cats", theta=False)


# ---------------------------------------------------------------- model log likelihood on the device
def test_model_log_likelihood_on_device(native, oracle, cats):
    """ggs_model_log_likelihood against the oracle's Java-order loop (UPLDA:1644-1758, MALLET logGammaStirling):
    same terms, reduced in a fixed tree instead of one running double => 1e-11 relative (stated tolerance);
    run-to-run identical; additive over document shards; and within 1e-6 of scipy's exact lgamma formula."""
    from ldagroupedgibbssampler_amd.sampler import model_log_likelihood
    for corpus, K, alpha, beta in ((cats, 20, 5.0, 7.0), (random_corpus(700, 900, 120, seed=4, empty_every=13), 33, 0.1, 0.01)):
        g, o = make_pair(native, oracle, corpus, K, alpha, beta, 5, zseed=6)
        g.sweep(3)
        o.sweep(3)
        gd, gt = g.model_log_likelihood()
        od, ot = o.model_log_likelihood()
        assert abs(gd - od) <= 1e-11 * abs(od) and abs(gt - ot) <= 1e-11 * abs(ot), ((gd, od), (gt, ot))
        assert g.model_log_likelihood() == (gd, gt)
        exact = model_log_likelihood(o.get_doc_topic_counts(), o.get_type_topic_counts(), o.get_topic_totals(), alpha, beta)
        assert abs((gd + gt) - exact) <= 1e-6 * abs(exact)
        # two document shards with the same z and counts: the document sides add up, the topic side is shared
        cut = corpus.num_docs // 3
        parts = []
        for lo, hi in ((0, cut), (cut, corpus.num_docs)):
            sub, db, tb = corpus.shard(lo, hi)
            h = native.GGSHandle(K, corpus.num_types, alpha, beta, 5)
            h.set_corpus(sub.doc_ptr, sub.tokens, db, tb)
            h.set_z(o.get_z()[tb:tb + sub.num_tokens], redraw_phi=False)
            parts.append(h.model_log_likelihood()[0])
        assert abs(sum(parts) - od) <= 1e-11 * abs(od)


def test_log_posterior_on_device(native, oracle, cats):
    """ggs_log_posterior (UPLDA:1573-1634, Doss and George 2025) against the oracle's Java-order loop: 1e-11 relative
    (fixed reduction tree instead of one running double; per-token terms instead of count * logPhi per cell),
    run-to-run identical, additive over document shards, and against the plain numpy formula."""
    for corpus, K, alpha, beta in ((cats, 20, 5.0, 7.0), (random_corpus(500, 700, 110, seed=8, empty_every=17), 33, 0.1, 0.01)):
        g, o = make_pair(native, oracle, corpus, K, alpha, beta, 15, zseed=16)
        g.sweep(2)
        o.sweep(2)
        gd, gt = g.log_posterior()
        od, ot = o.log_posterior()
        assert abs(gd - od) <= 1e-11 * abs(od) and abs(gt - ot) <= 1e-11 * abs(ot), ((gd, od), (gt, ot))
        assert g.log_posterior() == (gd, gt)
        phi, theta, z = o.get_phi(), o.get_theta(), o.get_z()
        doc_of = np.repeat(np.arange(corpus.num_docs), np.diff(corpus.doc_ptr))
        n_dk = o.get_doc_topic_counts().astype(np.float64)
        direct = (np.log(phi[z, corpus.tokens] + 1e-12).sum() + ((n_dk + alpha - 1.0) * np.log(theta + 1e-12)).sum()
                  + (beta - 1.0) * np.log(phi + 1e-12).sum())
        assert doc_of.size == z.size and abs((gd + gt) - direct) <= 1e-9 * abs(direct)
    # scheme=pcgs keeps no thetaMatrix: its diagnostic theta is a fresh Dir(n_d + alpha) draw (UPLDA:710-714), here from the Philox stream
    for K in (4, 300):
        p = native.GGSHandle(K, cats.num_types, 0.1, 0.1, 1, flags=native.FLAG_PCGS)
        o = oracle.OracleSampler(K, cats.num_types, 0.1, 0.1, 1)
        o.set_scheme("pcgs")
        for s in (p, o):
            s.set_corpus(cats.doc_ptr, cats.tokens)
            s.init_z_java_lcg(1)
            s.init_phi()
            s.sweep(2)
        o.draw_diagnostic_theta()
        od, ot = o.log_posterior()
        gd, gt = p.log_posterior()
        assert abs(gd - od) <= 1e-11 * abs(od) and abs(gt - ot) <= 1e-11 * abs(ot), (K, gd, od, gt, ot)
        assert_bit_equal(p.get_theta(), o.get_theta(), "pcgs diagnostic theta K=%d" % K)
        p.sweep(1)
        o.sweep(1)
        compare_state(p, o, "pcgs after the diagnostic", theta=False)      # the diagnostic does not disturb the chain
    c = native.GGSHandle(4, cats.num_types, 0.1, 0.1, 1, flags=native.FLAG_COLLAPSED)
    c.set_corpus(cats.doc_ptr, cats.tokens)
    c.init_z_java_lcg(1)
    c.init_phi()
    with pytest.raises(native.GGSError) as e:
        c.log_posterior()
    assert e.value.code == native.ERR_UNSUPPORTED


@pytest.mark.parametrize("scheme", ["ggs", "pcgs"])
def test_resume_from_exported_state(native, oracle, scheme):
    """The reference has no resume path, only the state exports (getZIndicators / setZIndicators-without-redraw,
    getPhi / setPhi) -- with counter-based streams keyed by the iteration they are enough: a second handle fed z, Phi and
    the iteration number of the first continues bit-identically to the uninterrupted run."""
    c = random_corpus(180, 260, 90, seed=44, empty_every=12)
    K, alpha, beta, seed = 14, 0.2, 0.05, 321
    flags = native.FLAG_PCGS if scheme == "pcgs" else 0
    a, o = make_pair(native, oracle, c, K, alpha, beta, seed, flags=flags, zseed=9)
    o.set_scheme(scheme)
    a.sweep(3)
    b = native.GGSHandle(K, c.num_types, alpha, beta, seed, flags=flags)
    b.set_corpus(c.doc_ptr, c.tokens)
    b.set_z(a.get_z(), redraw_phi=False)
    b.set_phi(a.get_phi())
    b.set_iteration(a.iteration)
    a.sweep(2)
    b.sweep(2)
    o.sweep(5)
    for h, tag in ((a, "uninterrupted"), (b, "resumed")):
        compare_state(h, o, tag, theta=scheme == "ggs")
