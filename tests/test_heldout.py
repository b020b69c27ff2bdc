"""Held-out log likelihood, MarginalProbEstimatorPlain.evaluateLeftToRight (MPE:85-121,123-519).

The reference has no test and no golden output for this estimator and draws from a clock-seeded Randoms: parity
unpinned against a JVM run.  What pins the oracle's restatement here: closed forms (one-token documents are exact;
a two-token document's exact marginal against many particles), and what pins the device: the oracle, bit for bit.
"""
import numpy as np
import pytest

from ldagroupedgibbssampler_amd.corpus import Corpus, random_corpus, synthetic_lda_corpus


def _trained_oracle(oracle, corpus, K, alpha, beta, seed, sweeps, threads=4):
    o = oracle.OracleSampler(K, corpus.num_types, alpha, beta, seed, threads=threads)
    o.set_corpus(corpus.doc_ptr, corpus.tokens)
    o.init_z_java_lcg(seed + 1)
    o.init_phi()
    if sweeps:
        o.sweep(sweeps)
    return o


def _docs(rows, num_types):
    ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    toks = np.array([t for r in rows for t in r], np.int32)
    return Corpus(ptr, toks, num_types)


# ---------------------------------------------------------------- the oracle against closed forms (CPU)
def test_one_token_documents_are_exact(oracle):
    """tokensSoFar = 0, no counts yet: wordProbability = sum_k alpha_k (beta + n_wk) / (n_k + betaSum) / alphaSum for
    every particle, whatever is drawn (MPE:75-78,352-365,399-401)."""
    c = random_corpus(60, 40, 30, seed=2)
    K, alpha, beta = 5, 0.3, 0.05
    o = _trained_oracle(oracle, c, K, alpha, beta, 11, 4)
    nwk, nk = o.get_type_topic_counts().astype(np.float64), o.get_topic_totals().astype(np.float64)
    test = _docs([[w] for w in range(c.num_types)], c.num_types)
    p = (alpha * (beta + nwk) / (nk + beta * c.num_types)).sum(1) / (alpha * K)
    tot1, ll1 = o.heldout_log_likelihood(test.doc_ptr, test.tokens, 1)
    np.testing.assert_allclose(ll1, np.log(p), rtol=1e-13)
    tot, ll = o.heldout_log_likelihood(test.doc_ptr, test.tokens, 100)
    np.testing.assert_allclose(ll, np.log(p), rtol=1e-12)
    assert tot == float(np.add.accumulate(ll)[-1])                       # the total is the running sum in document order


def test_two_token_document_against_the_exact_marginal(oracle):
    """p(w1, w2) = p(w1) * sum_z1 p(z1 | w1) sum_z2 (alpha + [z1 = z2]) / (alphaSum + 1) * phi_hat[z2][w2]; the estimator
    averages the inner sum over particles whose z1 is drawn from p(z1 | w1): unbiased, so 8000 particles land within 2 %.
    A wrong bucket walk (MPE:409-460) shows here."""
    c = random_corpus(80, 12, 25, seed=5)
    K, alpha, beta = 3, 0.4, 0.2
    o = _trained_oracle(oracle, c, K, alpha, beta, 3, 6)
    nwk, nk = o.get_type_topic_counts().astype(np.float64), o.get_topic_totals().astype(np.float64)
    phi_hat = (beta + nwk) / (nk + beta * c.num_types)                   # [V][K]
    pairs = [(0, 1), (3, 3), (7, 2), (11, 5)]
    test = _docs([list(p) for p in pairs], c.num_types)
    _, ll = o.heldout_log_likelihood(test.doc_ptr, test.tokens, 8000)
    for (w1, w2), got in zip(pairs, ll):
        p1k = alpha * phi_hat[w1]                                        # proportional to p(z1 = k, w1)
        p1 = p1k.sum() / (alpha * K)
        post = p1k / p1k.sum()
        p2 = sum(post[z1] * sum((alpha + (z1 == z2)) / (alpha * K + 1) * phi_hat[w2][z2] for z2 in range(K)) for z1 in range(K))
        assert abs(got - np.log(p1 * p2)) < 0.02, ((w1, w2), got, np.log(p1 * p2))


def test_oracle_out_of_vocabulary_and_threads(oracle):
    c = random_corpus(50, 30, 20, seed=8, empty_every=6)
    o = _trained_oracle(oracle, c, 4, 0.2, 0.1, 9, 3, threads=1)
    rows = [[1, 2, 3], [], [5, 30, 6, 999], [30]]                       # 30, 999: not in the training alphabet (MPE:341-345)
    test = _docs(rows, c.num_types)
    tot, ll = o.heldout_log_likelihood(test.doc_ptr, test.tokens, 10)
    assert ll[1] == 0.0 and ll[3] == 0.0
    same = _docs([[1, 2, 3], [], [5, 6], []], c.num_types)               # dropping the unknown words changes nothing:
    _, ll2 = o.heldout_log_likelihood(same.doc_ptr, same.tokens, 10)     # they take no draw and no tokensSoFar
    assert np.array_equal(ll, ll2)
    o.set_threads(4)
    assert o.heldout_log_likelihood(test.doc_ptr, test.tokens, 10)[0] == tot
    tot_b, ll_b = o.heldout_log_likelihood(test.doc_ptr, test.tokens, 10, doc_base=7)
    assert ll_b[0] != ll[0]                                              # another document index, another stream


def test_training_raises_the_heldout_likelihood(oracle):
    c = synthetic_lda_corpus(260, 300, 50, true_topics=8, seed=3)
    train, _, _ = c.shard(0, 220)
    test, _, _ = c.shard(220, 260)
    o = _trained_oracle(oracle, train, 8, 0.1, 0.01, 21, 0)
    before = o.heldout_log_likelihood(test.doc_ptr, test.tokens, 20)[0]
    o.sweep(40)
    after = o.heldout_log_likelihood(test.doc_ptr, test.tokens, 20)[0]
    assert after > before + 0.05 * abs(before), (before, after)


# ---------------------------------------------------------------- the device against the oracle (GPU), bit for bit
def _pair(native, oracle, corpus, K, alpha, beta, seed, sweeps):
    g = native.GGSHandle(K, corpus.num_types, alpha, beta, seed)
    g.set_corpus(corpus.doc_ptr, corpus.tokens)
    g.init_z_java_lcg(seed + 1)
    g.init_phi()
    o = _trained_oracle(oracle, corpus, K, alpha, beta, seed, 0)
    if sweeps:
        g.sweep(sweeps)
        o.sweep(sweeps)
    return g, o


def _same(g, o, test, particles, doc_base=0):
    g.set_test_corpus(test.doc_ptr, test.tokens, doc_base)
    gt, gl = g.heldout_log_likelihood(particles)
    ot, ol = o.heldout_log_likelihood(test.doc_ptr, test.tokens, particles, doc_base)
    bad = np.nonzero(gl.view(np.int64) != ol.view(np.int64))[0]
    assert bad.size == 0, (particles, bad[:5], gl[bad[:5]], ol[bad[:5]])
    assert gt == ot
    return gt


@pytest.mark.gpu
@pytest.mark.parametrize("K,alpha,beta,sweeps", [(3, 5.0, 7.0, 2), (20, 0.1, 0.01, 3), (64, 0.5, 0.1, 0), (100, 0.1, 0.01, 4), (200, 0.05, 0.01, 2),
                                                 (333, 0.1, 0.02, 1)])
def test_heldout_matches_oracle(native, oracle, K, alpha, beta, sweeps):
    c = random_corpus(260, 350, 90, seed=K, empty_every=17)
    train, _, _ = c.shard(0, 200)
    test, _, _ = c.shard(200, 260)
    g, o = _pair(native, oracle, train, K, alpha, beta, 40 + K, sweeps)
    for particles in (1, 64, 65, 100):
        _same(g, o, test, particles)
    a, b = g.heldout_log_likelihood(100), g.heldout_log_likelihood(100)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])                  # run-to-run identical
    # after another sweep the iteration (hence the stream) and the counts have moved, on both sides alike
    g.sweep(1)
    o.sweep(1)
    _same(g, o, test, 100)
    _same(g, o, test, 100, doc_base=12345)


@pytest.mark.gpu
def test_heldout_edge_cases(native, oracle, cats):
    K = 20
    g, o = _pair(native, oracle, cats, K, 5.0, 7.0, 2019, 2)
    V = cats.num_types
    rows = [[], [0], [V - 1, V, V + 5, 1, 2**31 - 1], [V], list(range(0, V, 3)), [], [7] * 300]
    _same(g, o, _docs(rows, V), 100)
    _same(g, o, _docs([], V), 100)                                       # no test documents: 0.0
    _same(g, o, cats, 100)                                               # the training set itself as test set
    g2 = native.GGSHandle(K, V, 5.0, 7.0, 1)
    g2.set_corpus(cats.doc_ptr, cats.tokens)
    g2.init_z_java_lcg(1)
    with pytest.raises(native.GGSError):
        g2.heldout_log_likelihood(100)                                   # no test set
    g2.set_test_corpus(np.array([0, 8193 * 2], np.int64), np.zeros(8193 * 2, np.int32))   # was refused (2 * GGS_MAX_BLOCKS uniforms per particle): the stream now has the counter's whole 24-bit block field
    with pytest.raises(native.GGSError):
        g2.set_test_corpus(np.array([0, 1], np.int64), np.array([-1], np.int32))
    g2.set_test_corpus(np.array([0, 2], np.int64), np.array([0, 1], np.int32))
    with pytest.raises(native.GGSError):
        g2.heldout_log_likelihood(0)


@pytest.mark.gpu
def test_heldout_through_the_python_mirror(oracle, cats):
    from ldagroupedgibbssampler_amd.sampler import SimpleLDAConfiguration, create_model
    train, _, _ = cats.shard(0, 18)
    test, _, _ = cats.shard(18, cats.num_docs)
    test = Corpus(test.doc_ptr, test.tokens, train.num_types)
    m = create_model(SimpleLDAConfiguration(topics=5, alpha=0.5, beta=0.1, seed=31, iterations=3, exec_time=1800))
    m.setRandomSeed(31)
    m.addInstances(train)
    with pytest.raises(ValueError):
        m.heldOutLogLikelihood()
    with pytest.raises(ValueError):
        m.addTestInstances(Corpus(test.doc_ptr, test.tokens, train.num_types + 1))
    m.addTestInstances(test)
    m.sample(3)
    o = oracle.OracleSampler(5, train.num_types, 0.5, 0.1, 31)
    o.set_corpus(train.doc_ptr, train.tokens)
    o.init_z_java_lcg(31)
    o.init_phi()
    o.sweep(3)
    assert m.heldOutLogLikelihood() == o.heldout_log_likelihood(test.doc_ptr, test.tokens, 100)[0]


@pytest.mark.gpu
def test_heldout_medium_slice(native, oracle):
    """A benchmark-shaped slice: K=100, V=50 000, 3 000 training and 400 test documents of ~200 tokens, 100 particles."""
    c = synthetic_lda_corpus(3400, 50000, 200, true_topics=100, seed=2019)
    train, _, _ = c.shard(0, 3000)
    test, _, _ = c.shard(3000, 3400)
    g = native.GGSHandle(100, c.num_types, 0.1, 0.01, 2019)
    g.set_corpus(train.doc_ptr, train.tokens)
    g.init_z_java_lcg(2019)
    g.init_phi()
    g.sweep(10)
    o = oracle.OracleSampler(100, c.num_types, 0.1, 0.01, 2019, threads=16)
    o.set_corpus(train.doc_ptr, train.tokens)
    o.set_z(g.get_z(), redraw_phi=False)                                 # same counts; the estimator reads nothing else
    o.set_iteration(10)
    _same(g, o, test, 100)


@pytest.mark.gpu
def test_heldout_wide_topic_rows(native, oracle):
    """K = 1024 (BASELINE config 3's width): the per-particle counts still fit LDS with a shallower coefficient table,
    for short and for long test documents; beyond 1024 topics only documents of at most 255 tokens do -- the longer ones,
    and everything beyond 1704 topics, keep the counts in global memory (heldout_particles_kernel<.., SPILL>)."""
    rng = np.random.default_rng(9)
    c = random_corpus(120, 400, 100, seed=61, empty_every=13)
    rows = [list(rng.integers(0, 400, int(n))) for n in (3, 40, 255, 256, 300, 0, 120)]
    test = _docs(rows, c.num_types)
    g, o = _pair(native, oracle, c, 1024, 0.05, 0.01, 17, 1)
    _same(g, o, test, 100)
    g, o = _pair(native, oracle, c, 1100, 0.05, 0.01, 18, 1)
    _same(g, o, _docs([r for r in rows if len(r) <= 255], c.num_types), 70)
    _same(g, o, test, 70)                                               # a 256-token document needs two-byte counts: too wide for LDS at K = 1100, spilled
    g, o = _pair(native, oracle, c, 2049, 0.05, 0.01, 19, 1)           # beyond 1704 topics every class is spilled
    _same(g, o, test, 40)


@pytest.mark.gpu
def test_heldout_long_test_documents(native, oracle):
    """Test documents beyond 8192 tokens (the old cap: 2 * GGS_MAX_BLOCKS uniforms of a particle's stream; now the
    counter's whole 24-bit block field) and beyond 65 535 (four-byte counts, in global memory)."""
    rng = np.random.default_rng(4)
    c = random_corpus(150, 200, 60, seed=3, empty_every=11)
    rows = [list(rng.integers(0, 200, int(n))) for n in (9000, 12, 70000, 0, 300)]
    test = _docs(rows, c.num_types)
    g, o = _pair(native, oracle, c, 12, 0.1, 0.02, 5, 1)
    _same(g, o, test, 6)


@pytest.mark.gpu
def test_heldout_in_several_batches(native, oracle, monkeypatch):
    """The word probabilities of a large test set are produced a batch of documents at a time (GGS_DEBUG_HELDOUT_CELLS
    shrinks the batch here; the default is 2^29 cells): same values, short and long documents mixed in every batch."""
    rng = np.random.default_rng(3)
    c = random_corpus(200, 300, 80, seed=31, empty_every=9)
    train, _, _ = c.shard(0, 150)
    rows = [list(rng.integers(0, 300, int(n))) for n in rng.integers(0, 120, 40)] + [list(rng.integers(0, 300, 400))] + \
           [list(rng.integers(0, 300, int(n))) for n in rng.integers(200, 300, 10)]
    test = _docs(rows, c.num_types)
    g, o = _pair(native, oracle, train, 30, 0.1, 0.02, 8, 2)
    whole = _same(g, o, test, 100)
    for cells in (1, 50000, 200000):
        monkeypatch.setenv("GGS_DEBUG_HELDOUT_CELLS", str(cells))
        assert _same(g, o, test, 100) == whole
    monkeypatch.delenv("GGS_DEBUG_HELDOUT_CELLS")


@pytest.mark.gpu
def test_heldout_sharded_test_set(native, oracle):
    """Three shards of the test documents, each with its doc_base (what ShardedGGS.set_test_corpus hands every rank):
    the per-document values concatenate to the one-handle result, so the ordered total is the same."""
    c = random_corpus(150, 200, 70, seed=77, empty_every=11)
    train, _, _ = c.shard(0, 110)
    test, _, _ = c.shard(110, 150)
    g, o = _pair(native, oracle, train, 24, 0.2, 0.05, 5, 2)
    whole = _same(g, o, test, 100)
    parts = []
    for lo, hi in ((0, 13), (13, 14), (14, 40)):
        sub, db, _ = test.shard(lo, hi)
        g.set_test_corpus(sub.doc_ptr, sub.tokens, db)
        parts.append(g.heldout_log_likelihood(100)[1])
    doc_ll = np.concatenate(parts)
    total = 0.0
    for v in doc_ll.tolist():
        total += v
    assert total == whole


@pytest.mark.gpu
def test_heldout_of_two_independent_chains_agree(native, oracle):
    """The pattern of the reference's MarginalProbEstimatorPlainTest (:36-92: two samplers' held-out log likelihoods
    within 10 % of each other), at the 1 % SURVEY 8c-5 asks for: the device chain and an oracle chain with ANOTHER seed
    (other z0, other Philox key) are trained on the same documents and evaluated on the same test documents."""
    c = synthetic_lda_corpus(1500, 800, 80, true_topics=16, seed=11)
    train, _, _ = c.shard(0, 1300)
    test, _, _ = c.shard(1300, 1500)
    K, alpha, beta, sweeps = 16, 0.1, 0.01, 80
    g = native.GGSHandle(K, c.num_types, alpha, beta, 1001)
    g.set_corpus(train.doc_ptr, train.tokens)
    g.init_z_java_lcg(1)
    g.init_phi()
    g.set_test_corpus(test.doc_ptr, test.tokens)
    start = g.heldout_log_likelihood(100)[0]
    g.sweep(sweeps)
    dev = g.heldout_log_likelihood(100)[0]
    o = _trained_oracle(oracle, train, K, alpha, beta, 2002, sweeps, threads=8)
    cpu = o.heldout_log_likelihood(test.doc_ptr, test.tokens, 100)[0]
    assert dev > start + 0.03 * abs(start), (start, dev)                 # training helped
    assert abs(dev - cpu) <= 0.01 * abs(cpu), (dev, cpu)
