"""scheme=collapsed on the device: the count-form conditional of ModifiedSimpleLDA.sampleTopicsForOneDoc (MSLDA:158-226,
the in-tree twin of what SerialCollapsedLDA runs), SURVEY.md 8(a) row 9 / 8(f) item 4.

  serial schedule     ggs_collapsed_serial_sweep == orc_collapsed_sweep bit for bit: the reference's own chain (counts in
                      place, the sampler's one java.util.Random stream continuing from the seeded initial topics) --
                      BASELINE config 1 (the bundled cats corpus, K = 20) among the cases
  parallel schedule   ggs_sweep with GGS_FLAG_COLLAPSED == orc_collapsed_parallel_sweep bit for bit: the SAME schedule
                      restated on the CPU (documents side by side on sweep-start counts, AD-LDA style).  Against the
                      serial chain that schedule is a different, approximate sampler: the north_star's "otherwise"
                      clause applies, held-out log likelihood within +-1 %.
Parity against a JVM run: unpinned, as for every path (no JVM, no fixture in the reference; tests/test_oracle_cpu.py)."""
import threading

import numpy as np
import pytest

from ldagroupedgibbssampler_amd.corpus import even_split, random_corpus, synthetic_lda_corpus
from tests.test_native_exchange_gpu import ThreadTransport, assert_bit_equal

pytestmark = pytest.mark.gpu


def pair(native, oracle, c, K, alpha, beta, seed, zseed):
    g = native.GGSHandle(K, c.num_types, alpha, beta, seed, flags=native.FLAG_COLLAPSED | native.FLAG_PARANOID)
    o = oracle.OracleSampler(K, c.num_types, alpha, beta, seed, threads=4)
    for s in (g, o):
        s.set_corpus(c.doc_ptr, c.tokens)
        s.init_z_java_lcg(zseed)
        s.init_phi()
    return g, o


def same_counts(g, o, tag):
    assert_bit_equal(g.get_z(), o.get_z(), tag + " z")
    assert_bit_equal(g.get_type_topic_counts(), o.get_type_topic_counts(), tag + " n_wk")
    assert_bit_equal(g.get_topic_totals(), o.get_topic_totals(), tag + " n_k")
    assert_bit_equal(g.get_doc_topic_counts(), o.get_doc_topic_counts(), tag + " n_dk")


@pytest.mark.parametrize("K", [3, 20])
def test_serial_chain_on_cats_is_the_java_loop(native, oracle, cats, K):
    """plda-cats-test.cfg:18-25 (alpha 5, beta 7, seed 2019) with K = 3 as in the file and K = 20 as BASELINE config 1 asks."""
    g, o = pair(native, oracle, cats, K, 5.0, 7.0, 2019, 2019)
    for it in range(3):
        g.collapsed_serial_sweep(2019, 1)
        o.collapsed_sweep(2019, 1)
        same_counts(g, o, "cats K=%d serial sweep %d" % (K, it + 1))
    g.collapsed_serial_sweep(2019, 4)                       # several sweeps in one call carry the stream on
    o.collapsed_sweep(2019, 4)
    same_counts(g, o, "cats K=%d serial sweep 7" % K)
    g.check_invariants()
    phi = g.get_phi()                                       # the point estimate (beta + n_wk)/(betaSum + n_k)
    nwk, nk = o.get_type_topic_counts().astype(np.float64), o.get_topic_totals().astype(np.float64)
    assert_bit_equal(phi, ((7.0 + nwk) / (7.0 * cats.num_types + nk)).T, "phi point estimate")


@pytest.mark.parametrize("K,alpha,beta", [(1, 0.5, 0.1), (7, 0.1, 0.01), (64, 0.05, 0.01), (100, 0.1, 0.01), (130, 1.5, 0.5)])
def test_serial_chain_on_ragged_corpora(native, oracle, K, alpha, beta):
    c = random_corpus(60, 200, 50, seed=K, empty_every=7)
    g, o = pair(native, oracle, c, K, alpha, beta, 11, K)
    g.collapsed_serial_sweep(0, 3)
    o.collapsed_sweep(0, 3)
    same_counts(g, o, "ragged K=%d serial" % K)
    # a state set with set_z starts a NEW Random(seed) at the first serial sweep
    z = g.get_z()
    g2 = native.GGSHandle(K, c.num_types, alpha, beta, 11, flags=native.FLAG_COLLAPSED)
    o2 = oracle.OracleSampler(K, c.num_types, alpha, beta, 11)
    for s in (g2, o2):
        s.set_corpus(c.doc_ptr, c.tokens)
        s.set_z(z, redraw_phi=True)
    g2.collapsed_serial_sweep(77, 2)
    o2.collapsed_sweep(77, 2)
    same_counts(g2, o2, "ragged K=%d serial after set_z" % K)


def test_set_z_after_the_seeded_start_begins_a_new_random_stream(native, oracle):
    """include/ggs_hip.h: after ggs_set_z the serial chain creates a new Random(java_seed) at its first call -- also when
    the handle had been started with ggs_init_z_java_lcg before (the stale stream of the seeded start must not go on)."""
    c = random_corpus(40, 120, 30, seed=2, empty_every=6)
    K = 9
    g, o = pair(native, oracle, c, K, 0.3, 0.05, 21, 5)          # init_z_java_lcg(5): the handle owns a stream now
    g.collapsed_serial_sweep(5, 1)
    o.collapsed_sweep(5, 1)
    z = np.random.default_rng(0).integers(0, K, c.num_tokens).astype(np.int32)
    g.set_z(z, redraw_phi=True)
    o2 = oracle.OracleSampler(K, c.num_types, 0.3, 0.05, 21)
    o2.set_corpus(c.doc_ptr, c.tokens)
    o2.set_z(z, redraw_phi=True)
    g.collapsed_serial_sweep(123, 2)                             # java_seed 123 is looked at: a fresh Random(123)
    o2.collapsed_sweep(123, 2)
    same_counts(g, o2, "serial chain after set_z on a seeded handle")


@pytest.mark.parametrize("K,alpha,beta", [(3, 5.0, 7.0), (7, 0.1, 0.01), (20, 5.0, 7.0), (64, 0.05, 0.01), (100, 0.1, 0.01), (200, 0.1, 0.01), (333, 0.1, 0.01)])
def test_parallel_schedule_matches_its_restatement(native, oracle, cats, K, alpha, beta):
    c = cats if K in (3, 20) else random_corpus(301, 500, 150, seed=K, empty_every=7)
    g, o = pair(native, oracle, c, K, alpha, beta, 42 + K, K)
    for it in range(3):
        g.sweep(1)
        o.collapsed_parallel_sweep(1)
        same_counts(g, o, "parallel K=%d sweep %d" % (K, it + 1))
    g.sweep_begin()
    g.sweep_end()
    o.collapsed_parallel_sweep(1)
    same_counts(g, o, "parallel K=%d split sweep" % K)
    with pytest.raises(native.GGSError) as e:
        g._chk(g._L.ggs_sample_z_given_phi(g._h, 1))
    assert e.value.code == native.ERR_UNSUPPORTED


@pytest.mark.parametrize("K,wave", [(96, None), (97, None), (130, "0"), (192, "0")])
def test_parallel_schedule_either_kernel_around_the_switch_point(native, oracle, monkeypatch, K, wave):
    """scheme=collapsed takes the wave-per-document kernel from 97 topics on (ggs_api.hip: the measured break-even); the
    lane-per-document variants above stay reachable (GGS_DEBUG_PCGS_WAVE=0) and give the same counts."""
    if wave is not None:
        monkeypatch.setenv("GGS_DEBUG_PCGS_WAVE", wave)
    c = random_corpus(130, 300, 90, seed=K + 1, empty_every=7)
    g, o = pair(native, oracle, c, K, 0.1, 0.01, 6 + K, K)
    monkeypatch.delenv("GGS_DEBUG_PCGS_WAVE", raising=False)
    want = "wave" if (wave is None and K > 96) else "lane"
    assert want in g.launch_info()["z_kernel"]
    for it in range(2):
        g.sweep(1)
        o.collapsed_parallel_sweep(1)
        same_counts(g, o, "parallel K=%d (%s per document) sweep %d" % (K, want, it + 1))


@pytest.mark.parametrize("K,force", [(500, False), (1024, False), (2049, False), (20, True), (130, True)])
def test_parallel_schedule_wide_topic_rows(native, oracle, monkeypatch, K, force):
    """The count-form conditional (MSLDA:158-226) in the parallel schedule above 320 topics, and with a document of more
    than 32 767 tokens: the wave-per-document kernel (ggs_z_pcgs_wave.hpp) over psi, the own-topic entry recomputed."""
    if force:
        monkeypatch.setenv("GGS_DEBUG_PCGS_WAVE", "1")
    c = random_corpus(120, 400, 80, seed=K, empty_every=7)
    g, o = pair(native, oracle, c, K, 0.1, 0.01, 4 + K, K)
    monkeypatch.delenv("GGS_DEBUG_PCGS_WAVE", raising=False)
    for it in range(2):
        g.sweep(1)
        o.collapsed_parallel_sweep(1)
        same_counts(g, o, "parallel wide K=%d sweep %d" % (K, it + 1))


def _shard_rank(native, tr, rank, world, whole, K, zseed, sweeps, out, errs):
    import torch
    from ldagroupedgibbssampler_amd.sharded import _DevPtr, java_lcg_initial_z
    try:
        dev = torch.device("cuda", 0)

        def view(ptr, n, typestr):
            return torch.as_tensor(_DevPtr(ptr, n, typestr), device=dev)

        def reduce_scatter_i32(send, recv, count, stream):
            torch.cuda.synchronize()
            parts = tr.exchange(rank, view(send, count * world, "<i4").cpu().numpy().reshape(world, count))
            view(recv, count, "<i4").copy_(torch.from_numpy(np.sum([p[rank] for p in parts], axis=0, dtype=np.int32)))
            torch.cuda.synchronize()
            return 0

        def all_gather(typestr):
            def cb(send, recv, count, stream):
                torch.cuda.synchronize()
                parts = tr.exchange(rank, view(send, count, typestr).cpu().numpy())
                view(recv, count * world, typestr).copy_(torch.from_numpy(np.concatenate(parts)))
                torch.cuda.synchronize()
                return 0
            return cb

        b = even_split(whole.num_docs, world)
        sub, doc_base, tok_base = whole.shard(b[rank], b[rank + 1])
        h = native.GGSHandle(K, whole.num_types, 0.1, 0.01, 5, flags=native.FLAG_COLLAPSED)
        h.attach_exchange(rank, world, reduce_scatter_i32, all_gather("<f8"), all_gather("<i4"))
        h.set_corpus(sub.doc_ptr, sub.tokens, doc_base, tok_base)
        h.set_global_token_count(whole.num_tokens)
        h.set_z(java_lcg_initial_z(whole.num_tokens, K, zseed)[tok_base:tok_base + sub.num_tokens], redraw_phi=True)
        h.sweep(sweeps)
        h.check_invariants()
        out[rank] = dict(z=h.get_z(), nwk=h.get_type_topic_counts(), nk=h.get_topic_totals())
        h.close()
    except BaseException as e:  # noqa: BLE001
        errs.append(e)
        tr.bar.abort()


def test_parallel_schedule_doc_sharded_is_ad_lda_with_a_merge_per_sweep(native, oracle):
    """Two doc shards joined by the exchange (the count slices are reduce-scattered and gathered: the AD-LDA merge,
    ADLDA.java:302-332) sample exactly what one handle samples: the schedule already conditions on sweep-start counts."""
    from ldagroupedgibbssampler_amd.sharded import java_lcg_initial_z
    whole = random_corpus(260, 400, 120, seed=9, empty_every=8)
    K, world, sweeps = 24, 2, 3
    tr, out, errs = ThreadTransport(world), [None] * world, []
    ts = [threading.Thread(target=_shard_rank, args=(native, tr, r, world, whole, K, 3, sweeps, out, errs)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    o = oracle.OracleSampler(K, whole.num_types, 0.1, 0.01, 5, threads=4)
    o.set_corpus(whole.doc_ptr, whole.tokens)
    o.set_z(java_lcg_initial_z(whole.num_tokens, K, 3), redraw_phi=True)
    o.collapsed_parallel_sweep(sweeps)
    assert_bit_equal(np.concatenate([p["z"] for p in out]), o.get_z(), "sharded collapsed z")
    for p in out:
        assert_bit_equal(p["nwk"], o.get_type_topic_counts(), "sharded collapsed n_wk")
        assert_bit_equal(p["nk"], o.get_topic_totals(), "sharded collapsed n_k")


def test_parallel_schedule_heldout_within_one_percent_of_the_serial_chain(native, oracle):
    """The north_star's "otherwise" clause: the device's parallel collapsed chain against the reference's serial chain
    (restated), different random streams, on the 1 000-document K=100 slice of the benchmark corpus with 100 held-out
    documents: left-to-right held-out log likelihood (MarginalProbEstimatorPlain, 100 particles) within +-1 %."""
    full = synthetic_lda_corpus(1100, 50000, 200, true_topics=100, seed=2019)
    train, _, _ = full.shard(0, 1000)
    test, _, _ = full.shard(1000, 1100)
    K, sweeps = 100, 60
    g = native.GGSHandle(K, train.num_types, 0.1, 0.01, 2019, flags=native.FLAG_COLLAPSED)
    g.set_corpus(train.doc_ptr, train.tokens)
    g.init_z_java_lcg(2019)
    g.init_phi()
    g.sweep(sweeps)
    g.set_test_corpus(test.doc_ptr, test.tokens)
    ll_dev = g.heldout_log_likelihood(100)[0]
    o = oracle.OracleSampler(K, train.num_types, 0.1, 0.01, 2019, threads=8)
    o.set_corpus(train.doc_ptr, train.tokens)
    o.init_z_java_lcg(2019)
    o.collapsed_sweep(2019, sweeps)
    ll_ref = o.heldout_log_likelihood(test.doc_ptr, test.tokens, 100)[0]
    assert ll_dev < 0 and ll_ref < 0
    assert abs(ll_dev - ll_ref) <= 0.01 * abs(ll_ref), (ll_dev, ll_ref)


def test_python_mirror_serial_collapsed_lda(native, oracle, cats):
    """create_model(config, "collapsed") -> SerialCollapsedLDA mirror: setRandomSeed / addInstances / sample as
    tui/ParallelLDA drives them (ParallelLDA.java:173-202); the serial schedule is the restated Java chain."""
    from ldagroupedgibbssampler_amd.sampler import SimpleLDAConfiguration, create_model
    cfg = SimpleLDAConfiguration(scheme="collapsed", topics=20, alpha=5.0, beta=7.0, seed=2019, iterations=5, exec_time=1800, start_diagnostic=1,
                                 compute_likelihood=True)
    m = create_model(cfg)
    m.setRandomSeed(2019)
    m.addInstances(cats)
    m.sample(5)
    o = oracle.OracleSampler(20, cats.num_types, 5.0, 7.0, 2019)
    o.set_corpus(cats.doc_ptr, cats.tokens)
    o.init_z_java_lcg(2019)
    o.collapsed_sweep(2019, 5)
    assert_bit_equal(np.concatenate(m.getZIndicators()), o.get_z(), "mirror z")
    assert_bit_equal(np.asarray(m.getTypeTopicMatrix()), o.get_type_topic_counts(), "mirror n_wk")
    assert len(m.loglikelihood) == 5 and m.logPosterior == []
    ref = sum(o.model_log_likelihood())
    assert abs(m.loglikelihood[-1] - ref) <= 1e-9 * abs(ref)


def test_parallel_schedule_two_pass_kernel_below_193_topics(native, oracle, monkeypatch):
    """K <= 192 normally keeps the token's scores in registers (pcgs_sliced_kernel<KMAX, true>); GGS_DEBUG_PCGS_STREAM=1
    forces the two-pass kernel that larger K use (pcgs_z_kernel<true>) -- same draws either way."""
    monkeypatch.setenv("GGS_DEBUG_PCGS_STREAM", "1")
    c = random_corpus(200, 300, 90, seed=4, empty_every=6)
    g, o = pair(native, oracle, c, 50, 0.1, 0.01, 8, 2)
    monkeypatch.delenv("GGS_DEBUG_PCGS_STREAM")
    g.sweep(3)
    o.collapsed_parallel_sweep(3)
    same_counts(g, o, "collapsed two-pass K=50")


def test_collapsed_and_uncollapsed_agree_on_state_and_likelihood(native, cats):
    """The reference's LogLikelihoodTest.testLogLikelihood (LogLikelihoodTest.java:29-131), on the bundled cats corpus in
    place of the missing nips.txt: a collapsed and an uncollapsed sampler given the same seed start from the same topic
    indicators and counts and report the same model log likelihood; topic indicators carried over from one to the other
    (getZIndicators -> setZIndicators) reproduce the counts and the likelihood, in both directions, after each has sampled."""
    K, alpha, beta, seed = 20, 1.0 / 20, 0.01, 4711
    col = native.GGSHandle(K, cats.num_types, alpha, beta, seed, flags=native.FLAG_COLLAPSED | native.FLAG_PARANOID)
    unc = native.GGSHandle(K, cats.num_types, alpha, beta, seed, flags=native.FLAG_PARANOID)
    for s in (col, unc):
        s.set_corpus(cats.doc_ptr, cats.tokens)
        s.init_z_java_lcg(seed)
        s.init_phi()

    def same_state(tag):
        assert_bit_equal(col.get_z(), unc.get_z(), tag + " z")
        assert_bit_equal(col.get_type_topic_counts(), unc.get_type_topic_counts(), tag + " type-topic counts")
        assert_bit_equal(col.get_topic_totals(), unc.get_topic_totals(), tag + " topic totals")
        a, b = col.model_log_likelihood(), unc.model_log_likelihood()
        assert a == b, (tag, a, b)                         # the Java test's epsilon is 1e-33: equality

    same_state("start")
    pc = native.GGSHandle(K, cats.num_types, alpha, beta, seed, flags=native.FLAG_PCGS)      # TestInitialization.testEqualInitialization
    pc.set_corpus(cats.doc_ptr, cats.tokens)                                                # (TestInitialization.java:10-200): every scheme
    pc.init_z_java_lcg(seed)                                                                # starts from the same indicators, counts, likelihood
    pc.init_phi()
    assert_bit_equal(pc.get_z(), col.get_z(), "pcgs start z")
    assert_bit_equal(pc.get_type_topic_counts(), col.get_type_topic_counts(), "pcgs start counts")
    assert pc.model_log_likelihood() == col.model_log_likelihood()
    pc.close()
    col.collapsed_serial_sweep(seed, 50)                   # "sample 50 iterations ... to something other than the start state"
    unc.set_z(col.get_z(), redraw_phi=True)
    same_state("collapsed -> uncollapsed")
    unc.sweep(5)                                           # "sample 5 iterations to change z"
    col.set_z(unc.get_z(), redraw_phi=True)
    same_state("uncollapsed -> collapsed")


def test_uncollapsed_heldout_within_ten_percent_of_adlda(native):
    """The reference's MarginalProbEstimatorPlainTest (MarginalProbEstimatorPlainTest.java:36-94): after 100 iterations the
    left-to-right held-out estimate (100 particles) from the uncollapsed sampler's counts lies within 10 % of the one
    from ADLDA's.  Here: scheme=ggs against the parallel collapsed schedule (the AD-LDA decomposition), K = 20,
    alphaSum = 1, beta = 0.01 as in the Java test, on a slice of the benchmark corpus in place of the missing nips.txt."""
    full = synthetic_lda_corpus(660, 5000, 120, true_topics=20, seed=4711)
    train, _, _ = full.shard(0, 600)
    test, _, _ = full.shard(600, 660)
    K, alpha, beta = 20, 1.0 / 20, 0.01
    out = {}
    for name, flags in (("uncollapsed", 0), ("adlda", native.FLAG_COLLAPSED)):
        g = native.GGSHandle(K, train.num_types, alpha, beta, 4711, flags=flags)
        g.set_corpus(train.doc_ptr, train.tokens)
        g.init_z_java_lcg(4711)
        g.init_phi()
        g.sweep(100)
        g.set_test_corpus(test.doc_ptr, test.tokens)
        out[name] = g.heldout_log_likelihood(100)[0]
        g.close()
    a, b = out["uncollapsed"], out["adlda"]
    assert a < 0 and b < 0 and abs(a - b) <= 0.1 * abs(b), out
