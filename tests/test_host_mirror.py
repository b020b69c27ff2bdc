"""The host-side mirrors of the reference interface (Python: sampler.py, C++: ggs_sampler.hpp)."""
import json
import os
import subprocess

import numpy as np
import pytest

from ldagroupedgibbssampler_amd.corpus import random_corpus
from ldagroupedgibbssampler_amd.sampler import (LDAGroupedGibbsSampler, SimpleLDAConfiguration, calc_theta_estimate, calc_zbar,
                                                create_model)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_vectors.json")))["modified_simple_lda_test"]


# ---- the reference's own known answers: src/test/java/cc/mallet/topics/ModifiedSimpleLDATest.java
def _counts():
    return np.bincount(KAT["oneDocTopics"], minlength=KAT["numTopics"]).astype(np.float64)


def test_theta_estimate_symmetric_alpha():              # testThetaEstimate, :29-44
    est = calc_theta_estimate(KAT["numTopics"], KAT["alpha"], KAT["oneDocTopics"])
    alpha_sum = 0.0
    for k in range(KAT["numTopics"]):
        alpha_sum += _counts()[k] + KAT["alpha"]
    assert (est != 0).all()
    assert np.array_equal(est, (_counts() + KAT["alpha"]) / alpha_sum)       # assertTrue(a == b): exact
    assert abs(est.sum() - 1.0) < 1e-10


def test_theta_estimate_asymmetric_alpha():             # testThetaEstimateNonSymAlpha / testThetaEstimates, :46-56,:94-104
    alphas = np.array(KAT["alphas"])
    est = calc_theta_estimate(KAT["numTopics"], alphas, KAT["oneDocTopics"])
    alpha_sum = 0.0
    for k in range(KAT["numTopics"]):
        alpha_sum += _counts()[k] + alphas[k]
    assert (est != 0).all() and np.array_equal(est, (_counts() + alphas) / alpha_sum)
    assert abs(est.sum() - 1.0) < 1e-10


def test_theta_estimate_empty_document():               # testThetaEstimate*EmptyDoc, :58-92
    alphas = np.array(KAT["alphas"])
    est = calc_theta_estimate(KAT["numTopics"], alphas, [])
    s = 0.0
    for a in alphas:
        s += a
    assert np.array_equal(est, alphas / s) and (est != 0).all()
    est = calc_theta_estimate(KAT["numTopics"], KAT["alpha"], [])
    assert np.array_equal(est, np.full(5, KAT["alpha"] / (5 * KAT["alpha"]))) or np.allclose(est, 0.2, atol=1e-15)
    assert abs(est.sum() - 1.0) < 1e-10


def test_zbar():                                        # testZbar, :106-113
    zb = calc_zbar(KAT["numTopics"], KAT["oneDocTopics"])
    assert abs(zb.sum() - 1.0) < 1e-10
    assert np.array_equal(zb, _counts() / 10)
    assert np.array_equal(calc_zbar(5, []), np.zeros(5))                    # docLength == 0 -> zeros, MSLDA:657-658


def test_configuration_defaults_and_registry():
    c = SimpleLDAConfiguration()
    assert (c.topics, c.alpha, c.beta, c.iterations, c.exec_time) == (10, 5.0, 0.01, 1500, 10)   # LDAConfiguration.java:11-17,35
    assert c.scheme == "ggs" and c.seed == 0 and c.get_seed() != 0        # seed 0 -> clock, ParsedLDAConfiguration.java:137-141
    with pytest.raises(TypeError):
        SimpleLDAConfiguration(no_such_key=1)
    m = create_model(SimpleLDAConfiguration(topics=3, seed=5))
    assert isinstance(m, LDAGroupedGibbsSampler) and m.getNoTopics() == 3 and m.getStartSeed() == 5
    with pytest.raises(ValueError):
        create_model(SimpleLDAConfiguration(), "adlda")
    assert type(create_model(SimpleLDAConfiguration(), "collapsed")).__name__ == "SerialCollapsedLDA"   # ParallelLDA.java:424-428
    with pytest.raises(RuntimeError):
        m.sample(1)                                      # before addInstances


# ---- on the GPU: the mirrors drive the C-ABI
@pytest.mark.gpu
def test_python_mirror_matches_oracle(oracle, cats, tmp_path, monkeypatch):
    cfg = SimpleLDAConfiguration(topics=3, alpha=5.0, beta=7.0, seed=2019, iterations=4, exec_time=1800, paranoid=True,
                                 save_phi_mean=True, phi_mean_burnin=25, phi_mean_thin=1)           # plda-cats-test.cfg:16-25
    m = create_model(cfg)
    m.setRandomSeed(2019)
    m.addInstances(cats)
    calls = []
    m.preIteration = lambda: calls.append("pre")
    m.postPhi = lambda: calls.append("phi")
    m.sample(4)
    assert calls == ["pre", "phi"] * 4 and m.getCurrentIteration() == 4
    o = oracle.OracleSampler(3, cats.num_types, 5.0, 7.0, 2019)
    o.set_phi_mean_gating(True, 1, 1)
    o.set_corpus(cats.doc_ptr, cats.tokens)
    o.init_z_java_lcg(2019)
    o.init_phi()
    o.sweep(4)
    assert np.array_equal(np.concatenate(m.getZIndicators()), o.get_z())
    assert np.array_equal(m.getTypeTopicMatrix(), o.get_type_topic_counts())
    assert np.array_equal(m.getTopicTotals(), o.get_topic_totals())
    assert np.array_equal(m.getDocumentTopicMatrix(), o.get_doc_topic_counts())
    assert np.array_equal(m.getPhi().view(np.int64), o.get_phi().view(np.int64))
    assert np.array_equal(m.getTheta().view(np.int64), o.get_theta().view(np.int64))
    assert m.getNoSampledPhi() == 3 and np.array_equal(m.getPhiMeans().view(np.int64), o.get_phi_mean()[0].view(np.int64))
    # z-bar / theta estimate against the per-document helpers
    zs = m.getZIndicators()
    assert np.array_equal(m.getZbar()[0], calc_zbar(3, zs[0]))
    assert np.array_equal(m.getThetaEstimate()[5], calc_theta_estimate(3, 5.0, zs[5]))
    assert m.zSamplingTimeCum > 0 and m.phiSamplingTimeCum > 0
    # setZIndicators round trip + its length check (UPLDA:1828-1830)
    m.setZIndicators(zs)
    assert np.array_equal(m.getTypeTopicMatrix(), o.get_type_topic_counts())
    with pytest.raises(ValueError):
        m.setZIndicators(zs[:-1])
    # exec_time budget: a zero budget stops after the first iteration (UPLDA:926-928)
    m2 = create_model(SimpleLDAConfiguration(topics=3, alpha=5.0, beta=7.0, seed=1, exec_time=0))
    m2.addInstances(cats)
    m2.sample(50)
    assert m2.getCurrentIteration() == 1
    # abort flag checked per iteration (UPLDA:645)
    m3 = create_model(SimpleLDAConfiguration(topics=3, alpha=5.0, beta=7.0, seed=1, exec_time=1800))
    m3.addInstances(cats)
    m3.postIteration = lambda: m3.abort() if m3.getCurrentIteration() >= 2 else None
    m3.sample(50)
    assert m3.getAbort() and m3.getCurrentIteration() == 2
    # the `abort` sentinel file in the working directory (UPLDA:131,908-910)
    m4 = create_model(SimpleLDAConfiguration(topics=3, alpha=5.0, beta=7.0, seed=1, exec_time=1800))
    m4.addInstances(cats)
    monkeypatch.chdir(tmp_path)
    m4.postIteration = lambda: open("abort", "w").close() if m4.getCurrentIteration() >= 3 else None
    m4.sample(50)
    assert m4.getAbort() and m4.getCurrentIteration() == 3


@pytest.mark.gpu
def test_cpp_mirror_matches_python_path(native, tmp_path):
    exe = os.path.join(ROOT, "examples", "ggs_host_demo")
    if not os.path.exists(exe):
        pytest.fail("examples/ggs_host_demo is not built: run __graft_entry__.build()")
    c = random_corpus(60, 90, 70, seed=12, empty_every=8)
    path = os.path.join(str(tmp_path), "corpus.txt")
    with open(path, "w") as f:
        f.write("%d %d\n" % (c.num_docs, c.num_types))
        for d in range(c.num_docs):
            t = c.tokens[c.doc_ptr[d]:c.doc_ptr[d + 1]]
            f.write(" ".join([str(len(t))] + [str(int(x)) for x in t]) + "\n")
    K, alpha, beta, seed, its = 6, 0.5, 0.1, 99, 3
    out = subprocess.run([exe, path, str(K), str(alpha), str(beta), str(seed), str(its)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    g = native.GGSHandle(K, c.num_types, alpha, beta, seed)
    g.set_corpus(c.doc_ptr, c.tokens)
    g.init_z_java_lcg(seed)
    g.init_phi()
    g.sweep(its)
    assert lines["iteration"] == "%d hooks %d %d" % (its, its, its)
    assert np.array_equal(np.array(lines["z"].split(), np.int32), g.get_z())
    assert np.array_equal(np.array(lines["nk"].split(), np.int32), g.get_topic_totals())
    assert abs(float(lines["theta_estimate_doc0_sum"]) - 1.0) < 1e-12
    g.set_test_corpus(c.doc_ptr, c.tokens)
    assert float(lines["heldout"]) == g.heldout_log_likelihood(100)[0]
    assert float(lines["loglik"]) == sum(g.model_log_likelihood())
    assert float(lines["logposterior"]) == sum(g.log_posterior())


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["ggs", "collapsed"])
def test_cpp_mirror_writes_the_java_drivers_files(native, tmp_path, scheme):
    """The C++ mirror with a log_dir (include/ggs_sampler.hpp + ggs_formats.hpp) leaves the same files as the Python
    mirror: log-likelihood.txt, test_held_out_log_likelihood.txt, z_<iteration>.csv byte for byte, log-posterior.txt
    up to its wall-clock column (LDAUtils.java:928-979; UPLDA:945-968)."""
    exe = os.path.join(ROOT, "examples", "ggs_host_demo")
    c = random_corpus(40, 70, 50, seed=3, empty_every=6)
    path = os.path.join(str(tmp_path), "corpus.txt")
    with open(path, "w") as f:
        f.write("%d %d\n" % (c.num_docs, c.num_types))
        for d in range(c.num_docs):
            t = c.tokens[c.doc_ptr[d]:c.doc_ptr[d + 1]]
            f.write(" ".join([str(len(t))] + [str(int(x)) for x in t]) + "\n")
    cdir, pdir = tmp_path / "cpp", tmp_path / "py"
    cdir.mkdir()
    pdir.mkdir()
    K, alpha, beta, seed, its = 5, 0.5, 0.1, 7, 3
    out = subprocess.run([exe, path, str(K), str(alpha), str(beta), str(seed), str(its), str(cdir), scheme], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = create_model(SimpleLDAConfiguration(scheme=scheme, topics=K, alpha=alpha, beta=beta, seed=seed, iterations=its, exec_time=1800, paranoid=True,
                                            compute_likelihood=True, start_diagnostic=1, log_topic_indicators=True, log_dir=str(pdir)))
    m.setRandomSeed(seed)
    m.addInstances(c)
    m.addTestInstances(c)
    m.sample(its)
    names = sorted(os.listdir(pdir))
    assert names == sorted(os.listdir(cdir))
    assert ("log-posterior.txt" in names) == (scheme == "ggs") and "z_3.csv" in names and "log-likelihood.txt" in names
    for n in names:
        a, b = (cdir / n).read_text(), (pdir / n).read_text()
        if n == "log-posterior.txt":
            a, b = ["\t".join(l.split("\t")[:2]) for l in a.splitlines()], ["\t".join(l.split("\t")[:2]) for l in b.splitlines()]
        assert a == b, n


def test_model_log_likelihood_formula():
    """UPLDA:1644-1758 against a direct evaluation of the Dirichlet-multinomial formula
    (UPLDA:1653-1659) on a tiny hand-made state."""
    from math import lgamma
    from ldagroupedgibbssampler_amd.sampler import model_log_likelihood
    n_dk = np.array([[2, 0, 1], [0, 3, 0]])
    n_wk = np.array([[1, 2, 0], [1, 0, 1], [0, 1, 0], [0, 0, 0]])
    n_k = n_wk.sum(0)
    alpha, beta = np.array([0.5, 0.25, 1.5]), 0.1
    K, V = 3, 4
    ll = 0.0
    for d in range(2):          # documents: logG(sum a) - logG(sum a + N_d) + sum_k [logG(a_k + n_dk) - logG(a_k)]
        ll += lgamma(alpha.sum()) - lgamma(alpha.sum() + n_dk[d].sum())
        ll += sum(lgamma(alpha[k] + n_dk[d, k]) - lgamma(alpha[k]) for k in range(K))
    for k in range(K):          # topics: logG(V b) - logG(V b + n_k) + sum_w [logG(b + n_wk) - logG(b)]
        ll += lgamma(V * beta) - lgamma(V * beta + n_k[k])
        ll += sum(lgamma(beta + n_wk[w, k]) - lgamma(beta) for w in range(V))
    assert abs(model_log_likelihood(n_dk, n_wk, n_k, alpha, beta) - ll) < 1e-10


@pytest.mark.gpu
def test_log_likelihood_within_one_percent_across_rng_streams(oracle):
    """The north star's "otherwise" clause: with DIFFERENT random streams the HIP sampler and the
    CPU restatement are the same Markov chain, so after burn-in their model log-likelihoods agree
    within 1 % (same stream => bit-identical, tested elsewhere)."""
    from ldagroupedgibbssampler_amd.corpus import synthetic_lda_corpus
    from ldagroupedgibbssampler_amd.sampler import model_log_likelihood
    c = synthetic_lda_corpus(400, 800, 80, true_topics=8, seed=3, topic_conc=0.05, doc_conc=0.2)
    K, alpha, beta, its = 8, 0.2, 0.05, 200     # independent CPU chains agree to ~0.2 % by 200 sweeps (2-3 % at 60)
    m = create_model(SimpleLDAConfiguration(topics=K, alpha=alpha, beta=beta, seed=111, iterations=its, exec_time=1800))
    m.addInstances(c)
    ll0 = m.modelLogLikelihood()
    m.sample(its)
    ll_hip = m.modelLogLikelihood()
    o = oracle.OracleSampler(K, c.num_types, alpha, beta, 987654321, threads=4)     # another Philox key, another z0
    o.set_corpus(c.doc_ptr, c.tokens)
    o.init_z_java_lcg(222)
    o.init_phi()
    o.sweep(its)
    ll_cpu = model_log_likelihood(o.get_doc_topic_counts(), o.get_type_topic_counts(), o.get_topic_totals(), alpha, beta)
    assert ll_hip > ll0 + 0.02 * abs(ll0)                   # the chain did move uphill from the random start
    assert abs(ll_hip - ll_cpu) <= 0.01 * abs(ll_cpu), (ll_hip, ll_cpu)


def test_registry_knows_pcgs():
    from ldagroupedgibbssampler_amd.sampler import LDAPartiallyCollapsedGibbsSampler
    m = create_model(SimpleLDAConfiguration(topics=4, seed=1, scheme="pcgs"))
    assert isinstance(m, LDAPartiallyCollapsedGibbsSampler) and m.getNoTopics() == 4
    with pytest.raises(NotImplementedError):
        m.getTheta()


@pytest.mark.gpu
def test_python_mirror_pcgs_matches_oracle(oracle, cats):
    m = create_model(SimpleLDAConfiguration(topics=5, alpha=0.5, beta=0.1, seed=31, iterations=3, exec_time=1800, paranoid=True), "pcgs")
    m.addInstances(cats)
    m.sample(3)
    o = oracle.OracleSampler(5, cats.num_types, 0.5, 0.1, 31)
    o.set_scheme("pcgs")
    o.set_corpus(cats.doc_ptr, cats.tokens)
    o.init_z_java_lcg(31)
    o.init_phi()
    o.sweep(3)
    assert np.array_equal(np.concatenate(m.getZIndicators()), o.get_z())
    assert np.array_equal(m.getPhi().view(np.int64), o.get_phi().view(np.int64))


@pytest.mark.gpu
def test_python_mirror_writes_the_driver_files(oracle, cats, tmp_path):
    """sample() with the diagnostics on: the device values land in the Java-named lists and, formatted like the Java
    driver's files (ldagroupedgibbssampler_amd/formats.py), under log_dir."""
    from ldagroupedgibbssampler_amd import formats as F
    from ldagroupedgibbssampler_amd.corpus import Corpus
    train, _, _ = cats.shard(0, 18)
    test, _, _ = cats.shard(18, cats.num_docs)
    d = str(tmp_path)
    m = create_model(SimpleLDAConfiguration(topics=5, alpha=0.5, beta=0.1, seed=31, iterations=4, exec_time=1800, compute_likelihood=True,
                                            start_diagnostic=3, log_topic_indicators=True, log_dir=d))
    m.setRandomSeed(31)
    m.addInstances(train)
    m.addTestInstances(Corpus(test.doc_ptr, test.tokens, train.num_types))
    m.sample(4)
    assert len(m.loglikelihood) == 4 and len(m.heldOutLoglikelihood) == 4 and len(m.logPosterior) == 2
    assert m.loglikelihood[-1] == m.modelLogLikelihood() and m.heldOutLoglikelihood[-1] == m.heldOutLogLikelihood()
    ll = open(os.path.join(d, "log-likelihood.txt")).read().splitlines()
    assert ll == ["%d\t%s" % (i + 1, F.java_double_to_string(v)) for i, v in enumerate(m.loglikelihood)]
    ho = open(os.path.join(d, "test_held_out_log_likelihood.txt")).read().splitlines()
    assert ho == ["%d\t%s" % (i + 1, F.java_double_to_string(v)) for i, v in enumerate(m.heldOutLoglikelihood)]
    lp = [l.split("\t") for l in open(os.path.join(d, "log-posterior.txt")).read().splitlines()]
    assert [l[0] for l in lp] == ["3", "4"] and [l[1] for l in lp] == [F.java_format_fixed(v, 6) for v in m.logPosterior]
    z4 = [list(map(int, l.split(","))) if l else [] for l in open(os.path.join(d, "z_4.csv")).read().splitlines()]
    assert [t for doc in z4 for t in doc] == [int(t) for doc in m.getZIndicators() for t in doc]
    assert sorted(f for f in os.listdir(d) if f.startswith("z_")) == ["z_1.csv", "z_2.csv", "z_3.csv", "z_4.csv"]
