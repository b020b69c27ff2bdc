"""Host-side mirror of the reference's sampler interface for the GGS path.

``LDAGroupedGibbsSampler`` keeps the method names, argument meaning and error behaviour of
``cc.mallet.topics.LDAGibbsSampler`` + ``LDASamplerWithPhi`` (LDAGibbsSampler.java:10-47,
LDASamplerWithPhi.java:5-12) for the calls the driver makes on the hot path
(tui/ParallelLDA.java:173-296): ctor(config) -> setRandomSeed -> addInstances -> sample ->
getters.  All numerics happen in libggs_hip.so through the C-ABI; this file is orchestration
(the per-iteration loop of UncollapsedParallelLDA.sample, UPLDA:645-930: abort flag, exec_time
budget, hooks) plus the cheap host-side helpers z-bar / theta estimate (MSLDA:617-778).

The reference toolchain (JDK, Maven, MALLET) is absent from the build image, so this mirror is
Python over ctypes for the tests and the benchmark; include/ggs_sampler.hpp is the same mirror
in C++, and INTEGRATION.md shows the JNI subclass a maintainer would add on the Java side.
"""
import os
import time

import numpy as np

from . import native
from .corpus import Corpus


class SimpleLDAConfiguration:
    """The keys of cc.mallet.configuration.LDAConfiguration the GGS path reads, with the
    reference's defaults (LDAConfiguration.java:10-56).  Mirrors the test-side POJO
    SimpleLDAConfiguration; INI parsing / CLI overrides stay in Java (out of scope)."""

    def __init__(self, **kw):
        self.scheme = kw.pop("scheme", "ggs")
        self.topics = int(kw.pop("topics", 10))                      # NO_TOPICS_DEFAULT
        self.alpha = float(kw.pop("alpha", 50.0 / 10))               # ALPHA_DEFAULT = 50.0 / NO_TOPICS_DEFAULT
        self.beta = float(kw.pop("beta", 0.01))                      # BETA_DEFAULT
        self.iterations = int(kw.pop("iterations", 1500))            # NO_ITER_DEFAULT
        self.seed = int(kw.pop("seed", 0))                           # SEED_DEFAULT: 0 = use the clock (ParsedLDAConfiguration.java:137-141)
        self.exec_time = kw.pop("exec_time", 10)                     # EXEC_TIME_DEFAULT seconds of z+Phi time (UPLDA:577,926-928)
        self.save_phi_mean = bool(kw.pop("save_phi_mean", False))    # SAVE_PHI_MEAN_DEFAULT
        self.phi_mean_burnin = int(kw.pop("phi_mean_burnin", 0))     # percent of iterations, PHI_BURN_IN_DEFAULT
        self.phi_mean_thin = int(kw.pop("phi_mean_thin", 1))         # PHI_THIN_DEFAULT
        self.paranoid = bool(kw.pop("paranoid", False))              # run the UPLDA:299-338 invariants every sweep
        self.device_id = int(kw.pop("device_id", 0))                 # optional gpu_* key; default first visible GPU
        # the diagnostics of the sampling loop (UPLDA:695-905), computed on the device and written in the Java driver's formats
        self.compute_likelihood = bool(kw.pop("compute_likelihood", False))   # model LL (+ held-out LL with a test set) every iteration, UPLDA:838-850
        self.start_diagnostic = int(kw.pop("start_diagnostic", 500))          # START_DIAG_DEFAULT; log posterior from this iteration on, UPLDA:706,818-821
        self.log_topic_indicators = bool(kw.pop("log_topic_indicators", False))  # z_<iteration>.csv, UPLDA:637-638,871-872
        self.log_dir = kw.pop("log_dir", None)                       # where the files go (LoggingUtils' run directory in Java); None = no files
        if kw:
            raise TypeError("unknown configuration keys: %s" % sorted(kw))

    def get_seed(self):
        if self.seed == 0:
            return int(time.time() * 1000) & 0x7FFFFFFF
        return self.seed


def calc_zbar(num_topics, one_doc_topics):
    """ModifiedSimpleLDA.calcZBar (MSLDA:647-668): topic frequencies of one document; an empty
    document gives zeros."""
    z = np.asarray(one_doc_topics, np.int64)
    counts = np.bincount(z, minlength=num_topics).astype(np.float64)
    if z.size == 0:
        return np.zeros(num_topics)
    return counts / z.size


def calc_theta_estimate(num_topics, alpha, one_doc_topics):
    """ModifiedSimpleLDA.calcThetaEstimate (MSLDA:709-753): (n_k + alpha_k) / sum_k (n_k + alpha_k),
    the normaliser summed in k order exactly as the Java loop does."""
    a = np.broadcast_to(np.asarray(alpha, np.float64), (num_topics,))
    counts = np.bincount(np.asarray(one_doc_topics, np.int64), minlength=num_topics).astype(np.float64)
    normalizer = 0.0
    for k in range(num_topics):
        normalizer += counts[k] + a[k]
    est = (counts + a) / normalizer
    if not np.all(np.isfinite(est)) or (est < 0).any():
        raise RuntimeError("theta estimate is broken")               # IllegalStateException, MSLDA:727-729
    return est


def model_log_likelihood(n_dk, n_wk, n_k, alpha, beta):
    """UncollapsedParallelLDA.modelLogLikelihood (UPLDA:1644-1758): Dirichlet-multinomial log
    likelihood of the topic assignments, from the count matrices alone.  Host-side diagnostic
    (it stays on the host in the reference too).  MALLET's Dirichlet.logGammaStirling is not
    in the reference tree; scipy's gammaln stands in for it (restate from the public
    formula, ~1e-6 relative), so against a JVM value this is pinned only to that accuracy."""
    from scipy.special import gammaln
    n_dk = np.asarray(n_dk, np.float64)
    n_wk = np.asarray(n_wk, np.float64)
    n_k = np.asarray(n_k, np.float64)
    K = n_dk.shape[1]
    V = n_wk.shape[0]
    a = np.broadcast_to(np.asarray(alpha, np.float64), (K,))
    alpha_sum = float(a.sum())
    ll = 0.0
    nz = n_dk > 0                                                    # UPLDA:1680-1685
    ll += float((gammaln(n_dk + a)[nz]).sum() - (np.broadcast_to(gammaln(a), n_dk.shape)[nz]).sum())
    ll -= float(gammaln(alpha_sum + n_dk.sum(1)).sum())              # UPLDA:1689
    ll += n_dk.shape[0] * float(gammaln(alpha_sum))                  # UPLDA:1694
    nzw = n_wk > 0                                                   # UPLDA:1701-1722
    ll += float(gammaln(beta + n_wk[nzw]).sum())
    ll -= float(gammaln(beta * V + n_k).sum())                       # UPLDA:1724-1728
    ll += float(gammaln(beta * V)) * K                               # UPLDA:1742-1743
    ll -= float(gammaln(beta)) * int(nzw.sum())                      # UPLDA:1746-1747
    return ll


class LDAGroupedGibbsSampler:
    """scheme=ggs on MI355X.  One instance drives one GPU (doc-sharded runs wrap the same
    native handle with ldagroupedgibbssampler_amd.sharded.ShardedGGS)."""
    _scheme_flags = 0

    def __init__(self, config):
        self.config = config
        self.numTopics = config.topics
        self.alpha = np.full(self.numTopics, config.alpha, np.float64)       # MSLDA:129-135
        self.alphaSum = config.alpha * self.numTopics
        self.beta = config.beta
        self.startSeed = config.get_seed()
        self.currentIteration = 0
        self._abort = False
        self._h = None
        self._corpus = None
        self.zSamplingTimeCum = 0.0      # ms, as UPLDA:642-693 accumulates them
        self.phiSamplingTimeCum = 0.0
        self.loglikelihood, self.heldOutLoglikelihood, self.logPosterior = [], [], []   # the Java lists (MSLDA:114-115; UPLDA:591,843,849) + the posterior values

    # ---- LDAGibbsSampler ----
    def setConfiguration(self, config):
        self.config = config

    def getConfiguration(self):
        return self.config

    def setRandomSeed(self, seed):
        """MSLDA:153-156: only the initial z depends on it (called before addInstances,
        tui/ParallelLDA.java:176,189); it also keys this build's Philox streams."""
        self.startSeed = int(seed)

    def addInstances(self, training):
        """UPLDA:357-456 + GGS:33-37.  `training` is a Corpus (integer CSR of the
        FeatureSequences in instance order)."""
        if not isinstance(training, Corpus):
            raise TypeError("addInstances expects a ldagroupedgibbssampler_amd.corpus.Corpus")
        cfg = self.config
        flags = (native.FLAG_PARANOID if cfg.paranoid else 0) | (native.FLAG_SAVE_PHI_MEAN if cfg.save_phi_mean else 0) | self._scheme_flags
        burn_in = int((cfg.phi_mean_burnin / 100.0) * cfg.iterations)        # UPLDA:206-207
        self._h = native.GGSHandle(self.numTopics, training.num_types, self.alpha, self.beta, self.startSeed,
                                   device_id=cfg.device_id, flags=flags, phi_burn_in=burn_in, phi_mean_thin=cfg.phi_mean_thin)
        self._corpus = training
        self._h.set_corpus(training.doc_ptr, training.tokens)
        self._h.init_z_java_lcg(self.startSeed)          # initialDrawTopicIndicator, UPLDA:458-460
        self._h.init_phi()                               # initialSamplePhi, UPLDA:1287-1294
        self.currentIteration = 0

    def addTestInstances(self, test_set):
        """MSLDA:918-923: the test set of the held-out estimator; it must share the training alphabet
        ("Alphabets on training and test sets do not match!")."""
        self._need_data()
        if not isinstance(test_set, Corpus):
            raise TypeError("addTestInstances expects a ldagroupedgibbssampler_amd.corpus.Corpus")
        if test_set.num_types != self._corpus.num_types:
            raise ValueError("Alphabets on training and test sets do not match!")
        self._h.set_test_corpus(test_set.doc_ptr, test_set.tokens)
        self._test_set = test_set

    def heldOutLogLikelihood(self, num_particles=100):
        """What the sampling loop logs when a test set is present (UPLDA:604-622,840-844):
        MarginalProbEstimatorPlain(...).evaluateLeftToRight(testSet, 100, null), on the device."""
        self._need_data()
        if getattr(self, "_test_set", None) is None:
            raise ValueError("no test set: call addTestInstances first")
        return self._h.heldout_log_likelihood(num_particles)[0]

    def sample(self, iterations):
        """UPLDA:552-943 without the host-side diagnostics: one native sweep per iteration so the
        abort flag and the exec_time budget keep their per-iteration granularity."""
        self._need_data()
        self.preSample()
        max_exec_ms = float(self.config.exec_time) * 1000.0 if self.config.exec_time is not None else float("inf")
        for iteration in range(1, int(iterations) + 1):
            if self._abort:
                break
            self.preIteration()
            self.preZ()
            t0 = self._h.get_timings()
            self._z_half()                               # loopOverBatches + the local half of updateCounts
            self.postZ()
            self.prePhi()
            self._phi_half()                             # samplePhi
            self.postPhi()
            self.currentIteration = self._h.iteration
            t1 = self._h.get_timings()
            self.zSamplingTimeCum += (t1["theta_ms"] - t0["theta_ms"]) + (t1["z_ms"] - t0["z_ms"]) + (t1["merge_ms"] - t0["merge_ms"])
            self.phiSamplingTimeCum += t1["phi_ms"] - t0["phi_ms"]
            self._diagnostics(iteration)
            self.postIteration()
            if os.path.exists("abort"):                  # the sentinel file of UPLDA:131,908-910 (relative to the working directory)
                self.abort()
            if self.zSamplingTimeCum + self.phiSamplingTimeCum >= max_exec_ms:   # UPLDA:926-928
                break
        self.postSample()

    def _z_half(self):
        self._h.sweep_begin()

    def _phi_half(self):
        self._h.sweep_end()

    def _diagnostics(self, iteration):
        """The per-iteration diagnostics of UPLDA:695-905 that have a device implementation, in the Java order: log
        posterior (ggs and pcgs, from start_diagnostic on), held-out and model log likelihood (compute_likelihood), topic
        indicators; each value is kept in the Java-named list and, with a log_dir, appended in the Java file format."""
        cfg = self.config
        from . import formats as F
        if cfg.start_diagnostic > 0 and iteration >= cfg.start_diagnostic and not (self._scheme_flags & native.FLAG_COLLAPSED):   # pcgs: with a fresh theta, UPLDA:710-714
            lp = self.computeLogPosterior()                               # UPLDA:818-821
            self.logPosterior.append(lp)
            if cfg.log_dir:
                F.append_log_posterior(cfg.log_dir, iteration, lp, int(time.time() * 1000))
        if cfg.compute_likelihood:
            if getattr(self, "_test_set", None) is not None:             # UPLDA:840-844
                ho = self.heldOutLogLikelihood(100)
                self.heldOutLoglikelihood.append(ho)
                if cfg.log_dir:
                    F.append_heldout_log_likelihood(cfg.log_dir, iteration, ho)
            ll = self.modelLogLikelihood()                                # UPLDA:846-850
            self.loglikelihood.append(ll)
            if cfg.log_dir:
                F.append_log_likelihood(cfg.log_dir, iteration, ll)
            if cfg.log_topic_indicators and cfg.log_dir:                  # UPLDA:871-872
                F.write_topic_indicators(self._corpus.doc_ptr, self._h.get_z(), cfg.log_dir, iteration)

    def sampleZGivenPhi(self, iterations):
        self._need_data()
        self.preSample()
        self._h.sample_z_given_phi(int(iterations))
        self.currentIteration = self._h.iteration
        self.postSample()

    def getNoTopics(self):
        return self.numTopics

    getNumTopics = getNoTopics

    def getNoTypes(self):
        self._need_data()
        return self._corpus.num_types

    def getCurrentIteration(self):
        return self.currentIteration

    def getCorpusSize(self):
        self._need_data()
        return self._corpus.num_tokens

    def getStartSeed(self):
        return self.startSeed

    def getZIndicators(self):
        """int[D][] (MSLDA:464-477)"""
        self._need_data()
        z = self._h.get_z()
        p = self._corpus.doc_ptr
        return [z[p[d]:p[d + 1]] for d in range(self._corpus.num_docs)]

    def setZIndicators(self, z_indicators):
        """UPLDA:1797-1843: rebuilds the counts and re-draws Phi; throws when the lengths do not
        add up to the corpus size (IllegalArgumentException, UPLDA:1828-1830)."""
        self._need_data()
        flat = np.concatenate([np.asarray(z, np.int32) for z in z_indicators]) if len(z_indicators) else np.zeros(0, np.int32)
        lens = [len(z) for z in z_indicators]
        if len(z_indicators) != self._corpus.num_docs or lens != list(np.diff(self._corpus.doc_ptr)):
            raise ValueError("Count does not sum to nr. types! Sumtotal: %d no.types: %d" % (flat.size, self._corpus.num_tokens))
        self._h.set_z(flat, redraw_phi=True)

    def getTypeTopicMatrix(self):
        self._need_data()
        return self._h.get_type_topic_counts()           # [V][K], UPLDA:226-234

    def getDocumentTopicMatrix(self):
        self._need_data()
        return self._h.get_doc_topic_counts()            # [D][K], MSLDA:536-547

    def getTopicTotals(self):
        self._need_data()
        return self._h.get_topic_totals()

    def getZbar(self):
        n_dk = self.getDocumentTopicMatrix().astype(np.float64)
        lens = np.diff(self._corpus.doc_ptr).astype(np.float64)
        out = np.zeros_like(n_dk)
        nz = lens > 0
        out[nz] = n_dk[nz] / lens[nz, None]              # MSLDA:655-660
        return out

    def getThetaEstimate(self):
        n_dk = self.getDocumentTopicMatrix()
        return np.stack([self._theta_estimate_from_counts(row) for row in n_dk]) if len(n_dk) else np.zeros((0, self.numTopics))

    def _theta_estimate_from_counts(self, counts):
        normalizer = 0.0
        for k in range(self.numTopics):
            normalizer += float(counts[k]) + self.alpha[k]
        return (counts.astype(np.float64) + self.alpha) / normalizer

    def getTheta(self):
        """thetaMatrix of the last z step (GGS:72; what UPLDA:716-720 copies for scheme ggs)."""
        self._need_data()
        return self._h.get_theta()

    def modelLogLikelihood(self):
        """UPLDA:1644-1758, on the device (ggs_model_log_likelihood): nothing is copied back but two doubles.
        model_log_likelihood() above is the host-side formula the tests check it against."""
        self._need_data()
        doc_side, topic_side = self._h.model_log_likelihood()
        return doc_side + topic_side

    def computeLogPosterior(self):
        """UPLDA:1573-1634 (Doss and George 2025), on the device (ggs_log_posterior)."""
        self._need_data()
        doc_side, topic_side = self._h.log_posterior()
        return doc_side + topic_side

    def getBeta(self):
        return self.beta

    def getAlpha(self):
        return self.alpha

    # ---- LDASamplerWithPhi ----
    def getPhi(self):
        self._need_data()
        return self._h.get_phi()                         # [K][V]

    def setPhi(self, phi, data_alphabet=None, target_alphabet=None):
        self._need_data()
        self._h.set_phi(np.asarray(phi, np.float64))

    def getPhiMeans(self):
        self._need_data()
        return self._h.get_phi_mean()[0]                 # None before the first accumulated sample, UPLDA:1955-1958

    def getNoSampledPhi(self):
        self._need_data()
        return self._h.get_phi_mean()[1]

    # ---- AbortableSampler ----
    def abort(self):
        self._abort = True                               # volatile flag, MSLDA:88,601-603

    def getAbort(self):
        return self._abort

    # ---- hooks (no-ops, MSLDA:783-810) ----
    def preIteration(self):
        pass

    def postIteration(self):
        pass

    def preSample(self):
        pass

    def postSample(self):
        pass

    def preZ(self):
        pass

    def postZ(self):
        pass

    def prePhi(self):
        pass

    def postPhi(self):
        pass

    def _need_data(self):
        if self._h is None:
            raise RuntimeError("addInstances has not been called")


class LDAPartiallyCollapsedGibbsSampler(LDAGroupedGibbsSampler):
    """scheme=pcgs (topics/LDAPartiallyCollapsedGibbsSampler.java): the same driver, counts and Phi draw;
    the z step is UPLDA:1466-1544 (theta integrated out, sequential inside a document).  SURVEY 8(f)-1."""
    _scheme_flags = native.FLAG_PCGS

    def getTheta(self):
        raise NotImplementedError("scheme=pcgs never draws theta; use getThetaEstimate() (UPLDA:716-720 does the same)")


class SerialCollapsedLDA(LDAGroupedGibbsSampler):
    """scheme=collapsed (topics/SerialCollapsedLDA.java; the conditional it samples from is sampleTopicsForOneDoc,
    MSLDA:158-226).  `schedule`:
      "serial"    the reference's own chain -- one pass over all tokens, counts moved in place, initial topics and
                  sampling uniforms from the ONE Randoms(seed) the Java class owns (SerialCollapsedLDA.java:60-65,789):
                  bit-identical to the restated Java loop, no parallelism across tokens (BASELINE config 1)
      "parallel"  documents side by side on the sweep-start counts, merged after the sweep (the AD-LDA decomposition):
                  the device-speed variant, approximate as ADLDA is
    No theta and no Phi are drawn: getPhi() is the point estimate (beta + n_wk)/(betaSum + n_k)."""
    _scheme_flags = native.FLAG_COLLAPSED

    def __init__(self, config, schedule="serial"):
        super().__init__(config)
        if schedule not in ("serial", "parallel"):
            raise ValueError("schedule must be 'serial' or 'parallel'")
        self.schedule = schedule

    def _z_half(self):
        if self.schedule == "serial":
            self._h.collapsed_serial_sweep(self.startSeed, 1)
        else:
            self._h.sweep_begin()

    def _phi_half(self):
        if self.schedule == "parallel":
            self._h.sweep_end()

    def getTheta(self):
        raise NotImplementedError("scheme=collapsed never draws theta; use getThetaEstimate()")

    def sampleZGivenPhi(self, iterations):
        raise NotImplementedError("scheme=collapsed has no Phi to condition on")


def create_model(config, scheme=None):
    """The `case "ggs"` / `case "pcgs"` / `case "collapsed"` of tui/ParallelLDA.createModel (ParallelLDA.java:401-490)."""
    scheme = scheme or config.scheme
    if scheme == "ggs":
        return LDAGroupedGibbsSampler(config)
    if scheme == "pcgs":
        return LDAPartiallyCollapsedGibbsSampler(config)
    if scheme == "collapsed":
        return SerialCollapsedLDA(config)
    raise ValueError("scheme %r is not provided by this build (only the ggs, pcgs and collapsed z loops are in scope)" % scheme)
