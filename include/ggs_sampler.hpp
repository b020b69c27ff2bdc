// ggs_sampler.hpp -- C++ host-side mirror of the reference's sampler interface for the GGS
// path, header-only over the C-ABI of ggs_hip.h.
//
// The reference is Java; its toolchain (JDK, Maven, MALLET 2.0.8) is not in the build image, so
// the host side above the C-ABI is written in C++ with the reference's names and call order:
//   cc.mallet.topics.LDAGibbsSampler   (topics/LDAGibbsSampler.java:10-47)
//   cc.mallet.topics.LDASamplerWithPhi (topics/LDASamplerWithPhi.java:5-12)
//   driver call order                  (topics/tui/ParallelLDA.java:173-296):
//     ctor(config) -> setRandomSeed -> addInstances -> sample(iterations) -> getters
// sample() is the per-iteration loop of UncollapsedParallelLDA.sample (UPLDA:645-930): abort flag, exec_time budget
// on cumulative z+Phi time, the eight hooks, and the diagnostics that have a device implementation (log posterior,
// held-out and model log likelihood, topic indicators), written in the Java driver's file formats by ggs_formats.hpp.
// Errors: the Java code throws IllegalStateException / IllegalArgumentException; here a
// non-zero C-ABI return becomes ggs::SamplerError carrying the code and ggs_last_error().
#pragma once
#include <atomic>
#include <fstream>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include <chrono>

#include "ggs_formats.hpp"
#include "ggs_hip.h"

namespace ggs {

struct SamplerError : std::runtime_error {
  int code;
  SamplerError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// keys of cc.mallet.configuration.LDAConfiguration read on the GGS path, reference defaults
// (LDAConfiguration.java:10-56)
struct LDAConfiguration {
  int topics = 10;                 // NO_TOPICS_DEFAULT
  double alpha = 50.0 / 10;        // ALPHA_DEFAULT
  double beta = 0.01;              // BETA_DEFAULT
  int iterations = 1500;           // NO_ITER_DEFAULT
  int seed = 0;                    // SEED_DEFAULT
  double exec_time = 10;           // seconds, EXEC_TIME_DEFAULT (UPLDA:577,926-928)
  bool save_phi_mean = false;
  int phi_mean_burnin = 0;         // percent of iterations
  int phi_mean_thin = 1;
  bool paranoid = false;
  bool pcgs = false;              // scheme=pcgs instead of ggs (ParallelLDA.java:414-416): the z step of UPLDA:1466-1544
  bool collapsed = false;         // scheme=collapsed (ParallelLDA.java:424-428, SerialCollapsedLDA): the serial chain of MSLDA:158-226
  int device_id = 0;
  // the diagnostics of the sampling loop (UPLDA:695-905), computed on the device, written as the Java driver writes them
  bool compute_likelihood = false;   // model LL (+ held-out LL with a test set) every iteration, UPLDA:838-850
  int start_diagnostic = 500;        // START_DIAG_DEFAULT; log posterior from this iteration on, UPLDA:706,818-821
  bool log_topic_indicators = false; // z_<iteration>.csv, UPLDA:637-638,871-872
  std::string log_dir;               // where the files go (LoggingUtils' run directory in Java); empty = no files
};

// integer CSR of the training InstanceList: FeatureSequence.getFeatures() in instance order
struct InstanceList {
  std::vector<int64_t> doc_ptr;    // D+1
  std::vector<int32_t> tokens;     // N
  int32_t num_types = 0;           // alphabet size
  int64_t size() const { return (int64_t)doc_ptr.size() - 1; }
};

class LDAGroupedGibbsSampler {
 public:
  explicit LDAGroupedGibbsSampler(const LDAConfiguration &config) : config_(config), startSeed_(config.seed) {}
  virtual ~LDAGroupedGibbsSampler() { if (h_) ggs_destroy(h_); }
  LDAGroupedGibbsSampler(const LDAGroupedGibbsSampler &) = delete;
  LDAGroupedGibbsSampler &operator=(const LDAGroupedGibbsSampler &) = delete;

  void setRandomSeed(int seed) { startSeed_ = seed; }                       // MSLDA:153-156
  int getStartSeed() const { return startSeed_; }

  void addInstances(const InstanceList &training) {                         // UPLDA:357-456, GGS:33-37
    if (h_) { ggs_destroy(h_); h_ = nullptr; }
    ggs_config c{};
    c.struct_size = (int32_t)sizeof(ggs_config);
    c.num_topics = config_.topics; c.num_types = training.num_types; c.device_id = config_.device_id;
    c.alpha = nullptr; c.alpha_scalar = config_.alpha; c.beta = config_.beta;
    c.seed = (uint64_t)(int64_t)startSeed_;
    c.flags = (config_.paranoid ? GGS_FLAG_PARANOID : 0) | (config_.save_phi_mean ? GGS_FLAG_SAVE_PHI_MEAN : 0) |
              (config_.pcgs ? GGS_FLAG_PCGS : 0) | (config_.collapsed ? GGS_FLAG_COLLAPSED : 0);
    c.phi_burn_in = (int32_t)(((double)config_.phi_mean_burnin / 100) * config_.iterations);   // UPLDA:206-207
    c.phi_mean_thin = config_.phi_mean_thin;
    int rc = ggs_create(&c, &h_);
    if (rc) { h_ = nullptr; throw SamplerError(rc, "ggs_create failed"); }
    D_ = training.size(); N_ = training.doc_ptr.back(); V_ = training.num_types;
    doc_ptr_ = training.doc_ptr;
    chk(ggs_set_corpus(h_, D_, training.doc_ptr.data(), training.tokens.data(), 0, 0));
    chk(ggs_init_z_java_lcg(h_, startSeed_));                               // UPLDA:458-460
    chk(ggs_init_phi(h_));                                                  // UPLDA:1287-1294
    currentIteration_ = 0;
  }

  void sample(int iterations) {                                             // UPLDA:552-943
    need();
    preSample();
    const double maxExecMs = config_.exec_time > 0 ? config_.exec_time * 1000.0 : std::numeric_limits<double>::infinity();
    for (int iteration = 1; iteration <= iterations && !abort_.load(); ++iteration) {
      preIteration();
      ggs_timings t0{}, t1{};
      chk(ggs_get_timings(h_, &t0));
      preZ();
      if (config_.collapsed) chk(ggs_collapsed_serial_sweep(h_, startSeed_, 1));   // SerialCollapsedLDA.java:159-172
      else chk(ggs_sweep_begin(h_));                                        // loopOverBatches (+ this device's counts)
      postZ();
      prePhi();
      if (!config_.collapsed) chk(ggs_sweep_end(h_));                       // samplePhi
      postPhi();
      chk(ggs_get_iteration(h_, &currentIteration_));
      chk(ggs_get_timings(h_, &t1));
      zSamplingTimeCum += (t1.theta_ms - t0.theta_ms) + (t1.z_ms - t0.z_ms) + (t1.merge_ms - t0.merge_ms);
      phiSamplingTimeCum += t1.phi_ms - t0.phi_ms;
      diagnostics(iteration);
      postIteration();
      if (std::ifstream("abort").good()) abort();                           // the sentinel file of UPLDA:131,908-910
      if (zSamplingTimeCum + phiSamplingTimeCum >= maxExecMs) break;        // UPLDA:926-928
    }
    postSample();
  }

  void sampleZGivenPhi(int iterations) {                                    // UPLDA:975-1014
    need();
    preSample();
    chk(ggs_sample_z_given_phi(h_, iterations));
    chk(ggs_get_iteration(h_, &currentIteration_));
    postSample();
  }

  int getNoTopics() const { return config_.topics; }
  int getNumTopics() const { return config_.topics; }
  int getNoTypes() const { return V_; }
  int getCurrentIteration() const { return currentIteration_; }
  int64_t getCorpusSize() const { return N_; }
  double getBeta() const { return config_.beta; }
  std::vector<double> getAlpha() const { return std::vector<double>((size_t)config_.topics, config_.alpha); }

  std::vector<std::vector<int32_t>> getZIndicators() {                      // MSLDA:464-477
    need();
    std::vector<int32_t> z((size_t)N_);
    chk(ggs_get_z(h_, z.data()));
    std::vector<std::vector<int32_t>> out((size_t)D_);
    for (int64_t d = 0; d < D_; ++d) out[(size_t)d].assign(z.begin() + doc_ptr_[(size_t)d], z.begin() + doc_ptr_[(size_t)d + 1]);
    return out;
  }
  void setZIndicators(const std::vector<std::vector<int32_t>> &zIndicators) {   // UPLDA:1797-1843
    need();
    std::vector<int32_t> flat;
    flat.reserve((size_t)N_);
    for (const auto &d : zIndicators) flat.insert(flat.end(), d.begin(), d.end());
    if ((int64_t)zIndicators.size() != D_ || (int64_t)flat.size() != N_)
      throw SamplerError(GGS_ERR_BAD_ARG, "Count does not sum to nr. types!");   // UPLDA:1828-1830
    chk(ggs_set_z(h_, flat.data(), 1));
  }
  std::vector<int32_t> getTypeTopicMatrix() {                               // [V][K], UPLDA:226-234
    need();
    std::vector<int32_t> m((size_t)V_ * config_.topics);
    chk(ggs_get_type_topic_counts(h_, m.data()));
    return m;
  }
  std::vector<int32_t> getDocumentTopicMatrix() {                           // [D][K], MSLDA:536-547
    need();
    std::vector<int32_t> m((size_t)D_ * config_.topics);
    chk(ggs_get_doc_topic_counts(h_, 0, D_, m.data()));
    return m;
  }
  std::vector<int32_t> getTopicTotals() {
    need();
    std::vector<int32_t> t((size_t)config_.topics);
    chk(ggs_get_topic_totals(h_, t.data()));
    return t;
  }
  std::vector<double> getZbar() {                                           // [D][K], MSLDA:647-668
    const std::vector<int32_t> ndk = getDocumentTopicMatrix();
    const int K = config_.topics;
    std::vector<double> out((size_t)D_ * K, 0.0);
    for (int64_t d = 0; d < D_; ++d) {
      const double len = (double)(doc_ptr_[(size_t)d + 1] - doc_ptr_[(size_t)d]);
      if (len > 0)
        for (int k = 0; k < K; ++k) out[(size_t)d * K + k] = (double)ndk[(size_t)d * K + k] / len;
    }
    return out;
  }
  std::vector<double> getThetaEstimate() {                                  // [D][K], MSLDA:709-753
    const std::vector<int32_t> ndk = getDocumentTopicMatrix();
    const int K = config_.topics;
    std::vector<double> out((size_t)D_ * K);
    for (int64_t d = 0; d < D_; ++d) {
      double normalizer = 0.0;
      for (int k = 0; k < K; ++k) normalizer += (double)ndk[(size_t)d * K + k] + config_.alpha;
      for (int k = 0; k < K; ++k) out[(size_t)d * K + k] = ((double)ndk[(size_t)d * K + k] + config_.alpha) / normalizer;
    }
    return out;
  }
  std::vector<double> getPhi() {                                            // [K][V], UPLDA:1946-1948
    need();
    std::vector<double> p((size_t)V_ * config_.topics);
    chk(ggs_get_phi(h_, p.data()));
    return p;
  }
  void setPhi(const std::vector<double> &phi) {                             // UPLDA:1897-1926
    need();
    if (phi.size() != (size_t)V_ * config_.topics) throw SamplerError(GGS_ERR_BAD_ARG, "phi must be [K][V]");
    chk(ggs_set_phi(h_, phi.data()));
  }
  // empty vector == the Java `null` before any Phi was accumulated (UPLDA:1955-1958)
  std::vector<double> getPhiMeans() {
    need();
    std::vector<double> p((size_t)V_ * config_.topics);
    int32_t n = 0;
    chk(ggs_get_phi_mean(h_, p.data(), &n));
    if (n == 0) p.clear();
    return p;
  }
  std::vector<double> getTheta() {                                          // thetaMatrix, GGS:72
    need();
    std::vector<double> t((size_t)D_ * config_.topics);
    chk(ggs_get_theta(h_, 0, D_, t.data()));
    return t;
  }

  double modelLogLikelihood() {                                             // UPLDA:1644-1758, on the device
    need();
    double a = 0, b = 0;
    chk(ggs_model_log_likelihood(h_, &a, &b));
    return a + b;
  }
  double computeLogPosterior() {                                            // UPLDA:1573-1634, on the device
    need();
    double a = 0, b = 0;
    chk(ggs_log_posterior(h_, &a, &b));
    return a + b;
  }
  void addTestInstances(const InstanceList &testSet) {                      // MSLDA:918-923; ids of the training alphabet
    need();
    chk(ggs_set_test_corpus(h_, testSet.size(), testSet.doc_ptr.data(), testSet.tokens.data(), 0));
    haveTestSet_ = true;
  }
  // what sample() logs at UPLDA:622,841: MarginalProbEstimatorPlain(...).evaluateLeftToRight(testSet, numParticles, null)
  double heldOutLogLikelihood(int numParticles = 100) {
    need();
    if (!haveTestSet_) throw SamplerError(GGS_ERR_STATE, "addTestInstances has not been called");
    double total = 0;
    chk(ggs_heldout_log_likelihood(h_, numParticles, nullptr, &total));
    return total;
  }

  // The per-iteration diagnostics of UPLDA:695-905 that have a device implementation, in the Java order; each value is
  // kept in the Java-named list and, with a log_dir, appended in the Java file format.
  std::vector<double> loglikelihood, heldOutLoglikelihood, logPosterior;    // MSLDA:114-115; UPLDA:591,843,849
  void diagnostics(int iteration) {
    const bool files = !config_.log_dir.empty();
    if (config_.start_diagnostic > 0 && iteration >= config_.start_diagnostic && !config_.collapsed) {   // pcgs: with a fresh theta, UPLDA:710-714
      const double lp = computeLogPosterior();                              // UPLDA:818-821
      logPosterior.push_back(lp);
      if (files)
        formats::append_log_posterior(config_.log_dir, iteration, lp,
                                      std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count());
    }
    if (config_.compute_likelihood) {
      if (haveTestSet_) {                                                   // UPLDA:840-844
        const double ho = heldOutLogLikelihood(100);
        heldOutLoglikelihood.push_back(ho);
        if (files) formats::append_heldout_log_likelihood(config_.log_dir, iteration, ho);
      }
      const double ll = modelLogLikelihood();                               // UPLDA:846-850
      loglikelihood.push_back(ll);
      if (files) formats::append_log_likelihood(config_.log_dir, iteration, ll);
      if (config_.log_topic_indicators && files) {                          // UPLDA:871-872
        std::vector<int32_t> z((size_t)N_);
        chk(ggs_get_z(h_, z.data()));
        formats::write_topic_indicators(doc_ptr_.data(), D_, z.data(), config_.log_dir, iteration);
      }
    }
  }

  void abort() { abort_.store(true); }                                      // MSLDA:601-603; may come from another thread
  bool getAbort() const { return abort_.load(); }

  // hooks: no-ops in the reference (MSLDA:783-810), virtual so a subclass can observe the loop
  virtual void preIteration() {}
  virtual void postIteration() {}
  virtual void preSample() {}
  virtual void postSample() {}
  virtual void preZ() {}
  virtual void postZ() {}
  virtual void prePhi() {}
  virtual void postPhi() {}

  double zSamplingTimeCum = 0, phiSamplingTimeCum = 0;                       // ms, UPLDA:642-693
  ggs_handle *native_handle() { return h_; }

 private:
  void need() const { if (!h_) throw SamplerError(GGS_ERR_STATE, "addInstances has not been called"); }
  void chk(int rc) { if (rc) throw SamplerError(rc, ggs_last_error(h_)); }
  LDAConfiguration config_;
  ggs_handle *h_ = nullptr;
  int startSeed_ = 0;
  int32_t currentIteration_ = 0;
  int64_t D_ = 0, N_ = 0;
  int32_t V_ = 0;
  std::vector<int64_t> doc_ptr_;
  std::atomic<bool> abort_{false};
  bool haveTestSet_ = false;
};

}  // namespace ggs
