// ggs_host_demo.cpp -- the C++ host mirror (include/ggs_sampler.hpp) driving the C-ABI the way
// tui/ParallelLDA.java:173-296 drives the Java sampler.  Reads an integer corpus
//   line 1: D V      then D lines: len tok tok ...
// runs `iterations` sweeps and prints z and tokensPerTopic so a test can compare it with the
// ctypes path.   usage: ggs_host_demo corpus.txt K alpha beta seed iterations [log_dir [ggs|collapsed]]
// With a log_dir the corpus doubles as the test set and the loop's diagnostics are on (compute_likelihood,
// start_diagnostic = 1, log_topic_indicators): the files of the Java driver appear there (ggs_formats.hpp).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "ggs_sampler.hpp"

int main(int argc, char **argv) {
  if (argc < 7 || argc > 9) { std::fprintf(stderr, "usage: %s corpus.txt K alpha beta seed iterations [log_dir [ggs|collapsed]]\n", argv[0]); return 2; }
  std::ifstream in(argv[1]);
  ggs::InstanceList inst;
  int64_t D;
  in >> D >> inst.num_types;
  inst.doc_ptr.push_back(0);
  for (int64_t d = 0; d < D; ++d) {
    int64_t len; in >> len;
    for (int64_t i = 0; i < len; ++i) { int32_t t; in >> t; inst.tokens.push_back(t); }
    inst.doc_ptr.push_back((int64_t)inst.tokens.size());
  }
  ggs::LDAConfiguration cfg;
  cfg.topics = std::atoi(argv[2]); cfg.alpha = std::atof(argv[3]); cfg.beta = std::atof(argv[4]);
  cfg.seed = std::atoi(argv[5]); cfg.iterations = std::atoi(argv[6]); cfg.exec_time = 1800; cfg.paranoid = true;
  const bool logging = argc >= 8;
  if (logging) { cfg.log_dir = argv[7]; cfg.compute_likelihood = true; cfg.start_diagnostic = 1; cfg.log_topic_indicators = true; }
  cfg.collapsed = argc >= 9 && std::string(argv[8]) == "collapsed";
  struct Counting : ggs::LDAGroupedGibbsSampler {
    using LDAGroupedGibbsSampler::LDAGroupedGibbsSampler;
    int pre = 0, post = 0;
    void preIteration() override { ++pre; }
    void postPhi() override { ++post; }
  };
  try {
    Counting model(cfg);
    model.setRandomSeed(cfg.seed);
    model.addInstances(inst);
    if (logging) model.addTestInstances(inst);
    model.sample(cfg.iterations);
    std::printf("iteration %d hooks %d %d\n", model.getCurrentIteration(), model.pre, model.post);
    std::printf("z");
    for (const auto &doc : model.getZIndicators()) for (int32_t z : doc) std::printf(" %d", z);
    std::printf("\nnk");
    for (int32_t n : model.getTopicTotals()) std::printf(" %d", n);
    std::printf("\n");
    const auto est = model.getThetaEstimate();
    double s = 0; for (int k = 0; k < cfg.topics; ++k) s += est[(size_t)k];
    std::printf("theta_estimate_doc0_sum %.17g\n", s);
    if (!logging) model.addTestInstances(inst);         // the diagnostics of the sampling loop, on the device
    std::printf("heldout %.17g\nloglik %.17g\n", model.heldOutLogLikelihood(100), model.modelLogLikelihood());
    if (!cfg.collapsed) std::printf("logposterior %.17g\n", model.computeLogPosterior());
  } catch (const ggs::SamplerError &e) {
    std::fprintf(stderr, "SamplerError %d: %s\n", e.code, e.what());
    return 1;
  }
  return 0;
}
