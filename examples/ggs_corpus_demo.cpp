// ggs_corpus_demo -- loads a dataset with include/ggs_corpus.hpp and prints it for tests/test_frontend.py:
//   usage: ggs_corpus_demo dataset.txt stoplist|- rare_threshold|tfidf:<n> keep_numbers(0|1) max_doc_buf_size keep_connectors(0|1) [alphabet.txt frozen(0|1)]
//          (tfidf:<n> takes LDAUtils.loadInstancesKeep with tfidf_vocab_size = n instead of loadInstancesPrune)
//          (alphabet.txt: one word per line -- a test set loaded against the training vocabulary, LDAUtils.java:252-257)
//   output: "D V N", then the doc_ptr, the token ids, the label ids (one line each), then V vocabulary lines,
//           then D name lines.  Exit code 3 + "overflow" on stderr for the tokenizer's ArrayIndexOutOfBoundsException.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "ggs_corpus.hpp"

int main(int argc, char **argv) {
  if (argc != 7 && argc != 9) { std::fprintf(stderr, "usage: %s dataset stoplist|- rare_threshold keep_numbers buf keep_connectors\n", argv[0]); return 2; }
  ggs::corpus::LoadOptions opt;
  if (std::string(argv[2]) != "-") opt.stoplist_file = argv[2];
  const bool keep = std::string(argv[3]).rfind("tfidf:", 0) == 0;
  if (keep) opt.keep_count = std::atoi(argv[3] + 6);
  else opt.prune_count = std::atoi(argv[3]);
  opt.keep_numbers = std::atoi(argv[4]) != 0;
  opt.buffer_size = std::atoi(argv[5]);
  opt.keep_connectors = std::atoi(argv[6]) != 0;
  try {
    std::vector<std::string> alphabet;
    if (argc == 9) {
      std::ifstream f(argv[7]);
      std::string w;
      while (std::getline(f, w)) alphabet.push_back(w);
    }
    const bool frozen = argc == 9 && std::atoi(argv[8]) != 0;
    const ggs::corpus::Dataset ds = keep ? ggs::corpus::load_instances_keep(argv[1], opt, argc == 9 ? &alphabet : nullptr, frozen)
                                         : ggs::corpus::load_instances_prune(argv[1], opt, argc == 9 ? &alphabet : nullptr, frozen);
    std::printf("%lld %zu %zu\n", (long long)ds.size(), ds.vocab.size(), ds.tokens.size());
    for (int64_t p : ds.doc_ptr) std::printf("%lld ", (long long)p);
    std::printf("\n");
    for (int32_t t : ds.tokens) std::printf("%d ", t);
    std::printf("\n");
    for (int32_t l : ds.labels) std::printf("%d ", l);
    std::printf("\n");
    for (const auto &w : ds.vocab) std::printf("%s\n", w.c_str());
    for (const auto &n : ds.names) std::printf("%s\n", n.c_str());
  } catch (const ggs::corpus::TokenBufferOverflow &e) {
    std::fprintf(stderr, "overflow: %s\n", e.what());
    return 3;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
