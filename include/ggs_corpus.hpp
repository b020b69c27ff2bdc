// ggs_corpus.hpp -- native corpus front-end: the reference's dataset loader restated in C++17 (second
// half), so that a corpus of BASELINE config 5's size reaches the sampler's integer CSR without a JVM in between.
// Header-only; pairs with ggs_sampler.hpp (the result is an ggs::InstanceList).  Same rules, line for line, as the
// Python restatement ldagroupedgibbssampler_amd/frontend.py, which documents the reference sites:
//
//   util/LDAUtils.loadInstancesPrune (LDAUtils.java:233-330): CsvIterator regex "^(\S*)[\s,]*([^\t]+)[\s,]*(.*)$"
//   (name, label, data) -> CharSequenceLowercase -> tokenizer chosen by initTokenizer (:532-563; the four classes of
//   cc/mallet/pipe: SimpleTokenizerLarge.java:52-135, NumericAlsoTokenizer, KeepConnectorPunctuation{TokenizerLarge,
//   NumericAlsoTokenizer}) -> stoplist -> alphabet in first-appearance order; rare_threshold > 0 adds a counting pass
//   whose rare types (count < threshold) join the stoplist (:243-289).
//   util/LDAUtils.loadInstancesKeep (:355-452) + pipe/TfIdfPipe.java: the same with a TF-IDF cut of the vocabulary.
//
// Unicode: general categories and lower-casing come from include/ggs_unicode_tables.hpp (generated, Unicode 13.0); a
// JDK speaks its own Unicode version (Java 8: 6.2) and lower-cases in the default locale -- corpora in ASCII / Latin-1
// are unaffected.  tests/test_frontend.py: the reference's own known answers (LDAUtilsTest, SimpleTokenizerLargeTest)
// and this loader against the Python one on the bundled datasets and on random Unicode text.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "ggs_unicode_tables.hpp"

namespace ggs {
namespace corpus {

// java.lang.ArrayIndexOutOfBoundsException out of the tokenizer: a token longer than max_doc_buf_size code points
// (SimpleTokenizerLarge.java:62,76; expected by SimpleTokenizerLargeTest.java:118-136)
struct TokenBufferOverflow : std::out_of_range {
  using std::out_of_range::out_of_range;
};

template <size_t N>
inline uint8_t range_lookup(const unicode::UnicodeRange (&tab)[N], uint32_t cp) {
  size_t lo = 0, hi = N;                              // last entry with start <= cp
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) / 2;
    if (tab[mid].start <= cp) lo = mid; else hi = mid;
  }
  return tab[lo].value;
}
inline int token_class(uint32_t cp) { return cp < 0x110000 ? range_lookup(unicode::kClassRanges, cp) : 0; }
inline uint32_t simple_lower(uint32_t cp) {
  constexpr size_t n = sizeof(unicode::kLowerPairs) / sizeof(unicode::kLowerPairs[0]);
  const unicode::LowerPair *b = unicode::kLowerPairs, *e = b + n;
  const unicode::LowerPair *it = std::lower_bound(b, e, cp, [](const unicode::LowerPair &p, uint32_t v) { return p.from < v; });
  return (it != e && it->from == cp) ? it->to : cp;
}

// UTF-8 -> code points (malformed bytes become U+FFFD, as Java's decoder substitutes them)
inline std::vector<uint32_t> decode_utf8(const std::string &s) {
  std::vector<uint32_t> out;
  out.reserve(s.size());
  for (size_t i = 0; i < s.size();) {
    const unsigned char c = (unsigned char)s[i];
    uint32_t cp = 0xFFFD;
    int extra = 0;
    if (c < 0x80) { cp = c; }
    else if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; extra = 1; }
    else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; extra = 2; }
    else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; extra = 3; }
    size_t j = i + 1;
    bool ok = extra == 0 ? c < 0x80 : true;
    for (int k = 0; k < extra && ok; ++k, ++j) {
      if (j >= s.size() || ((unsigned char)s[j] & 0xC0) != 0x80) { ok = false; break; }
      cp = (cp << 6) | ((unsigned char)s[j] & 0x3F);
    }
    if (!ok) { out.push_back(0xFFFD); i += 1; continue; }
    out.push_back(cp);
    i = j;
  }
  return out;
}
inline void append_utf8(std::string &out, uint32_t cp) {
  if (cp < 0x80) out.push_back((char)cp);
  else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
  else { out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
}

// String.toLowerCase(): one-to-one pairs, U+0130 -> "i" + U+0307, and the Final_Sigma rule
// (\p{cased}\p{case-ignorable}* U+03A3 !(\p{case-ignorable}*\p{cased}) -> U+03C2, else U+03C3)
inline std::vector<uint32_t> to_lower(const std::vector<uint32_t> &in) {
  std::vector<uint32_t> out;
  out.reserve(in.size() + 4);
  auto sig = [](uint32_t cp) { return cp < 0x110000 ? range_lookup(unicode::kSigmaRanges, cp) : 0; };   // 1 case-ignorable, 2 cased
  for (size_t i = 0; i < in.size(); ++i) {
    const uint32_t cp = in[i];
    if (cp == 0x130) { out.push_back('i'); out.push_back(0x307); continue; }
    if (cp == 0x3A3) {
      long j = (long)i - 1;
      while (j >= 0 && sig(in[(size_t)j]) == 1) --j;
      bool fin = j >= 0 && sig(in[(size_t)j]) == 2;
      if (fin) {
        size_t k = i + 1;
        while (k < in.size() && sig(in[k]) == 1) ++k;
        fin = k == in.size() || sig(in[k]) != 2;
      }
      out.push_back(fin ? 0x3C2 : 0x3C3);
      continue;
    }
    out.push_back(simple_lower(cp));
  }
  return out;
}

using Stoplist = std::unordered_set<std::string>;

// SimpleTokenizer(File): every line of the file is a stop word (UTF-8, untrimmed)
inline Stoplist read_stoplist(const std::string &path) {
  Stoplist s;
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open stoplist " + path);
  std::string line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    s.insert(line);
  }
  return s;
}

struct TokenizerOptions {
  bool keep_numbers = true;        // NumericAlsoTokenizer instead of SimpleTokenizerLarge (cfg key keep_numbers)
  bool keep_connectors = false;    // the KeepConnectorPunctuation* variants (cfg key keep_connecting_punctuation)
  int buffer_size = 10000;         // cfg key max_doc_buf_size, LDAConfiguration.java:39
};

// The pipe() loop of the four tokenizers on lower-cased code points.  Faithful to the Java indexing: the loop runs
// codePointCount steps but reads codePointAt(chars, i) at UTF-16 UNIT i, so text behind supplementary characters
// loses its tail -- reproduced, not repaired.
template <class Emit>
inline void tokenize(const std::vector<uint32_t> &cps, const Stoplist &stoplist, const TokenizerOptions &opt, Emit &&emit) {
  std::vector<uint32_t> units;                         // UTF-16 units, surrogate pairs as two entries
  units.reserve(cps.size());
  for (uint32_t cp : cps) {
    if (cp >= 0x10000) { units.push_back(0xD800 + ((cp - 0x10000) >> 10)); units.push_back(0xDC00 + ((cp - 0x10000) & 0x3FF)); }
    else units.push_back(cp);
  }
  const size_t total = cps.size();
  std::vector<uint32_t> buf;
  std::string token;
  auto flush = [&]() {
    if (buf.empty()) return;
    token.clear();
    for (uint32_t c : buf) append_utf8(token, c);
    if (!stoplist.count(token)) emit(token);
    buf.clear();
  };
  for (size_t i = 0; i < total && i < units.size(); ++i) {
    uint32_t cp = units[i];
    if (cp >= 0xD800 && cp < 0xDC00 && i + 1 < units.size() && units[i + 1] >= 0xDC00 && units[i + 1] < 0xE000)
      cp = 0x10000 + ((cp - 0xD800) << 10) + (units[i + 1] - 0xDC00);
    const int k = (cp >= 0xD800 && cp < 0xE000) ? 0 : token_class(cp);
    if (k == 1 || (k == 3 && opt.keep_connectors) || (k == 4 && opt.keep_numbers)) {
      if ((int)buf.size() >= opt.buffer_size) throw TokenBufferOverflow("token longer than the token buffer");
      buf.push_back(cp);
    } else if (k == 2 || k == 3) {
      flush();
    }                                                  // everything else is skipped without ending the token
  }
  flush();
}

// CsvIterator's line regex ^(\S*)[\s,]*([^\t]+)[\s,]*(.*)$ (java.util.regex: \s is [ \t\n\x0B\f\r]), matched by hand with
// the regex engine's greedy-then-backtrack order.  false = the line does not match (CsvIterator throws).
inline bool match_line(const std::string &line, std::string &name, std::string &label, std::string &data) {
  auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\x0B' || c == '\f' || c == '\r'; };
  auto is_sep = [&](char c) { return is_space(c) || c == ','; };
  const size_t n = line.size();
  size_t maxrun = 0;
  while (maxrun < n && !is_space(line[maxrun])) ++maxrun;
  for (size_t g1 = maxrun + 1; g1-- > 0;) {
    size_t q = g1;
    while (q < n && is_sep(line[q])) ++q;
    for (size_t sep_end = q + 1; sep_end-- > g1;) {
      size_t r = sep_end;
      while (r < n && line[r] != '\t') ++r;
      if (r > sep_end) {
        size_t s = r;
        while (s < n && is_sep(line[s])) ++s;
        name = line.substr(0, g1); label = line.substr(sep_end, r - sep_end); data = line.substr(s);
        return true;
      }
    }
  }
  return false;
}

struct LoadOptions : TokenizerOptions {
  std::string stoplist_file;       // empty = USE_EMPTY_STOPLIST
  int prune_count = 0;             // cfg key rare_threshold
  int keep_count = 0;              // cfg key tfidf_vocab_size (load_instances_keep)
};

struct Dataset {
  std::vector<int64_t> doc_ptr{0}; // D+1
  std::vector<int32_t> tokens;     // N type ids
  std::vector<std::string> vocab;  // the data alphabet, id -> word
  std::vector<std::string> names;  // instance names (regex group 1)
  std::vector<int32_t> labels;     // label ids (Target2Label), first-appearance order
  std::vector<std::string> label_alphabet;
  int64_t size() const { return (int64_t)doc_ptr.size() - 1; }
};

template <class PerLine>
inline void for_each_instance(const std::string &path, PerLine &&fn) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open dataset " + path);
  std::string line, name, label, data;
  int64_t lineno = 0;
  while (std::getline(f, line)) {
    ++lineno;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (!match_line(line, name, label, data)) throw std::runtime_error("Line #" + std::to_string(lineno) + " does not match regex:\n" + line);
    fn(name, label, data);
  }
}

// The second halves of loadInstancesPrune and loadInstancesKeep (LDAUtils.java:291-330, 413-452)
inline Dataset load_with_stoplist(const std::string &path, const Stoplist &stoplist, const LoadOptions &opt, const std::vector<std::string> *alphabet, bool frozen) {
  Dataset ds;
  std::unordered_map<std::string, int32_t> index, label_index;
  if (alphabet) {
    ds.vocab = *alphabet;
    for (size_t i = 0; i < ds.vocab.size(); ++i) index.emplace(ds.vocab[i], (int32_t)i);
  }
  for_each_instance(path, [&](const std::string &name, const std::string &label, const std::string &data) {
    tokenize(to_lower(decode_utf8(data)), stoplist, opt, [&](const std::string &t) {
      auto it = index.find(t);
      if (it == index.end()) {
        if (frozen) return;
        it = index.emplace(t, (int32_t)ds.vocab.size()).first;
        ds.vocab.push_back(t);
      }
      ds.tokens.push_back(it->second);
    });
    ds.doc_ptr.push_back((int64_t)ds.tokens.size());
    ds.names.push_back(name);
    auto li = label_index.find(label);
    if (li == label_index.end()) {
      li = label_index.emplace(label, (int32_t)ds.label_alphabet.size()).first;
      ds.label_alphabet.push_back(label);
    }
    ds.labels.push_back(li->second);
  });
  return ds;
}

// LDAUtils.loadInstancesPrune.  `alphabet`: an existing vocabulary to extend (a test set against the training
// alphabet); `frozen` = Alphabet.stopGrowth(): unknown words are dropped.
inline Dataset load_instances_prune(const std::string &path, const LoadOptions &opt, const std::vector<std::string> *alphabet = nullptr, bool frozen = false) {
  Stoplist stoplist;
  if (!opt.stoplist_file.empty()) stoplist = read_stoplist(opt.stoplist_file);
  if (opt.prune_count > 0) {
    std::unordered_map<std::string, int64_t> counts;
    for_each_instance(path, [&](const std::string &, const std::string &, const std::string &data) {
      tokenize(to_lower(decode_utf8(data)), stoplist, opt, [&](const std::string &t) { ++counts[t]; });
    });
    for (const auto &kv : counts)
      if (kv.second < opt.prune_count) stoplist.insert(kv.first);          // FeatureCountPipe.addPrunedWordsToStoplist
  }
  return load_with_stoplist(path, stoplist, opt, alphabet, frozen);
}

// LDAUtils.loadInstancesKeep (LDAUtils.java:355-452): the vocabulary cut by TF-IDF.  With opt.keep_count > 0 a first pass
// counts, per type, its occurrences (tf) and the documents it occurs in (df) -- indexing the caller's alphabet when one
// is given, as Java does --, every type ranked keep_count or later by tf * ln(D / df) (pipe/TfIdfPipe.java:73-104; 0 when
// either count is 0) joins the stoplist and the file is read again.  Equal weights rank by FALLING id: MALLET's
// IDSorter.compareTo (not under /root/reference) breaks ties that way, pinned by the reference's TfIdfPipeTest.testRank
// (the three weight-0 types of tfidf-samples.txt, ids 0, 1, 2, rank 5, 4, 3; TfIdfPipeTest.java:124-145).
// `grown`: receives the alphabet as the first pass left it (Java grows the caller's Alphabet object in place).
inline Dataset load_instances_keep(const std::string &path, const LoadOptions &opt, const std::vector<std::string> *alphabet = nullptr, bool frozen = false,
                                   std::vector<std::string> *grown = nullptr) {
  Stoplist stoplist;
  if (!opt.stoplist_file.empty()) stoplist = read_stoplist(opt.stoplist_file);
  std::vector<std::string> vocab;
  if (alphabet) vocab = *alphabet;
  if (opt.keep_count > 0) {
    std::unordered_map<std::string, int32_t> index;
    for (size_t i = 0; i < vocab.size(); ++i) index.emplace(vocab[i], (int32_t)i);
    std::vector<int64_t> tf(vocab.size(), 0), df(vocab.size(), 0), last_doc(vocab.size(), -1);
    int64_t docs = 0;
    for_each_instance(path, [&](const std::string &, const std::string &, const std::string &data) {
      tokenize(to_lower(decode_utf8(data)), stoplist, opt, [&](const std::string &t) {
        auto it = index.find(t);
        if (it == index.end()) {
          if (frozen) return;
          it = index.emplace(t, (int32_t)vocab.size()).first;
          vocab.push_back(t); tf.push_back(0); df.push_back(0); last_doc.push_back(-1);
        }
        const size_t i = (size_t)it->second;
        ++tf[i];
        if (last_doc[i] != docs) { last_doc[i] = docs; ++df[i]; }
      });
      ++docs;
    });
    std::vector<double> w(vocab.size());
    for (size_t i = 0; i < w.size(); ++i) w[i] = (tf[i] == 0 || df[i] == 0) ? 0.0 : (double)tf[i] * std::log((double)docs / (double)df[i]);
    std::vector<int32_t> ranks(vocab.size());
    for (size_t i = 0; i < ranks.size(); ++i) ranks[i] = (int32_t)i;
    std::sort(ranks.begin(), ranks.end(), [&](int32_t a, int32_t b) { return w[(size_t)a] != w[(size_t)b] ? w[(size_t)a] > w[(size_t)b] : a > b; });
    for (size_t r = (size_t)opt.keep_count; r < ranks.size(); ++r) stoplist.insert(vocab[(size_t)ranks[r]]);
  }
  if (grown) *grown = vocab;
  return load_with_stoplist(path, stoplist, opt, alphabet ? &vocab : nullptr, frozen);   // a caller's alphabet: as the first pass grew it
}

}  // namespace corpus
}  // namespace ggs
