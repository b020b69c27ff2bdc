"""Corpora for tests and the benchmark: integer CSR (doc_ptr, tokens).

The sampler consumes what MALLET's FeatureSequence.getFeatures() holds: one int type id
per token, documents in instance order.  Tokenisation, stop-lists and pruning stay on
the Java side (cc/mallet/util/LDAUtils.java:233-330) -- out of scope here; the small
loader below only covers the trivially tokenised bundled corpus (lower-case,
whitespace-separated) so the cats.txt case can be reproduced.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class Corpus:
    doc_ptr: np.ndarray   # int64 [D+1]
    tokens: np.ndarray    # int32 [N]
    num_types: int
    vocab: list = None

    @property
    def num_docs(self):
        return self.doc_ptr.size - 1

    @property
    def num_tokens(self):
        return int(self.doc_ptr[-1])

    def shard(self, doc_begin, doc_end):
        """Documents [doc_begin, doc_end) as a corpus of their own, plus the global index
        of its first document / token (what ggs_set_corpus calls doc_base / tok_base)."""
        b, e = int(self.doc_ptr[doc_begin]), int(self.doc_ptr[doc_end])
        sub = Corpus((self.doc_ptr[doc_begin:doc_end + 1] - b).astype(np.int64), self.tokens[b:e].copy(), self.num_types)
        return sub, doc_begin, b


def load_tsv_corpus(path):
    """``name<TAB>label<TAB>text`` lines (the CsvIterator regex of LDAUtils.java:236);
    type id = order of first appearance, as an empty-stoplist / rare_threshold=0 /
    keep_numbers=true MALLET pipe assigns it (plda-cats-test.cfg:21-24)."""
    vocab, index, doc_ptr, toks = [], {}, [0], []
    with open(path, encoding="utf-8") as f:
        for line in f:
            line = line.rstrip("\n")
            if not line.strip():
                continue
            parts = line.split("\t", 2)
            text = parts[2] if len(parts) == 3 else ""
            for w in text.lower().split():
                i = index.get(w)
                if i is None:
                    i = index[w] = len(vocab)
                    vocab.append(w)
                toks.append(i)
            doc_ptr.append(len(toks))
    return Corpus(np.asarray(doc_ptr, np.int64), np.asarray(toks, np.int32), len(vocab), vocab)


def synthetic_lda_corpus(num_docs, num_types, mean_doc_len, true_topics=100, seed=2019, zipf_s=1.07,
                         topic_conc=0.01, doc_conc=0.1, shuffle_within_doc=True):
    """Deterministic LDA-generated corpus: len_d = max(1, Poisson(mean)),
    theta*_d ~ Dir(doc_conc), phi*_k ~ Dir(topic_conc * V * m) with m a Zipf(zipf_s) base
    measure over a random permutation of the vocabulary (Zipfian word marginals => realistic
    head-word contention on the count updates)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    D, V, Kt = int(num_docs), int(num_types), int(true_topics)
    lens = np.maximum(1, rng.poisson(mean_doc_len, D)).astype(np.int64)
    doc_ptr = np.zeros(D + 1, np.int64)
    np.cumsum(lens, out=doc_ptr[1:])
    N = int(doc_ptr[-1])
    base = 1.0 / np.arange(1, V + 1, dtype=np.float64) ** zipf_s
    base /= base.sum()
    base = base[rng.permutation(V)]
    theta = rng.dirichlet(np.full(Kt, doc_conc), D)
    counts = rng.multinomial(lens, theta)                    # [D][Kt] tokens of doc d from topic k
    del theta
    # slot of (d,k)'s first token inside the CSR: documents laid out topic-major first
    within = np.cumsum(counts, axis=1) - counts
    tokens = np.empty(N, np.int32)
    for k in range(Kt):
        g = rng.standard_gamma(np.maximum(topic_conc * V * base, 1e-300))
        s = g.sum()
        phi = g / s if s > 0 else base
        cdf = np.cumsum(phi)
        cdf /= cdf[-1]
        ck = counts[:, k]
        nk = int(ck.sum())
        if nk == 0:
            continue
        words = np.searchsorted(cdf, rng.random(nk), side="right").astype(np.int32)
        np.minimum(words, V - 1, out=words)
        start = doc_ptr[:-1] + within[:, k]
        # position of every topic-k token: start[d] + 0..ck[d]-1
        rep = np.repeat(start - (np.cumsum(ck) - ck), ck)
        tokens[rep + np.arange(nk)] = words
    if shuffle_within_doc:
        key = rng.random(N)
        doc_of = np.repeat(np.arange(D, dtype=np.int64), lens)
        order = np.lexsort((key, doc_of))
        tokens = tokens[order]
    return Corpus(doc_ptr, tokens, V)


def zipf_unigram_corpus(num_docs, num_types, mean_doc_len, seed=2019, zipf_s=1.07):
    """Cheap corpus for Wikipedia-scale shapes (BASELINE config 5; SURVEY.md 8d allows unigram words there for the
    generation cost): len_d = max(1, Poisson(mean)), every token an independent draw from a Zipf(zipf_s) law over a
    random permutation of the vocabulary.  Generated in slabs so that the 1e8-token case stays within a few GB."""
    rng = np.random.Generator(np.random.PCG64(seed))
    D, V = int(num_docs), int(num_types)
    lens = np.maximum(1, rng.poisson(mean_doc_len, D)).astype(np.int64)
    doc_ptr = np.zeros(D + 1, np.int64)
    np.cumsum(lens, out=doc_ptr[1:])
    N = int(doc_ptr[-1])
    p = 1.0 / np.arange(1, V + 1, dtype=np.float64) ** zipf_s
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    perm = rng.permutation(V).astype(np.int32)
    tokens = np.empty(N, np.int32)
    slab = 1 << 24
    for b in range(0, N, slab):
        e = min(N, b + slab)
        r = np.searchsorted(cdf, rng.random(e - b), side="right")
        np.minimum(r, V - 1, out=r)
        tokens[b:e] = perm[r]
    return Corpus(doc_ptr, tokens, V)


def random_corpus(num_docs, num_types, max_len, seed=0, empty_every=0):
    """Small ragged test corpus: uniform lengths in [0, max_len], Zipf-ish words."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, num_docs)
    if empty_every:
        lens[::empty_every] = 0
    doc_ptr = np.zeros(num_docs + 1, np.int64)
    np.cumsum(lens, out=doc_ptr[1:])
    p = 1.0 / np.arange(1, num_types + 1) ** 1.0
    p /= p.sum()
    tokens = rng.choice(num_types, int(doc_ptr[-1]), p=p).astype(np.int32)
    return Corpus(doc_ptr, tokens, int(num_types))


def even_split(n, parts):
    """Even contiguous split: sizes n//parts + (remainder > b), the rule of
    randomscan/document/EvenSplitBatchBuilder.java:30-44 (documents -> batches there,
    documents -> GPUs here).  Returns the parts+1 boundaries."""
    size, rem = divmod(int(n), int(parts))
    bounds = [0]
    for b in range(parts):
        bounds.append(bounds[-1] + size + (1 if rem > b else 0))
    return bounds
