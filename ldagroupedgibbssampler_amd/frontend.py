"""Corpus front-end: the reference's dataset loader restated -- text file in, integer CSR out (second half).

What `tui/ParallelLDA` does before the sampler sees a token (ParallelLDA.java:170 -> util/LDAUtils.loadDataset,
LDAUtils.java:140-182 -> loadInstancesPrune, :233-330):

    line  --CsvIterator regex-->  (name, label, data)               LDAUtils.java:236-239
    data  --CharSequenceLowercase-->  lower case                    :259,307
          --tokenizer-->  tokens (Unicode general categories)       pipe/SimpleTokenizerLarge.java:52-135 and its three
                                                                    variants (NumericAlsoTokenizer, KeepConnectorPunctuation*),
                                                                    chosen by LDAUtils.initTokenizer, :532-563
          --stoplist-->  dropped if listed                          one word per line of the stoplist file
          --StringList2FeatureSequence(alphabet)-->  type ids       first appearance order
    rare_threshold > 0: a first pass counts every type; types seen fewer than `rare_threshold` times join the stoplist
    (FeatureCountPipe.addPrunedWordsToStoplist) and the file is read again with a fresh alphabet      :243-289
    tfidf_vocab_size > 0 (loadInstancesKeep, :355-452) instead ranks the types by tf * ln(D / df) (pipe/TfIdfPipe.java) and
    stops all but the first tfidf_vocab_size of them

Not under /root/reference (cc.mallet:mallet:2.0.8, pom.xml:130-134), restated from the published sources: CsvIterator
(one instance per line, `find()` of the regex, IllegalStateException on a line that does not match), CharSequenceLowercase
(`toString().toLowerCase()`), SimpleTokenizer(File) (every line of the file is a stop word, untrimmed),
StringList2FeatureSequence, FeatureCountPipe (prunes counts `< minimumCount`).

Pinned by the reference's own known answers (tests/test_frontend.py): LDAUtilsTest.testLoadInstances /
testLoadInstancesPrune (SmallTexts.txt: 5 instances; rare_threshold 2 leaves 7 types), SimpleTokenizerLargeTest.testSpecialChars
(`but_i_can` only with keep_connecting_punctuation), the token-buffer ArrayIndexOutOfBoundsException of
testIntegrationRareWordPrune, and the bundled cats corpus (D=23, V=303, N=7788).

Two things follow the JDK rather than this file's host and are stated where they matter: `Character.getType` speaks
the Unicode version of the running JDK (Java 8: 6.2) while `unicodedata` here speaks Python's (code points assigned
since then are UNASSIGNED to an older JDK, i.e. dropped); `String.toLowerCase()` uses the default locale (tr/az/lt
have special dotted-i rules) while `str.lower()` is locale-free.  ASCII and Latin-1 corpora are unaffected.
"""
import re
import unicodedata

import numpy as np

from .corpus import Corpus

# "^(\\S*)[\\s,]*([^\\t]+)[\\s,]*(.*)$", LDAUtils.java:236; java.util.regex \s and \S are ASCII-only by default
LINE_REGEX = re.compile(r"^(\S*)[\s,]*([^\t]+)[\s,]*(.*)$", re.ASCII)

_LETTERS = {"Ll", "Lu"}
_DELIMS = {"Zs", "Zl", "Zp", "Pe", "Pd", "Pc", "Ps", "Pi", "Pf", "Po"}
_WORD_PARTS = {"Mc", "Me", "Mn", "Lt", "Lm", "Lo"}


class TokenBufferOverflow(IndexError):
    """java.lang.ArrayIndexOutOfBoundsException out of the tokenizer: a token longer than max_doc_buf_size code points
    (SimpleTokenizerLarge.java:62,76; the reference's tests expect exactly this, SimpleTokenizerLargeTest.java:118-136)."""


def read_stoplist(path):
    """SimpleTokenizer(File): every line is a stop word (UTF-8, untrimmed).  None -> the empty stoplist."""
    if path is None:
        return set()
    with open(path, encoding="utf-8") as f:
        return {line.rstrip("\n").rstrip("\r") for line in f}


def _utf16_units(s):
    b = s.encode("utf-16-le", "surrogatepass")
    return np.frombuffer(b, "<u2").tolist()


def tokenize(text, stoplist=frozenset(), keep_numbers=True, keep_connectors=False, buffer_size=10000):
    """SimpleTokenizerLarge.pipe / NumericAlsoTokenizer.pipe / KeepConnectorPunctuation*.pipe on an already lower-cased
    string.  Faithful to the Java loop including its indexing: `for (i < codePointCount) codePointAt(chars, i)` walks
    UTF-16 *units* but stops after as many steps as there are code points, so text behind supplementary characters
    loses its tail -- reproduced, not repaired."""
    units = _utf16_units(text)
    n_units = len(units)
    total_code_points = n_units - sum(1 for i in range(n_units - 1) if 0xD800 <= units[i] < 0xDC00 and 0xDC00 <= units[i + 1] < 0xE000)
    tokens, buf = [], []

    def flush():
        if buf:
            token = "".join(chr(c) for c in buf)
            if token not in stoplist:
                tokens.append(token)
            buf.clear()

    for i in range(total_code_points):
        cp = units[i]
        if 0xD800 <= cp < 0xDC00 and i + 1 < n_units and 0xDC00 <= units[i + 1] < 0xE000:
            cp = 0x10000 + ((cp - 0xD800) << 10) + (units[i + 1] - 0xDC00)
        cat = "Cs" if 0xD800 <= cp < 0xE000 else unicodedata.category(chr(cp))
        if cat in _LETTERS or cat in _WORD_PARTS or (keep_connectors and cat == "Pc") or (keep_numbers and cat == "Nd"):
            if len(buf) >= buffer_size:
                raise TokenBufferOverflow("token longer than the token buffer (%d code points)" % buffer_size)
            buf.append(cp)
        elif cat in _DELIMS:
            flush()
        # everything else -- controls (tab and newline among them), symbols, other numbers, unassigned -- is skipped
        # without ending the token (the last else branch of the Java loop)
    flush()
    return tokens


def iter_instances(path):
    """CsvIterator(new FileReader(file), lineRegex, data=3, label=2, name=1): one (name, label, data) per line."""
    with open(path, encoding="utf-8", newline=None) as f:
        for lineno, line in enumerate(f, 1):
            line = line.rstrip("\n")
            m = LINE_REGEX.search(line)
            if m is None:
                raise ValueError("Line #%d does not match regex:\n%s" % (lineno, line))      # CsvIterator's IllegalStateException
            yield m.group(1), m.group(2), m.group(3)


class LoadedDataset:
    """What an InstanceList carries for the sampler and the driver's outputs."""

    def __init__(self, corpus, names, labels, label_alphabet):
        self.corpus, self.names, self.labels, self.label_alphabet = corpus, names, labels, label_alphabet


def load_instances_prune(path, stoplist_file=None, prune_count=0, keep_numbers=True, max_buf_size=10000, keep_connectors=False,
                         data_alphabet=None):
    """LDAUtils.loadInstancesPrune (LDAUtils.java:233-330).  `data_alphabet`: an existing vocabulary list (a test set
    loaded against the training alphabet, LDAUtils.java:252-257,298-303); it grows unless the caller froze it by
    passing a tuple (Alphabet.stopGrowth: unknown words are then dropped, as lookupIndex returns -1 and
    StringList2FeatureSequence skips them)."""
    stoplist = read_stoplist(stoplist_file)
    if prune_count > 0:
        counts = {}
        for _, _, data in iter_instances(path):
            for t in tokenize(data.lower(), stoplist, keep_numbers, keep_connectors, max_buf_size):
                counts[t] = counts.get(t, 0) + 1
        stoplist = stoplist | {t for t, c in counts.items() if c < prune_count}           # FeatureCountPipe.addPrunedWordsToStoplist
    return _load_with_stoplist(path, stoplist, keep_numbers, max_buf_size, keep_connectors, data_alphabet)


def _load_with_stoplist(path, stoplist, keep_numbers, max_buf_size, keep_connectors, data_alphabet):
    """The second halves of loadInstancesPrune and loadInstancesKeep (LDAUtils.java:291-330, 413-452): lower case,
    tokenise, drop stop words, index against the alphabet, label."""
    frozen = isinstance(data_alphabet, tuple)
    vocab = list(data_alphabet) if data_alphabet is not None else []
    index = {w: i for i, w in enumerate(vocab)}
    doc_ptr, toks, names, labels, label_alphabet, label_index = [0], [], [], [], [], {}
    for name, label, data in iter_instances(path):
        for t in tokenize(data.lower(), stoplist, keep_numbers, keep_connectors, max_buf_size):
            i = index.get(t)
            if i is None:
                if frozen:
                    continue
                i = index[t] = len(vocab)
                vocab.append(t)
            toks.append(i)
        doc_ptr.append(len(toks))
        names.append(name)
        if label not in label_index:                                                      # Target2Label
            label_index[label] = len(label_alphabet)
            label_alphabet.append(label)
        labels.append(label_index[label])
    corpus = Corpus(np.asarray(doc_ptr, np.int64), np.asarray(toks, np.int32), len(vocab), vocab)
    return LoadedDataset(corpus, names, np.asarray(labels, np.int32), label_alphabet)


def tfidf_ranks(tf, df, corpus_size):
    """TfIdfPipe.getTfIdf + freqSortWords (pipe/TfIdfPipe.java:73-104): weight = tf * ln(corpusSize / df) (0 when either
    count is 0), types ordered by falling weight; EQUAL weights by falling id -- MALLET's IDSorter.compareTo (not under
    /root/reference) breaks ties that way, which the reference's own TfIdfPipeTest.testRank pins: the three weight-0
    types of tfidf-samples.txt, ids 0, 1, 2, rank 5, 4, 3 (TfIdfPipeTest.java:124-145)."""
    import math
    w = [0.0 if (t == 0 or d == 0) else float(t) * math.log(corpus_size / float(d)) for t, d in zip(tf, df)]
    return sorted(range(len(w)), key=lambda i: (-w[i], -i)), w


def load_instances_keep(path, stoplist_file=None, keep_count=0, keep_numbers=True, max_buf_size=10000, keep_connectors=False,
                        data_alphabet=None):
    """LDAUtils.loadInstancesKeep (LDAUtils.java:355-452): the vocabulary cut by TF-IDF.  With keep_count > 0 a first
    pass over the file counts, per type, its occurrences (tf) and the documents it occurs in (df); every type ranked
    keep_count or later by tf * ln(D / df) joins the stoplist (TfIdfPipe.addPrunedWordsToStoplist) and the file is read
    again.  As in Java the first pass indexes a caller's alphabet when one is given (and grows it unless it is frozen:
    a tuple), so a test set is cut among the training vocabulary's ids; without one each pass starts a fresh alphabet."""
    stoplist = read_stoplist(stoplist_file)
    if keep_count > 0:
        frozen = isinstance(data_alphabet, tuple)
        vocab = data_alphabet if isinstance(data_alphabet, list) else (list(data_alphabet) if data_alphabet is not None else [])
        index = {w: i for i, w in enumerate(vocab)}
        tf, df, docs = [0] * len(vocab), [0] * len(vocab), 0
        for _, _, data in iter_instances(path):
            seen = set()
            for t in tokenize(data.lower(), stoplist, keep_numbers, keep_connectors, max_buf_size):
                i = index.get(t)
                if i is None:
                    if frozen:
                        continue
                    i = index[t] = len(vocab)
                    vocab.append(t); tf.append(0); df.append(0)
                tf[i] += 1
                if i not in seen:
                    seen.add(i); df[i] += 1
            docs += 1
        ranks, _ = tfidf_ranks(tf, df, docs)
        stoplist = stoplist | {vocab[i] for i in ranks[keep_count:]}
    return _load_with_stoplist(path, stoplist, keep_numbers, max_buf_size, keep_connectors, data_alphabet)


def load_dataset(path, stoplist="stoplist.txt", rare_threshold=0, keep_numbers=True, max_doc_buf_size=10000,
                 keep_connecting_punctuation=False, alphabet=None, tfidf_vocab_size=-1):
    """LDAUtils.loadDataset for a single file (LDAUtils.java:140-182; defaults of LDAConfiguration.java: stoplist.txt,
    rare_threshold 0, max_doc_buf_size 10000, tfidf_vocab_size -1): the TF-IDF cut when tfidf_vocab_size > 0, the
    rare-word cut otherwise.  A DIRECTORY of documents (loadInstanceDirectory: another tokenisation, and the files in
    File.listFiles order, which Java leaves to the file system) is not provided."""
    if tfidf_vocab_size > 0:
        return load_instances_keep(path, stoplist, tfidf_vocab_size, keep_numbers, max_doc_buf_size, keep_connecting_punctuation, alphabet)
    return load_instances_prune(path, stoplist, rare_threshold, keep_numbers, max_doc_buf_size, keep_connecting_punctuation, alphabet)
