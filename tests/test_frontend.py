"""Corpus front-end (second half): util/LDAUtils.loadInstancesPrune + the tokenizers of cc/mallet/pipe,
restated in Python (ldagroupedgibbssampler_amd/frontend.py) and natively in C++ (include/ggs_corpus.hpp).

Pinned by the REFERENCE'S OWN known answers, on the data files its tests read (copied as data under tests/golden/datasets):
  LDAUtilsTest.testLoadInstances           SmallTexts.txt, rare_threshold 0            -> 5 instances        (LDAUtilsTest.java:291-295)
  LDAUtilsTest.testLoadInstancesPrune      SmallTexts.txt, rare_threshold 2, numbers   -> 7 types            (:297-301)
  SimpleTokenizerLargeTest.testSpecialChars  special_chars.txt: "but_i_can" is a type only with keep_connecting_punctuation (:78-98)
  SimpleTokenizerLargeTest.testIntegrationRareWordPrune  a token longer than max_doc_buf_size -> ArrayIndexOutOfBoundsException (:118-136)
  the bundled cats corpus                  D=23, V=303, N=7788, ids in first-appearance order
  TfIdfPipeTest (tfidf-samples.txt)        tf, df, ranks incl. the tie rule, cut sizes of loadInstancesKeep          (TfIdfPipeTest.java:41-243)
The C++ loader must produce exactly what the Python one produces, on those files and on random Unicode text."""
import os
import subprocess

import numpy as np
import pytest

from ldagroupedgibbssampler_amd import frontend as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "datasets")


def test_reference_known_answers():
    a = F.load_instances_prune(os.path.join(DATA, "SmallTexts.txt"), None, 0, True)
    assert a.corpus.num_docs == 5                                                        # testLoadInstances
    b = F.load_instances_prune(os.path.join(DATA, "SmallTexts.txt"), None, 2, True)
    assert b.corpus.num_types == 7                                                       # testLoadInstancesPrune
    assert b.corpus.vocab == ["but", "the", "intel", "inside", "is", "a", "warning"]     # first-appearance order of the survivors
    c = F.load_instances_prune(os.path.join(DATA, "special_chars.txt"), None, 0, True, 10000, False)
    assert "but_i_can" not in c.corpus.vocab and "but" in c.corpus.vocab                 # testSpecialChars
    c = F.load_instances_prune(os.path.join(DATA, "special_chars.txt"), None, 0, True, 10000, True)
    assert "but_i_can" in c.corpus.vocab
    with pytest.raises(F.TokenBufferOverflow):                                           # testIntegrationRareWordPrune (buffer of 10)
        F.load_instances_prune(os.path.join(DATA, "SmallTexts.txt"), None, 0, True, 3)
    assert a.names == ["docno:%d" % i for i in range(1, 6)] and a.label_alphabet == ["X"]


def test_tfidf_vocabulary_cut(tmp_path):
    """LDAUtils.loadInstancesKeep.  Reference-held answers: SimpleTokenizerLargeTest.testSpecialChars runs it with
    tfidf_vocab_size = 7700 (special_chars.cfg:15) and finds `but_i_can` only with keep_connecting_punctuation (:78-98);
    testIntegrationTfIdfPrune expects the token-buffer ArrayIndexOutOfBoundsException out of the FIRST pass (:50-75).
    TfIdfPipeTest (tfidf-samples.txt) holds the counts, the ranking INCLUDING its tie rule, and the cut sizes.
    The cut on a larger file (which types survive) is checked against TfIdfPipe's formula recomputed here."""
    ts = os.path.join(DATA, "tfidf-samples.txt")
    full = F.load_instances_prune(ts, None, 0, True)
    assert full.corpus.vocab == ["this", "is", "a", "sample", "another", "example"]
    tf = np.bincount(full.corpus.tokens, minlength=6).tolist()
    df = [sum(1 for d in range(2) if i in full.corpus.tokens[full.corpus.doc_ptr[d]:full.corpus.doc_ptr[d + 1]]) for i in range(6)]
    assert tf == [2, 2, 2, 1, 2, 3]                                                       # TfIdfPipeTest.testTf
    assert df == [2, 2, 2, 1, 1, 1]                                                       # testIdf
    assert F.tfidf_ranks(tf, df, 2)[0] == [5, 4, 3, 2, 1, 0]                              # testRank: weight-0 ties by falling id
    assert F.load_instances_keep(ts, None, 3, True).corpus.num_types == 3                 # testLoadInstances
    assert F.load_instances_keep(ts, None, 3, True).corpus.vocab == ["sample", "another", "example"]
    assert F.load_instances_keep(ts, None, 2, True).corpus.num_types == 2                 # testCutoff: two types left unstopped
    assert F.load_instances_keep(ts, None, -1, True).corpus.num_types == 6                # testLoadInstancesNegCnt
    st = F.load_instances_keep(os.path.join(DATA, "SmallTexts.txt"), None, 20, True)
    assert st.corpus.num_types == 20                                                      # LDAUtilsTest.testLoadInstancesKeep (:303-307)
    again = F.load_instances_keep(os.path.join(DATA, "SmallTexts.txt"), None, 20, True, 1000, True, tuple(st.corpus.vocab))
    assert again.corpus.vocab == st.corpus.vocab and again.corpus.num_docs == 5           # the pattern of testLoadTestInstancesKeep (:309-320): frozen training alphabet
    sc = os.path.join(DATA, "special_chars.txt")
    c = F.load_instances_keep(sc, None, 7700, True, 10000, False)
    assert "but_i_can" not in c.corpus.vocab and "but" in c.corpus.vocab
    c = F.load_instances_keep(sc, None, 7700, True, 10000, True)
    assert "but_i_can" in c.corpus.vocab
    with pytest.raises(F.TokenBufferOverflow):
        F.load_instances_keep(os.path.join(DATA, "SmallTexts.txt"), None, 7700, True, 3)
    # keep_count <= 0: no cut at all (the loader then equals loadInstancesPrune without a threshold)
    cats_path = os.path.join(DATA, "cats.txt")
    full = F.load_instances_prune(cats_path, None, 0, True)
    same = F.load_instances_keep(cats_path, None, 0, True)
    assert same.corpus.vocab == full.corpus.vocab and np.array_equal(same.corpus.tokens, full.corpus.tokens)
    # the cut on cats: tf, df recomputed from the uncut encoding, weight = tf * ln(D / df), equal weights by falling id
    V, D = full.corpus.num_types, full.corpus.num_docs
    tf = np.bincount(full.corpus.tokens, minlength=V)
    df = np.zeros(V, np.int64)
    for d in range(D):
        df[np.unique(full.corpus.tokens[full.corpus.doc_ptr[d]:full.corpus.doc_ptr[d + 1]])] += 1
    w = tf * np.log(D / df)
    for keep in (1, 17, 100, 302, 303, 5000):
        kept_ids = sorted(sorted(range(V), key=lambda i: (-w[i], -i))[:keep])
        want_vocab_set = {full.corpus.vocab[i] for i in kept_ids}
        k = F.load_instances_keep(cats_path, None, keep, True)
        assert set(k.corpus.vocab) == want_vocab_set and k.corpus.num_types == min(keep, V)
        # survivors keep their relative order of first appearance; the token stream is the uncut one minus the dropped types
        keep_mask = np.isin(full.corpus.tokens, kept_ids)
        assert [k.corpus.vocab[i] for i in k.corpus.tokens] == [full.corpus.vocab[i] for i in full.corpus.tokens[keep_mask]]
        assert k.corpus.num_docs == D
    # a type that occurs in every document weighs tf * ln(1) = 0 however frequent it is, and goes first when the cut bites;
    # equal weights rank by falling id (y before x: one occurrence in one of three documents each)
    small = tmp_path / "every.txt"
    small.write_text("d1\tL\tthe the the x\nd2\tL\tthe y\nd3\tL\tthe z z\n", encoding="utf-8")
    assert F.load_instances_keep(str(small), None, 3, True).corpus.vocab == ["x", "y", "z"]
    assert F.load_instances_keep(str(small), None, 2, True).corpus.vocab == ["y", "z"]        # z: 2 ln 3; x, y: ln 3 each, y (the later id) first
    assert F.load_instances_keep(str(small), None, 1, True).corpus.vocab == ["z"]
    # load_dataset dispatches on tfidf_vocab_size as LDAUtils.loadDataset does (LDAUtils.java:163-181)
    assert F.load_dataset(cats_path, stoplist=None, tfidf_vocab_size=50).corpus.num_types == 50
    # a test set against the training alphabet: cut among the training ids, unknown words dropped when the alphabet is frozen
    t = F.load_instances_keep(os.path.join(DATA, "SmallTexts.txt"), None, 5, True, data_alphabet=tuple(full.corpus.vocab))
    assert t.corpus.num_types == V and len(set(t.corpus.tokens.tolist())) <= 5


def test_cats_is_the_golden_encoding(cats):
    d = F.load_dataset(os.path.join(DATA, "cats.txt"), stoplist=None)                    # plda-cats-test.cfg:21-24: empty stoplist, keep numbers
    assert (d.corpus.num_docs, d.corpus.num_types, d.corpus.num_tokens) == (23, 303, 7788)
    assert np.array_equal(d.corpus.tokens, cats.tokens) and np.array_equal(d.corpus.doc_ptr, cats.doc_ptr)


def test_tokenizer_rules():
    tk = F.tokenize
    assert tk("the cat's pyjamas, e-mail x_y 42nd") == ["the", "cat", "s", "pyjamas", "e", "mail", "x", "y", "42nd"]
    assert tk("x_y 42nd", keep_connectors=True, keep_numbers=False) == ["x_y", "nd"]
    assert tk("a\tb\nc d") == ["abc", "d"]                         # controls are skipped WITHOUT ending the token (Character.CONTROL falls through)
    assert tk("3.14 + 2 = 5.14") == ["3", "14", "2", "5", "14"]    # '+' and '=' are MATH_SYMBOL: skipped, the spaces delimit
    assert tk("a+b") == ["ab"]
    assert tk("naïve café ñu") == ["naïve", "café", "ñu"]
    assert tk("stop me now", stoplist={"me"}) == ["stop", "now"]
    assert tk("ab \U0001D400cd ef gh") == ["ab", "\U0001D400cd", "ef", "g"]   # the codePointAt(i) indexing quirk: one supplementary character costs the tail one unit
    assert "ΟΔΥΣΣΕΥΣ ΣΟΦΟΣ".lower() == "οδυσσευς σοφος"            # final sigma: what CharSequenceLowercase hands the tokenizer
    assert F.LINE_REGEX.search("n1\tlabel\tsome text").groups() == ("n1", "label", "some text")
    assert F.LINE_REGEX.search("n1 lab, the rest").groups() == ("n1", "lab, the rest", "")   # no tab: everything is label, the data is empty
    assert F.LINE_REGEX.search("") is None


@pytest.fixture(scope="module")
def corpus_demo(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fe") / "ggs_corpus_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "ggs_corpus_demo.cpp"), "-o", exe])
    return exe


def run_cpp(exe, path, stoplist, prune, numbers, buf, connectors, extra=()):
    out = subprocess.run([exe, path, stoplist or "-", str(prune), str(int(numbers)), str(buf), str(int(connectors))] + list(extra), capture_output=True)
    if out.returncode == 3:
        raise F.TokenBufferOverflow(out.stderr.decode())
    assert out.returncode == 0, out.stderr.decode()
    lines = out.stdout.decode("utf-8").split("\n")
    D, V, N = map(int, lines[0].split())
    return dict(doc_ptr=np.array(lines[1].split(), np.int64), tokens=np.array(lines[2].split(), np.int32), labels=np.array(lines[3].split(), np.int32),
                vocab=lines[4:4 + V], names=lines[4 + V:4 + V + D], D=D, V=V, N=N)


def same(cpp, py):
    assert (cpp["D"], cpp["V"], cpp["N"]) == (py.corpus.num_docs, py.corpus.num_types, py.corpus.num_tokens)
    assert np.array_equal(cpp["doc_ptr"], py.corpus.doc_ptr) and np.array_equal(cpp["tokens"], py.corpus.tokens)
    assert cpp["vocab"] == py.corpus.vocab and cpp["names"] == py.names and np.array_equal(cpp["labels"], py.labels)


@pytest.mark.parametrize("name,stop,prune,numbers,connectors", [
    ("SmallTexts.txt", None, 0, True, False), ("SmallTexts.txt", None, 2, True, False), ("SmallTexts.txt", "stoplist.txt", 0, False, False),
    ("special_chars.txt", None, 0, True, True), ("special_chars.txt", None, 0, False, False), ("small.txt", None, 0, True, False),
    ("small.txt", None, 0, False, False), ("cats.txt", None, 0, True, False), ("cats.txt", "stoplist.txt", 3, True, True)])
def test_cpp_loader_equals_python_loader_on_the_bundled_datasets(corpus_demo, name, stop, prune, numbers, connectors):
    path = os.path.join(DATA, name)
    stop = os.path.join(DATA, stop) if stop else None
    same(run_cpp(corpus_demo, path, stop, prune, numbers, 10000, connectors), F.load_instances_prune(path, stop, prune, numbers, 10000, connectors))


@pytest.mark.parametrize("name,stop,keep,numbers,connectors", [
    ("special_chars.txt", None, 7700, True, False), ("special_chars.txt", None, 7700, True, True), ("SmallTexts.txt", None, 6, True, False),
    ("tfidf-samples.txt", None, 3, True, False), ("tfidf-samples.txt", None, 2, True, False), ("tfidf-samples.txt", None, 4, True, False),
    ("cats.txt", None, 100, True, False), ("cats.txt", "stoplist.txt", 40, False, True), ("small.txt", None, 25, True, False)])
def test_cpp_tfidf_cut_equals_python(corpus_demo, name, stop, keep, numbers, connectors, tmp_path):
    path = os.path.join(DATA, name)
    stop = os.path.join(DATA, stop) if stop else None
    same(run_cpp(corpus_demo, path, stop, "tfidf:%d" % keep, numbers, 10000, connectors), F.load_instances_keep(path, stop, keep, numbers, 10000, connectors))
    if name == "cats.txt" and stop is None:             # a test set cut among the training alphabet's ids (grown, and frozen)
        train = F.load_instances_prune(os.path.join(DATA, "SmallTexts.txt"), None, 0, True)
        alpha = tmp_path / "alphabet.txt"
        alpha.write_text("\n".join(train.corpus.vocab) + "\n", encoding="utf-8")
        same(run_cpp(corpus_demo, path, None, "tfidf:%d" % keep, True, 10000, False, [str(alpha), "1"]),
             F.load_instances_keep(path, None, keep, True, 10000, False, tuple(train.corpus.vocab)))
        same(run_cpp(corpus_demo, path, None, "tfidf:%d" % keep, True, 10000, False, [str(alpha), "0"]),
             F.load_instances_keep(path, None, keep, True, 10000, False, list(train.corpus.vocab)))
    with pytest.raises(F.TokenBufferOverflow):          # testIntegrationTfIdfPrune: out of the first pass
        run_cpp(corpus_demo, os.path.join(DATA, "SmallTexts.txt"), None, "tfidf:7700", True, 3, False)


def test_cpp_loader_known_answers_and_overflow(corpus_demo):
    assert run_cpp(corpus_demo, os.path.join(DATA, "SmallTexts.txt"), None, 2, True, 10000, False)["V"] == 7
    with pytest.raises(F.TokenBufferOverflow):
        run_cpp(corpus_demo, os.path.join(DATA, "SmallTexts.txt"), None, 0, True, 3, False)


def test_cpp_loader_equals_python_loader_on_random_unicode(corpus_demo, tmp_path):
    """Random documents over an alphabet that exercises every branch: ASCII, Latin-1, Greek with capital sigmas in and at
    the end of words, the dotted capital I, combining marks, CJK, digits of several scripts, every punctuation class,
    symbols, controls, supplementary-plane letters, and separators of the line regex in odd places."""
    rng = np.random.default_rng(123)
    pool = (list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ") * 3 + list("0123456789") + list("  \t,;.!?'\"()[]{}-_+=*/<>|~`@#$%^&") * 2
            + list("àéîõüñçßÀÉÎÕÜÑÇ") + list("αβγδεζσςΑΒΓΔΣΣΣ") + ["İ", "ı", "ǅ", "ʰ", "́", "̈", "҉", "ः"] + list("日本語中文한국") + list("٠١٢३४५")
            + ["‐", "–", "—", "‘", "’", "“", "”", "«", "»", "‿", "⁀", " ", " ", "　", "\x0b", "\x1f", "­", "€", "√", "Ⅷ", "½"]
            + ["\U0001D400", "\U00010400", "\U0001F600", "\U00020000"])
    lines = []
    for d in range(300):
        n = int(rng.integers(0, 60))
        text = "".join(rng.choice(pool, n).tolist()).replace("\n", " ").replace("\r", " ")
        sep1, sep2 = rng.choice(["\t", " \t", "\t ", ",\t"], 2)
        lines.append("doc%d%sL%d%s%s" % (d, sep1, d % 3, sep2, text))
    path = str(tmp_path / "random.txt")
    with open(path, "w", encoding="utf-8", newline="\n") as f:
        f.write("\n".join(lines) + "\n")
    for prune, numbers, connectors in [(0, True, False), (2, False, True), (3, True, True)]:
        same(run_cpp(corpus_demo, path, None, prune, numbers, 10000, connectors), F.load_instances_prune(path, None, prune, numbers, 10000, connectors))


def test_test_set_against_the_training_alphabet(corpus_demo, tmp_path):
    """LDAUtils.loadInstancesPrune(..., dataAlphabet) (LDAUtils.java:252-257,298-303; the pattern of LDAUtilsTest.testLoadTestInstancesPrune):
    a test set loaded against the training vocabulary keeps the training ids; with the alphabet frozen (Alphabet.stopGrowth)
    unknown words are dropped, otherwise they extend it."""
    train = F.load_instances_prune(os.path.join(DATA, "SmallTexts.txt"), None, 0, True)
    path = os.path.join(DATA, "special_chars.txt")
    frozen = F.load_instances_prune(path, None, 0, True, 10000, True, tuple(train.corpus.vocab))
    assert frozen.corpus.vocab == train.corpus.vocab and frozen.corpus.num_docs == 5
    assert frozen.corpus.num_tokens == train.corpus.num_tokens - 3            # "but_i_can" is one unknown word now: dropped
    grown = F.load_instances_prune(path, None, 0, True, 10000, True, list(train.corpus.vocab))
    assert grown.corpus.vocab[:train.corpus.num_types] == train.corpus.vocab and "but_i_can" in grown.corpus.vocab[train.corpus.num_types:]
    alpha = tmp_path / "alphabet.txt"
    alpha.write_text("\n".join(train.corpus.vocab) + "\n", encoding="utf-8")
    same(run_cpp(corpus_demo, path, None, 0, True, 10000, True, [str(alpha), "1"]), frozen)
    same(run_cpp(corpus_demo, path, None, 0, True, 10000, True, [str(alpha), "0"]), grown)
