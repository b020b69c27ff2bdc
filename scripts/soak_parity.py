#!/usr/bin/env python3
"""Longer parity run than the test suite affords: the benchmark corpus (10 % held out), N sweeps on the device and in
the oracle (all host cores), state compared bit for bit after every sweep; the held-out estimate over the whole test set
at the end.   python scripts/soak_parity.py [--sweeps 10] [--topics 100]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweeps", type=int, default=10)
    ap.add_argument("--docs", type=int, default=100000)
    ap.add_argument("--topics", type=int, default=100)
    ap.add_argument("--scheme", default="ggs", choices=["ggs", "pcgs"])
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    from ldagroupedgibbssampler_amd import native
    from ldagroupedgibbssampler_amd.corpus import synthetic_lda_corpus
    from oracle import oracle as O

    c = synthetic_lda_corpus(args.docs, 50000, 200, true_topics=100, seed=2019)
    cut = args.docs - args.docs // 10
    train, _, _ = c.shard(0, cut)
    test, _, _ = c.shard(cut, args.docs)
    K = args.topics
    g = native.GGSHandle(K, c.num_types, 0.1, 0.01, 2019, flags=native.FLAG_PCGS if args.scheme == "pcgs" else 0)
    o = O.OracleSampler(K, c.num_types, 0.1, 0.01, 2019, threads=min(32, os.cpu_count() or 1))   # the GPU box gives a process a fraction of its 256 hardware threads: more are slower
    o.set_scheme(args.scheme)
    for s in (g, o):
        s.set_corpus(train.doc_ptr, train.tokens)
        s.init_z_java_lcg(2019)
        s.init_phi()
    for it in range(1, args.sweeps + 1):
        t0 = time.perf_counter()
        g.sweep(1)
        o.sweep(1)
        same = (np.array_equal(g.get_z(), o.get_z()) and np.array_equal(g.get_topic_totals(), o.get_topic_totals())
                and np.array_equal(g.get_phi().view(np.int64), o.get_phi().view(np.int64)))
        print("sweep %d: %s (%.1f s)" % (it, "bit-identical" if same else "DIFFERENT", time.perf_counter() - t0), flush=True)
        if not same:
            sys.exit(1)
    g.set_test_corpus(test.doc_ptr, test.tokens)
    gt, gl = g.heldout_log_likelihood(100)
    ot, ol = o.heldout_log_likelihood(test.doc_ptr, test.tokens, 100)
    ok = gt == ot and np.array_equal(gl.view(np.int64), ol.view(np.int64))
    print("held-out over %d documents: %s (%.6f)" % (test.num_docs, "bit-identical" if ok else "DIFFERENT", gt), flush=True)
    ll_g, ll_o = sum(g.model_log_likelihood()), sum(o.model_log_likelihood())
    print("model log likelihood: device %.6f oracle %.6f rel %.2e" % (ll_g, ll_o, abs(ll_g - ll_o) / abs(ll_o)), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
