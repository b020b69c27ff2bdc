#!/usr/bin/env python3
"""Time the held-out estimator (ggs_heldout_log_likelihood; MarginalProbEstimatorPlain.evaluateLeftToRight,
100 particles as at UPLDA:615) on the benchmark-shaped corpus with 10 % of the documents held out, beside the
oracle's restatement on all host cores (on a sample of the test documents).  Prints one JSON line.

    python scripts/bench_heldout.py [--docs 100000] [--topics 100] [--sweeps 20] [--particles 100]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=100000)
    ap.add_argument("--types", type=int, default=50000)
    ap.add_argument("--topics", type=int, default=100)
    ap.add_argument("--sweeps", type=int, default=20)
    ap.add_argument("--particles", type=int, default=100)
    ap.add_argument("--cpu-docs", type=int, default=1000)
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401  (one HIP runtime per process)
    from ldagroupedgibbssampler_amd import native
    from ldagroupedgibbssampler_amd.corpus import synthetic_lda_corpus
    from oracle import oracle as O

    c = synthetic_lda_corpus(args.docs, args.types, 200, true_topics=100, seed=2019)
    cut = args.docs - args.docs // 10
    train, _, _ = c.shard(0, cut)
    test, _, _ = c.shard(cut, args.docs)
    K = args.topics
    h = native.GGSHandle(K, c.num_types, 0.1, 0.01, 2019)
    h.set_corpus(train.doc_ptr, train.tokens)
    h.init_z_java_lcg(2019)
    h.init_phi()
    out = {"workload": "held-out left-to-right, D_test=%d N_test=%d particles=%d K=%d V=%d" % (test.num_docs, test.num_tokens, args.particles, K, c.num_types)}
    h.set_test_corpus(test.doc_ptr, test.tokens)
    for label, sweeps in (("random_init", 0), ("after_%d_sweeps" % args.sweeps, args.sweeps)):
        if sweeps:
            h.sweep(sweeps)
        nnz = float((h.get_type_topic_counts() != 0).sum()) / c.num_types
        h.heldout_log_likelihood(args.particles)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            tot, ll = h.heldout_log_likelihood(args.particles)
            ts.append(time.perf_counter() - t0)
        out[label] = {"gpu_ms": round(min(ts) * 1e3, 2), "total_ll": tot, "ll_per_token": tot / test.num_tokens,
                      "mean_nonzero_topics_per_word": round(nnz, 1),
                      "G_particle_tokens_per_s": round(test.num_tokens * args.particles / min(ts) / 1e9, 2)}
    cores = os.cpu_count() or 1
    o = O.OracleSampler(K, c.num_types, 0.1, 0.01, 2019, threads=cores)
    o.set_corpus(train.doc_ptr, train.tokens)
    o.set_z(h.get_z(), redraw_phi=False)
    o.set_iteration(args.sweeps)
    sub, _, _ = test.shard(0, min(args.cpu_docs, test.num_docs))
    t0 = time.perf_counter()
    ot, ol = o.heldout_log_likelihood(sub.doc_ptr, sub.tokens, args.particles)
    dt = time.perf_counter() - t0
    out["cpu_oracle"] = {"cores": cores, "docs": sub.num_docs, "seconds": round(dt, 3),
                         "G_particle_tokens_per_s": round(sub.num_tokens * args.particles / dt / 1e9, 4),
                         "bit_identical_to_gpu": bool(np.array_equal(ol.view(np.int64), ll[:sub.num_docs].view(np.int64)))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
