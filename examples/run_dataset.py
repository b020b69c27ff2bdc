#!/usr/bin/env python3
"""End to end on one MI355X: a MALLET-style text file (one document per line: name, label, text) through the restated
front-end (LDAUtils.loadDataset), the sampler mirror (tui/ParallelLDA.java:170-296: setRandomSeed, addInstances, sample)
and the driver's output files (LDAUtils.java:1120-1254 formats).  Not a CLI: the reference's configuration parsing and
command line stay in Java (out of scope); this is the order of calls a caller makes.

    python examples/run_dataset.py tests/golden/datasets/cats.txt --topics 20 --iterations 200 --out /tmp/cats_run
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dataset")
    ap.add_argument("--stoplist", default=None, help="one stop word per line (LDAConfiguration: stoplist.txt)")
    ap.add_argument("--rare-threshold", type=int, default=0)
    ap.add_argument("--scheme", default="ggs", choices=["ggs", "pcgs", "collapsed"])
    ap.add_argument("--topics", type=int, default=20)
    ap.add_argument("--alpha", type=float, default=0.1)
    ap.add_argument("--beta", type=float, default=0.01)
    ap.add_argument("--iterations", type=int, default=200)
    ap.add_argument("--seed", type=int, default=4711)
    ap.add_argument("--top-words", type=int, default=8)
    ap.add_argument("--out", default=None, help="directory for the driver's files (phi / theta / counts / likelihood)")
    args = ap.parse_args()

    from ldagroupedgibbssampler_amd import formats as F
    from ldagroupedgibbssampler_amd.frontend import load_dataset
    from ldagroupedgibbssampler_amd.sampler import SimpleLDAConfiguration, create_model

    ds = load_dataset(args.dataset, stoplist=args.stoplist, rare_threshold=args.rare_threshold)
    c = ds.corpus
    print("%s: %d documents, %d types, %d tokens" % (os.path.basename(args.dataset), c.num_docs, c.num_types, c.num_tokens))
    if args.out:
        os.makedirs(args.out, exist_ok=True)
    cfg = SimpleLDAConfiguration(scheme=args.scheme, topics=args.topics, alpha=args.alpha, beta=args.beta, iterations=args.iterations,
                                 seed=args.seed, exec_time=None, compute_likelihood=bool(args.out), log_dir=args.out)
    model = create_model(cfg)
    model.setRandomSeed(cfg.get_seed())
    model.addInstances(c)
    model.sample(args.iterations)
    print("%d iterations: z %.1f ms, Phi %.1f ms in all; model log likelihood %.2f" %
          (model.getCurrentIteration(), model.zSamplingTimeCum, model.phiSamplingTimeCum, model.modelLogLikelihood()))

    n_wk = np.asarray(model.getTypeTopicMatrix())           # [V][K] counts, as LDAUtils.getTopWords ranks them
    for k in range(args.topics):
        top = np.argsort(-n_wk[:, k], kind="stable")[:args.top_words]
        print("topic %2d: %s" % (k, " ".join(c.vocab[i] for i in top if n_wk[i, k] > 0)))

    if args.out:
        K, V, D, it = args.topics, c.num_types, c.num_docs, model.getCurrentIteration()
        if args.scheme != "collapsed":
            F.write_binary_double_matrix(np.asarray(model.getPhi()), F.binary_matrix_name(os.path.join(args.out, "phi"), K, V, it))
        F.write_ascii_double_matrix(np.asarray(model.getThetaEstimate()), F.ascii_matrix_name(args.out, "Theta_DxK", D, K, it))
        F.write_ascii_int_matrix(n_wk.T, os.path.join(args.out, "type_topic_counts.csv"))
        print("wrote", ", ".join(sorted(os.listdir(args.out))))


if __name__ == "__main__":
    main()
