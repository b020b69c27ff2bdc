#!/usr/bin/env python3
"""Benchmark of the Grouped Gibbs sweep on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full sweep of BASELINE.json's configs[1]: synthetic LDA corpus D=100k,
V=50k, mean 200 tokens/doc (~20M tokens), K=100, alpha=0.1, beta=0.01 -- theta draw +
z draw + count merge (+ RCCL all-reduce of the deltas when N>1) + Phi re-draw, i.e. what
UncollapsedParallelLDA.sample does between preZ() and postPhi() (UPLDA:659-687).
Inputs are resident in HBM before the timed region; rank 0 prints ONE JSON line.

N>1 shards the documents across the ranks (one count all-reduce per sweep).  Default
`--scaling weak`: every rank holds a configs[1]-sized shard (rank r generates its D documents
with seed+r; rank 0's shard IS the N=1 corpus), i.e. the corpus grows with the node, V and K
stay -- the per-GPU work is fixed, `value` = tokens of all ranks / time.  `--scaling strong`
keeps the N=1 corpus and splits it (2.5 M tokens per GPU at N=8: the replicated Phi draw and
the all-reduce then dominate).  At N>1 the default run measures that split too, after the
headline measurement, and reports it in the same line as "strong_scaling" (--no-strong-leg skips it).

N>1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_token(K):
    """SURVEY.md 8(d) with fp64 Phi (the reference's double[][] phi): one K-row of 8-byte
    values + token id + z read + z write + two 4-byte read-modify-write count updates."""
    return 8 * K + 28


def measured_traffic(kernel_prefixes):
    """HBM bytes per z step (summed over its kernels) from the committed PMC passes of this same
    command (profiles/r*_pmc_counters.txt; FETCH_SIZE and WRITE_SIZE are collected in separate
    rocprofv3 runs, in KiB).  gfx950: FETCH_SIZE counts 128-byte requests at 64 bytes for wide
    coalesced reads, so it is doubled (MI355X_MICROARCH.md, HBM).  None if no profile is present."""
    import ast
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_counters.txt")))
    if not files:
        return None
    total = 0.0
    for kernel_prefix in kernel_prefixes:
        fetch = write = None
        for line in open(files[-1]):
            if kernel_prefix not in line.split("{")[0]:
                continue
            try:
                d = ast.literal_eval(line[line.index("{"):line.rindex("}") + 1])
            except (ValueError, SyntaxError):
                continue
            fetch = d.get("FETCH_SIZE", fetch)
            write = d.get("WRITE_SIZE", write)
        if fetch is None or write is None:
            return None
        total += 2.0 * fetch + write
    return int(total * 1024)


def cpu_baseline(corpus, K, alpha, beta, seed, z0, sample_docs):
    """The oracle (a C restatement of the Java GGS sweep, Java layouts kept: phi[K][V],
    atomic [K][V] deltas, dynamic chunks of 100 documents) on all host cores, on the first
    `sample_docs` documents with the full vocabulary.  kind = "port": not a JVM run."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    sub, _, _ = corpus.shard(0, min(sample_docs, corpus.num_docs))
    o = O.OracleSampler(K, corpus.num_types, alpha, beta, seed, threads=cores)
    o.set_corpus(sub.doc_ptr, sub.tokens)
    o.set_z(z0[:sub.num_tokens], redraw_phi=True)
    o.sweep(1)                     # warm-up (page in, first-touch)
    t0 = time.perf_counter()
    n_sw = 2
    for _ in range(n_sw):
        o.set_iteration(o.iteration + 1)
        t_a = time.perf_counter()
        o.z_step()
        t_b = time.perf_counter()
        o.update_counts()
        o.sample_phi()
        t_c = time.perf_counter()
    dt = time.perf_counter() - t0
    o.close()
    return {
        "value": round(sub.num_tokens * n_sw / dt / 1e6, 3),
        "unit": "M tokens/s",
        "cores": cores,
        "kind": "port",
        "sample": "first %d docs (%d tokens) of the same corpus, full V=%d and K=%d, %d full sweeps "
                  "(z step %.2fs + merge/Phi %.2fs in the last one; the K*V Phi draw does not shrink with the sample)"
                  % (sub.num_docs, sub.num_tokens, corpus.num_types, K, n_sw, t_b - t_a, t_c - t_b),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=100000)
    ap.add_argument("--types", type=int, default=50000)
    ap.add_argument("--mean-len", type=int, default=200)
    ap.add_argument("--topics", type=int, default=100)
    ap.add_argument("--alpha", type=float, default=0.1)
    ap.add_argument("--beta", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=2019)
    ap.add_argument("--cpu-sample-docs", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scheme", default="ggs", choices=["ggs", "pcgs"], help="ggs = the headline path; pcgs = the partially collapsed z loop (SURVEY 8f-1), for comparison")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = D documents PER RANK (default), strong = the N=1 corpus split across the ranks")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process group backend; gloo + --single-device rehearses N ranks on ONE GPU (RCCL refuses two ranks per device)")
    ap.add_argument("--single-device", action="store_true", help="every rank uses cuda:0 (rehearsal only; the number is not a multi-GPU result)")
    ap.add_argument("--no-strong-leg", action="store_true", help="N>1, weak scaling: skip the extra measurement of the N=1 corpus split across the ranks")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the doc-sharded path (process group + RCCL all-reduce) even with one rank; for testing")
    args = ap.parse_args()

    import numpy as np
    import torch  # device plumbing + torch.distributed (RCCL); imported before libggs_hip so both share one HIP runtime

    from ldagroupedgibbssampler_amd import native
    from ldagroupedgibbssampler_amd.corpus import synthetic_lda_corpus
    from ldagroupedgibbssampler_amd.sharded import (ShardedGGS, TorchHipExchange, gather_shard_sizes, java_lcg_initial_z,
                                                    java_lcg_initial_z_slice)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("no GPU visible: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:          # single-rank self test without a launcher
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29531"
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    K = args.topics
    weak = sharded and args.scaling == "weak"
    corpus = synthetic_lda_corpus(args.docs, args.types, args.mean_len, true_topics=100, seed=args.seed + (rank if weak else 0))
    if weak:
        sizes = gather_shard_sizes(corpus, rank, world, device="cuda")
        total_docs, total_tokens = sum(d for d, _ in sizes), sum(t for _, t in sizes)
        z0 = java_lcg_initial_z_slice(sum(t for _, t in sizes[:rank]), corpus.num_tokens, K, args.seed)   # one sequential stream over the global corpus
    else:
        total_docs, total_tokens = corpus.num_docs, corpus.num_tokens
        z0 = java_lcg_initial_z(corpus.num_tokens, K, args.seed)

    h = native.GGSHandle(K, corpus.num_types, args.alpha, args.beta, args.seed, device_id=local_rank,
                         flags=native.FLAG_PCGS if args.scheme == "pcgs" else 0)
    if sharded:
        if weak:
            sh = ShardedGGS.from_local_shard(h, TorchHipExchange, corpus, sizes, rank, world)
            sh.set_z_local(z0)
        else:
            sh = ShardedGGS(h, TorchHipExchange, corpus, rank, world)
            sh.set_z_global(z0)
        def run(n):
            # one count exchange per sweep, enqueued back to back; the host waits once per batch of 5
            for i in range(0, n, 5):
                sh.sweep(min(5, n - i))
        n_local = sh.local.num_tokens
    else:
        h.set_corpus(corpus.doc_ptr, corpus.tokens)
        h.set_z(z0, redraw_phi=True)
        def run(n):
            # batches of 5 sweeps: only the last sweep of a ggs_sweep call is waited for and timed by the library
            # (no per-sweep host round trip); 20 steps still give 4 samples of every phase
            for i in range(0, n, 5):
                h.sweep(min(5, n - i))
        n_local = corpus.num_tokens

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    h.reset_timings()
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tm = h.get_timings()
    h.check_invariants()

    # N>1, weak scaling: the same run also measures the N=1 corpus SPLIT across the ranks (the strong-scaling reading of
    # BASELINE's "reported at 1, 2, 4 and 8 GPUs"), reported beside the headline value as "strong_scaling".
    strong = None
    if weak and world > 1 and not args.no_strong_leg:
        h.close()                     # its streams first: hardware queues are few, and a second handle beside it runs serialised
        corpus1 = corpus if rank == 0 else synthetic_lda_corpus(args.docs, args.types, args.mean_len, true_topics=100, seed=args.seed)
        h1 = native.GGSHandle(K, corpus1.num_types, args.alpha, args.beta, args.seed, device_id=local_rank,
                              flags=native.FLAG_PCGS if args.scheme == "pcgs" else 0)
        sh1 = ShardedGGS(h1, TorchHipExchange, corpus1, rank, world)
        sh1.set_z_global(java_lcg_initial_z(corpus1.num_tokens, K, args.seed))
        for i in range(0, args.warmup, 5):
            sh1.sweep(min(5, args.warmup - i))
        fence()
        t1 = time.perf_counter()
        for i in range(0, args.steps, 5):
            sh1.sweep(min(5, args.steps - i))
        fence()
        dt1 = time.perf_counter() - t1
        t = torch.tensor([dt1], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt1 = float(t.item())
        h1.check_invariants()
        tm1 = h1.get_timings()
        strong = {"value": round(corpus1.num_tokens * args.steps / dt1 / 1e6, 3), "unit": "M tokens/s", "ms_per_step": round(dt1 / args.steps * 1e3, 4),
                  "phase_ms_per_sweep": {k: round(tm1[k] / max(tm1["sweeps"], 1), 4) for k in ("theta_ms", "z_ms", "merge_ms", "phi_ms")},
                  "workload": "the N=1 corpus (D=%d, N=%d tokens) split across the %d ranks" % (corpus1.num_docs, corpus1.num_tokens, world)}
        h1.close()

    if rank == 0:
        btok = algorithmic_bytes_per_token(K)
        z_ms = tm["z_ms"] / max(tm["sweeps"], 1)            # HIP events on the handle's stream, over the timed region
        achieved = n_local * btok / (z_ms * 1e-3) / 1e9 if z_ms > 0 else 0.0
        # the z step: for K <= 192 the cold-chunk kernel and, beside it on a second stream, the hot-chunk kernel
        kmax = 8 * ((K + 7) // 8)
        zkernels = ((["pcgs_sliced_kernel<%d>" % kmax] if K <= 192 else ["pcgs_z_kernel"]) if args.scheme == "pcgs" else
                    ["z_sliced_kernel<%d>" % kmax, "z_hot_kernel<%d>" % kmax] if K <= 192 else ["z_stream_kernel"])
        line = {
            "metric": "M tokens sampled/sec (whole node) per Gibbs sweep at K=%d" % K + ("" if args.scheme == "ggs" else " (scheme=%s)" % args.scheme),
            "value": round(total_tokens * args.steps / dt / 1e6, 3),
            "unit": "M tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "LDAGroupedGibbsSampler sweep, synthetic LDA corpus D=%d V=%d N=%d tokens K=%d alpha=%g beta=%g seed=%d"
                            % (total_docs, corpus.num_types, total_tokens, K, args.alpha, args.beta, args.seed)
                            + (" (%d documents per rank)" % args.docs if weak else ""),
                "parallelism": ("doc-sharded x%d, int32 count all-reduce (%s) per sweep%s"
                                % (world, "RCCL" if args.backend == "nccl" else "gloo", ", ALL RANKS ON ONE GPU (rehearsal)" if args.single_device else ""))
                               if sharded else "1 GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": " + ".join(zkernels),
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": measured_traffic(["ggs::" + k.split("<")[0] for k in zkernels]),
                "bytes_per_token": btok,
                "tokens_per_launch": n_local,
                "avg_launch_ms": round(z_ms, 4),
            },
            "phase_ms_per_sweep": {k: round(tm[k] / max(tm["sweeps"], 1), 4) for k in ("theta_ms", "z_ms", "merge_ms", "phi_ms")},
        }
        if strong is not None:
            line["strong_scaling"] = strong
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(corpus, K, args.alpha, args.beta, args.seed, z0, args.cpu_sample_docs)
        print(json.dumps(line), flush=True)
    if strong is None:
        h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
