#!/usr/bin/env python3
"""Benchmark of the Grouped Gibbs sweep on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full sweep of BASELINE.json's configs[1]: synthetic LDA corpus D=100k,
V=50k, mean 200 tokens/doc (~20M tokens), K=100, alpha=0.1, beta=0.01 -- theta draw +
z draw + count merge (+ the exchange when N>1) + Phi re-draw, i.e. what
UncollapsedParallelLDA.sample does between preZ() and postPhi() (UPLDA:659-687).
Inputs are resident in HBM before the timed region; rank 0 prints ONE JSON line.

N>1 shards the documents across the ranks, one process per GPU, joined by the library's NATIVE exchange
(ggs_attach_rccl: RCCL reduce-scatter of the int32 counts by topic slice, Phi drawn for the rank's own topics,
all-gather of the fp64 Phi slices; torch.distributed only carries the unique id, the barrier and the max over
ranks).  Default `--scaling strong`, BASELINE.json's reading ("... at 1, 2, 4 and 8 GPUs ... >=6x strong scaling"):
the N=1 corpus is split across the ranks, `value` = its tokens / time.  The same run then measures the weak-scaling
regime (every rank brings a configs[1]-sized shard, rank r generated with seed+r) and reports it as "weak_scaling"
(--no-weak-leg skips it); `--scaling weak` makes that the headline instead.

At N=1 the line also carries `extra_configs` (BASELINE configs 3 and the config-4 stand-in: ms per sweep, phase times,
row-byte accounting) and `cpu_baseline` (the CPU restatement in three variants, on a bounded sample).

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
# the pool's host driver only supports dmabuf IPC: without this RCCL's buffer exchange between the ranks' processes fails
# (hipIpcGetMemHandle: invalid argument).  Exported on the boxes already; set here, before anything initialises HIP, for a
# launcher that does not pass the environment on.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MALL_GATHER_PEAK_GBS = 8600.0  # measured Infinity-Cache random-row gather ceiling, same guide ("Indexed rows: gather into LDS")
L2_GATHER_PEAK_GBS = 17800.0   # measured L2-resident row gather, same table (16.8-18.8 TB/s)


def algorithmic_bytes_per_token(K):
    """SURVEY.md 8(d) with fp64 Phi (the reference's double[][] phi): one K-row of 8-byte
    values + token id + z read + z write + two 4-byte read-modify-write count updates."""
    return 8 * K + 28


def csrc_sha16():
    """Hash of the kernel sources this run was built from: stamps profiles, so a counter file is only ever quoted for
    the build and workload it was collected on."""
    import hashlib
    d = os.path.join(ROOT, "ldagroupedgibbssampler_amd", "csrc")
    hs = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            hs.update(f.encode())
            hs.update(open(os.path.join(d, f), "rb").read())
    return hs.hexdigest()[:16]


def measured_traffic(kernel_prefixes, workload, sha):
    """HBM bytes per z step (summed over its kernels) from a committed PMC profile of THIS workload and THIS build:
    profiles/r*_pmc_counters.txt whose first line (written by scripts/collect_profiles.sh) carries the same workload
    string and csrc hash -- anything else is somebody else's counter and the answer is None.  FETCH_SIZE and WRITE_SIZE
    are collected in separate rocprofv3 runs, in KiB; gfx950 counts 128-byte read requests at 64 bytes for wide coalesced
    reads, so FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM).  Returns (bytes, file name) or (None, reason)."""
    import ast
    import glob
    want = "# workload: %s | csrc_sha16: %s" % (workload, sha)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc_counters.txt")), reverse=True):
        lines = open(path).read().splitlines()
        if not lines or lines[0].strip() != want:
            continue
        total = 0.0
        for kernel_prefix in kernel_prefixes:
            fetch = write = None
            for line in lines[1:]:
                if kernel_prefix not in line.split("{")[0]:
                    continue
                try:
                    d = ast.literal_eval(line[line.index("{"):line.rindex("}") + 1])
                except (ValueError, SyntaxError):
                    continue
                fetch = d.get("FETCH_SIZE", fetch)
                write = d.get("WRITE_SIZE", write)
            if fetch is None or write is None:
                return None, "profile %s lacks FETCH/WRITE_SIZE of %s" % (os.path.basename(path), kernel_prefix)
            total += 2.0 * fetch + write
        return int(total * 1024), os.path.basename(path)
    return None, "no profiles/r*pmc_counters.txt stamped with this workload and csrc hash"


SLICED_MAX_K = 160      # ggs_api.hip: the score-register z kernels up to here, the one-pass streaming kernel above


def z_kernels(K, scheme, warm_tiers=0):
    kmax = 8 * ((K + 7) // 8)
    nb = 1
    while nb * 128 < K + (K & 1):
        nb *= 2
    if scheme == "collapsed":                            # ggs_api.hip: kCollapsedWaveFromTopics / kPcgsWaveFromTopics, the measured switch points
        return ["pcgs_sliced_kernel<%d, true>" % kmax] if K <= 96 else ["pcgs_wave_kernel<%d, true>" % nb]
    if scheme == "pcgs":
        return ["pcgs_sliced_kernel<%d>" % kmax] if K <= 176 else ["pcgs_wave_kernel<%d, false>" % nb]
    if K > SLICED_MAX_K:
        return ["z_stream1_kernel"]
    return ["z_sliced_kernel<%d>" % kmax, "z_hot_kernel<%d>" % kmax] + (["z_warm_kernel<%d>" % kmax] if warm_tiers else [])


def row_stats(corpus, K, num_hot):
    """Byte accounting of one z step over `corpus` (per launch):
      compulsory   what must cross HBM at least once: the chunk lists / token ids, z read + two z writes, theta rows
                   once per document, one pass over phiT
      cold_rows    phiT row bytes the kernels gather from L2 / Infinity Cache / HBM (tokens of the num_hot most
                   frequent words are served from the LDS table instead; 0 hot words for K > 192)"""
    import numpy as np
    N, D, V = corpus.num_tokens, corpus.num_docs, corpus.num_types
    freq = np.bincount(corpus.tokens, minlength=V)
    hot_tokens = int(np.sort(freq)[::-1][:num_hot].sum()) if num_hot > 0 else 0
    row = 8 * (K + (K & 1))
    ns = (K + 15) // 16
    gathered = row + (0 if K <= SLICED_MAX_K else 128 * (1 if ns <= 16 else 2 if ns <= 32 else 4))   # the one-pass kernel streams the row once plus one checkpoint group again
    return {
        "compulsory_bytes": int(N * (3 * 4 + 4 + 2 * 4) + D * K * 8 + V * row),
        "cold_row_bytes": int((N - hot_tokens) * gathered),
        "hot_token_frac": round(hot_tokens / max(N, 1), 4),
    }


def roofline_block(corpus, K, scheme, n_local, z_ms, workload, sha, num_hot, z_parts=1, warm_tiers=0):
    """What bounds the dominant kernel (the z step), in three consistent readings:
      achieved / frac    HBM bytes per launch over the launch time against the HBM peak -- from the PMC counters when a
                         profile of this very workload and build is committed (`traffic`), else from the compulsory
                         bytes (a lower bound of the traffic).  Never above 1 by construction of what is counted.
      algorithmic        SURVEY 8(d)'s per-token figure x tokens: what a kernel without on-chip reuse would move.  The
                         phiT rows are served from the LDS hot-word table and L2 / Infinity Cache, so this exceeds the
                         HBM peak at K=100 -- it is a statement about reuse, not a fraction of a roofline.
      row_gather         phiT row bytes the kernels actually pull into LDS over the launch time, beside the guide's two
                         measured ceilings for exactly this access pattern (indexed rows gathered into LDS): rows
                         served by the XCD's L2 and rows served by the Infinity Cache.  The Zipfian head of the
                         vocabulary hits L2, the tail goes to the Infinity Cache (K=100: phiT is 40 MB) or HBM
                         (K=1024: 410 MB), so the rate sits between the two: this is the memory-side limit that
                         applies, and the PMC profile's L2 hit rate says where between."""
    btok = algorithmic_bytes_per_token(K)
    zk = z_kernels(K, scheme, warm_tiers)
    rs = row_stats(corpus, K, num_hot)
    traffic, src = measured_traffic(["ggs::" + k.split("<")[0] for k in zk], workload, sha)
    if traffic is not None:
        traffic *= z_parts                                   # the profile's counters are per launch; the step is z_parts launches
    secs = z_ms * 1e-3

    def gbs(b):
        return round(b / secs / 1e9, 1) if secs > 0 else 0.0
    hbm_bytes = traffic if traffic is not None else rs["compulsory_bytes"]
    alg = n_local * btok
    return {
        # what the counters say bounds the kernel (profiles/): up to 160 topics one wave per SIMD issuing on half of its cycles
        # on top of a row gather served by L2 / Infinity Cache; above, the latency of that gather.  `frac` is still
        # the memory-side traffic against the HBM peak -- the one roofline the byte counters can be read against.
        "bound": "issue+cache-gather" if (K <= SLICED_MAX_K and scheme == "ggs") else "cache-gather-latency",
        "kernel": " + ".join(zk),
        "achieved": gbs(hbm_bytes),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(gbs(hbm_bytes) / HBM_PEAK_GBS, 4),
        "basis": ("PMC FETCH_SIZE/WRITE_SIZE of the committed profile of this workload and build: the L2's memory-side traffic, Infinity-Cache hits "
                  "included (an upper bound of the HBM bytes)") if traffic is not None else "compulsory bytes (lower bound of the HBM traffic; no matching PMC profile)",
        "traffic": traffic,
        "traffic_source": src,
        "compulsory_bytes": rs["compulsory_bytes"],
        "limiter": ("instruction issue + L2/Infinity-Cache row gather (far from the HBM roofline: see row_gather and profiles/)"
                    if K <= SLICED_MAX_K else "L2/Infinity-Cache/HBM row gather latency (one pass over the rows; see row_gather and profiles/)"),
        "row_gather": {"row_bytes": rs["cold_row_bytes"], "GBps": gbs(rs["cold_row_bytes"]), "l2_ceiling_GBps": L2_GATHER_PEAK_GBS,
                       "infinity_cache_ceiling_GBps": MALL_GATHER_PEAK_GBS, "frac_of_l2_ceiling": round(gbs(rs["cold_row_bytes"]) / L2_GATHER_PEAK_GBS, 4),
                       "hot_token_frac_in_lds": rs["hot_token_frac"]},
        "algorithmic": {"bytes_per_token": btok, "bytes_per_launch": alg, "GBps": gbs(alg), "over_hbm_peak": round(gbs(alg) / HBM_PEAK_GBS, 4),
                        "note": "SURVEY 8(d) figure; rows served from LDS/L2/MALL make it exceed the HBM peak -- not a roofline fraction"},
        "tokens_per_z_step": n_local,
        "launches_per_z_step": z_parts,
        "avg_z_step_ms": round(z_ms, 4),
    }


def cpu_baseline(corpus, K, alpha, beta, seed, z0, sample_docs):
    """BASELINE.md section 3, timed on this box's host cores on the first `sample_docs` documents with the full vocabulary:
      cpu_ref_mt    the oracle: C restatement of the Java GGS sweep with the Java layouts kept (phi[K][V] column gather,
                    atomic [K][V] deltas, dynamic chunks of 100 documents, the Phi draw one topic per thread as GGS:139-171
                    hands it out) -- the headline `value`
      cpu_ref_1t    the same on one thread, on a fiftieth of the sample, the z step (what scales with the tokens) and the Phi
                    draw (K*V gammas whatever the sample) timed apart; its `value` is the z-step rate
      cpu_tuned_mt  what a good CPU implementation does with the same arithmetic: transposed phiT rows, no per-document
                    allocation, no atomics (counts rebuilt per word), the Phi draw spread over (topic, 1024-type tile)
                    units so that every thread has work at K = 100
    The two multi-threaded variants are timed at 16, 32, 64, 128 and all hardware threads (those the box has) and report
    the best with its thread count -- measured on the 256-thread GPU box: 96 M tokens/s on 32 threads, 16 on 256 (the
    process's CPU share is a fraction of the host's threads, and oversubscribed spinning barriers are slow).
    kind = "port": not a JVM run (no JDK on the box)."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    out = {}

    def run(docs, threads, tuned, n_sw):
        sub, _, _ = corpus.shard(0, min(docs, corpus.num_docs))
        o = O.OracleSampler(K, corpus.num_types, alpha, beta, seed, threads=threads)
        o.set_corpus(sub.doc_ptr, sub.tokens)
        o.set_z(z0[:sub.num_tokens], redraw_phi=True)
        sweep = o.sweep_tuned if tuned else o.sweep
        sweep(1)                       # warm-up (page in, first-touch)
        t0 = time.perf_counter()
        sweep(n_sw)
        dt = time.perf_counter() - t0
        o.close()
        return {"value": round(sub.num_tokens * n_sw / dt / 1e6, 3), "unit": "M tokens/s", "threads": threads,
                "sample": "first %d docs (%d tokens), full V=%d, K=%d, %d full sweeps" % (sub.num_docs, sub.num_tokens, corpus.num_types, K, n_sw)}

    counts = sorted({t for t in (16, 32, 64, 128, cores) if t <= cores}) or [cores]   # a GPU box's CPU share may be far below its hardware threads
    for tag, tuned, n_sw in (("cpu_ref_mt", False, 1), ("cpu_tuned_mt", True, 2)):
        sweep = [run(sample_docs, t, tuned, n_sw) for t in counts]
        best = max(sweep, key=lambda r: r["value"])
        out[tag] = dict(best, thread_sweep={str(r["threads"]): r["value"] for r in sweep})

    def run_1t(docs):
        """One thread, the phases timed apart: the z step scales with the tokens of the sample, the Phi draw (K*V gammas) does
        not -- a single tokens/s figure over both would say more about the sample size than about the sampler."""
        sub, _, _ = corpus.shard(0, min(docs, corpus.num_docs))
        o = O.OracleSampler(K, corpus.num_types, alpha, beta, seed, threads=1)
        o.set_corpus(sub.doc_ptr, sub.tokens)
        o.set_z(z0[:sub.num_tokens], redraw_phi=True)
        o.set_iteration(1)
        t0 = time.perf_counter()
        o.z_step()                     # GGS:47-132 for every document of the sample: theta draw + token loop
        t1 = time.perf_counter()
        o.update_counts()              # UPLDA:1107-1221
        t2 = time.perf_counter()
        o.sample_phi()                 # GGS:139-198: K*V gammas whatever the sample
        t3 = time.perf_counter()
        o.close()
        return {"value": round(sub.num_tokens / (t1 - t0) / 1e6, 3), "unit": "M tokens/s (z step only: theta draw + token loop)", "threads": 1,
                "z_step_s": round(t1 - t0, 3), "update_counts_s": round(t2 - t1, 3), "phi_draw_s": round(t3 - t2, 3),
                "phi_draw_gammas_per_s": round(K * corpus.num_types / max(t3 - t2, 1e-9)),
                "sample": "first %d docs (%d tokens), full V=%d, K=%d, one sweep with its three phases timed apart" % (sub.num_docs, sub.num_tokens, corpus.num_types, K)}
    out["cpu_ref_1t"] = run_1t(max(sample_docs // 50, 200))
    return {
        "value": out["cpu_ref_mt"]["value"],
        "unit": "M tokens/s",
        "cores": out["cpu_ref_mt"]["threads"],
        "kind": "port",
        "sample": out["cpu_ref_mt"]["sample"] + " (the K*V Phi draw does not shrink with the sample); best of the thread counts in variants.cpu_ref_mt.thread_sweep",
        "variants": out,
    }


_T0 = time.time()


def stage(rank, msg):
    """A timestamped line per rank on stderr: a stall of an N-rank run is then attributable to a stage."""
    sys.stderr.write("[bench %s +%6.1fs rank %d] %s\n" % (time.strftime("%H:%M:%S"), time.time() - _T0, rank, msg))
    sys.stderr.flush()


def sha256_of(a):
    import hashlib

    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def state_digests(h, z_local=None):
    """sha256 of what a rank holds after its sweeps: its z shard, the (replicated) Phi, tokensPerTopic (a COLLECTIVE
    getter with an exchange attached), its theta rows.  Bit patterns, not values."""
    z = h.get_z() if z_local is None else z_local
    return {"z": sha256_of(z), "phi": sha256_of(h.get_phi()), "n_k": sha256_of(h.get_topic_totals()), "theta": sha256_of(h.get_theta()), "tokens": int(z.size)}


def make_handle(native, K, V, args, local_rank):
    return native.GGSHandle(K, V, args.alpha, args.beta, args.seed, device_id=local_rank,
                            flags={"ggs": 0, "pcgs": native.FLAG_PCGS, "collapsed": native.FLAG_COLLAPSED}[args.scheme])


def phases(tm):
    n = max(tm["sweeps"], 1)
    return {k: round(tm[k] / n, 4) for k in ("theta_ms", "z_ms", "merge_ms", "phi_ms", "exchange_ms", "exchange_rs_ms", "exchange_ag_ms")}


def run_single(native, corpus, z0, K, args, local_rank, steps, warmup, fence):
    """One handle over the whole corpus: (seconds for `steps` sweeps, per-sweep phase ms, launch info)."""
    h = make_handle(native, K, corpus.num_types, args, local_rank)
    if args.simulate_world > 1:
        h.attach_null_exchange(args.simulate_rank, args.simulate_world)
    h.set_corpus(corpus.doc_ptr, corpus.tokens)
    h.set_z(z0, redraw_phi=True)

    def run(n):
        # batches of 5 sweeps: only the last sweep of a ggs_sweep call is waited for (no per-sweep host round trip)
        for i in range(0, n, 5):
            h.sweep(min(5, n - i))
    run(warmup)
    h.reset_timings()
    fence()
    t0 = time.perf_counter()
    run(steps)
    fence()
    dt = time.perf_counter() - t0
    tm = h.get_timings()
    if args.simulate_world <= 1:
        h.check_invariants()
    info = h.launch_info()
    h.close()
    return dt, phases(tm), info


def workload_string(D, V, N, K, args):
    return ("LDAGroupedGibbsSampler sweep, synthetic LDA corpus D=%d V=%d N=%d tokens K=%d alpha=%g beta=%g seed=%d"
            % (D, V, N, K, args.alpha, args.beta, args.seed))


def extra_configs(native, corpus2, args, local_rank, fence, sha):
    """BASELINE configs 3 (the same corpus at K=1024) and 4 (stand-in for 20-Newsgroups, whose file is not in the image:
    D=18 846, V=60 000, mean 150 tokens, K=200) on one GPU: a few sweeps each, same timing discipline as the headline."""
    import copy

    from ldagroupedgibbssampler_amd.corpus import synthetic_lda_corpus
    from ldagroupedgibbssampler_amd.sharded import java_lcg_initial_z
    out = {}
    for tag, K, corpus, steps, warmup in (("config3_K1024", 1024, corpus2, 5, 1), ("config4_standin_20ng_K200", 200, None, 10, 2)):
        a = copy.copy(args)
        a.topics, a.simulate_world = K, 1
        if corpus is None:
            corpus = synthetic_lda_corpus(18846, 60000, 150, true_topics=100, seed=args.seed)
        z0 = java_lcg_initial_z(corpus.num_tokens, K, args.seed)
        dt, ph, info = run_single(native, corpus, z0, K, a, local_rank, steps, warmup, fence)
        workload = workload_string(corpus.num_docs, corpus.num_types, corpus.num_tokens, K, args)
        out[tag] = {"workload": workload, "value": round(corpus.num_tokens * steps / dt / 1e6, 3), "unit": "M tokens/s", "steps": steps,
                    "ms_per_step": round(dt / steps * 1e3, 4), "phase_ms_per_sweep": ph,
                    "roofline": roofline_block(corpus, K, "ggs", corpus.num_tokens, ph["z_ms"], workload, sha, info.get("num_hot", 0), info.get("z_parts", 1), info.get("warm_tiers", 0))}
    return out


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
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the extra_configs legs (configs 3 and 4 stand-in) of the default N=1 run")
    ap.add_argument("--scheme", default="ggs", choices=["ggs", "pcgs", "collapsed"],
                    help="ggs = the headline path; pcgs = the partially collapsed z loop (SURVEY 8f-1); collapsed = the count-form conditional, parallel schedule (SURVEY 8f-4)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N>1: strong = the N=1 corpus split across the ranks (default, BASELINE.json), weak = D documents PER RANK")
    ap.add_argument("--exchange", default="native", choices=["native", "torch"],
                    help="N>1: native = the library's own RCCL exchange, topic-sliced Phi draw (default); torch = dense count all-reduce "
                         "through torch.distributed, Phi re-drawn on every rank (the round-1 form, kept as a cross-check)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process group backend; gloo + --single-device rehearses N ranks on ONE GPU (RCCL refuses two ranks per device): "
                         "the native exchange then runs over its callback provider with host staging")
    ap.add_argument("--single-device", action="store_true", help="every rank uses cuda:0 (rehearsal only; the number is not a multi-GPU result)")
    ap.add_argument("--no-weak-leg", action="store_true", help="N>1: skip the second measurement (the other scaling regime)")
    ap.add_argument("--count-exchange", default="auto", choices=["auto", "dense", "sparse"],
                    help="N>1, native exchange: how the counts travel (auto: (cell, count) pairs where the dense buffer is large and mostly zero)")
    ap.add_argument("--large-factor", type=int, default=3,
                    help="N>1: a third leg, the strong split of a corpus this many times the headline's (0/1: none), with its own one-GPU time from the same run")
    ap.add_argument("--no-large-leg", action="store_true", help="N>1: skip that leg")
    ap.add_argument("--no-verify", action="store_true",
                    help="N>1 / --force-sharded: skip the comparison of the N-rank end state with one handle over the whole corpus (parity_vs_one_gpu)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the doc-sharded path (process group + RCCL exchange) even with one rank; for testing")
    ap.add_argument("--simulate-rank", type=int, default=0)
    ap.add_argument("--simulate-world", type=int, default=1,
                    help="N=1 only, a TIMING AID: run as rank --simulate-rank of this many with the peers missing (ggs_attach_null_exchange) on that "
                         "rank's share of the corpus; the sampler's results are wrong by construction, the per-rank phase times are what is read")
    args = ap.parse_args()

    import torch  # device plumbing + torch.distributed (barrier, max over ranks, the unique id); imported before libggs_hip so both share one HIP runtime and one RCCL

    from ldagroupedgibbssampler_amd import native
    from ldagroupedgibbssampler_amd.corpus import even_split, synthetic_lda_corpus
    from ldagroupedgibbssampler_amd.sharded import (ShardedGGS, TorchHipExchange, gather_shard_sizes, gloo_callback_exchange, java_lcg_initial_z,
                                                    java_lcg_initial_z_slice, rccl_exchange)

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

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if dist is None:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def exchange_factory():
        if args.exchange == "torch":
            return TorchHipExchange
        if os.environ.get("GGS_BENCH_FAIL_NATIVE"):      # rehearsal of run_sharded_or_fall_back
            raise RuntimeError("native exchange refused (GGS_BENCH_FAIL_NATIVE)")
        return rccl_exchange(rank, world) if args.backend == "nccl" else gloo_callback_exchange(rank, world)

    K = args.topics

    def run_sharded(weak, docs=None):
        """One measurement over `world` ranks: dict(dt, phases, docs, tokens, n_local, V, info, local corpus)"""
        corpus = synthetic_lda_corpus(docs or args.docs, args.types, args.mean_len, true_topics=100, seed=args.seed + (rank if weak else 0))
        h = make_handle(native, K, corpus.num_types, args, local_rank)
        try:
            return run_sharded_on(h, corpus, weak)
        except Exception:
            try:
                h.close()
            except Exception:       # noqa: BLE001 -- the first error is the one to report
                pass
            raise

    def run_sharded_on(h, corpus, weak):
        stage(rank, "%s leg: corpus built (D=%d, N=%d), attaching the exchange" % ("weak" if weak else "strong", corpus.num_docs, corpus.num_tokens))
        if weak:
            sizes = gather_shard_sizes(corpus, rank, world, device="cuda" if args.backend == "nccl" else None)
            total_docs, total_tokens = sum(d for d, _ in sizes), sum(t for _, t in sizes)
            sh = ShardedGGS.from_local_shard(h, exchange_factory(), corpus, sizes, rank, world)
            sh.set_z_local(java_lcg_initial_z_slice(sh.tok_base, corpus.num_tokens, K, args.seed))   # one sequential stream over the global corpus
        else:
            total_docs, total_tokens = corpus.num_docs, corpus.num_tokens
            sh = ShardedGGS(h, exchange_factory(), corpus, rank, world)
            sh.set_z_global(java_lcg_initial_z(corpus.num_tokens, K, args.seed))
        if args.exchange == "native" and args.count_exchange != "auto":
            h.set_count_exchange(args.count_exchange)
        xinfo = h.exchange_info() if args.exchange == "native" else {"provider": "torch.distributed all-reduce", "comm_nranks": world, "comm_rank": rank, "nranks": world}
        stage(rank, "exchange attached, shard uploaded (%d documents, %d tokens), initial z / counts / Phi in place; exchange: %s"
              % (sh.local.num_docs, sh.local.num_tokens, json.dumps(xinfo)))

        def run(n):
            # the sweeps are enqueued back to back, exchange included; the host waits once per batch of 5
            for i in range(0, n, 5):
                sh.sweep(min(5, n - i))
        run(args.warmup)
        stage(rank, "warm-up done (%d sweeps)" % args.warmup)
        h.reset_timings()
        fence()
        t0 = time.perf_counter()
        run(args.steps)
        fence()
        dt = max_over_ranks(time.perf_counter() - t0)
        stage(rank, "timed region done: %d sweeps in %.2f ms (max over ranks)" % (args.steps, dt * 1e3))
        tm = h.get_timings()
        h.check_invariants()            # with the native exchange: a collective call (gathers the corpus-wide counts)
        if args.exchange == "native":
            xinfo = dict(xinfo, count_exchange=h.count_exchange())     # dense reduce-scatter or (cell, count) pairs, and how many pairs the last sweep sent
        res = dict(dt=dt, phases=phases(tm), docs=total_docs, tokens=total_tokens, n_local=sh.local.num_tokens, V=corpus.num_types,
                   info=h.launch_info(), local=sh.local, exchange=xinfo)
        if not weak and not args.no_verify:
            # what this rank ends with, for the comparison with ONE handle over the whole corpus (verify_against_one_gpu)
            res["digests"] = state_digests(h)
            res["tok_range"] = (int(sh.tok_base), int(sh.tok_base + sh.local.num_tokens))
            res["corpus"] = corpus
        h.close()                       # before a second leg builds its handle: hardware queues are few
        return res

    def verify_against_one_gpu(r):
        """The N-rank state against ONE handle over the whole corpus after the same number of sweeps from the same z0:
        every rank's z shard, theta rows and Phi, and tokensPerTopic, as sha256 of the bit patterns.  Sharding is exact
        (identical Philox element ids on every rank), so anything but equality is a bug in the exchange.
        Rank 0 runs the one-GPU handle (the same --warmup + --steps sweeps; the others wait at the barrier)."""
        mine = dict(r["digests"], tok_range=r["tok_range"], rank=rank, z_form=r["info"].get("z_form"), comm_nranks=r["exchange"].get("comm_nranks"))
        every = [None] * world
        if world > 1:
            dist.all_gather_object(every, mine)
        else:
            every = [mine]
        out = None
        if rank == 0:
            stage(rank, "verification: one handle over the whole corpus, %d sweeps" % (args.warmup + args.steps))
            corpus = r["corpus"]
            h1 = make_handle(native, K, corpus.num_types, args, local_rank)
            try:
                h1.set_corpus(corpus.doc_ptr, corpus.tokens)
                h1.set_z(java_lcg_initial_z(corpus.num_tokens, K, args.seed), redraw_phi=True)
                one_dt = None
                for n in (args.warmup, args.steps):             # the same batches; only the number of sweeps matters to the state
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for i in range(0, n, 5):
                        h1.sweep(min(5, n - i))
                    torch.cuda.synchronize()
                    one_dt = time.perf_counter() - t1           # of the second batch: the same --steps sweeps the N ranks were timed on
                z1, th1 = h1.get_z(), h1.get_theta()
                one = state_digests(h1, z1)
                bounds = even_split(corpus.num_docs, world)
                bad = []
                for e in every:
                    a, b = e["tok_range"]
                    d0, d1 = bounds[e["rank"]], bounds[e["rank"] + 1]
                    if sha256_of(z1[a:b]) != e["z"]:
                        bad.append("z of rank %d" % e["rank"])
                    if sha256_of(th1[d0:d1]) != e["theta"]:
                        bad.append("theta of rank %d" % e["rank"])
                    if e["phi"] != one["phi"]:
                        bad.append("phi on rank %d" % e["rank"])
                    if e["n_k"] != one["n_k"]:
                        bad.append("n_k on rank %d" % e["rank"])
                out = {"parity_vs_one_gpu": not bad,
                       "compared": "sha256 of the bit patterns of z and theta (per shard), phi and n_k (on every rank) after %d sweeps from the same z0, against ONE handle "
                                   "over the whole corpus run by rank 0 after the timed region" % (args.warmup + args.steps),
                       "mismatches": bad, "one_gpu_z_form": h1.launch_info().get("z_form"), "z_form_per_rank": [e["z_form"] for e in every],
                       "comm_nranks_per_rank": [e["comm_nranks"] for e in every],
                       # the same corpus, the same sweeps, ONE GPU (rank 0's), in this very run: the strong-scaling ratio without a second invocation
                       "one_gpu_ms_per_step": round(one_dt / args.steps * 1e3, 4),
                       "speedup_vs_one_gpu_same_run": round(one_dt / r["dt"], 3)}
            finally:
                h1.close()
            stage(rank, "verification %s" % ("OK: bit-identical to the one-GPU run" if out["parity_vs_one_gpu"] else "FAILED: %s" % out["mismatches"]))
        if world > 1:
            dist.barrier()
        return out

    fallback = []

    def run_sharded_or_fall_back(weak, docs=None):
        """A native exchange that cannot be set up on every rank (librccl not loadable, ncclCommInitRank refused) would cost
        the whole scaling record: all ranks then agree to repeat the leg over torch.distributed (round 1's form: dense
        count all-reduce, Phi re-drawn everywhere) and the line says so.  Only failures every rank sees before its first
        sweep are covered -- one rank failing inside a collective leaves the others waiting, as with any collective."""
        err = None
        try:
            res = run_sharded(weak, docs)
        except Exception as e:      # noqa: BLE001 -- whatever it was, the other ranks must hear of it
            res, err = None, "%s: %s" % (type(e).__name__, e)
        bad = torch.tensor([0 if err is None else 1], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) == 0:
            return res
        if args.exchange != "native":
            raise RuntimeError(err or "another rank failed")
        stage(rank, "native exchange failed (%s); falling back to --exchange torch" % (err or "on another rank"))
        fallback.append(err or "failed on another rank")
        args.exchange = "torch"
        return run_sharded(weak, docs)

    sha = csrc_sha16()
    other = None
    large = None
    corpus = z0 = None
    verification = None
    if sharded:
        weak_first = args.scaling == "weak"
        r = run_sharded_or_fall_back(weak_first)
        r2 = None
        if world > 1 and not args.no_weak_leg:
            r2 = run_sharded_or_fall_back(not weak_first)
        strong_leg = r2 if weak_first else r
        if strong_leg is not None and "digests" in strong_leg:
            verification = verify_against_one_gpu(strong_leg)
            strong_leg.pop("corpus", None)
        # The strong split of a corpus --large-factor times the headline's: where the per-rank budget of DESIGN.md section 6
        # puts 8 GPUs at >= 6x one (the headline corpus is 1.5 ms of work on ONE GPU: its split is bound by what does not
        # shrink -- the Phi slice's chain and the collectives).  Timed and verified against one handle like the headline leg.
        if world > 1 and args.large_factor > 1 and not args.no_large_leg and not args.no_verify:
            r3 = run_sharded_or_fall_back(False, docs=args.large_factor * args.docs)
            v3 = verify_against_one_gpu(r3)
            r3.pop("corpus", None)
            if rank == 0:
                large = {"workload": "the strong split of a %dx corpus: D=%d, N=%d tokens over %d ranks" % (args.large_factor, r3["docs"], r3["tokens"], world),
                         "value": round(r3["tokens"] * args.steps / r3["dt"] / 1e6, 3), "unit": "M tokens/s", "ms_per_step": round(r3["dt"] / args.steps * 1e3, 4),
                         "phase_ms_per_sweep": r3["phases"], "parity_vs_one_gpu": v3["parity_vs_one_gpu"], "one_gpu_ms_per_step": v3["one_gpu_ms_per_step"],
                         "speedup_vs_one_gpu_same_run": v3["speedup_vs_one_gpu_same_run"]}
                if not v3["parity_vs_one_gpu"]:
                    verification = dict(verification or {}, parity_vs_one_gpu=False, mismatches=(verification or {}).get("mismatches", []) + ["large leg: %s" % v3["mismatches"]])
        if r2 is not None:
            other = {"value": round(r2["tokens"] * args.steps / r2["dt"] / 1e6, 3), "unit": "M tokens/s",
                     "ms_per_step": round(r2["dt"] / args.steps * 1e3, 4), "phase_ms_per_sweep": r2["phases"],
                     "workload": ("every rank brings D=%d documents: D=%d, N=%d tokens over %d ranks" % (args.docs, r2["docs"], r2["tokens"], world)) if not weak_first
                     else "the N=1 corpus (D=%d, N=%d tokens) split across the %d ranks" % (r2["docs"], r2["tokens"], world)}
    else:
        corpus = synthetic_lda_corpus(args.docs, args.types, args.mean_len, true_topics=100, seed=args.seed)
        z0 = java_lcg_initial_z(corpus.num_tokens, K, args.seed)
        run_corpus, run_z0 = corpus, z0
        if args.simulate_world > 1:
            b = even_split(corpus.num_docs, args.simulate_world)
            run_corpus, _, tb = corpus.shard(b[args.simulate_rank], b[args.simulate_rank + 1])
            run_z0 = z0[tb:tb + run_corpus.num_tokens]
        dt, ph, info = run_single(native, run_corpus, run_z0, K, args, local_rank, args.steps, args.warmup, fence)
        r = dict(dt=dt, phases=ph, docs=run_corpus.num_docs, tokens=run_corpus.num_tokens, n_local=run_corpus.num_tokens, V=corpus.num_types,
                 info=info, local=run_corpus)

    if rank == 0:
        workload = (workload_string(r["docs"], r["V"], r["tokens"], K, args)
                    + (" (%d documents per rank)" % args.docs if sharded and args.scaling == "weak" else "")
                    + (" [TIMING AID: rank %d of a simulated %d-rank split, peers missing]" % (args.simulate_rank, args.simulate_world) if args.simulate_world > 1 else ""))
        if sharded:
            how = ("native exchange: RCCL reduce-scatter of int32 counts by topic slice + Phi slice draw + fp64 all-gather" if args.exchange == "native" and args.backend == "nccl"
                   else "native exchange over the callback provider (gloo, host staged)" if args.exchange == "native"
                   else "torch.distributed int32 count all-reduce (%s), Phi re-drawn on every rank" % ("RCCL" if args.backend == "nccl" else "gloo"))
            par = "doc-sharded x%d, %s%s" % (world, how, ", ALL RANKS ON ONE GPU (rehearsal)" if args.single_device else "")
            if fallback:
                par += " [FELL BACK from the native exchange: %s]" % fallback[0][:200]
        else:
            par = "1 GPU"
        line = {
            "metric": "M tokens sampled/sec (whole node) per Gibbs sweep at K=%d" % K + ("" if args.scheme == "ggs" else " (scheme=%s)" % args.scheme),
            "value": round(r["tokens"] * args.steps / r["dt"] / 1e6, 3),
            "unit": "M tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(r["dt"] / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling if sharded else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "parallelism": par},
            "roofline": roofline_block(r["local"], K, args.scheme, r["n_local"], r["phases"]["z_ms"], workload, sha, r["info"].get("num_hot", 0), r["info"].get("z_parts", 1), r["info"].get("warm_tiers", 0)),
            "phase_ms_per_sweep": r["phases"],
            # which z kernel(s) ran, and which of the two forms of the K <= 160 step the first z step's timed comparison kept
            "z_step": {k: r["info"].get(k) for k in ("z_kernel", "z_form", "z_form_calibrated", "z_parts", "num_hot", "warm_tiers", "num_warm", "warm_docs_per_chunk")},
            "build": {"csrc_sha16": sha},
        }
        if sharded:
            # who carried the collectives, as the library reports it: for RCCL the rank count read back from the communicator
            x = r.get("exchange", {})
            line["exchange"] = {"provider": x.get("provider"), "rccl_nranks": x.get("comm_nranks") if x.get("provider") == "rccl" else None,
                                "nranks": x.get("nranks"), "topic_slice_rank0": [x.get("k_begin"), x.get("k_end")], "count_exchange": x.get("count_exchange")}
            if verification is not None:
                line.update({"parity_vs_one_gpu": verification["parity_vs_one_gpu"], "verification": verification})
            else:
                line["parity_vs_one_gpu"] = None
        if other is not None:
            line["weak_scaling" if args.scaling == "strong" else "strong_scaling"] = other
        if sharded and large is not None:
            line["strong_scaling_large_corpus"] = large
        if not sharded and args.simulate_world <= 1:
            if not args.no_extra_configs and args.scheme == "ggs" and K == 100 and args.docs == 100000:
                line["extra_configs"] = extra_configs(native, corpus, args, local_rank, fence, sha)
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(corpus, K, args.alpha, args.beta, args.seed, z0, args.cpu_sample_docs)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and verification is not None and not verification["parity_vs_one_gpu"]:
        sys.exit("bench.py: the %d-rank end state differs from the one-GPU run: %s" % (world, verification["mismatches"]))


if __name__ == "__main__":
    main()
