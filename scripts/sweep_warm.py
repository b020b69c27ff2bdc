"""Timing of the warm tiers (z_warm_kernel) on the benchmark corpus: one corpus, one handle per setting of the
GGS_DEBUG_WARM* / GGS_DEBUG_HOT knobs, `--steps` sweeps each; prints ms per sweep, the phases and what the lists look like.
Every setting's end state is hashed: all settings must agree (the lists only regroup tokens).

  python scripts/sweep_warm.py --settings "WARM=0;WARM=1;WARM=2;WARM=4;WARM=8;WARM=8,WARM_FILL=25"
"""
import argparse
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=100000)
    ap.add_argument("--types", type=int, default=50000)
    ap.add_argument("--mean-len", type=int, default=200)
    ap.add_argument("--topics", type=int, default=100)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--simulate-world", type=int, default=1)
    ap.add_argument("--settings", default="WARM=0;WARM=1;WARM=2;WARM=3;WARM=4;WARM=6;WARM=8")
    args = ap.parse_args()
    os.environ["GGS_DEBUG"] = "1"
    import numpy as np
    import torch
    from ldagroupedgibbssampler_amd.sharded import java_lcg_initial_z
    from ldagroupedgibbssampler_amd import native
    from ldagroupedgibbssampler_amd.corpus import even_split, synthetic_lda_corpus

    c = synthetic_lda_corpus(args.docs, args.types, args.mean_len, true_topics=100, seed=2019)
    z0 = java_lcg_initial_z(c.num_tokens, args.topics, 2019)
    if args.simulate_world > 1:                      # rank 0 of an N-way split, the peers missing (bench.py --simulate-world)
        b = even_split(c.num_docs, args.simulate_world)
        c, _, t0 = c.shard(b[0], b[1])
        z0 = z0[:c.num_tokens]
    print("corpus D=%d V=%d N=%d K=%d" % (c.num_docs, c.num_types, c.num_tokens, args.topics), flush=True)
    digests = set()
    for setting in args.settings.split(";"):
        env = {}
        for kv in filter(None, setting.split(",")):
            k, v = kv.split("=")
            env["GGS_DEBUG_" + k] = v
        for k, v in env.items():
            os.environ[k] = v
        h = native.GGSHandle(args.topics, c.num_types, 0.1, 0.01, 2019, device_id=0)
        if args.simulate_world > 1:
            h.attach_null_exchange(0, args.simulate_world)
        h.set_corpus(c.doc_ptr, c.tokens)
        for k in env:
            del os.environ[k]
        h.set_z(z0, redraw_phi=True)
        for i in range(0, args.warmup, 5):
            h.sweep(min(5, args.warmup - i))
        h.reset_timings()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(0, args.steps, 5):
            h.sweep(min(5, args.steps - i))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tm = h.get_timings()
        n = max(tm["sweeps"], 1)
        info = h.launch_info()
        dg = hashlib.sha256(h.get_z().tobytes()).hexdigest()[:16] if args.simulate_world <= 1 else "n/a"
        digests.add(dg)
        print(json.dumps({"setting": setting, "ms_per_sweep": round(dt / args.steps * 1e3, 4), "z_ms": round(tm["z_ms"] / n, 4),
                          "theta_ms": round(tm["theta_ms"] / n, 4), "merge_ms": round(tm["merge_ms"] / n, 4), "phi_ms": round(tm["phi_ms"] / n, 4),
                          "num_hot+warm": info["num_hot"], "warm_tiers": info["warm_tiers"], "num_warm": info["num_warm"], "z_form": info["z_form"],
                          "chunks": info["num_chunks"], "z_sha": dg}), flush=True)
        h.close()
    print("all settings agree on z: %s" % (len(digests) == 1), flush=True)
    return 0 if len(digests) == 1 else 1


if __name__ == "__main__":
    sys.exit(main())
