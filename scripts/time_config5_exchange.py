#!/usr/bin/env python3
"""One rank of BASELINE config 5 (D=5M, V=1M, K=500 over 8 GPUs: a shard of 625 000 documents, 125 M tokens) on ONE GPU with the
peers missing (ggs_attach_null_exchange: the collectives become local copies), the count exchange dense and sparse: what the
two forms cost the rank itself per sweep (the wire is not measured here; DESIGN.md section 6 prices it).
usage: python scripts/time_config5_exchange.py [sweeps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ldagroupedgibbssampler_amd import native  # noqa: E402
from ldagroupedgibbssampler_amd.corpus import zipf_unigram_corpus  # noqa: E402


def main():
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    D, V, K, world, rank = 625000, 1000000, 500, 8, 3
    t0 = time.time()
    c = zipf_unigram_corpus(D, V, 200, seed=2019)
    N = c.num_tokens
    z0 = np.random.default_rng(1).integers(0, K, N, dtype=np.int32)
    print("corpus: %d tokens in %.0f s" % (N, time.time() - t0), flush=True)
    out = {"workload": "one rank of 8 of BASELINE config 5: D=%d V=%d K=%d N=%d, peers missing (null exchange)" % (D, V, K, N)}
    for mode in ("dense", "sparse"):
        h = native.GGSHandle(K, V, 0.1, 0.01, 2019)
        h.attach_null_exchange(rank, world)
        h.set_count_exchange(mode)
        h.set_corpus(c.doc_ptr, c.tokens, rank * D, rank * N)
        h.set_global_token_count(world * N)
        h.set_z(z0, redraw_phi=True)
        h.sweep(1)
        h.reset_timings()
        t1 = time.perf_counter()
        h.sweep(sweeps)
        dt = (time.perf_counter() - t1) / sweeps
        tm = h.get_timings()
        how = h.count_exchange()
        out[mode] = {"ms_per_sweep": round(dt * 1e3, 3), "phase_ms": {k: round(tm[k] / sweeps, 3) for k in ("theta_ms", "z_ms", "merge_ms", "phi_ms", "exchange_rs_ms", "exchange_ag_ms")},
                     "count_exchange": how, "bytes_sent_dense_int32": how["dense_cells"] * 4, "bytes_sent_pairs": how["pairs_last"] * 8}
        print(mode, json.dumps(out[mode]), flush=True)
        h.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
