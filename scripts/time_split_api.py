#!/usr/bin/env python3
"""Times the sweep through the split entry points the Java binding uses (ggs_sweep_begin + ggs_sweep_end_async per
iteration: GGSDevice.zStep / phiStep) beside whole sweeps (ggs_sweep), on BASELINE config 2."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device plumbing, imported before the library as everywhere else)
from ldagroupedgibbssampler_amd import native  # noqa: E402
from ldagroupedgibbssampler_amd.corpus import synthetic_lda_corpus  # noqa: E402
from ldagroupedgibbssampler_amd.sharded import java_lcg_initial_z  # noqa: E402

K = 100
c = synthetic_lda_corpus(100000, 50000, 200, true_topics=100, seed=2019)
h = native.GGSHandle(K, c.num_types, 0.1, 0.01, 2019, device_id=0)
h.set_corpus(c.doc_ptr, c.tokens)
h.set_z(java_lcg_initial_z(c.num_tokens, K, 2019), redraw_phi=True)
for mode in ("whole", "split", "whole", "split"):
    def run(n):
        if mode == "whole":
            h.sweep(n)
        else:
            for _ in range(n - 1):
                h.sweep_begin()
                h.sweep_end_async()
            h.sweep_begin()
            h.sweep_end()
    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(20)
    torch.cuda.synchronize()
    print("%s: %.4f ms per sweep" % (mode, (time.perf_counter() - t0) / 20 * 1e3))
h.check_invariants()
h.close()
