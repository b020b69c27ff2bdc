"""Per-draw cost of the device gamma sampler by shape (run under rocprofv3 --kernel-trace: one
debug_draw_kernel launch per shape, in this order)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ldagroupedgibbssampler_amd import native
N = 4 << 20
for kind, a in [("uniform", None), ("gaussian", None), ("gamma", 0.01), ("gamma", 0.1), ("gamma", 0.5), ("gamma", 1.01), ("gamma", 5.0), ("gamma", 200.0)]:
    if kind == "gamma":
        out, st = native.debug_draw("gamma", 7, 1, native.PURPOSE_PHI, 0, shape=np.full(N, a))
    else:
        out, st = native.debug_draw(kind, 7, 1, native.PURPOSE_PHI, 0, n=N)
    print(kind, a, st, float(out.mean()))
