"""Why is sum_walk_kernel<int,true> slow on converged K=1024 counts?  Times ggs_debug_column_sum on the real counts and on variants."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ldagroupedgibbssampler_amd import native
from ldagroupedgibbssampler_amd.corpus import synthetic_lda_corpus
from ldagroupedgibbssampler_amd.sharded import java_lcg_initial_z
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
c = synthetic_lda_corpus(100000, 50000, 200, true_topics=100, seed=2019)
h = native.GGSHandle(K, c.num_types, 0.1, 0.01, 2019)
h.set_corpus(c.doc_ptr, c.tokens)
h.set_z(java_lcg_initial_z(c.num_tokens, K, 2019), redraw_phi=True)
def timeit(name, counts):
    native.debug_column_sum(counts=counts, beta=0.01)
    t0 = time.perf_counter()
    for _ in range(3):
        out = native.debug_column_sum(counts=counts, beta=0.01)
    dt = (time.perf_counter() - t0) / 3
    print("%-40s %.1f ms (incl. upload)" % (name, dt * 1e3), flush=True)
    return out
n0 = h.get_type_topic_counts()
timeit("initial counts", n0)
h.sweep(8)
n = h.get_type_topic_counts()
nk = n.sum(0)
print("n_k quantiles", np.quantile(nk, [0, 0.01, 0.1, 0.5, 0.9, 0.99, 1]).astype(int), "empty topics", int((nk == 0).sum()), "nnz frac", float((n > 0).mean()))
print("max count", int(n.max()))
timeit("after 8 sweeps", n)
rng = np.random.default_rng(0)
timeit("rows permuted", n[rng.permutation(n.shape[0])])
timeit("clipped to 1000", np.minimum(n, 1000))
timeit("clipped to 50", np.minimum(n, 50))
timeit("zeros", np.zeros_like(n))
timeit("only columns with n_k>20000", np.where(nk[None, :] > 20000, n, 0))
timeit("only columns with n_k<=20000", np.where(nk[None, :] <= 20000, n, 0))
