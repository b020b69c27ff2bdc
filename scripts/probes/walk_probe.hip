// walk_probe -- the exact column sums of ggs_exact_sum.hpp on their own (no theta draw beside them): kernel times by
// hipEvent and, for the wave of topic 0, 100 MHz timestamps at the walk's phases + how its steps split into accepted
// runs / element-path segments with rows fetched ahead / fetched on the spot.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -DGGS_WALK_TRACE=1 -I ldagroupedgibbssampler_amd/csrc -o scripts/bin/walk_probe scripts/probes/walk_probe.hip
//   (-DGGS_WALK_TRACE=2 adds the step counters)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "ggs_exact_sum.hpp"
using namespace ggs;

int main(int argc, char **argv) {
  const int V = argc > 1 ? atoi(argv[1]) : 50000, K = argc > 2 ? atoi(argv[2]) : 13;
  const int nseg = (V + 63) / 64;
  std::mt19937_64 rng(7);
  std::vector<int32_t> cnt((size_t)V * K);
  std::vector<double> zipf(V);
  for (int v = 0; v < V; ++v) zipf[v] = 1.0 / std::pow(1.0 + (rng() % V), 1.07);
  for (int v = 0; v < V; ++v)
    for (int k = 0; k < K; ++k) {
      const double lam = 200000.0 * zipf[v] / 12.0;
      cnt[(size_t)v * K + k] = (rng() % 1000 < 150) ? (int32_t)(lam * ((rng() % 1000) / 500.0)) : 0;
    }
  int32_t *d_cnt; double *d_guess, *d_fn, *d_out; int32_t *d_nk;
  hipMalloc(&d_cnt, cnt.size() * 4); hipMalloc(&d_guess, (size_t)(nseg + 1) * K * 8); hipMalloc(&d_fn, (size_t)nseg * K * 32);
  hipMalloc(&d_out, K * 8); hipMalloc(&d_nk, K * 4);
  hipMemcpy(d_cnt, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice);
  SumParams sp{};
  sp.src = d_cnt; sp.guess = d_guess; sp.fn = d_fn; sp.out = d_out; sp.n_k = d_nk; sp.beta = 0.01; sp.pitch = K; sp.K = K; sp.V = V; sp.nseg = nseg; sp.write_pref = 1;
  const dim3 rows((unsigned)nseg, (unsigned)((K + kSumBlock - 1) / kSumBlock)), fnr((unsigned)nseg, (unsigned)((K + kSegFnCols - 1) / kSegFnCols));
  hipLaunchKernelGGL((sum_seg_kernel<int32_t, true>), rows, dim3(kSumBlock), 0, 0, sp);
  hipLaunchKernelGGL(sum_prefix_kernel, dim3((unsigned)K), dim3(64), 0, 0, sp);
  hipEvent_t e[4]; for (auto &x : e) hipEventCreate(&x);
  float t_fn = 0, t_walk = 0;
  for (int it = 0; it < 12; ++it) {
    hipEventRecord(e[0]);
    hipLaunchKernelGGL((sum_segfn_kernel<int32_t, true>), fnr, dim3(256), 0, 0, sp);
    hipEventRecord(e[1]);
    hipLaunchKernelGGL((sum_walk_kernel<int32_t, true>), dim3((unsigned)K), dim3(64), 0, 0, sp);
    hipEventRecord(e[2]);
    hipEventSynchronize(e[2]);
    float a, b; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]);
    if (it >= 2) { t_fn += a; t_walk += b; }
    if (it == 0) { unsigned long long z[128] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_walk_trace), z, sizeof z); }   // counters of the traced launches only
  }
  printf("V=%d K=%d nseg=%d: segfn %.1f us, walk %.1f us (event to event, mean of 10)\n", V, K, nseg, t_fn * 100, t_walk * 100);
  unsigned long long tr[128]; hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_walk_trace), sizeof tr);
  printf("topic 0, last launch (us since kernel entry): prologue done %.2f", (tr[1] - tr[0]) / 100.0);
  for (int it = 0; it * kWalkSuper < nseg && it < 18; ++it) printf(" | sg%d walked %.2f staged %.2f", it, (tr[2 + 2 * it] - tr[0]) / 100.0, (tr[3 + 2 * it] - tr[0]) / 100.0);
  printf("\n");
#if GGS_WALK_TRACE > 1   // the step counters are global read-modify-writes: they distort the times above, so they are a build of their own
  printf("steps over 11 launches: %llu accepted runs, %llu element-path segments with rows ahead, %llu fetched on the spot\n", tr[40], tr[41], tr[42]);
  printf("per super-group (runs/rows ahead/on the spot):");
  for (int it = 0; it * kWalkSuper < nseg && it < 16; ++it) printf(" sg%d %llu/%llu/%llu", it, tr[64 + it], tr[80 + it], tr[96 + it]);
  printf("\n");
#endif
  std::vector<double> out(K); hipMemcpy(out.data(), d_out, K * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int k = 0; k < K; ++k) {
    double want = 0; for (int v = 0; v < V; ++v) want += 0.01 + (double)cnt[(size_t)v * K + k];
    if (out[k] != want) { if (!bad) printf("topic %d: %.17g, sequential %.17g (DIFFERENT)\n", k, out[k], want); ++bad; }
  }
  printf("%d of %d topics differ from the sequential sum\n", bad, K);
  return 0;
}
