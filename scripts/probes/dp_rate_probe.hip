// fp64 VALU throughput of a CU as a function of waves per SIMD (single-wave workgroups).
// Question: a lone wave issues one v_add_f64 per 8 cycles -- do 2 or 4 waves on the same SIMD
// interleave to a higher aggregate rate, or is 8 cycles per wave-op the SIMD's rate?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, double x, int n) {
  double a = out[threadIdx.x], b = a + 1, c = a + 2, d = a + 3;
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) { a += x; b += x; c += x; d += x; a += x; b += x; c += x; d += x; }
    if (MODE == 1) { a = __builtin_fma(b, x, a); b = __builtin_fma(c, x, b); c = __builtin_fma(d, x, c); d = __builtin_fma(a, x, d);
                     a = __builtin_fma(b, x, a); b = __builtin_fma(c, x, b); c = __builtin_fma(d, x, c); d = __builtin_fma(a, x, d); }
    if (MODE == 2) { a += b * x; b += c * x; c += d * x; d += a * x; }   // mul + add, contract off: 8 ops
  }
  out[threadIdx.x] = a + b + c + d;
}
template <int MODE>
void run(double *d, int grid, int n) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 1e-9, 1000);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 1e-9, n);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double waveops = 8.0 * n * grid;
  printf("mode %d grid %5d (%.1f waves/SIMD): %.3f ms, %.2f ns per wave-op per SIMD-slot, %.2f T lane-ops/s\n", MODE, grid, grid / 1024.0, ms,
         ms * 1e6 / (8.0 * n) / (grid > 1024 ? 1.0 : 1.0), waveops * 64 / (ms * 1e-3) / 1e12);
}
int main() {
  double *d; hipMalloc(&d, 8 * 4096); hipMemset(d, 0, 8 * 4096);
  for (int g : {256, 1024, 2048, 4096, 8192}) run<0>(d, g, 200000);
  for (int g : {1024, 2048, 4096}) run<1>(d, g, 200000);
  for (int g : {1024, 2048, 4096}) run<2>(d, g, 200000);
}
