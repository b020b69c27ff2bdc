#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, double x, int n) {
  double a = out[threadIdx.x], b = a + 1, c = a + 2, d = a + 3;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) { a += x; a += x; a += x; a += x; a += x; a += x; a += x; a += x; }            // dependent add chain
    if (MODE == 1) { a += x; b += x; c += x; d += x; a += x; b += x; c += x; d += x; }            // 4 independent chains
    if (MODE == 2) { a += b * x; a += c * x; a += d * x; a += b * x; a += c * x; a += d * x; a += b * x; a += c * x; } // mul indep + dep add (contract off)
    if (MODE == 3) { a = __builtin_fma(b, x, a); a = __builtin_fma(c, x, a); a = __builtin_fma(d, x, a); a = __builtin_fma(b, x, a); a = __builtin_fma(c, x, a); a = __builtin_fma(d, x, a); a = __builtin_fma(b, x, a); a = __builtin_fma(c, x, a);} // dep fma
    if (MODE == 4) { float fa = (float)a; for (int j = 0; j < 8; ++j) fa += (float)x; a = fa; }
  }
  long long t1 = clock64();
  out[threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0) out[64 + blockIdx.x] = (double)(t1 - t0) / (8.0 * n);
}
int main() {
  double* d; hipMalloc(&d, 8 * 4096); hipMemset(d, 0, 8 * 4096);
  double h[4096];
  #define RUN(M, G) { hipLaunchKernelGGL(k<M>, dim3(G), dim3(64), 0, 0, d, 1e-9, 100000); hipDeviceSynchronize(); hipMemcpy(h, d, 8*4096, hipMemcpyDeviceToHost); printf("mode %d grid %d: %.2f clock64-ticks per op\n", M, G, h[64]); }
  RUN(0,1) RUN(1,1) RUN(2,1) RUN(3,1)
  RUN(0,1024) RUN(2,1024) RUN(0,2048) RUN(2,2048)
  // wall-clock version to calibrate clock64 ticks
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, 1e-9, 1000000); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); printf("8e6 dependent adds: %.3f ms -> %.2f ns per add\n", ms, ms * 1e6 / 8e6);
}
