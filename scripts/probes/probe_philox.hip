// throughput of one Philox4x32-10 block per lane, three ways of forming the 32x32->64 products
// hipcc --offload-arch=gfx950 -O3 -o probe_philox probe_philox.hip && ./probe_philox
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

struct U4 { uint32_t x, y, z, w; };
constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;

template <int MODE>
__device__ __forceinline__ void mulhilo(uint32_t m, uint32_t c, uint32_t &hi, uint32_t &lo) {
  if (MODE == 0) { hi = __umulhi(m, c); lo = m * c; }
  else if (MODE == 1) { const uint64_t p = (uint64_t)m * c; hi = (uint32_t)(p >> 32); lo = (uint32_t)p; }
  else {
    uint64_t p;
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(c), "s"(m) : "vcc");
    hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
  }
}

template <int MODE>
__device__ __forceinline__ U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    mulhilo<MODE>(M0, c0, hi0, lo0);
    mulhilo<MODE>(M1, c2, hi1, lo1);
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  return U4{c0, c1, c2, c3};
}

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters) {
  uint32_t a = threadIdx.x + blockIdx.x * 256, acc = 0;
  for (int i = 0; i < iters; ++i) {
    const U4 o = philox<MODE>(a, i, 7, acc, 2019, 5);
    acc ^= o.x ^ o.y ^ o.z ^ o.w;
  }
  out[threadIdx.x + blockIdx.x * 256] = acc;
}

__global__ __launch_bounds__(256) void kfma(double *out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 100; ++j) a = __builtin_fma(a, b, c);
  }
  out[threadIdx.x + blockIdx.x * 256] = a;
}

template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  const int blocks = 256 * 8, iters = 2000;     // 8 waves per SIMD
  uint32_t *d; hipMalloc(&d, blocks * 256 * 8);
  uint32_t h[3][4];
  float t0 = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters); }); hipMemcpy(h[0], d, 16, hipMemcpyDeviceToHost);
  float t1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters); }); hipMemcpy(h[1], d, 16, hipMemcpyDeviceToHost);
  float t2 = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters); }); hipMemcpy(h[2], d, 16, hipMemcpyDeviceToHost);
  float tf = timeit([&] { hipLaunchKernelGGL(kfma, dim3(blocks), dim3(256), 0, 0, (double *)d, iters); });
  const double waves = blocks * 4.0, per = 1e6 / (waves * iters) * 1024;   // ns of one SIMD per wave-block
  printf("mul_hi+mul_lo : %.3f ms  %.1f SIMD-ns per wave block  (%08x)\n", t0, t0 * per, h[0][1]);
  printf("u64 product   : %.3f ms  %.1f SIMD-ns per wave block  (%08x)\n", t1, t1 * per, h[1][1]);
  printf("v_mad_u64_u32 : %.3f ms  %.1f SIMD-ns per wave block  (%08x)\n", t2, t2 * per, h[2][1]);
  printf("100 dependent fp64 fma: %.3f ms  %.1f SIMD-ns per wave per 100\n", tf, tf * per);
  return 0;
}
