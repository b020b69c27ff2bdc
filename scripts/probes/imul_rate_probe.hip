// Issue cost of the 32-bit multiplies Philox4x32 is made of, and of two ways to write a round:
//   mode 0  v_mul_lo_u32 chain          mode 1  v_mul_hi_u32 chain        mode 2  v_mad_u64_u32 (full 64-bit product)
//   mode 3  v_add_u32 (baseline)        mode 4  philox round as mul_hi + mul_lo   mode 5  philox round from one 64-bit product
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t x, int n) {
  uint32_t a = out[threadIdx.x], b = a + 1, c = a + 2, d = a + 3;
  uint32_t k0 = x, k1 = x * 3;
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) { a *= x; b *= x; c *= x; d *= x; a *= x; b *= x; c *= x; d *= x; }
    if (MODE == 1) { a = __umulhi(a, x) + 1; b = __umulhi(b, x) + 1; c = __umulhi(c, x) + 1; d = __umulhi(d, x) + 1; }   // 4 mulhi + 4 add
    if (MODE == 2) {
      uint64_t p = (uint64_t)a * x, q = (uint64_t)c * x;
      a = (uint32_t)(p >> 32) ^ b; b = (uint32_t)p; c = (uint32_t)(q >> 32) ^ d; d = (uint32_t)q;
      p = (uint64_t)a * x; q = (uint64_t)c * x;
      a = (uint32_t)(p >> 32) ^ b; b = (uint32_t)p; c = (uint32_t)(q >> 32) ^ d; d = (uint32_t)q;
    }
    if (MODE == 3) { a += x; b += x; c += x; d += x; a += b; b += c; c += d; d += a; }
    if (MODE == 4) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, a), lo0 = 0xD2511F53u * a;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c), lo1 = 0xCD9E8D57u * c;
        const uint32_t n0 = hi1 ^ b ^ k0, n2 = hi0 ^ d ^ k1;
        a = n0; b = lo1; c = n2; d = lo0; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
      }
    }
    if (MODE == 5) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * a, p1 = (uint64_t)0xCD9E8D57u * c;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ b ^ k0, n2 = (uint32_t)(p0 >> 32) ^ d ^ k1;
        a = n0; b = (uint32_t)p1; c = n2; d = (uint32_t)p0; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
      }
    }
  }
  out[threadIdx.x] = a + b + c + d + k0 + k1;
}
template <int MODE>
void run(uint32_t *d, int grid, int n, const char *what) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 12345u, 1000);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 12345u, n);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("mode %d (%s) grid %5d: %.3f ms, %.2f ns per loop body per wave-slot\n", MODE, what, grid, ms, ms * 1e6 / n / (grid > 1024 ? grid / 1024.0 : 1.0));
}
int main() {
  uint32_t *d; (void)hipMalloc(&d, 4 * 4096); (void)hipMemset(d, 1, 4 * 4096);
  for (int g : {1024, 4096}) {
    run<0>(d, g, 200000, "8 mul_lo");
    run<1>(d, g, 200000, "4 mul_hi + 4 add");
    run<2>(d, g, 200000, "4 x 64-bit product + 4 xor");
    run<3>(d, g, 200000, "8 add");
    run<4>(d, g, 200000, "2 philox rounds, mul_hi+mul_lo");
    run<5>(d, g, 200000, "2 philox rounds, 64-bit product");
  }
}
