// Where does the immediate offset of global_load_lds_dwordx4 go: the global address only, or the
// LDS destination as well?  A wave DMA-copies 1 KiB from src + OFFSET with LDS base 4096.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
typedef const void __attribute__((address_space(1))) glb_cvoid_t;
typedef void __attribute__((address_space(3))) lds_void_t;
__global__ __launch_bounds__(64) void k(const uint32_t *src, uint32_t *out) {
  __shared__ uint32_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  __builtin_amdgcn_global_load_lds((glb_cvoid_t *)((const unsigned char *)src + threadIdx.x * 16), (lds_void_t *)((unsigned char *)lds + 4096), 16, 256, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
  uint32_t *s, *o, h[4096];
  hipMalloc(&s, 16384); hipMalloc(&o, 16384);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  hipMemcpy(s, h, 16384, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, o);
  hipMemcpy(h, o, 16384, hipMemcpyDeviceToHost);
  int first = -1, last = -1;
  for (int i = 0; i < 4096; ++i) if (h[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
  printf("LDS words written: [%d, %d] (byte %d..), first value %u (global byte %u)\n", first, last, first * 4, h[first], h[first] * 4);
  printf("expected if offset applies to global only: LDS byte 4096, global byte 256; if to both: LDS byte 4352, global byte 256\n");
}
