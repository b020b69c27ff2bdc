#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(int* out, int n) {
  extern __shared__ unsigned char smem[];
  smem[threadIdx.x] = 1;
  __syncthreads();
  // census: count co-resident blocks per CU via s_getreg HW_ID? simply spin for a while
  long long t0 = clock64();
  while (clock64() - t0 < 2000000) {}
  if (threadIdx.x == 0) out[blockIdx.x] = smem[0] + __builtin_amdgcn_s_getreg(4 | (8<<6) | (3 << 11)); // HW_ID cu_id bits
}
int main() {
  for (int lds : {13000, 26912, 40000, 52224, 65536, 80000}) {
    int nb = 0;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64, lds);
    int* d; hipMalloc(&d, 4 * 4096);
    for (int wpc : {1, 2, 3, 4, 6, 8}) {
      int grid = 256 * wpc;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, 0, d, 0);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("lds=%d occ_api=%d grid=256x%d time=%.3f ms\n", lds, nb, wpc, ms);
    }
    hipFree(d);
  }
}
