// How much LDS can ONE workgroup per CU use before a 256-workgroup launch no longer fits the 256 CUs in a single round?  (run on the GPU box)
// The specialised per-frame program wants every byte of the 160 KB; its launch took 162 us at 160.0 KB against 98 us at 150.2 KB while workgroup 0's
// own timeline stayed at 81 us — i.e. some workgroups waited for a CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_lds_occupancy tools/microbench_lds_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(1024) void spin_k(unsigned long long* out, int spin_us) {
  extern __shared__ float smem[];
  smem[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin_us * 100) {}
  if (threadIdx.x == 0) out[blockIdx.x] = t0 + (unsigned long long)smem[5];
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 4096 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(spin_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int wgs : {256, 512}) {
    for (int kb4 = 4 * 144; kb4 <= 4 * 160; kb4 += 2) {                 // 0.5 KB steps
      const size_t lds = (size_t)kb4 * 256;
      spin_k<<<wgs, 1024, lds>>>(d, 50);
      hipDeviceSynchronize();
      hipEventRecord(a);
      spin_k<<<wgs, 1024, lds>>>(d, 50);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      printf("%4d workgroups x 1024 lanes, %6.1f KB LDS each: %7.1f us (%s)\n", wgs, lds / 1024.0, ms * 1e3, hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}
