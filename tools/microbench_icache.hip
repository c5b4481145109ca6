// What does COLD straight-line code cost on gfx950?  (run on the GPU box)
// The per-frame network program executes every op body once per launch on every CU; whether that is an instruction-fetch problem decides
// between "one specialised straight-line kernel per graph" and "few compact, re-used loops".
//   one 1024-lane (or 256-lane) workgroup per CU runs S KB of straight-line VALU code twice; per-pass shader cycles of wave 0 are recorded.
//   pass 1 of the first launch = cold (instruction cache and L2 miss), pass 2 = whatever the 64 KB instruction cache kept.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_icache tools/microbench_icache.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define STR2(x) #x
#define STR(x) STR2(x)
// one block = 8 independent 4-byte VALU instructions = 32 bytes; REPS blocks per body
#define BODY(REPS)                                                                                                        \
  asm volatile(".rept " STR(REPS) "\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n" \
               " v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n .endr"        \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(inc))

template <int KB>
__global__ void code_k(unsigned long long* out, float inc, int passes) {
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  unsigned long long t[5];
  __syncthreads();
  t[0] = __builtin_readcyclecounter();
#pragma unroll 1
  for (int p = 0; p < passes; p++) {
    if constexpr (KB == 8) BODY(256);
    else if constexpr (KB == 32) BODY(1024);
    else BODY(3072);        // 96 KB: more than the 64 KB instruction cache (a loop body cannot exceed the 128 KB branch reach)
    __syncthreads();
    if (p < 4) t[p + 1] = __builtin_readcyclecounter();
  }
  if (threadIdx.x == 0) for (int p = 0; p < 4 && p < passes; p++) out[blockIdx.x * 4 + p] = t[p + 1] - t[p];
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[0] = 1;
}

__global__ void evict_k(float* p, long n) {          // something else between launches (its own code, its own data)
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

template <int KB>
void run(int threads, unsigned long long* d, float* scratch) {
  std::vector<unsigned long long> h(256 * 4);
  for (int launch = 0; launch < 3; launch++) {
    evict_k<<<4096, 256>>>(scratch, 1 << 20);
    hipMemset(d, 0, 256 * 4 * 8);
    code_k<KB><<<256, threads>>>(d, 1.f, 3);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 256 * 4 * 8, hipMemcpyDeviceToHost);
    for (int p = 0; p < 3; p++) {
      std::vector<unsigned long long> v;
      for (int b = 0; b < 256; b++) v.push_back(h[b * 4 + p]);
      std::sort(v.begin(), v.end());
      const double ninst = KB * 1024.0 / 4;
      printf("  %3d KB code, %4d lanes, launch %d pass %d: median %8llu cycles (%.2f cyc/instr, %.1f cyc/KB)  min %llu max %llu\n", KB, threads, launch, p,
             v[128], v[128] / ninst, v[128] / (double)KB, v[0], v[255]);
    }
  }
}

int main() {
  unsigned long long* d;
  float* scratch;
  hipMalloc(&d, 256 * 4 * 8);
  hipMalloc(&scratch, 4 << 20);
  for (int threads : {64, 256, 1024}) {
    run<8>(threads, d, scratch);
    run<32>(threads, d, scratch);
    run<96>(threads, d, scratch);
  }
  return 0;
}
