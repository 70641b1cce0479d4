// Micro-benchmarks of gfx950 issue behaviour that the k_tile design questions depend on (run on the GPU box):
//   dependent / independent VALU, transcendental, MFMA f32 16x16x4, MFMA+VALU interleaved in ONE wave,
//   and MFMA-wave + VALU-wave sharing a SIMD.  Output: cycles per instruction (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP16(x) x x x x x x x x x x x x x x x x
__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

// mode: 0 dep fma | 1 8x indep fma | 2 dep exp | 3 8x indep exp | 4 7 indep-acc mfma | 5 mfma + 8 indep fma interleaved
//       6 mfma + 4 fma | 7 mfma + 16 fma | 8 split: waves with (wl & mask) do fma, others mfma
__global__ void k(int mode, int mask, int iters, unsigned long long* out, float* sink) {
  const int wl = threadIdx.x >> 6;
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 0.999f;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0;
  const float ma = a0, mb = a1;
  int which = mode;
  if (mode == 8) which = (wl & mask) ? 1 : 4;
  __syncthreads();
  const unsigned long long t0 = now();
  for (int it = 0; it < iters; it++) {
    switch (which) {
      case 0: REP16(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(b));) break;
      case 1:
        REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                           "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        break;
      case 2: REP16(asm volatile("v_exp_f32 %0, %0" : "+v"(a0));) break;
      case 3:
        REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n"
                           "v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        break;
      case 4:
        REP16(asm volatile("v_mfma_f32_16x16x4_f32 %0, %7, %8, %0\n v_mfma_f32_16x16x4_f32 %1, %7, %8, %1\n v_mfma_f32_16x16x4_f32 %2, %7, %8, %2\n"
                           "v_mfma_f32_16x16x4_f32 %3, %7, %8, %3\n v_mfma_f32_16x16x4_f32 %4, %7, %8, %4\n v_mfma_f32_16x16x4_f32 %5, %7, %8, %5\n"
                           "v_mfma_f32_16x16x4_f32 %6, %7, %8, %6"
                           : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6) : "v"(ma), "v"(mb));)
        break;
      case 5:
        REP16(asm volatile("v_mfma_f32_16x16x4_f32 %0, %9, %10, %0\n"
                           "v_fma_f32 %1, %1, %11, %11\n v_fma_f32 %2, %2, %11, %11\n v_fma_f32 %3, %3, %11, %11\n v_fma_f32 %4, %4, %11, %11\n"
                           "v_fma_f32 %5, %5, %11, %11\n v_fma_f32 %6, %6, %11, %11\n v_fma_f32 %7, %7, %11, %11\n v_fma_f32 %8, %8, %11, %11"
                           : "+v"(c0), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ma), "v"(mb), "v"(b));)
        break;
      case 6:
        REP16(asm volatile("v_mfma_f32_16x16x4_f32 %0, %5, %6, %0\n"
                           "v_fma_f32 %1, %1, %7, %7\n v_fma_f32 %2, %2, %7, %7\n v_fma_f32 %3, %3, %7, %7\n v_fma_f32 %4, %4, %7, %7"
                           : "+v"(c0), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(ma), "v"(mb), "v"(b));)
        break;
      case 7:
        REP16(asm volatile("v_mfma_f32_16x16x4_f32 %0, %9, %10, %0\n"
                           "v_fma_f32 %1, %1, %11, %11\n v_fma_f32 %2, %2, %11, %11\n v_fma_f32 %3, %3, %11, %11\n v_fma_f32 %4, %4, %11, %11\n"
                           "v_fma_f32 %5, %5, %11, %11\n v_fma_f32 %6, %6, %11, %11\n v_fma_f32 %7, %7, %11, %11\n v_fma_f32 %8, %8, %11, %11\n"
                           "v_fma_f32 %1, %1, %11, %11\n v_fma_f32 %2, %2, %11, %11\n v_fma_f32 %3, %3, %11, %11\n v_fma_f32 %4, %4, %11, %11\n"
                           "v_fma_f32 %5, %5, %11, %11\n v_fma_f32 %6, %6, %11, %11\n v_fma_f32 %7, %7, %11, %11\n v_fma_f32 %8, %8, %11, %11"
                           : "+v"(c0), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ma), "v"(mb), "v"(b));)
        break;
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = now();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + wl] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2];
}

int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 8 * 64); hipMalloc(&sink, 4 * 1024);
  const int iters = 256;
  auto run = [&](const char* name, int mode, int mask, int threads, double per_iter_instrs) {
    k<<<1, threads>>>(mode, mask, iters, out, sink);
    k<<<1, threads>>>(mode, mask, iters, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(threads / 64);
    hipMemcpy(h.data(), out, 8 * h.size(), hipMemcpyDeviceToHost);
    printf("%-58s threads=%4d :", name, threads);
    for (auto v : h) printf(" %7.2f", (double)v / (iters * 16.0 * per_iter_instrs));
    printf("   [cycles per instruction, per wave]\n");
  };
  for (int threads : {64, 256, 512, 1024}) {
    run("dependent v_fma_f32", 0, 0, threads, 1);
    run("8 independent v_fma_f32", 1, 0, threads, 8);
    run("dependent v_exp_f32", 2, 0, threads, 1);
    run("8 independent v_exp_f32", 3, 0, threads, 8);
    run("7 independent-acc v_mfma_f32_16x16x4_f32", 4, 0, threads, 7);
    run("1 mfma + 4 fma  (per group of 5; /5)", 6, 0, threads, 5);
    run("1 mfma + 8 fma  (per group of 9; /9)", 5, 0, threads, 9);
    run("1 mfma + 16 fma (per group of 17; /17)", 7, 0, threads, 17);
  }
  // waves sharing a SIMD: some waves MFMA (7/iter), others VALU (8/iter) -- numbers are cycles per instruction of each wave's own kind
  run("split by wave bit 0: even=mfma(/7) odd=fma(/8) [printed /7]", 8, 1, 512, 7);
  run("split by wave bit 1", 8, 2, 512, 7);
  run("split by wave bit 2", 8, 4, 512, 7);
  run("split by wave bit 0, 256 threads", 8, 1, 256, 7);
  run("split by wave bit 1, 256 threads", 8, 2, 256, 7);
  return 0;
}
