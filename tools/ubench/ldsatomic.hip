// LDS atomic-add rates on gfx950 (cycles per wave-instruction, conflict-free addresses, N waves per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int mode, int iters, unsigned long long* out, unsigned long long* sink) {
  __shared__ unsigned long long tab[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned* t32 = reinterpret_cast<unsigned*>(tab);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int row = (it * 16 + u + w * 5) & 31;
      if (mode == 0) atomicAdd(&tab[row * 64 + lane], (unsigned long long)(it + lane));
      else if (mode == 1) atomicAdd(&t32[row * 64 + lane], (unsigned)(it + lane));
      else if (mode == 2) tab[row * 64 + lane] += (unsigned long long)(it + lane);   // plain read-modify-write (racy across waves; timing only)
      else { atomicAdd(&t32[row * 128 + lane], (unsigned)(it + lane)); atomicAdd(&t32[row * 128 + 64 + lane], (unsigned)(it >> 3)); }
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[w] = t1 - t0;
  sink[threadIdx.x] = tab[threadIdx.x];
}
int main() {
  unsigned long long *out, *sink;
  (void)hipMalloc(&out, 8 * 64); (void)hipMalloc(&sink, 8 * 1024);
  const char* names[4] = {"ds_add_u64", "ds_add_u32", "plain u64 rmw", "2 x ds_add_u32"};
  for (int threads : {64, 256, 512, 1024}) for (int mode = 0; mode < 4; mode++) {
    k<<<1, threads>>>(mode, 256, out, sink);
    k<<<1, threads>>>(mode, 256, out, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[16];
    (void)hipMemcpy(h, out, 8 * (threads / 64), hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (256.0 * 16.0);
    printf("%-16s waves/CU=%2d : %7.1f cycles per wave-instruction (wave 0) -> %6.2f cycles per instruction for the CU\n", names[mode], threads / 64, per,
           per / (threads / 64));
  }
  return 0;
}
