// Store-shape micro-benchmark (gfx950): how expensive are the R-row stores of the block-update kernel?
//   A: 28 x global_store_dword   per 16-row tile (lane (g,c): row 4g+reg, column 16ct+c)        -- current k_tile
//   B:  8 x global_store_dwordx4 per tile (lane (gg,cc): rows gg+4i, float4 columns cc, cc+16)  -- after an LDS transpose
//   C:  7 x global_store_dwordx4 per tile (lane (g,c): row c, float4 column 4ct+g)              -- transposed MFMA layout
// Rows are K=100 floats, 16 rows of a tile are `stride` rows apart (block update: ~20), tiles dealt round-robin to waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* R, long nrows, int K, int stride, int ntiles, int shape, unsigned long long* clk) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int tile = wave; tile < ntiles; tile += nw) {
    const long base = (long)tile * 16 * stride;
    const float v = (float)tile;
    if (shape == 0) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        float* row = R + ((base + (long)(4 * g + reg) * stride) % nrows) * K + c;
#pragma unroll
        for (int ct = 0; ct < 7; ct++) if (ct < 6 || 16 * ct + c < K) row[16 * ct] = v;
      }
    } else if (shape == 1) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float* row = R + ((base + (long)(g + 4 * i) * stride) % nrows) * K;
        const f32x4 q = {v, v, v, v};
        *reinterpret_cast<f32x4*>(row + 4 * c) = q;
        if (4 * (c + 16) < K) *reinterpret_cast<f32x4*>(row + 4 * (c + 16)) = q;
      }
    } else {
      float* row = R + ((base + (long)c * stride) % nrows) * K;
      const f32x4 q = {v, v, v, v};
#pragma unroll
      for (int ct = 0; ct < 7; ct++) if (16 * ct + 4 * g < K) *reinterpret_cast<f32x4*>(row + 16 * ct + 4 * g) = q;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) clk[wave] = t1 - t0;
}
int main(int argc, char** argv) {
  const long nrows = argc > 1 ? atol(argv[1]) : 10000000;
  const int K = 100, stride = 20, ntiles = (int)(nrows / stride / 16);
  float* R; unsigned long long* clk;
  if (hipMalloc(&R, sizeof(float) * nrows * K) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMalloc(&clk, 8 * 2048 * 4);
  (void)hipMemset(R, 0, sizeof(float) * nrows * K);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[3] = {"A 28 x dword  ", "B 8 x dwordx4 ", "C 7 x dwordx4T"};
  for (int blocks : {256, 512}) for (int shape = 0; shape < 3; shape++) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      (void)hipEventRecord(e0);
      k<<<blocks, 512>>>(R, nrows, K, stride, ntiles, shape, clk);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = (double)ntiles * 16 * K * 4;
    printf("%s blocks=%d: %8.1f us for %d tiles (%.0f MB) -> %6.2f TB/s, %.0f ns per tile per CU-wave-slot\n", names[shape], blocks,
           best * 1e3, ntiles, bytes / 1e6, bytes / (best * 1e-3) / 1e12, best * 1e6 / ntiles * (blocks * 8));
  }
  return 0;
}
