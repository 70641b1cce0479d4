// hmx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the Harmony
// clustering + correction loop.  MI355X only: no CUDA paths, no portability layer.
//
// Data layout (DESIGN.md "HBM layout"): cells are stored cell-major, exactly the
// reference's column-major d x N / K x N matrices (src/harmony.h:50), but in an
// internal order sorted by covariate-level combination so that every streaming pass sees
// long runs of cells sharing one combination.  "Cluster-lane" mapping: lane l of a wave
// owns clusters l, l+64, ... (KPL per lane); a cell's PCs are broadcast lane->SGPR with
// v_readlane.  Cross-cell sums (O, E, objective, ridge statistics) are accumulated per
// lane over a run and flushed with one coalesced atomic per run: 64-bit fixed point for
// R sums (exact, order-independent => bit-reproducible and shard-count independent),
// fp64 for the rest.
#include "hmx_internal.h"
#include <float.h>
#include <hip/hip_ext.h>

#ifndef HMX_USE_DPP
#define HMX_USE_DPP 1
#endif
#ifndef HMX_CHAIN_PRE2
#define HMX_CHAIN_PRE2 1      // the chain runs the MFMAs of a wave's second tile of a block ahead of the flag too
#endif
#ifndef HMX_PAIR_PRE2
// the wave-pair chain (MODE 6) hoists ONE tile's MFMAs ahead of the flag: with the second accumulator set live across the flag as well the kernel needs 87
// spilled registers instead of 13 and a block step at K = 200 takes 44.8 instead of 42.6 us (profiles/r6_pair_chain_tuning.txt)
#define HMX_PAIR_PRE2(PAIR) (!(PAIR))
#endif
#ifndef HMX_CHAIN_BALANCE
#define HMX_CHAIN_BALANCE 1
#endif
#ifndef HMX_TILE_BF
#define HMX_TILE_BF 0         // 1 (hmx_tile_bf.hip): this translation unit builds ONLY k_tile, with the split-bf16 distance GEMM, and its three launchers
#endif
#ifndef HMX_TILE_LB
#define HMX_TILE_LB(NCT) 1   // waves/SIMD the tile kernel is register-budgeted for; 3 measured slower than unconstrained
#endif

namespace hmx {
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };     // 12-byte row segment (global_store_dwordx3)
struct __attribute__((aligned(8))) F2 { float x, y; };

// --------------------------------------------------------------------------------------
// device helpers
// --------------------------------------------------------------------------------------
__device__ __forceinline__ float rlane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ double wsumd(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ unsigned long long wmin64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), m, 64);
    unsigned hi = __shfl_xor((unsigned)(v >> 32), m, 64);
    unsigned long long o = ((unsigned long long)hi << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}
// UNCONDITIONAL load from a clamped (always valid) index, masked afterwards.  `cond ? p[i] : 0` makes hipcc branch
// around every load and wait vmcnt(0) at each join -- fully serialised memory latency (measured: 2-3x slower).
template <class T> __device__ __forceinline__ T ld_or(const T* __restrict__ p, size_t safe_idx, bool ok, T dflt) {
  const T v = p[safe_idx];
  return ok ? v : dflt;
}
__device__ __forceinline__ float trunc_logf_dev(float x) {  // arma::trunc_log (src/utils.cpp:78)
  return (x > 0.0f) ? logf(x) : logf(FLT_MIN);
}
// R in [0,1] -> fixed point: ONE function of the float for every kernel that adds or removes a cell's R (what a block update files is exactly what
// the next round takes out).  fma + truncating convert: two instructions (round half up; from 2^23 on r * 2^29 is an integer already).
__device__ __forceinline__ unsigned fx32_of(float r) { return (unsigned)__builtin_fmaf(r, FX_SCALE, 0.5f); }
__device__ __forceinline__ unsigned long long fx_of(float r) { return (unsigned long long)fx32_of(r); }

// diversity penalty ((2E+1)/(O+E+1))^theta (src/harmony.cpp:319-321) with a SHORT dependent chain: rcp, mul, log2, mul, exp2
// (~2e-7 relative; powf's ~100 dependent instructions cost >1 us in the serial prologue of every block step at gfx950's
// 26-cycle dependent-issue latency).  ONE definition for the fused and the stand-alone fold kernels: the sharded and the
// single-GPU paths must produce bit-identical penalty tables.
// one int64 of a K x B table into every peer's inbox: two self-validating granules {tag, half}, written through at system scope
// (par = plane * 8: the plane's first source slot)
__device__ __forceinline__ void p2p_send(const Dev& D, size_t par, int i, unsigned tag, long long v) {
  const unsigned long long tb = (unsigned long long)tag << 32;
  const unsigned long long lo = tb | ((unsigned long long)v & 0xffffffffull), hi = tb | ((unsigned long long)v >> 32);
#pragma unroll
  for (int gq = 0; gq < 8; gq++) if (gq < D.p2p_world && gq != D.p2p_rank) {
    unsigned long long* dst = D.p2p_inbox[gq] + ((par + D.p2p_rank) * P2P_CAP + i) * 2;
    __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__device__ __forceinline__ float pen_pow(float num, float den, float theta) {
  const float x = num * __builtin_amdgcn_rcpf(den);
  return __builtin_amdgcn_exp2f(theta * __builtin_amdgcn_logf(x));
}

// counter-based generators -- same SPEC as include/harmony_mi355x.h documents
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
struct FeistelKeys { uint32_t k[6]; int half; uint32_t mask; };
__host__ __device__ __forceinline__ uint64_t feistel_apply(const FeistelKeys& fk, uint64_t N, uint64_t g) {
  uint64_t x = g;
  do {
    uint32_t L = (uint32_t)(x >> fk.half), Rr = (uint32_t)(x & fk.mask);
#pragma unroll
    for (int r = 0; r < 6; r++) {
      uint32_t t = L ^ (fmix32(Rr * 0x9E3779B1u + fk.k[r]) & fk.mask);
      L = Rr; Rr = t;
    }
    x = ((uint64_t)L << fk.half) | Rr;
  } while (x >= N);
  return x;
}
static FeistelKeys make_keys(uint64_t seed, uint64_t round, uint64_t N) {
  FeistelKeys fk;
  int bits = 2;
  while (((uint64_t)1 << bits) < N) bits += 2;
  fk.half = bits / 2;
  fk.mask = (uint32_t)(((uint64_t)1 << fk.half) - 1);
  for (int r = 0; r < 6; r++)
    fk.k[r] = (uint32_t)(splitmix64(seed ^ (round * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(r + 1) << 56)) >> 32);
  return fk;
}

// Stage the centroid table Yt[d][K] into LDS as [d][KP] (zero padded), k fastest:
// lane l then reads ldsY[j*KP + l + 64q] -- consecutive dwords, conflict-free.
__device__ __forceinline__ void stage_Y(float* ldsY, const float* __restrict__ Yt, int d, int K, int KP) {
  for (int i = threadIdx.x; i < d * KP; i += blockDim.x) {
    int j = i / KP, k = i - j * KP;
    ldsY[i] = ld_or(Yt, (size_t)j * K + min(k, K - 1), k < K, 0.0f);
  }
  __syncthreads();
}

// dots of CB cells (rows broadcast from lanes) with all clusters of this lane.
template <int KPL, int DPL, int CB>
__device__ __forceinline__ void group_dots(const float* __restrict__ ldsY, int d, int KP, int lane,
                                           const float (&z)[CB][DPL], float (&acc)[CB][KPL]) {
#pragma unroll
  for (int c = 0; c < CB; c++)
#pragma unroll
    for (int q = 0; q < KPL; q++) acc[c][q] = 0.0f;
  const int d0 = d < 64 ? d : 64;
  for (int j = 0; j < d0; ++j) {
    float y[KPL];
#pragma unroll
    for (int q = 0; q < KPL; q++) y[q] = ldsY[j * KP + lane + 64 * q];
#pragma unroll
    for (int c = 0; c < CB; c++) {
      const float zj = rlane(z[c][0], j);
#pragma unroll
      for (int q = 0; q < KPL; q++) acc[c][q] = fmaf(zj, y[q], acc[c][q]);
    }
  }
  if constexpr (DPL > 1) {
    for (int j = 64; j < d; ++j) {
      float y[KPL];
#pragma unroll
      for (int q = 0; q < KPL; q++) y[q] = ldsY[j * KP + lane + 64 * q];
#pragma unroll
      for (int c = 0; c < CB; c++) {
        const float zj = rlane(z[c][DPL - 1], j - 64);
#pragma unroll
        for (int q = 0; q < KPL; q++) acc[c][q] = fmaf(zj, y[q], acc[c][q]);
      }
    }
  }
}

template <int DPL>
__device__ __forceinline__ void load_row(const float* __restrict__ Z, size_t cell, int zs, int d, int lane, float (&z)[DPL]) {
  z[0] = ld_or(Z, cell * zs + min(lane, d - 1), lane < d, 0.0f);
  if constexpr (DPL > 1) z[DPL - 1] = ld_or(Z, cell * zs + min(64 + lane, d - 1), 64 + lane < d, 0.0f);
}

// flush a lane-private fixed-point run sum into a [B][K] table, once per covariate level
template <int KPL>
__device__ __forceinline__ void flush_fx(long long* __restrict__ tab, const int* __restrict__ qlev, int q, int C,
                                         int K, int lane, unsigned long long (&oacc)[KPL]) {
  for (int c = 0; c < C; c++) {
    const int b = qlev[q * C + c];
#pragma unroll
    for (int qq = 0; qq < KPL; qq++) {
      const int k = lane + 64 * qq;
      if (k < K && oacc[qq]) atomicAdd((unsigned long long*)&tab[(size_t)b * K + k], oacc[qq]);
    }
  }
#pragma unroll
  for (int qq = 0; qq < KPL; qq++) oacc[qq] = 0ull;
}

// --------------------------------------------------------------------------------------
// ingest / egress
// The kernels live in four include files (one translation unit: they share the helpers above and each other's device functions):
//   hmx_k_stream.inc   ingest / egress, shuffle and sort, old-contribution passes, fold / penalty, first-generation kernels
//   hmx_k_tile.inc     the MFMA tile machinery and k_tile (block update, head, Lloyd, seeding, persistent chain) -- also built by hmx_tile_bf.hip
//   hmx_k_correct.inc  objective tables / terms, MoE ridge correction (statistics, solve, apply), VALU fallbacks
//   hmx_k_launch.inc   launchers, peer-inbox self-test and all-reduces -- also built by hmx_tile_bf.hip
#include "hmx_k_stream.inc"
#include "hmx_k_tile.inc"
#include "hmx_k_correct.inc"
#include "hmx_k_launch.inc"
}  // namespace hmx
