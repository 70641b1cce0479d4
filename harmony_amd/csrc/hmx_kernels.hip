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
// --------------------------------------------------------------------------------------
// src: [n][d] doubles or floats in local original order (host slab staged in HBM, or the caller's device buffer)
// -> dst: [n][zs] floats in internal order
#if !HMX_TILE_BF   // (the split-bf16 translation unit builds k_tile and its launchers only)
template <class T>
__global__ void k_convert_in(const T* __restrict__ src, float* __restrict__ dst, const int* __restrict__ invperm,
                             int n, int d, int zs) {
  const size_t total = (size_t)n * d;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t cell = i / d; const int j = (int)(i - cell * d);
    dst[(size_t)invperm[cell] * zs + j] = (float)src[i];
  }
}
// egress of a slab of cells in ORIGINAL order: dst[i][0..w) = src[invperm[i]][0..w) (rows gathered, output written
// contiguously -> the slab can be copied to the host while the next one is converted).  T = double (the R seam) or float.
template <class T>
__global__ void k_convert_out(const float* __restrict__ src, T* __restrict__ dst, const int* __restrict__ invperm,
                              int n, int w, int ws) {
  const size_t total = (size_t)n * w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t cell = i / w; const int j = (int)(i - cell * w);
    dst[i] = (T)src[(size_t)invperm[cell] * ws + j];
  }
}
__global__ void k_copy(const float* __restrict__ src, float* __restrict__ dst, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
// arma::normalise(Z, 2, 0) (src/harmony.cpp:42,220): one wave per cell
__global__ __launch_bounds__(TPB) void k_normalize(float* __restrict__ Z, int n, int d, int zs) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int cell = wave; cell < n; cell += nw) {
    float* z = Z + (size_t)cell * zs;
    const float a = ld_or(z, (size_t)min(lane, d - 1), lane < d, 0.0f), b = ld_or(z, (size_t)min(64 + lane, d - 1), 64 + lane < d, 0.0f);
    float nrm = sqrtf(wsum(a * a + b * b));
    if (nrm == 0.0f) nrm = 1.0f;
    if (lane < d) z[lane] = a / nrm;
    if (64 + lane < d) z[64 + lane] = b / nrm;
  }
}

// --------------------------------------------------------------------------------------
// E-step head: dist = 2(1 - Y^T Z), R = softmax_k(-dist/sigma), O, objective partials
// (src/harmony.cpp:141-150 and :221-227).  One wave per static work item (a run of cells
// with one combination).  MODE 0: write R and accumulate O_fx.  MODE 1: objective only,
// R is read (harmony::compute_objective on the current state).
// --------------------------------------------------------------------------------------
template <int KPL, int DPL, int MODE>
__global__ __launch_bounds__(TPB) void k_head(Dev D) {
  extern __shared__ __attribute__((aligned(16))) float ldsY[];
  stage_Y(ldsY, D.Yt, D.d, D.K, D.KP);
  constexpr int CB = 4;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const int d = D.d, K = D.K;
  float sig[KPL];
#pragma unroll
  for (int q = 0; q < KPL; q++) sig[q] = ld_or(D.sigma, (size_t)min(lane + 64 * q, K - 1), lane + 64 * q < K, 1.0f);
  double od = 0.0, oe = 0.0;
  for (int it = wave; it < D.nitems; it += nw) {
    const Item item = D.items[it];
    unsigned long long oacc[KPL];
#pragma unroll
    for (int q = 0; q < KPL; q++) oacc[q] = 0ull;
    for (int p = 0; p < item.cnt; p += CB) {
      const int nc = min(CB, item.cnt - p);
      float z[CB][DPL];
#pragma unroll
      for (int c = 0; c < CB; c++) {
        if (c < nc) load_row<DPL>(D.Zc, (size_t)(item.start + p + c), D.zs, d, lane, z[c]);
        else {
#pragma unroll
          for (int t = 0; t < DPL; t++) z[c][t] = 0.0f;
        }
      }
      float acc[CB][KPL];
      group_dots<KPL, DPL, CB>(ldsY, d, D.KP, lane, z, acc);
#pragma unroll
      for (int c = 0; c < CB; c++) {
        if (c < nc) {
          const size_t cell = (size_t)(item.start + p + c);
          float r[KPL], dist[KPL];
          if (MODE == 0) {
            float s = 0.0f;
#pragma unroll
            for (int q = 0; q < KPL; q++) {
              const int k = lane + 64 * q;
              dist[q] = 2.0f * (1.0f - acc[c][q]);
              r[q] = (k < K) ? expf(-dist[q] / sig[q]) : 0.0f;
              s += r[q];
            }
            s = wsum(s);
#pragma unroll
            for (int q = 0; q < KPL; q++) {
              const int k = lane + 64 * q;
              r[q] = r[q] / s;
              if (k < K) { D.R[cell * K + k] = r[q]; oacc[q] += fx_of(r[q]); }
            }
          } else {
#pragma unroll
            for (int q = 0; q < KPL; q++) {
              const int k = lane + 64 * q;
              dist[q] = 2.0f * (1.0f - acc[c][q]);
              r[q] = ld_or(D.R, cell * K + min(k, K - 1), k < K, 0.0f);
            }
          }
#pragma unroll
          for (int q = 0; q < KPL; q++) {
            if (lane + 64 * q < K) {
              od += (double)(r[q] * dist[q]);
              oe += (double)((r[q] * trunc_logf_dev(r[q])) * sig[q]);
            }
          }
        }
      }
    }
    if (MODE == 0) flush_fx<KPL>(D.O_fx, D.qlev, item.q, D.C, K, lane, oacc);
  }
  od = wsumd(od); oe = wsumd(oe);
  if (lane == 0) { D.objpart[2 * wave] += od; D.objpart[2 * wave + 1] += oe; }  // private slot (slot row 0): no atomics
}

// --------------------------------------------------------------------------------------
// per-round block membership + stable counting sort by block
// --------------------------------------------------------------------------------------
// Stable counting sort of the cells by block, with every (block, combination) bin padded to a multiple of
// 16 positions (dummy entries = -1) so that a 16-cell MFMA tile never straddles two combinations.
// Sort chunks are static runs of <= SORT_CHUNK cells of ONE combination (D.schunks, built at setup), so the
// per-chunk histogram counts[blk][chunk] also yields the per-(block, combination) bin sizes.
// one wave per sort chunk.  FUSED: the block id is computed here from the Feistel bijection (no separate block-id kernel: saves a
// launch and a write + read of blk); the histogram is one ds_add_u32 per 64 cells instead of a ballot loop over the
// distinct block values.
// D.nxt (own shuffle only): the sort key of a cell is (block of this round, block of the NEXT round) -- nV = nb * nb keys --, so that
// every 16-cell tile also has ONE next block and the tile kernels can file the tile's new R sums as that block's old contribution
// (flush_tile_fx): 16 padding slots per (block, combination, next block) instead of per (block, combination).
struct BlockIdArgs { FeistelKeys fk, fk2; uint64_t Nglob, goff, cpb; float inv_cpb; };
struct BlockIdBatch { BlockIdArgs a[4]; };      // one entry per round of a batched sort (blockIdx.y)
// block of a position: min(pos / cells_per_block, n_blocks - 1) (src/harmony.cpp:296-300) without the 64-bit division (~100 instructions
// per cell, twice per cell and round): a float estimate, corrected exactly by two integer comparisons
__device__ __forceinline__ int block_of(uint64_t pos, const BlockIdArgs& A, int nb) {
  long long b = (long long)((float)pos * A.inv_cpb);
  if (b > nb) b = nb;                                        // (keeps the products below in range)
  if ((uint64_t)b * A.cpb > pos) b--;
  else if ((uint64_t)(b + 1) * A.cpb <= pos) b++;
  return (int)(b < (long long)(nb - 1) ? b : (long long)(nb - 1));
}
template <bool FUSED>
__global__ __launch_bounds__(WAVE) void k_sort_hist(Dev D, BlockIdBatch AB, SortBatch S) {
  extern __shared__ int cnt[];
  const BlockIdArgs& A = AB.a[blockIdx.y]; const SortPtrs& P = S.p[blockIdx.y];      // (blockIdx.y: the round of a batched sort)
  const int lane = threadIdx.x, chunk = blockIdx.x, nb = D.nb, nV = D.nxt ? nb * nb : nb;
  for (int v = lane; v < nV; v += WAVE) cnt[v] = 0;
  __syncthreads();
  const Item ch = D.schunks[chunk];
  const int s = ch.start, e = ch.start + ch.cnt;
  constexpr int NSTEP = SORT_CHUNK / WAVE;
  int pm[NSTEP];      // the chunk's cell ids, all loads in flight together (one exposed latency per chunk instead of one per step)
#pragma unroll
  for (int u = 0; u < NSTEP; u++) pm[u] = FUSED ? D.perm[min(s + u * WAVE + lane, e - 1)] : 0;
#pragma unroll
  for (int u = 0; u < NSTEP; u++) {
    const int base = s + u * WAVE;
    if (base >= e) break;
    const int i = base + lane;
    int b = -1;
    if (i < e) {
      if constexpr (FUSED) {
        const uint64_t pos = feistel_apply(A.fk, A.Nglob, A.goff + (uint64_t)pm[u]);
        b = block_of(pos, A, nb);
        P.blk[i] = b;
        if (D.nxt) {
          const uint64_t pos2 = feistel_apply(A.fk2, A.Nglob, A.goff + (uint64_t)pm[u]);
          b = b * nb + block_of(pos2, A, nb);
          P.blkv[i] = b;
        }
      } else b = P.blk[i];
      atomicAdd(&cnt[b], 1);
    }
  }
  __syncthreads();
  for (int v = lane; v < nV; v += WAVE) P.counts[(size_t)v * D.nchunks + chunk] = cnt[v];
}
// one wave per (block, combination) bin: exclusive prefix of the bin's chunk counts -> offs (offset inside the bin),
// padded bin size -> binoff[bin].  The chunks of a combination are contiguous, so the loads are coalesced.
__global__ __launch_bounds__(WAVE) void k_sort_binscan(Dev D, SortBatch S) {
  const SortPtrs& P = S.p[blockIdx.y];
  const int lane = threadIdx.x, bin = blockIdx.x, Q = D.Q, nch = D.nchunks, nV = D.nxt ? D.nb * D.nb : D.nb;
  const int v = bin / Q, q = bin - v * Q;
  const int lo = D.qchunk[q], hi = D.qchunk[q + 1];
  const int* __restrict__ cin = P.counts + (size_t)v * nch;
  int* __restrict__ cout = P.offs + v;      // offs[chunk][key]: the scatter kernel reads a chunk's nV offsets as one contiguous run
  int run = 0;
  for (int base = lo; base < hi; base += WAVE) {
    const int i = base + lane;
    const int c = ld_or(cin, (size_t)min(i, hi - 1), i < hi, 0);
    int incl = c;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_up(incl, m, 64); if (lane >= m) incl += o; }
    if (i < hi) cout[(size_t)i * nV] = run + incl - c;
    run += __shfl(incl, 63, 64);
  }
  if (lane == 0) { P.binoff[bin] = (run + 15) & ~15; P.bincnt[bin] = run; }
}
// single workgroup: exclusive scan of the padded bin sizes (block-major) -> binoff; boff[v] = padded start of block v
__global__ __launch_bounds__(1024) void k_sort_binoff(Dev D, SortBatch S) {
  const SortPtrs& P = S.p[blockIdx.x];
  __shared__ int part[1024];
  const int t = threadIdx.x, nb = D.nb, Q = D.Q, vpb = D.nxt ? nb : 1, nbins = nb * vpb * Q;   // vpb: sort keys per block
  int* bins = P.binoff;
  const int per = (nbins + 1023) / 1024;
  const int s = t * per, e = min(nbins, s + per);
  int sum = 0;
  for (int i = s; i < e; i++) sum += bins[i];
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;
  for (int i = s; i < e; i++) { const int c = bins[i]; bins[i] = run; run += c; }
  if (t == 1023) bins[nbins] = part[1023];
  __syncthreads();
  for (int v = t; v <= nb; v += 1024) P.boff[v] = bins[v < nb ? v * vpb * Q : nbins];
}
// Placement of a chunk's cells into their (key, combination) bins, 64 cells per step; the chunk's first slot in every bin comes from
// k_sort_binscan / k_sort_binoff, the rank inside the chunk from an LDS atomic (see below).
__global__ __launch_bounds__(WAVE) void k_sort_scatter(Dev D, SortBatch S) {
  extern __shared__ int base_[];
  const SortPtrs& P = S.p[blockIdx.y];
  const int lane = threadIdx.x, chunk = blockIdx.x, nb = D.nb, nV = D.nxt ? nb * nb : nb;
  const Item ch = D.schunks[chunk];
  const int* __restrict__ key = D.nxt ? P.blkv : P.blk;
  const int s = ch.start, e = ch.start + ch.cnt;
  constexpr int NSTEP = SORT_CHUNK / WAVE;
  for (int v0 = 0; v0 < nV; v0 += 8 * WAVE) {     // (eight independent pairs of loads in flight per pass)
    int t0[8], t1[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int v = min(v0 + u * WAVE + lane, nV - 1);
      t0[u] = P.binoff[v * D.Q + ch.q]; t1[u] = P.offs[(size_t)chunk * nV + v];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) if (v0 + u * WAVE + lane < nV) base_[v0 + u * WAVE + lane] = t0[u] + t1[u];
  }
  __syncthreads();
  int kk[NSTEP];      // the chunk's keys, all loads in flight together
#pragma unroll
  for (int u = 0; u < NSTEP; u++) kk[u] = ld_or(key, (size_t)min(s + u * WAVE + lane, e - 1), s + u * WAVE + lane < e, 0x00FFFFFF);   // (past the end: sorts behind every key)
#pragma unroll
  for (int u = 0; u < NSTEP; u++) {
    const int base = s + u * WAVE;
    if (base >= e) break;
    // rank of a cell among the chunk's cells of the same key = the value its LDS atomic returns: the 64 lanes of one ds_add_rtn are
    // served one after the other, the chunk's steps run in order on this one wave -- every cell gets a distinct slot of its bin, and
    // which slot is irrelevant (a bin only has to be PURE: same block, combination and next block).  (The first version ranked the
    // cells with a 21-stage bitonic sort of (key, lane) per step to keep the sort stable: 44 us per round, all of it on the critical
    // path between two chain launches.)
    const int ks = kk[u];
    if (base + lane < e) {
      const int dst = atomicAdd(&base_[ks], 1), cell = base + lane;
      if (D.need_lorder) { P.lorder[dst] = cell; P.lcombo[dst] = ch.q; }
      P.lpair[dst] = make_int2(cell, D.nxt ? (ch.q | ((ks % nb) << 19) | ((ks / nb) << 25)) : ch.q);     // (combination, next block, block): see flush_run in k_tile
    }
  }
  // the padding slots (< 16 per bin) of this combination's bins: "no cell" -- written here, so that no memset of the whole order
  // precedes every shuffle
  // (the keys are dealt over the combination's chunks: a couple of bins per wave)
  const int ci = chunk - D.qchunk[ch.q], ncq = D.qchunk[ch.q + 1] - D.qchunk[ch.q];
  for (int v = ci + ncq * lane; v < nV; v += ncq * WAVE) {
    const int bin = v * D.Q + ch.q, st = P.binoff[bin], cnt = P.bincnt[bin], pad = (cnt + 15) & ~15;
    for (int k = cnt; k < pad; k++) { if (D.need_lorder) P.lorder[st + k] = -1; P.lpair[st + k] = make_int2(-1, -1); }
  }
}

// --------------------------------------------------------------------------------------
// The same padded order WITHOUT sorting the cells by block (round 4).  The round's shuffle is a bijection of positions with a closed-form
// inverse, so the cells of block b are simply the inverse images of the positions [b * cpb, (b + 1) * cpb): the cells arrive grouped by
// block for free, and only the (next block, combination) bins INSIDE a block are left to count.  Four wide launches serve up to four rounds
// (blockIdx.y = round of the batch; k_shuf_blocks -- block id per cell, D.blk -- runs only for a round whose old contributions were not
// carried and must be summed from R); the unit of work is a PART of a block (SHUF_PART consecutive positions, one 1024-thread workgroup):
//   k_shuf_count   per (round, part): cell = inverse image of the position (-1: another rank's), its next block = block of its image under the
//                  NEXT round's bijection, its rank inside (part, bin) from a returning LDS atomic -> posr[round][position] = (cell, rank |
//                  combination | next block); the part's offset inside every bin from a returning atomic on the bin's size
//   k_shuf_scan    per round, one workgroup: bins padded to 16 -> first slot of every bin, padded block offsets
//   k_shuf_place   per (round, part): lpair[first slot of the bin + part offset + rank] = (cell, keys); part 0 of a block writes the padding
// Against the counting sort above (four dependent launches of ONE-WAVE workgroups over an (nb^2 keys) x (N / 512 chunks) count matrix: 150 us
// per four rounds at 1M cells) the count matrix is (nb * Q bins) x (N / SHUF_PART parts).  The counting sort stays for host-provided
// orders and sharded runs.
// --------------------------------------------------------------------------------------
constexpr int SHUF_PART = 4096;
__host__ __device__ __forceinline__ uint64_t feistel_invert(const FeistelKeys& fk, uint64_t N, uint64_t pos) {
  uint64_t x = pos;
  do {
    uint32_t L = (uint32_t)(x >> fk.half), Rr = (uint32_t)(x & fk.mask);
#pragma unroll
    for (int r = 5; r >= 0; r--) {
      const uint32_t t = Rr ^ (fmix32(L * 0x9E3779B1u + fk.k[r]) & fk.mask);
      Rr = L; L = t;
    }
    x = ((uint64_t)L << fk.half) | Rr;
  } while (x >= N);
  return x;
}
struct ShufBatch {
  FeistelKeys fk[5];
  int2* posr[4];                // [position] (internal cell | -1, rank inside its (part, bin))
  int2* lpair[4]; int* lorder[4]; int* lcombo[4]; int* boff[4];
  int* partcnt[4];              // [block][part][bin] the part's offset inside the bin
  int* binbase[4]; int* bincnt[4]; int* binacc[4];     // [block][bin] first slot / cells / cells, accumulated by the parts (zero between batches)
  uint64_t Nglob, goff, cpb; float inv_cpb; int nr, P;
};
__global__ __launch_bounds__(256) void k_shuf_blocks(Dev D, BlockIdArgs A) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= D.n) return;
  D.blk[i] = block_of(feistel_apply(A.fk, A.Nglob, A.goff + (uint64_t)D.perm[i]), A, D.nb);
}
// positions [s, e) of (block b, part p)
__device__ __forceinline__ void shuf_range(const ShufBatch& S, int nb, int b, int p, uint64_t& s, uint64_t& e) {
  const uint64_t p0 = min((uint64_t)b * S.cpb, S.Nglob), p1 = (b == nb - 1) ? S.Nglob : min((uint64_t)(b + 1) * S.cpb, S.Nglob);
  s = min(p0 + (uint64_t)p * SHUF_PART, p1); e = min(s + SHUF_PART, p1);
}
// One pass of the six-round network over x: forward with keys k[0..5], or -- swap the halves, keys in reverse order, swap back -- its inverse.
__device__ __forceinline__ uint64_t feistel_pass(const FeistelKeys& kf, const FeistelKeys& ki, const bool inv, const uint64_t x) {
  const int half = kf.half; const uint32_t mask = kf.mask;          // (same domain: the two key sets differ in the keys only)
  uint32_t hi = (uint32_t)(x >> half), lo = (uint32_t)(x & mask);
  uint32_t L = inv ? lo : hi, Rr = inv ? hi : lo;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const uint32_t kr = inv ? ki.k[5 - r] : kf.k[r];
    const uint32_t t = L ^ (fmix32(Rr * 0x9E3779B1u + kr) & mask);
    L = Rr; Rr = t;
  }
  hi = inv ? Rr : L; lo = inv ? L : Rr;
  return ((uint64_t)hi << half) | lo;
}
constexpr int SHUF_THREADS = 256, SHUF_U = SHUF_PART / SHUF_THREADS;
// The bijection walks cycles: an image outside [0, N) is mapped again (up to 3/4 of all images when N is just above a power of four).  With
// one position per lane and pass, a wave repeats a pass until its UNLUCKIEST lane is inside -- 13 passes instead of 3.3 at 1.25M cells.  Here
// every lane works through its 16 positions as a little state machine (position -> cell: inverse passes; cell -> next block: forward
// passes of the next round's keys) and starts its next position the moment one is done: a wave then runs for the lane with the largest SUM of
// passes, which is close to the mean.  Results go to LDS; cell lookup, bin count and rank follow in a second, unrolled phase.
__global__ __launch_bounds__(SHUF_THREADS) void k_shuf_count(Dev D, ShufBatch S) {
  extern __shared__ int sm_[];
  const int r = blockIdx.y, b = blockIdx.x / S.P, p = blockIdx.x - b * S.P, tid = threadIdx.x, nb = D.nb, Q = D.Q;
  const int nbin = (D.nxt ? nb : 1) * Q;
  int* const cnt = sm_; int* const qf = sm_ + nbin;
  unsigned* const gbuf = reinterpret_cast<unsigned*>(qf + (Q + 1)); unsigned char* const nbuf = reinterpret_cast<unsigned char*>(gbuf + SHUF_PART);
  for (int v = tid; v < nbin; v += SHUF_THREADS) cnt[v] = 0;
  for (int v = tid; v <= Q; v += SHUF_THREADS) qf[v] = D.qstart[v];
  uint64_t s, e; shuf_range(S, nb, b, p, s, e);
  int2* __restrict__ const pr = S.posr[r];
  const bool nxt = D.nxt != 0;
  BlockIdArgs A; A.cpb = S.cpb; A.inv_cpb = S.inv_cpb;
  const FeistelKeys ki = S.fk[r], kf = S.fk[r + 1];
  const int nitem = (s + tid < e) ? min(SHUF_U, (int)((e - s - tid + SHUF_THREADS - 1) / SHUF_THREADS)) : 0;
  {
    int u = 0; bool inv = true; unsigned gcur = 0;
    uint64_t x = s + tid;
    while (u < nitem) {
      const uint64_t y = feistel_pass(kf, ki, inv, x);
      if (y >= S.Nglob) { x = y; continue; }
      if (inv) {
        const bool local = y >= S.goff && y < S.goff + (uint64_t)D.n;
        if (local && nxt) { gcur = (unsigned)(y - S.goff); inv = false; x = y; continue; }
        gbuf[u * SHUF_THREADS + tid] = local ? (unsigned)(y - S.goff) : 0xFFFFFFFFu; nbuf[u * SHUF_THREADS + tid] = 0;
      } else {
        gbuf[u * SHUF_THREADS + tid] = gcur; nbuf[u * SHUF_THREADS + tid] = (unsigned char)block_of(y, A, nb);
        inv = true;
      }
      u++; x = s + (uint64_t)u * SHUF_THREADS + tid;
    }
  }
  __syncthreads();                                  // (cnt / qf initialised; a lane reads back only what it wrote itself)
  int ci[SHUF_U];
#pragma unroll
  for (int u = 0; u < SHUF_U; u++) {
    const unsigned g = u < nitem ? gbuf[u * SHUF_THREADS + tid] : 0xFFFFFFFFu;
    ci[u] = ld_or(D.invperm, (size_t)(g != 0xFFFFFFFFu ? g : 0u), g != 0xFFFFFFFFu, -1);
  }
#pragma unroll
  for (int u = 0; u < SHUF_U; u++) if (u < nitem) {
    int rank = 0;
    if (ci[u] >= 0) {
      const int nbk = nbuf[u * SHUF_THREADS + tid];
      int lo = 0, hi = Q;                         // qf[q] <= cell < qf[q + 1]
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (qf[mid] <= ci[u]) lo = mid; else hi = mid; }
      rank = atomicAdd(&cnt[nbk * Q + lo], 1) | (lo << 12) | (nbk << 23);     // (a part has 4096 positions: the rank fits 12 bits; combination (11) and next block (6) ride along)
    }
    pr[s + (uint64_t)u * SHUF_THREADS + tid] = make_int2(ci[u], rank);
  }
  __syncthreads();
  // the part's offset inside every bin of its block: whatever a returning atomic on the bin's size hands out (a bin has to be pure, the
  // order of the parts inside it is free); k_shuf_scan reads the sizes and leaves them zero for the next batch into this order set
  for (int v = tid; v < nbin; v += SHUF_THREADS) {
    const int c = cnt[v];
    S.partcnt[r][((size_t)b * S.P + p) * nbin + v] = c ? atomicAdd(&S.binacc[r][b * nbin + v], c) : 0;
  }
}
__global__ __launch_bounds__(1024) void k_shuf_scan(Dev D, ShufBatch S) {
  __shared__ int red[16];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nb = D.nb, P = S.P;
  const int nbin = (D.nxt ? nb : 1) * D.Q, nbt = nb * nbin;
  const int per = (nbt + 1023) >> 10;
  int own = 0;
  for (int k = 0; k < per; k++) {
    const int t = min(tid * per + k, nbt - 1);
    const int c = S.binacc[r][t];
    if (tid * per + k < nbt) { S.binacc[r][t] = 0; S.bincnt[r][t] = c; own += (c + 15) & ~15; }
  }
  int incl = own;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
  if (lane == 63) red[w] = incl;
  __syncthreads();
  int wbase = 0, total = 0;
  for (int x = 0; x < 16; x++) { const int t = red[x]; if (x < w) wbase += t; total += t; }
  int run = wbase + incl - own;
  for (int k = 0; k < per; k++) {
    const int t = tid * per + k;
    if (t < nbt) {
      S.binbase[r][t] = run;
      if (t % nbin == 0) S.boff[r][t / nbin] = run;
      run += (S.bincnt[r][t] + 15) & ~15;
    }
  }
  if (tid == 0) S.boff[r][nb] = total;
}
__global__ __launch_bounds__(1024) void k_shuf_place(Dev D, ShufBatch S) {
  extern __shared__ int sm_[];
  const int r = blockIdx.y, b = blockIdx.x / S.P, p = blockIdx.x - b * S.P, tid = threadIdx.x, nb = D.nb, Q = D.Q;
  const int nbin = (D.nxt ? nb : 1) * Q;
  int* const base = sm_;
  for (int v = tid; v < nbin; v += 1024) base[v] = S.binbase[r][b * nbin + v] + S.partcnt[r][((size_t)b * S.P + p) * nbin + v];
  __syncthreads();
  uint64_t s, e; shuf_range(S, nb, b, p, s, e);
  const int2* __restrict__ const pr = S.posr[r];
  const bool nxt = D.nxt != 0;
  int2* __restrict__ const lp = S.lpair[r];
  int* __restrict__ const lo_ = S.lorder[r]; int* __restrict__ const lc_ = S.lcombo[r];
  constexpr int U = SHUF_PART / 1024;
  int2 cr[U];
#pragma unroll
  for (int u = 0; u < U; u++) { const uint64_t pos = s + (uint64_t)u * 1024 + tid; cr[u] = pos < e ? pr[pos] : make_int2(-1, 0); }
#pragma unroll
  for (int u = 0; u < U; u++) if (cr[u].x >= 0) {
    const int q = (cr[u].y >> 12) & 0x7FF, nbk = cr[u].y >> 23, dst = base[nbk * Q + q] + (cr[u].y & 0xFFF);
    if (D.need_lorder) { lo_[dst] = cr[u].x; lc_[dst] = q; }
    lp[dst] = make_int2(cr[u].x, nxt ? (q | (nbk << 19) | (b << 25)) : q);        // (combination, next block, block): see flush_run in k_tile
  }
  if (p == 0)                                    // the padding slots of the block's bins: "no cell"
    for (int v = tid; v < nbin; v += 1024) {
      const int c = S.bincnt[r][b * nbin + v], st = S.binbase[r][b * nbin + v], pad = (c + 15) & ~15;
      for (int k = c; k < pad; k++) { if (D.need_lorder) lo_[st + k] = -1; lp[st + k] = make_int2(-1, -1); }
    }
}

// --------------------------------------------------------------------------------------
// update_R (src/harmony.cpp:269-342) split into:
//   k_oldsum   one pass: old contribution of EVERY block of this round (:312-313 for all blocks)
//   k_prepare  tiny: O <- O + new(prev block) - old(this block); penalty table (:322)
//   k_update   the block's cells: R <- normalise(exp(-dist/sigma)); R *= penalty; normalise;
//              accumulate the new contribution (:318-330) and the objective partials (:160-161)
// --------------------------------------------------------------------------------------
#ifndef HMX_OLDSUM_CB
#define HMX_OLDSUM_CB 4
#endif
template <int KPL>
__global__ __launch_bounds__(TPB) void k_oldsum(Dev D) {
  constexpr int CB = HMX_OLDSUM_CB;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const int K = D.K;
  const int total = D.boff[D.nb];  // padded length of this round's order
  const int per = (total + nw - 1) / nw;
  const int s = wave * per, e = min(total, s + per);
  if (s >= e) return;
  unsigned long long oacc[KPL];
#pragma unroll
  for (int q = 0; q < KPL; q++) oacc[q] = 0ull;
  int curq = -1, curb = -1;
  for (int p = s; p < e; p += CB) {
    const int nc = min(CB, e - p);
    int cell[CB]; float r[CB][KPL];
#pragma unroll
    for (int c = 0; c < CB; c++) {
      cell[c] = ld_or(D.lorder, (size_t)min(p + c, e - 1), c < nc, -1);  // -1: padding slot
#pragma unroll
      for (int q = 0; q < KPL; q++) {
        const int k = lane + 64 * q;
        r[c][q] = ld_or(D.R, (size_t)max(cell[c], 0) * K + min(k, K - 1), cell[c] >= 0 && k < K, 0.0f);
      }
    }
#pragma unroll
    for (int c = 0; c < CB; c++) {
      if (cell[c] >= 0) {
        const int b = D.blk[cell[c]], q0 = D.lcombo[p + c];
        if (b != curb || q0 != curq) {
          if (curq >= 0) flush_fx<KPL>(D.Sold_fx + (size_t)curb * D.B * K, D.qlev, curq, D.C, K, lane, oacc);
          curb = b; curq = q0;
        }
#pragma unroll
        for (int q = 0; q < KPL; q++) oacc[q] += fx_of(r[c][q]);
      }
    }
  }
  if (curq >= 0) flush_fx<KPL>(D.Sold_fx + (size_t)curb * D.B * K, D.qlev, curq, D.C, K, lane, oacc);
}

// k_oldsum_stream: the same sums, but R is read in INTERNAL cell order -- a pure sequential stream of the K-float rows
// instead of a gather through this round's sorted order.  The target table of a cell depends on its block id, so the
// accumulators live in LDS ([nb][K] 64-bit fixed point, ds_add_u64; the 64 lanes of a wave hit 64 different clusters of
// one block: conflict free) and are flushed with global atomics when the combination changes and at the end of the
// workgroup's contiguous range of sort chunks (<= SORT_CHUNK cells of one combination each).  Integer sums: exact,
// order independent, identical to k_oldsum's.
template <int KPL>
__global__ __launch_bounds__(256) void k_oldsum_stream(Dev D) {
  extern __shared__ unsigned long long otab[];   // [nb][K]
  const int K = D.K, nT = D.nb * K;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int cpw = (D.nchunks + gridDim.x - 1) / gridDim.x;
  const int c0 = blockIdx.x * cpw, c1 = min(D.nchunks, c0 + cpw);
  if (c0 >= c1) return;
  for (int i = threadIdx.x; i < nT; i += blockDim.x) otab[i] = 0ull;
  __syncthreads();
  auto flush = [&](int q) {   // + zero
    for (int i = threadIdx.x; i < nT; i += blockDim.x) {
      const unsigned long long v = otab[i];
      if (v) {
        const int blk = i / K, k = i - blk * K;
        for (int cc = 0; cc < D.C; cc++)
          atomicAdd((unsigned long long*)&D.Sold_fx[((size_t)blk * D.B + D.qlev[q * D.C + cc]) * K + k], v);
        otab[i] = 0ull;
      }
    }
  };
  int curq = D.schunks[c0].q;
  for (int ch = c0; ch < c1; ch++) {
    const Item it = D.schunks[ch];
    if (it.q != curq) { __syncthreads(); flush(curq); __syncthreads(); curq = it.q; }
    const int per = (it.cnt + 3) >> 2;                          // the chunk's cells: one contiguous quarter per wave
    const int s = it.start + w * per, e = min(it.start + it.cnt, s + per);
    for (int base = s; base < e; base += 64) {
      const int n = min(64, e - base);
      const int bv = D.blk[min(base + lane, e - 1)];           // block ids of the next 64 cells, one per lane
      for (int u = 0; u < n; u += 8) {                          // eight rows (3.2 KB at K = 100) in flight per wave
        float r[8][KPL];
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const size_t row = (size_t)min(base + u + c, e - 1) * K;
#pragma unroll
          for (int q = 0; q < KPL; q++) r[c][q] = D.R[row + min(lane + 64 * q, K - 1)];
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
          if (u + c < n) {
            const int b = __builtin_amdgcn_readlane(bv, u + c);
#pragma unroll
            for (int q = 0; q < KPL; q++)
              if (lane + 64 * q < K) atomicAdd(&otab[b * K + lane + 64 * q], fx_of(r[c][q]));
          }
        }
      }
    }
  }
  __syncthreads();
  flush(curq);
}

// k_oldsum_stream4: as k_oldsum_stream for K % 4 == 0 -- the rows of a sort chunk are ONE contiguous run of floats, read
// with 16-byte loads (four in flight per lane); a lane's four floats belong to one cell, whose block id selects the LDS row.
// 1024-thread workgroups over ~4 chunks each: full occupancy with a quarter of the flush atomics (measured: the 4M global
// atomics of one-chunk workgroups cost 32 of 128 us).
__global__ __launch_bounds__(1024) void k_oldsum_stream4(Dev D) {
  extern __shared__ unsigned long long otab[];   // [nb][K]
  const int K = D.K, nT = D.nb * K;
  const int cpw = (D.nchunks + gridDim.x - 1) / gridDim.x;
  const int c0 = blockIdx.x * cpw, c1 = min(D.nchunks, c0 + cpw);
  if (c0 >= c1) return;
  for (int i = threadIdx.x; i < nT; i += blockDim.x) otab[i] = 0ull;
  __syncthreads();
  // LDS layout of a block's row: entry (k % 4) * K/4 + k / 4 holds cluster k -- a lane owns four CONSECUTIVE clusters of a cell, and
  // with the natural layout its four ds_add_u64 would sit 32 bytes from its neighbours' (8 lanes per bank pair: rocprof counted
  // 77 % of the LDS cycles of this kernel as bank conflicts); transposed, the lanes of one atomic instruction hit consecutive slots
  const int K4 = K >> 2;
  auto flush = [&](int q) {   // + zero
    for (int i = threadIdx.x; i < nT; i += blockDim.x) {
      const unsigned long long v = otab[i];
      if (v) {
        const int blk = i / K, kp = i - blk * K, k = (kp % K4) * 4 + kp / K4;
#ifndef HMX_OS_NOFLUSH
        for (int cc = 0; cc < D.C; cc++)
          atomicAdd((unsigned long long*)&D.Sold_fx[((size_t)blk * D.B + D.qlev[q * D.C + cc]) * K + k], v);
#endif
        otab[i] = 0ull;
      }
    }
  };
  const unsigned magic = (unsigned)((0x100000000ull + (unsigned)K - 1) / (unsigned)K);   // floor(n / K) = umulhi(n, magic), n < 2^32 / K
  int curq = D.schunks[c0].q;
  for (int ch = c0; ch < c1; ch++) {
    const Item it = D.schunks[ch];
    if (it.q != curq) { __syncthreads(); flush(curq); __syncthreads(); curq = it.q; }
    const f32x4* src = reinterpret_cast<const f32x4*>(D.R + (size_t)it.start * K);
    const int n4 = it.cnt * (K >> 2);                            // float4 groups of this chunk
    for (int base = 0; base < n4; base += 4 * (int)blockDim.x) {
      f32x4 v[4]; int bl[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int f = min(base + u * (int)blockDim.x + (int)threadIdx.x, n4 - 1);
        v[u] = src[f];
        bl[u] = D.blk[it.start + (int)__umulhi((unsigned)(4 * f), magic)];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int f = base + u * (int)blockDim.x + (int)threadIdx.x;
        if (f < n4) {
          const int cell = (int)__umulhi((unsigned)(4 * f), magic);
          unsigned long long* row = otab + bl[u] * K + ((4 * f - cell * K) >> 2);
#pragma unroll
          for (int e = 0; e < 4; e++) {
#ifdef HMX_OS_NOATOM   // timing experiment (tools/oldsum_cmp.sh): no LDS atomics
            if (fx_of(v[u][e]) == 0x123456789ull) row[e * K4] = 1;
#else
            atomicAdd(&row[e * K4], fx_of(v[u][e]));
#endif
          }
        }
      }
    }
  }
  __syncthreads();
  flush(curq);
}

// fold the finished block into O and build the penalty table of block j (j < 0: fold only).
// Phase 1 (k_fold): O_fx[b][k] += Snew - Sold[j]; Snew = 0.   Phase 2 (k_penalty): pen[b][k].
// One thread per table entry; two launches because phase 2 reads a column sum of phase 1.
// mode 0: O += sum_rep Snew[rep] - Sold[j] (single GPU).  mode 1: Snew[0] = sum_rep Snew[rep] only (the
// sharded path all-reduces Snew[0] next).  mode 2: O += Snew[0] - Sold[j] (after that all-reduce).
__global__ void k_fold(Dev D, int j, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = D.B * D.K;
  if (i >= n) return;
  long long sn = D.Snew_fx[i];
  if (mode != 2) for (int r = 1; r < D.nrep; r++) { sn += D.Snew_fx[(size_t)r * n + i]; D.Snew_fx[(size_t)r * n + i] = 0; }
  if (mode == 1) { D.Snew_fx[i] = sn; return; }
  long long o = D.O_fx[i] + sn;
  if (j >= 0) o -= D.Sold_fx[(size_t)j * n + i];
  D.O_fx[i] = o;
  D.Snew_fx[i] = 0;
}
// objective partial slots -> obj[0..1]
__global__ __launch_bounds__(1024) void k_obj_reduce(Dev D) {
  __shared__ double ra[1024], rb[1024];
  double a = 0.0, b = 0.0;
  double* row = D.objpart + (size_t)blockIdx.x * D.nwmax * 2;
  for (int i = threadIdx.x; i < D.nwmax; i += 1024) { a += row[2 * i]; b += row[2 * i + 1]; row[2 * i] = 0.0; row[2 * i + 1] = 0.0; }
  ra[threadIdx.x] = a; rb[threadIdx.x] = b;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) { ra[threadIdx.x] += ra[threadIdx.x + off]; rb[threadIdx.x] += rb[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { D.objrow[2 * blockIdx.x] = ra[0]; D.objrow[2 * blockIdx.x + 1] = rb[0]; }
}
// fixed-order sum of the slot rows -> obj[0..1]
__global__ void k_obj_final(Dev D) {
  double a = 0.0, b = 0.0;
  for (int s = 0; s < D.objslots; s++) { a += D.objrow[2 * s]; b += D.objrow[2 * s + 1]; }
  D.obj[0] = a; D.obj[1] = b;
}
// Single-launch variant (one GPU): fold + penalty with ping-pong tables, so no thread reads what another writes;
// one workgroup per 16 clusters, the new O column block goes through LDS for the covariate-0 row sums.
//   Oin/Sin: O and the replicas filled by the previous block update;  Oout: new O;  Szero: the replica set the NEXT
//   update accumulates into (zeroed here).  j < 0: fold only.
__global__ __launch_bounds__(256) void k_foldpen(Dev D, int j, const long long* __restrict__ Oin, long long* __restrict__ Oout,
                                                 const long long* __restrict__ Sin, long long* __restrict__ Szero) {
  extern __shared__ long long shO[];  // [B][16] new O of this workgroup's 16 clusters
  const int K = D.K, B = D.B, n = B * K;
  const int kk = threadIdx.x & 15, k = blockIdx.x * 16 + kk;
  const long long* sold = (j >= 0) ? D.Sold_fx + (size_t)j * n : nullptr;
  for (int b = threadIdx.x >> 4; b < B; b += 16) {
    long long o = 0;
    if (k < K) {
      const size_t i = (size_t)b * K + k;
      o = Oin[i];
      for (int r = 0; r < D.nrep; r++) { o += Sin[(size_t)r * n + i]; Szero[(size_t)r * n + i] = 0; }
      if (sold) o -= sold[i];
      Oout[i] = o;
    }
    shO[b * 16 + kk] = o;
  }
  if (j < 0) return;
  __syncthreads();
  long long rs = 0;  // rowsum(R) of the cells currently "in" = sum over the levels of covariate 0 of the NEW O
  for (int b0 = 0; b0 < D.B0; b0++) rs += shO[b0 * 16 + kk];
  const double rsd = (double)rs * FX_INV;
  if (k < K)
    for (int b = threadIdx.x >> 4; b < B; b += 16) {
      const float of = (float)((double)shO[b * 16 + kk] * FX_INV);
      const float ef = (float)(rsd * (double)D.Pr_b[b]);
      D.pen[(size_t)b * K + k] = pen_pow((2.0f * ef) + 1.0f, of + ef + 1.0f, D.theta[b]);
    }
}
__global__ void k_penalty(Dev D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.B * D.K) return;
  const int K = D.K, b = i / K, k = i - b * K;
  // rowsum(R) over the cells currently "in" = sum over the levels of covariate 0
  // (every cell has exactly one level per covariate)
  long long rs = 0;
  for (int b0 = 0; b0 < D.B0; b0++) rs += D.O_fx[(size_t)b0 * K + k];
  const float o = (float)((double)D.O_fx[i] * FX_INV);
  const float e = (float)(((double)rs * FX_INV) * (double)D.Pr_b[b]);
  D.pen[i] = pen_pow((2.0f * e) + 1.0f, o + e + 1.0f, D.theta[b]);
}

template <int KPL, int DPL>
__global__ __launch_bounds__(TPB) void k_update(Dev D, int j) {
  extern __shared__ __attribute__((aligned(16))) float ldsY[];
  stage_Y(ldsY, D.Yt, D.d, D.K, D.KP);
  constexpr int CB = 4;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const int d = D.d, K = D.K;
  const int p0 = D.boff[j], p1 = D.boff[j + 1];
  const int per = (p1 - p0 + nw - 1) / nw;
  const int s = p0 + wave * per, e = min(p1, s + per);
  if (s >= e) return;
  float sig[KPL], penv[KPL];
  unsigned long long oacc[KPL];
#pragma unroll
  for (int q = 0; q < KPL; q++) { sig[q] = ld_or(D.sigma, (size_t)min(lane + 64 * q, K - 1), lane + 64 * q < K, 1.0f); penv[q] = 0.0f; oacc[q] = 0ull; }
  double od = 0.0, oe = 0.0;
  int curq = -1;
  for (int p = s; p < e; p += CB) {
    const int nc = min(CB, e - p);
    int cell[CB]; float z[CB][DPL];
#pragma unroll
    for (int c = 0; c < CB; c++) {
      cell[c] = ld_or(D.lorder, (size_t)min(p + c, e - 1), c < nc, -1);  // -1: padding slot
      if (cell[c] >= 0) load_row<DPL>(D.Zc, (size_t)cell[c], D.zs, d, lane, z[c]);
      else {
#pragma unroll
        for (int t = 0; t < DPL; t++) z[c][t] = 0.0f;
      }
    }
    float acc[CB][KPL];
    group_dots<KPL, DPL, CB>(ldsY, d, D.KP, lane, z, acc);
#pragma unroll
    for (int c = 0; c < CB; c++) {
      if (cell[c] >= 0) {
        const int q0 = D.combo[cell[c]];
        if (q0 != curq) {
          if (curq >= 0) flush_fx<KPL>(D.Snew_fx, D.qlev, curq, D.C, K, lane, oacc);
          curq = q0;
#pragma unroll
          for (int q = 0; q < KPL; q++) penv[q] = 0.0f;
          for (int cc = 0; cc < D.C; cc++) {  // penalty of a cell = SUM over its covariates (:322 is a matrix product)
            const int b = D.qlev[q0 * D.C + cc];
#pragma unroll
            for (int q = 0; q < KPL; q++) if (lane + 64 * q < K) penv[q] += D.pen[(size_t)b * K + lane + 64 * q];
          }
        }
        float r[KPL], dist[KPL];
        float s1 = 0.0f;
#pragma unroll
        for (int q = 0; q < KPL; q++) {
          dist[q] = 2.0f * (1.0f - acc[c][q]);
          r[q] = (lane + 64 * q < K) ? expf(-dist[q] / sig[q]) : 0.0f;
          s1 += fabsf(r[q]);
        }
        s1 = wsum(s1);
        if (s1 == 0.0f) s1 = 1.0f;
        float s2 = 0.0f;
#pragma unroll
        for (int q = 0; q < KPL; q++) { r[q] = (r[q] / s1) * penv[q]; s2 += fabsf(r[q]); }
        s2 = wsum(s2);
        if (s2 == 0.0f) s2 = 1.0f;
#pragma unroll
        for (int q = 0; q < KPL; q++) {
          const int k = lane + 64 * q;
          r[q] = r[q] / s2;
          if (k < K) {
            D.R[(size_t)cell[c] * K + k] = r[q];
            oacc[q] += fx_of(r[q]);
            od += (double)(r[q] * dist[q]);
            oe += (double)((r[q] * trunc_logf_dev(r[q])) * sig[q]);
          }
        }
      }
    }
  }
  if (curq >= 0) flush_fx<KPL>(D.Snew_fx, D.qlev, curq, D.C, K, lane, oacc);
  od = wsumd(od); oe = wsumd(oe);
  if (lane == 0) { D.objpart[2 * wave] += od; D.objpart[2 * wave + 1] += oe; }  // private slot (slot row 0): no atomics
}

#endif  // !HMX_TILE_BF
// --------------------------------------------------------------------------------------
// MFMA tile variant of the block update (the dominant kernel).
//
// One wave owns tiles of 16 gathered cells.  dist(16 cells x 16 clusters) accumulates on the
// matrix cores with v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain):
//   A = the cells' embedding rows   lane l supplies A[cell = l&15][k-slot = l>>4]
//   B = centroids                   lane l supplies B[k-slot = l>>4][cluster = l&15]
//   D                               lane l holds   D[cell = 4*(l>>4)+reg][cluster = l&15]
// PCs are assigned to k-slots so that every lane fetches its A operands with 16-byte loads:
// in float4 group t the lane with slot p holds PCs 16t+4p+{0..3}, used in steps 4t+{0..3};
// the centroid image D.Yimg is laid out on the host in exactly that order (and as float4 per
// lane per step -> conflict-free ds_read_b128).  The softmax normalisations are 16-lane DPP
// row reductions, the penalty is a per-run register vector, O is accumulated per lane in
// 64-bit fixed point and flushed once per run of equal covariate combination.
// --------------------------------------------------------------------------------------

__device__ __forceinline__ float dpp_f(float v, const int ctrl_id) {
  // ctrl_id: 0 quad_perm[1,0,3,2]  1 quad_perm[2,3,0,1]  2 row_half_mirror  3 row_mirror
  // (old = 0 + bound_ctrl: every source lane is in range, and this form lets hipcc fold the move into v_add_f32_dpp)
  const int x = __float_as_int(v);
  int r;
  switch (ctrl_id) {
    case 0: r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); break;
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true); break;
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true); break;
  }
  return __int_as_float(r);
}
// sum over the 16 lanes of a DPP row (lanes sharing l>>4); every lane gets the total
__device__ __forceinline__ float rowsum16(float v) {
#if HMX_USE_DPP
  v += dpp_f(v, 0);
  v += dpp_f(v, 1);
  v += dpp_f(v, 2);
  v += dpp_f(v, 3);
  return v;
#else
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
#endif
}

// arma::normalise(Z, 2, 0) with 16-byte accesses: 16 lanes per row (lane c: float4 c [and c + 16]), four rows per wave instruction, four
// instructions in flight; the row's sum of squares is a 16-lane DPP reduction.  Rows are zero beyond d (zs = d rounded up to 4).
template <int NF>
__global__ __launch_bounds__(TPB) void k_normalize4(const float* Zsrc, float* Z, int n, int nq) {   // Zsrc == Z: in place
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (long long base = (long long)wave * 16; base < n; base += (long long)nw * 16) {
    f32x4 v[4][NF];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long long cell = min(base + 4 * u + g, (long long)n - 1);
      const f32x4* row = reinterpret_cast<const f32x4*>(Zsrc) + cell * nq;
#pragma unroll
      for (int f = 0; f < NF; f++) v[u][f] = row[min(c + 16 * f, nq - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      float ss = 0.0f;
#pragma unroll
      for (int f = 0; f < NF; f++) {
        if (c + 16 * f >= nq) v[u][f] = zero4;
        ss += v[u][f][0] * v[u][f][0] + v[u][f][1] * v[u][f][1] + v[u][f][2] * v[u][f][2] + v[u][f][3] * v[u][f][3];
      }
      float nrm = sqrtf(rowsum16(ss));
      if (nrm == 0.0f) nrm = 1.0f;
      const long long cell = base + 4 * u + g;
      if (cell < n) {
        f32x4* row = reinterpret_cast<f32x4*>(Z) + cell * nq;
#pragma unroll
        for (int f = 0; f < NF; f++)
          if (c + 16 * f < nq) { f32x4 o; for (int e = 0; e < 4; e++) o[e] = v[u][f][e] / nrm; row[c + 16 * f] = o; }
      }
    }
  }
}

// N independent row sums, the DPP steps interleaved (ILP N: see epi_rows)
template <int N>
__device__ __forceinline__ void rowsum16xN(float (&v)[N]) {
#if HMX_USE_DPP
#pragma unroll
  for (int st = 0; st < 4; st++) {
    float t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = (st == 0) ? dpp_f(v[i], 0) : (st == 1) ? dpp_f(v[i], 1) : (st == 2) ? dpp_f(v[i], 2) : dpp_f(v[i], 3);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += t[i];
  }
#else
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = rowsum16(v[i]);
#endif
}

template <int NCT>
__device__ __forceinline__ void tile_dots(const f32x4* __restrict__ ldsY4, const float* __restrict__ zrow, bool valid,
                                          int g, int lane, int NS, int NT4, int tail, f32x4 (&acc)[NCT],
                                          bool have_first = false, f32x4 zfirst = f32x4{0.f, 0.f, 0.f, 0.f}) {
  constexpr int NQ = (NCT + 3) / 4;
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // zrow always points at a real row (row 0 for padding lanes): unconditional loads, masked afterwards
  f32x4 zt = zero4;
  if (have_first) zt = zfirst;                  // fetched by the caller before its LDS-staging barrier
  else if (NT4 > 0) zt = *reinterpret_cast<const f32x4*>(zrow + 4 * g);
  for (int t = 0; t < NT4; ++t) {
    const f32x4 zc = valid ? zt : zero4;
    if (t + 1 < NT4) zt = *reinterpret_cast<const f32x4*>(zrow + 16 * (t + 1) + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s = 4 * t + e;
#pragma unroll
      for (int qd = 0; qd < NQ; ++qd) {
        const f32x4 y = ldsY4[(qd * NS + s) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 * qd + i < NCT) acc[4 * qd + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc[e], y[i], acc[4 * qd + i], 0, 0, 0);
      }
    }
  }
  for (int u = 0; u < tail; ++u) {
    const int s = 4 * NT4 + u;
    const float zl = zrow[16 * NT4 + 4 * u + g];
    const float zv = valid ? zl : 0.0f;
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) {
      const f32x4 y = ldsY4[(qd * NS + s) * 64 + lane];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * qd + i < NCT) acc[4 * qd + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv, y[i], acc[4 * qd + i], 0, 0, 0);
    }
  }
}

// LDS-DMA: every lane copies 16 bytes from its own global address to  lds_base + lane * 16  (wave-uniform base) without
// touching a VGPR; completion is counted by vmcnt like an ordinary load.
__device__ __forceinline__ void glds16(const float* gsrc, f32x4* lds_base) {
  // inline asm on purpose: with the builtin hipcc treats the DMA as a store to LDS and drains it (vmcnt(0)) in front of the next
  // LDS read -- the epilogue's penalty-table reads -- which exposes the whole copy latency.  The asm statement is invisible
  // to the wait-count pass; the caller waits with a counted s_waitcnt (the copies are older than everything it leaves in flight).
  // M0 = wave-uniform LDS byte address of the destination; written and restored inside the statement (compiler-reserved).
  const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)(uintptr_t)lds_base));
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const int* gsrc, int* lds_base) {   // 4 bytes per lane, same contract as glds16
  const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)(uintptr_t)lds_base));
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the MFMA chain of one tile with the A operands in an LDS image filled by glds16: group t < NT4 holds each lane's PCs
// 16t+4g..+3, group NT4+u the 16 bytes at PC 16*NT4+4u (the same for the four k-slot lanes of a cell: lane g uses component g)
template <int NCT>
__device__ __forceinline__ void tile_dots_lds(const f32x4* __restrict__ ldsY4, const f32x4* __restrict__ rows, bool valid, int g,
                                              int lane, int NS, int NT4, int tail, f32x4 (&acc)[NCT]) {
  constexpr int NQ = (NCT + 3) / 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) acc[ct] = zero4;
  for (int t = 0; t < NT4; ++t) {
    const f32x4 zl = rows[t * 64 + lane];
    const f32x4 zc = valid ? zl : zero4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s = 4 * t + e;
#pragma unroll
      for (int qd = 0; qd < NQ; ++qd) {
        const f32x4 y = ldsY4[(qd * NS + s) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 * qd + i < NCT) acc[4 * qd + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc[e], y[i], acc[4 * qd + i], 0, 0, 0);
      }
    }
  }
  for (int u = 0; u < tail; ++u) {
    const int s = 4 * NT4 + u;
    const f32x4 zl = rows[(NT4 + u) * 64 + lane];
    const float zg = (g == 0) ? zl[0] : (g == 1) ? zl[1] : (g == 2) ? zl[2] : zl[3];
    const float zv = valid ? zg : 0.0f;
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) {
      const f32x4 y = ldsY4[(qd * NS + s) * 64 + lane];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * qd + i < NCT) acc[4 * qd + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv, y[i], acc[4 * qd + i], 0, 0, 0);
    }
  }
}

// A-operand rows of one tile held in registers (d <= 76: at most 4 float4 groups + 3 single steps), so that the rows of
// tile i+1 can be requested while tile i is computed.  Loads are unconditional with clamped offsets (see ld_or).
struct RowRegs { f32x4 v[4]; float t[3]; };
__device__ __forceinline__ void load_rows(const float* __restrict__ zrow, int g, int NT4, int tail, RowRegs& r) {
  const int tmax = NT4 > 0 ? NT4 - 1 : 0, umax = tail > 0 ? tail - 1 : 0;
  if (NT4 > 0) {
#pragma unroll
    for (int t = 0; t < 4; t++) r.v[t] = *reinterpret_cast<const f32x4*>(zrow + 16 * min(t, tmax) + 4 * g);
  }
  if (tail > 0) {
#pragma unroll
    for (int u = 0; u < 3; u++) r.t[u] = zrow[16 * NT4 + 4 * min(u, umax) + g];
  }
}
template <int NCT>
__device__ __forceinline__ void tile_dots_regs(const f32x4* __restrict__ ldsY4, const RowRegs& r, bool valid, int lane,
                                               int NS, int NT4, int tail, f32x4 (&acc)[NCT]) {
  constexpr int NQ = (NCT + 3) / 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) acc[ct] = zero4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < NT4) {
      const f32x4 zc = valid ? r.v[t] : zero4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = 4 * t + e;
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
          const f32x4 y = ldsY4[(qd * NS + s) * 64 + lane];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (4 * qd + i < NCT) acc[4 * qd + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(zc[e], y[i], acc[4 * qd + i], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    if (u < tail) {
      const int s = 4 * NT4 + u;
      const float zv = valid ? r.t[u] : 0.0f;
#pragma unroll
      for (int qd = 0; qd < NQ; ++qd) {
        const f32x4 y = ldsY4[(qd * NS + s) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 * qd + i < NCT) acc[4 * qd + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv, y[i], acc[4 * qd + i], 0, 0, 0);
      }
    }
  }
}

// ---- split-bf16 form of the distance GEMM ---------------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate on gfx950 (32 cycles per SIMD, on the datapath the epilogue's VALU work needs);
// v_mfma_f32_16x16x32_bf16 takes ~17 cycles for 8x the products on the matrix cores proper.  With x = hi + mid + lo (three bf16 parts,
// exact: bf3_split) the 16 x 16 x 32 product of fp32 operands is six bf16 MFMAs (the three dropped cross terms are < 2^-21 |x||y| at worst, ~2^-24 typically: hmx_internal.h),
// fp32 accumulation: 12 NCT MFMAs per 64 PCs instead of 16 NCT per 64, at half the cycles each, and VALU work of the SIMD's other
// wave overlaps them.  A operand (cells): lane (c, g) holds PCs 32 s + 8 g + {0..7} of cell c in step s -- two 16-byte loads -- and
// splits them in registers (5.5 VALU per value); B operand (centroids): the three parts from the LDS image D.Yimg3, one ds_read_b128 each.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Bf3 { u32x4 p[3]; };
__device__ __forceinline__ Bf3 bf3_split8(const f32x4 lo4, const f32x4 hi4) {
  Bf3 o;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float x0 = i < 2 ? lo4[2 * i] : hi4[2 * i - 4], x1 = i < 2 ? lo4[2 * i + 1] : hi4[2 * i - 3];
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    o.p[0][i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);      // { bf16(x0), bf16(x1) }: the upper halves
    o.p[1][i] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    o.p[2][i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
  return o;
}
// one step (32 PCs) of all cluster tiles: smallest terms first
template <int NCT>
__device__ __forceinline__ void bf_step(const u32x4* __restrict__ ldsB, const Bf3& a, int s, int NS2, int lane, f32x4 (&acc)[NCT]) {
  auto mf = [](const u32x4 A, const u32x4 B, const f32x4 C) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0);
  };
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) {
    const u32x4* b = ldsB + ((size_t)(ct * NS2 + s) * 3) * 64 + lane;
    const u32x4 b0 = b[0], b1 = b[64], b2 = b[128];
    f32x4 v = acc[ct];
    v = mf(a.p[0], b2, v);
    v = mf(a.p[2], b0, v);
    v = mf(a.p[1], b1, v);
    v = mf(a.p[0], b1, v);
    v = mf(a.p[1], b0, v);
    v = mf(a.p[0], b0, v);
    acc[ct] = v;
  }
}
// rows in registers: r.v[2 s + h] = the 16 bytes at PC 32 s + 8 g + 4 h (clamped into the row; rowmask bit 2 s + h = inside the row)
__device__ __forceinline__ void load_rows_bf(const float* __restrict__ zrow, int g, int zs, RowRegs& r) {
#pragma unroll
  for (int t = 0; t < 4; t++) r.v[t] = *reinterpret_cast<const f32x4*>(zrow + min(32 * (t >> 1) + 8 * g + 4 * (t & 1), zs - 4));
}
__device__ __forceinline__ int rows_mask_bf(int g, int zs) {
  int m = 0;
#pragma unroll
  for (int t = 0; t < 8; t++) m |= (32 * (t >> 1) + 8 * g + 4 * (t & 1) < zs) ? (1 << t) : 0;
  return m;
}
template <int NCT>
__device__ __forceinline__ void tile_dots_bf_regs(const u32x4* __restrict__ ldsB, const RowRegs& r, bool valid, int rowmask, int lane,
                                                  int NS2, f32x4 (&acc)[NCT]) {
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) acc[ct] = zero4;
  const int m = valid ? rowmask : 0;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    if (s < NS2) {
      const Bf3 a = bf3_split8((m >> (2 * s)) & 1 ? r.v[2 * s] : zero4, (m >> (2 * s + 1)) & 1 ? r.v[2 * s + 1] : zero4);
      bf_step<NCT>(ldsB, a, s, NS2, lane, acc);
    }
  }
}
// TWO tiles against ONE read of the centroid image (round 5, the chain's hoisted MFMA phase): a wave that owns two tiles of a block used to
// walk the 42 KB image twice -- the LDS operand reads of a SIMD's three tiles were 1.6 of the 5.4 us the workers need behind an arrival
// (DESIGN 7.1); with both tiles' rows in registers every (cluster tile, step) operand triple is read once and feeds twelve MFMAs, two
// independent accumulator chains.
template <int NCT>
__device__ __forceinline__ void tile_dots_bf_regs2(const u32x4* __restrict__ ldsB, const RowRegs& ra, bool valida, const RowRegs& rb, bool validb,
                                                   int rowmask, int lane, int NS2, f32x4 (&acca)[NCT], f32x4 (&accb)[NCT]) {
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto mf = [](const u32x4 A, const u32x4 B, const f32x4 C) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0);
  };
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) { acca[ct] = zero4; accb[ct] = zero4; }
  const int ma = valida ? rowmask : 0, mb = validb ? rowmask : 0;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    if (s < NS2) {
      const Bf3 a = bf3_split8((ma >> (2 * s)) & 1 ? ra.v[2 * s] : zero4, (ma >> (2 * s + 1)) & 1 ? ra.v[2 * s + 1] : zero4);
      const Bf3 b = bf3_split8((mb >> (2 * s)) & 1 ? rb.v[2 * s] : zero4, (mb >> (2 * s + 1)) & 1 ? rb.v[2 * s + 1] : zero4);
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) {
        const u32x4* y = ldsB + ((size_t)(ct * NS2 + s) * 3) * 64 + lane;
        const u32x4 y0 = y[0], y1 = y[64], y2 = y[128];
        f32x4 va = acca[ct], vb = accb[ct];          // (the same order of the six terms as bf_step: smallest first -- bit-identical to the one-tile form)
        va = mf(a.p[0], y2, va); vb = mf(b.p[0], y2, vb);
        va = mf(a.p[2], y0, va); vb = mf(b.p[2], y0, vb);
        va = mf(a.p[1], y1, va); vb = mf(b.p[1], y1, vb);
        va = mf(a.p[0], y1, va); vb = mf(b.p[0], y1, vb);
        va = mf(a.p[1], y0, va); vb = mf(b.p[1], y0, vb);
        va = mf(a.p[0], y0, va); vb = mf(b.p[0], y0, vb);
        acca[ct] = va; accb[ct] = vb;
      }
    }
  }
}
// rows streamed from memory one step ahead (any NS2 <= 4)
template <int NCT>
__device__ __forceinline__ void tile_dots_bf(const u32x4* __restrict__ ldsB, const float* __restrict__ zrow, bool valid, int rowmask, int g,
                                             int lane, int NS2, int zs, f32x4 (&acc)[NCT]) {
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) acc[ct] = zero4;
  const int m = valid ? rowmask : 0;
  f32x4 c0 = *reinterpret_cast<const f32x4*>(zrow + min(8 * g, zs - 4)), c1 = *reinterpret_cast<const f32x4*>(zrow + min(8 * g + 4, zs - 4));
  for (int s = 0; s < NS2; s++) {
    const f32x4 a0 = (m >> (2 * s)) & 1 ? c0 : zero4, a1 = (m >> (2 * s + 1)) & 1 ? c1 : zero4;
    if (s + 1 < NS2) {
      c0 = *reinterpret_cast<const f32x4*>(zrow + min(32 * (s + 1) + 8 * g, zs - 4));
      c1 = *reinterpret_cast<const f32x4*>(zrow + min(32 * (s + 1) + 8 * g + 4, zs - 4));
    }
    const Bf3 a = bf3_split8(a0, a1);
    bf_step<NCT>(ldsB, a, s, NS2, lane, acc);
  }
}

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int srclane) {
  const unsigned lo = (unsigned)__shfl((int)(unsigned)v, srclane, 64), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), srclane, 64);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  const unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), m, 64);
  const unsigned hi = __shfl_xor((unsigned)(v >> 32), m, 64);
  return ((unsigned long long)hi << 32) | lo;
}
// The four 16-lane groups of a wave hold partial sums of the SAME clusters (different cells): add them
// first, then one atomic per cluster from group 0 -- into one of several table replicas, so that the
// hundreds of waves of a launch do not serialise on the same few L2 atomic addresses.
template <int NCT>
__device__ __forceinline__ void flush_tile_fx(long long* __restrict__ tab, long long* __restrict__ tab2, const int* __restrict__ qlev, int q, int C,
                                              int K, int c, int g, unsigned long long (&oacc)[NCT]) {
  constexpr int NFULL = NCT >> 2, RT = NCT & 3, NG = NFULL + (RT ? 1 : 0);
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) {
    unsigned long long v = oacc[ct];
    v += shfl_xor_u64(v, 16);
    v += shfl_xor_u64(v, 32);
    oacc[ct] = v;
  }
  // Every lane now holds the sums of its clusters (kcol: 4c+{0..3} of a quad).  Redistribute so that lane (g, c) owns cluster
  // 64 G + 16 g + c of group G: ONE atomic instruction per group and level with all 64 lanes on consecutive addresses, instead of
  // one per cluster tile with 16 lanes.
  const int kk = 16 * g + c;
  unsigned long long mine[NG];
#pragma unroll
  for (int G = 0; G < NG; G++) {
    const int RG = (G < NFULL) ? 4 : RT;
    const int src = min(kk / RG, 15), jsel = kk - (kk / RG) * RG;
    unsigned long long v = 0ull;
#pragma unroll
    for (int jj = 0; jj < RG; jj++) {
      const unsigned long long t = shfl_u64(oacc[4 * G + jj], 16 * g + src);
      v = (jsel == jj) ? t : v;
    }
    mine[G] = (kk < 16 * RG) ? v : 0ull;
  }
  for (int cc = 0; cc < C; cc++) {
    const int b = qlev[q * C + cc];
#pragma unroll
    for (int G = 0; G < NG; G++) {
      const int k = 64 * G + kk;
      if (k < K && mine[G]) {
        atomicAdd((unsigned long long*)&tab[(size_t)b * K + k], mine[G]);
        // the same sums are the cells' OLD contribution to their block of the NEXT round (tiles are keyed by it, D.Sold_next)
        if (tab2) atomicAdd((unsigned long long*)&tab2[(size_t)b * K + k], mine[G]);
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) oacc[ct] = 0ull;
}

// Instantiated cluster-tile counts are {1..8,10,12,13,14,16} (hmx_setup picks the smallest >= ceil(K/16)): cluster tiles
// below this index are always completely inside K, only the ones from it on can hold k >= K.
// (cluster tiles are grouped in quads by kcol: the tiles of the last group can hold k >= K)
constexpr int first_partial_ct(int nct) { return (nct & 3) ? 4 * (nct >> 2) : 4 * ((nct >> 2) - 1); }
// Static-tile launches (head / Lloyd / seeding) use 256-thread workgroups, capped at one resident generation
// (D.static_maxblocks); measured: 768-thread workgroups (one per CU) are 30% slower (tail effect).
constexpr int tile_threads(int nct) { return 256; }
// MODE 0: block update (cells gathered through lpair, penalty in the exponent, ONE normalisation)  update_R :318-330
// MODE 1: head (static 16-cell tiles of the internal order, plain softmax)                 :141-150 / :221-227
// Register budget: K > 64 runs 2 waves per SIMD (<= 256 VGPRs: row prefetch + two accumulator sets, K <= 112);
// K <= 64 with a uniform sigma runs the 128-VGPR variant (WPS = 4).
// MODE 2: one Lloyd iteration of kmeans_centers (nearest centre, fixed-point sums in LDS)  src/utils.cpp:56-61
// MODE 3: the seeding race of kmeans_centers: for every anchor k  argmin_n -log(u_kn) / |2(1 - y_k.x_n)|  src/utils.cpp:24-34
// MODE 4: the WHOLE block chain of one update_R round in ONE persistent launch (one workgroup per CU): the workers run MODE 0's
//         tile pipeline for block j, arrive on a counter, and -- while a dedicated folder workgroup folds the block's
//         contribution into O and publishes the next penalty table -- already gather the rows and run the MFMAs of their
//         first tile of block j+1.  Only  + log2 pen -> exp2 -> normalise -> store -> flush  stays on the chain's critical
//         path; the centroid image is staged once per round instead of once per block step.
// WPS: waves per SIMD the register budget is cut for (update workgroup = 256*WPS threads).  USIG: one sigma for all clusters
// (the reference's default, R/ui.R:219-221): ce / cl become scalars, 2-3 register arrays of NCT floats disappear.
template <int NCT, int MODE, int WPS = 2, bool USIG = false, bool BF = (HMX_TILE_BF != 0)>
__global__ __launch_bounds__(256 * WPS) void k_tile(Dev D, int j) {
  constexpr bool LEAN = WPS > 2;
  constexpr bool CHAIN = (MODE == 4 || MODE == 5);     // MODE 5: the chain of a round whose R rows nobody reads (Dev::r_store == 0), see below
  constexpr bool NOSTORE = (MODE == 5);
  // ONE LDS object (a second __shared__ object de-pipelines hipcc's waits):
  //   [ centroid image: NQ*NS*64 float4 | MODE 0: pen[B][K] + qlev[Q][C] (if they fit) | MODE 2: int64 sums[K][d] + counts[K] ]
  extern __shared__ __attribute__((aligned(16))) f32x4 lds4[];
  const int K = D.K, C = D.C, zs = D.zs;
  const int nY4 = BF ? NCT * D.NS2 * 3 * 64 : D.NQ * D.NS * 64;     // 16-byte entries of the centroid image (f32 steps | three bf16 parts)
  const u32x4* const ldsB = reinterpret_cast<const u32x4*>(lds4);
  // MODE 0 with D.fused_fold: [ image | O' int64 [B][K] | pen | qlev ] -- the fold + penalty of this block step is
  // recomputed by EVERY workgroup in its own LDS (k_foldpen's launch and its boundary disappear); workgroup 0 also
  // publishes O' and zeroes the replica set of the NEXT launch (three sets rotate, so nobody reads what is zeroed).
  const int nBK = D.B * K;
  long long* ldsO = reinterpret_cast<long long*>(lds4 + nY4);
  float* ldsPen = ((MODE == 0 && D.fused_fold) || CHAIN) ? reinterpret_cast<float*>(ldsO + nBK) : reinterpret_cast<float*>(lds4 + nY4);
  int* ldsQlev = reinterpret_cast<int*>(ldsPen + ((nBK + 3) & ~3));
  long long* ltab = reinterpret_cast<long long*>(lds4 + nY4);
  constexpr bool UPD = (MODE == 0 || CHAIN);   // block update modes (gathered cells, penalty, O contributions)
  int p0 = 0, ntiles;
  if constexpr (MODE == 0) { p0 = D.boff[j]; ntiles = (D.boff[j + 1] - p0) >> 4; }  // padded: combination-pure tiles
  else if constexpr (CHAIN) { p0 = D.boff[0]; ntiles = (D.boff[1] - p0) >> 4; }
  else if (MODE == 1 && D.head_gather) ntiles = D.boff[D.nb] >> 4;     // the head runs over the NEXT round's padded order (see flush_run)
  else ntiles = D.ntitems;
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  // wave index as a SCALAR: tile numbers and all loop control become SALU work (no exec-mask branches in the tile loop)
  // (MODE 4: the last workgroup is the folder, the others are the workers)
  const int nw = ((gridDim.x - (CHAIN ? 1 : 0)) * blockDim.x) >> 6;
  int wave_ = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if constexpr (CHAIN) {
    // tiles are dealt t = wave + i * nw, so the waves with the low indices get the extra tile of a block: number the first
    // wave of every SIMD (all workgroups) before the second ones, and each SIMD hosts one heavy and one light wave -- the
    // same tile count on every SIMD instead of 4 tiles on half the CUs and 2 on the others
    const int wpg = (int)blockDim.x >> 6, wib = (int)threadIdx.x >> 6, half = wpg >> 1, nwg = (int)gridDim.x - 1;
    // (second waves numbered workgroup-minor: the few SIMDs that must take a fourth tile are spread one per workgroup instead of
    //  filling whole CUs -- a CU's store / atomic queues are shared by its SIMDs)
    if (HMX_CHAIN_BALANCE && half >= 1) wave_ = (wib < half) ? (int)blockIdx.x * half + wib : nwg * half + (wib - half) * nwg + (int)blockIdx.x;
  }
  const int wave = __builtin_amdgcn_readfirstlane(wave_);
  auto stamp = [&](int slot) {  // diagnostics build only (-DHMX_TRACE, tools/trace_update.py): per-wave phase stamps
#ifdef HMX_TRACE
    if constexpr (MODE == 0) {
      if (D.trace && lane == 0) D.trace[(size_t)wave * 16 + slot] = (slot == 0 || slot == 7) ? wall_clock64() : __builtin_readcyclecounter();
    }
#endif
  };
  stamp(0); stamp(1);
  // MODE 0 (short launches, grid capped at the resident capacity): tiles are dealt round-robin, tile = wave + i*nw, so
  // every wave gets 1-2 tiles and no CU runs a second round of workgroups.  Static modes: contiguous ranges.
  // (dealing tiles workgroup-major instead -- equal tiles per CU -- measured 25% SLOWER: the 8 consecutive tiles of a
  //  workgroup share lorder/lcombo cache lines and the lighter half of the CUs finishing early helps the tail)
  const int per = (ntiles + nw - 1) / nw;
  // MODE 0 with many tiles per wave (D.upd_contig): contiguous ranges -- consecutive tiles share their (combination, next block)
  // key, so the wave flushes its O contributions once per RUN of tiles instead of once per tile (the flush atomics were 15 % of
  // a 10M-cell block step once every tile filed its sums twice, see flush_run)
  // MODE 4 (two-accumulator chain): contiguous BALANCED ranges, re-derived for every block -- T = base nw + rem tiles: the rem
  // lowest-numbered waves own base + 1 consecutive tiles, the others base.  A wave's tiles of a block are neighbours in the padded
  // order: they mostly share their (combination, next block) key, i.e. one contribution flush and one penalty fetch per wave and
  // block instead of one per tile, and their pair loads share cache lines.
  constexpr bool CHAIN_CONTIG = (CHAIN) && !LEAN;
  auto chain_range = [&](const int T, int& s0, int& e0) {
    const int base = T / nw, rem = T - base * nw;
    s0 = wave * base + min(wave, rem);
    e0 = s0 + base + (wave < rem ? 1 : 0);
  };
  const bool strided = (CHAIN && !CHAIN_CONTIG) || (MODE == 0 && !D.upd_contig);
  int ts = strided ? wave : wave * per;
  int te = strided ? ntiles : min(ntiles, ts + per);     // (MODE 4 re-derives it for every block)
  if constexpr (CHAIN_CONTIG) chain_range(ntiles, ts, te);
  if (CHAIN && blockIdx.x == gridDim.x - 1) te = ts;   // the folder owns no tiles
  const int tstep = strided ? nw : 1;
  // MODE 0: the first tile's cell ids and the first 16 bytes of their embedding rows are requested BEFORE the
  // LDS staging below, so the two dependent HBM round trips overlap with it
  // Software pipeline over tiles (when the rows fit in registers, D.NT4 <= 4): cell ids two tiles ahead, embedding rows
  // one tile ahead.  MODE 0 requests the first tile's ids and rows BEFORE the LDS staging below.
  const bool pre = !LEAN && D.NT4 <= 4 && NCT <= 8;    // (K > 128: the extra row registers would spill)
  // the three forms of a tile's distance GEMM (rows in registers | rows streamed), fp32 MFMA or split bf16 (BF; hmx_setup offers BF
  // only where the register form exists for the same shapes: NS2 <= 2 whenever NT4 <= 4)
  const int rowmask = BF ? rows_mask_bf(g, zs) : 0;
  auto ld_rows = [&](const float* __restrict__ zr, RowRegs& r) __attribute__((always_inline)) {
    if constexpr (BF) load_rows_bf(zr, g, zs, r); else load_rows(zr, g, D.NT4, D.tail, r);
  };
  auto dots_regs = [&](const RowRegs& r, const bool valid, f32x4 (&acc)[NCT]) __attribute__((always_inline)) {
    if constexpr (BF) tile_dots_bf_regs<NCT>(ldsB, r, valid, rowmask, lane, D.NS2, acc);
    else tile_dots_regs<NCT>(lds4, r, valid, lane, D.NS, D.NT4, D.tail, acc);
  };
  auto dots_stream = [&](const float* __restrict__ zr, const bool valid, f32x4 (&acc)[NCT]) __attribute__((always_inline)) {
    if constexpr (BF) tile_dots_bf<NCT>(ldsB, zr, valid, rowmask, g, lane, D.NS2, zs, acc);
    else tile_dots<NCT>(lds4, zr, valid, g, lane, D.NS, D.NT4, D.tail, acc);
  };
  // (cell id, combination) of this lane's A-operand row, ONE vector load per tile: a separate uniform load of the tile's
  // combination ends in a readfirstlane right behind the load, i.e. a vmcnt(0) -- a drain of the 28 outstanding R stores
  // of the previous tile plus a full memory latency -- in every iteration.
  int2 cellN = make_int2(-1, -1), cellNN = make_int2(-1, -1);
  RowRegs rowsN;
  auto tile_cell = [&](int tile) -> int2 {     // (-1, .): padding slot / beyond the end
    if (tile >= te) return make_int2(-1, -1);
    if constexpr (UPD) return D.lpair[p0 + 16 * tile + c];
    else if (MODE == 1 && D.head_gather) return D.lpair[16 * tile + c];
    else {
      // keep this a per-lane (vector) load: with the uniform tile index hipcc would emit load + readfirstlane, i.e. a
      // vmcnt(0) -- a full memory latency per tile that also drains the prefetched rows
      const Item* tp = D.titems + tile;
      asm volatile("" : "+v"(tp));
      const Item it = *tp;
      return make_int2((c < it.cnt) ? it.start + c : -1, it.q);
    }
  };
  {
    // Staging.  Every global load of this prologue is ISSUED before the first one is consumed (the launch is a link of
    // the block-step chain, its prologue a serial section): first the fold inputs -- replica tables written by the
    // previous launch's atomics, i.e. L2 misses -- then the centroid image, then they are consumed in the same order.
    constexpr int FE = 2;                     // fold entries per thread and chunk (B*K <= 2 * blockDim in one chunk)
    const bool fold = (MODE == 0) && D.fused_fold;
    const int bd = blockDim.x, tid = threadIdx.x;
    long long fv[FE][10];
    float fth[FE], fpr[FE];
    const long long* sold = fold ? D.Sold_fx + (size_t)j * nBK : nullptr;
    auto fold_issue = [&](int base) {
#pragma unroll
      for (int e = 0; e < FE; e++) {
        const int ic = min(base + tid + e * bd, nBK - 1);
#pragma unroll
        for (int r = 0; r < 8; r++) fv[e][r] = D.fold_prev[(size_t)min(r, D.nrep - 1) * nBK + ic];
        fv[e][8] = D.O_fx[ic]; fv[e][9] = sold[ic];
        const int b = ic / K;
        fth[e] = D.theta[b]; fpr[e] = D.Pr_b[b];
      }
    };
    auto fold_consume = [&](int base) {
#pragma unroll
      for (int e = 0; e < FE; e++) {
        const int i = base + tid + e * bd;
        if (i < nBK) {
          long long o = fv[e][8] - fv[e][9];
#pragma unroll
          for (int r = 0; r < 8; r++) if (r < D.nrep) o += fv[e][r];
          ldsO[i] = o;
          if (blockIdx.x == 0) {
            D.O_alt[i] = o;
            for (int r = 0; r < D.nrep; r++) D.fold_zero[(size_t)r * nBK + i] = 0;
          }
        }
      }
    };
    auto pen_entry = [&](int i, float th, float pr) {   // same arithmetic as k_foldpen (sharded path): tables must agree bitwise
      const int b = i / K, k = i - b * K;
      long long rs = 0;
      for (int b0 = 0; b0 < D.B0; b0++) rs += ldsO[b0 * K + k];
      const float of = (float)((double)ldsO[i] * FX_INV);
      const float ef = (float)(((double)rs * FX_INV) * (double)pr);
      ldsPen[i] = pen_pow((2.0f * ef) + 1.0f, of + ef + 1.0f, th);
    };
    if (fold) fold_issue(0);
    const int nQC = D.Q * C;
    int qlv0 = 0;
    if (fold) qlv0 = D.qlev[min(tid, nQC - 1)];
    if (ts < te) { cellN = tile_cell(ts); cellNN = tile_cell(ts + tstep); }   // first tiles' (cell, combination) pairs
    const f32x4* src = BF ? reinterpret_cast<const f32x4*>(D.Yimg3) : reinterpret_cast<const f32x4*>(D.Yimg);
    {   // first chunk straight-line (a loop header here would make hipcc drain the loads above before the first image load)
      f32x4 t[4];
#pragma unroll
      for (int k = 0; k < 4; k++) t[k] = src[min(tid + k * bd, nY4 - 1)];
#pragma unroll
      for (int k = 0; k < 4; k++) if (tid + k * bd < nY4) lds4[tid + k * bd] = t[k];
    }
    stamp(12);
    if (ts < te && pre) ld_rows(D.Zc + (size_t)(cellN.x >= 0 ? cellN.x : 0) * zs, rowsN);
    stamp(13);
    for (int base = 4 * bd; base < nY4; base += 4 * bd) {
      f32x4 t[4];
#pragma unroll
      for (int k = 0; k < 4; k++) t[k] = src[min(base + tid + k * bd, nY4 - 1)];
#pragma unroll
      for (int k = 0; k < 4; k++) if (base + tid + k * bd < nY4) lds4[base + tid + k * bd] = t[k];
    }
    if constexpr (MODE == 0) {
      if (D.fused_fold) {
        fold_consume(0);
        for (int base = FE * bd; base < nBK; base += FE * bd) { fold_issue(base); fold_consume(base); }
        if (tid < nQC) ldsQlev[tid] = qlv0;
        for (int i = tid + bd; i < nQC; i += bd) ldsQlev[i] = D.qlev[i];
        stamp(14);
        __syncthreads();
        stamp(15);
        if (nBK <= FE * bd) {
#pragma unroll
          for (int e = 0; e < FE; e++) if (tid + e * bd < nBK) pen_entry(tid + e * bd, fth[e], fpr[e]);
        } else {
          for (int i = tid; i < nBK; i += bd) pen_entry(i, D.theta[i / K], D.Pr_b[i / K]);
        }
      } else if (D.pen_lds) {
        for (int i = threadIdx.x; i < D.B * K; i += blockDim.x) ldsPen[i] = D.pen[i];
        for (int i = threadIdx.x; i < D.Q * C; i += blockDim.x) ldsQlev[i] = D.qlev[i];
      }
    }
    if constexpr (CHAIN) for (int i = threadIdx.x; i < D.Q * C; i += blockDim.x) ldsQlev[i] = D.qlev[i];
    if constexpr (MODE == 2) for (int i = threadIdx.x; i < K * D.d + K; i += blockDim.x) ltab[i] = 0;
    __syncthreads();
  }
  stamp(2);
  const float* penT = ((MODE == 0 && (D.pen_lds || D.fused_fold)) || CHAIN) ? ldsPen : D.pen;
  const int* qlevT = ((MODE == 0 && (D.pen_lds || D.fused_fold)) || CHAIN) ? ldsQlev : D.qlev;
  long long* snew = D.Snew_fx + (size_t)(wave & (D.nrep - 1)) * D.B * K;  // this wave's table replica
  // per-lane cluster constants: exp(-dist/sigma) = exp2(dist * ce), ce = -log2(e)/sigma;  sigma r ln r = cl r log2 r,
  // cl = sigma ln 2;  lpen = log2(penalty of the current combination), clp = cl * lpen (general sigma only).
  constexpr int NSIG = USIG ? 1 : NCT;
  float ce[NSIG], cl[NSIG], clp[NSIG], lpen[NCT];
  unsigned long long oacc[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) {
    const bool kv = kcol(NCT, ct, c) < K;
    const size_t ks = (size_t)min(kcol(NCT, ct, c), K - 1);
    if (ct < NSIG) {
      if constexpr (MODE == 2) { ce[ct] = ld_or(D.ynorm, ks, kv, 0.0f); cl[ct] = 0.0f; }
      else if constexpr (USIG) { ce[ct] = D.ce[0]; cl[ct] = D.cl[0]; }
      else { ce[ct] = ld_or(D.ce, ks, kv, 0.0f); cl[ct] = ld_or(D.cl, ks, kv, 0.0f); }
      clp[ct] = 0.0f;
    }
    lpen[ct] = 0.0f; oacc[ct] = 0ull;
  }
  // MODE 3: per-lane running minima of the packed (key bits, global cell) race values and the per-anchor hash keys
  unsigned long long best[MODE == 3 ? NCT : 1], sk[MODE == 3 ? NCT : 1];
  if constexpr (MODE == 3) {
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
      best[ct] = ~0ull;
      sk[ct] = splitmix64(D.seed_key ^ ((uint64_t)(1 + kcol(NCT, ct, c)) * 0xD1342543DE82EF95ull));
    }
  }
  auto CE = [&](int ct) -> float { return ce[USIG ? 0 : ct]; };
  auto CL = [&](int ct) -> float { return cl[USIG ? 0 : ct]; };
  double od = 0.0, oe = 0.0;
  int curq = -1;
  // ---- per-tile stages ------------------------------------------------------------------------------------------
  // the tile's combination: slot 0 of a tile is always a real cell (static tiles: every lane holds it)
  auto tile_q = [&](const int2 cq) -> int { return __builtin_amdgcn_readfirstlane(cq.y); };
  const RowRegs* erows = nullptr;   // MODE 2: the A-operand registers of the tile whose epilogue runs (single-accumulator loop)
  // MODE 0/1 epilogue, split so that the fused loop below can interleave it with the next tile's MFMAs:
  // epi_begin: run change -> flush the O contributions of the finished combination, fetch the new penalty row
  // A tile's combination word (lpair.y) when the shuffle keyed the tiles by it (D.nxt): bits 0..18 the covariate combination,
  // bits 19..24 the block its cells belong to in the NEXT round, bits 25..30 their block in THIS round.  The R rows a block update
  // writes are at the same time the cells' old contribution to their next block (D.Sold_next), and the R rows the head writes --
  // run over this round's order -- are their old contribution to this round's blocks (D.Sold_head): the pass over R that used to
  // collect them (k_oldsum) disappears.
  const int QMASK = D.qmask;      // 0x7FFFF when shuffles may be keyed (then Q < 2^19), else all bits (Q up to 2^24)
  auto flush_run = [&]() __attribute__((always_inline)) {
    long long* t2 = nullptr;
    if (UPD && D.Sold_next) t2 = D.Sold_next + (size_t)((curq >> 19) & 63) * D.B * K;
    if (MODE == 1 && D.head_gather && D.Sold_head) t2 = D.Sold_head + (size_t)((curq >> 25) & 63) * D.B * K;
#ifdef HMX_TRACE
    if (D.upd_debug & 16) t2 = nullptr;                   // timing experiments (WRONG RESULTS): no carry atomics | no contribution atomics at all
    if (D.upd_debug & 8) {
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) oacc[ct] = 0ull;
      return;
    }
#endif
    flush_tile_fx<NCT>(snew, t2, qlevT, curq & QMASK, C, K, c, g, oacc);
  };
  auto epi_begin = [&](const int q0) __attribute__((always_inline)) {
    if (q0 != curq) {
      if (curq >= 0) flush_run();
      curq = q0;
      if constexpr (UPD) {
        float penv[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) penv[ct] = 0.0f;
        for (int cc = 0; cc < C; cc++) {  // penalty of a cell = SUM over its covariates (:322 is a matrix product)
          const int b = qlevT[(q0 & QMASK) * C + cc];
          const float* __restrict__ pr = penT + (size_t)b * K;
          if (D.rvec) {
            // a lane's clusters are consecutive (kcol): one 16-byte read per quad of cluster tiles, ALL reads of the level in
            // flight before the first is used (one latency per level instead of one per cluster tile)
            constexpr int NFULL = NCT >> 2, RT = NCT & 3;
            f32x4 vq[NFULL > 0 ? NFULL : 1];
            float vt[RT > 0 ? RT : 1];
#pragma unroll
            for (int q = 0; q < NFULL; q++) vq[q] = *reinterpret_cast<const f32x4*>(pr + min(64 * q + 4 * c, K - 4));
#pragma unroll
            for (int jj = 0; jj < RT; jj++) vt[jj] = pr[min(64 * NFULL + RT * c + jj, K - 1)];
#pragma unroll
            for (int q = 0; q < NFULL; q++) {
              const bool ok = 4 * q < first_partial_ct(NCT) || 64 * q + 4 * c < K;
#pragma unroll
              for (int jj = 0; jj < 4; jj++) penv[4 * q + jj] += ok ? vq[q][jj] : 0.0f;
            }
#pragma unroll
            for (int jj = 0; jj < RT; jj++) penv[4 * NFULL + jj] += (64 * NFULL + RT * c + jj < K) ? vt[jj] : 0.0f;
          } else {
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) penv[ct] += ld_or(penT, (size_t)b * K + min(kcol(NCT, ct, c), K - 1), kcol(NCT, ct, c) < K, 0.0f);
          }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) { lpen[ct] = __builtin_amdgcn_logf(fmaxf(penv[ct], FLT_MIN)); if constexpr (!USIG) clp[ct] = cl[ct] * lpen[ct]; }
      }
    }
  };
  // put_row: one R row (register `reg` of every cluster tile) -> memory.  A lane's columns are consecutive clusters (kcol): one
  // 16-byte store per quad of cluster tiles and one 12/8/4-byte store for the rest -- 2 store instructions per row at K = 100
  // instead of 7 (the epilogue is store-ISSUE-bound).  Rows are 16-byte aligned when K % 4 == 0 (D.rvec).
  auto put_row = [&](float* __restrict__ row, const f32x4 (&acc)[NCT], const int reg) __attribute__((always_inline)) {
    constexpr int NFULL = NCT >> 2, RT = NCT & 3;
    if (D.rvec) {
#pragma unroll
      for (int q = 0; q < NFULL; q++) {
        const int k0 = 64 * q + 4 * c;
        if (4 * q < first_partial_ct(NCT) || k0 < K) {      // (K % 4 == 0: a lane's four clusters are all inside K or all outside)
          f32x4 v; v[0] = acc[4 * q][reg]; v[1] = acc[4 * q + 1][reg]; v[2] = acc[4 * q + 2][reg]; v[3] = acc[4 * q + 3][reg];
          *reinterpret_cast<f32x4*>(row + k0) = v;
        }
      }
      if constexpr (RT > 0) {
        const int k0 = 64 * NFULL + RT * c;
        if (k0 + RT <= K) {
          if constexpr (RT == 3) { F3 v; v.x = acc[4 * NFULL][reg]; v.y = acc[4 * NFULL + 1][reg]; v.z = acc[4 * NFULL + 2][reg]; *reinterpret_cast<F3*>(row + k0) = v; }
          else if constexpr (RT == 2) { F2 v; v.x = acc[4 * NFULL][reg]; v.y = acc[4 * NFULL + 1][reg]; *reinterpret_cast<F2*>(row + k0) = v; }
          else row[k0] = acc[4 * NFULL][reg];
        } else {
#pragma unroll
          for (int jj = 0; jj < RT; jj++) if (k0 + jj < K) row[k0 + jj] = acc[4 * NFULL + jj][reg];
        }
      }
    } else {
#pragma unroll
      for (int ct = 0; ct < NCT; ct++)
        if (ct < first_partial_ct(NCT) || kcol(NCT, ct, c) < K) row[kcol(NCT, ct, c)] = acc[ct][reg];
    }
  };
  // epi_rows: the four accumulator registers (rows 4g..4g+3 of the tile; their cell ids live in lanes 4g+reg of cellA)
  // TOGETHER, stage by stage: a gfx950 wave issues a dependent VALU instruction only every ~26 cycles (measured,
  // tools/ubench/issue.hip) and few waves share a SIMD here, so the code must carry its own instruction-level
  // parallelism -- up to 28 independent values per stage instead of one row after the other.
  //   t_k   = x_k ce_k + log2 pen_k,  e_k = exp2(t_k)      x_k = 2 - 2 z.y_k                   (:141-150, :318-322)
  //   r_k   = e_k / sum_k e_k                   (ONE L1 normalisation: the reference's first one cancels, :323-326)
  //   sum_k r_k x_k              = inv * sum e_k x_k                                       (objective_kmeans_dist, :160)
  //   sum_k sigma_k r_k ln r_k   = inv * sum cl_k e_k (t_k + log2 inv)                     (entropy, :161; log2 r_k = t_k + log2 inv)
  //        uniform sigma:  = inv cl (ce sum e_k x_k + sum e_k lpen_k + log2(inv) sum e_k)
  //        general sigma:  = inv (sum e_k clp_k - sum e_k x_k + log2(inv) sum e_k cl_k)     (cl_k ce_k = -1)
  //   -- no logarithm per value, one per row
  // DEFER (std::true_type, MODE 4's last tile of a block): the normalised values replace `acc` instead of being stored, so that the
  // O contributions can be flushed BEFORE the tile's 28 stores are issued (store_rows) -- the arrival then waits for the
  // atomics only, not for the block's R rows to reach HBM (vector memory operations retire in issue order on gfx9).
  auto epi_rows = [&](const int cellA, f32x4 (&acc)[NCT], auto defer_tag) __attribute__((always_inline)) {
    constexpr bool DEFER = decltype(defer_tag)::value;
    constexpr int RB = (NCT <= 7 && !LEAN) ? 4 : 2;   // rows per batch: all four while the registers last
#pragma unroll
    for (int r0 = 0; r0 < 4; r0 += RB) {
      float se[RB], sx[RB], sp[RB], sc[RB];   // sums of e, e x, e lpen (or e clp), e (lane local) (or e cl)
      float* Rrow[RB];
      bool cv[RB];
#pragma unroll
      for (int i = 0; i < RB; i++) {
        const int cell = __shfl(cellA, 4 * g + r0 + i, 64);
        cv[i] = cell >= 0;
        Rrow[i] = D.R + (size_t)(cv[i] ? cell : D.n) * K;       // invalid rows: the dummy row behind R (zeros)
        se[i] = 0.0f; sx[i] = 0.0f; sp[i] = 0.0f; sc[i] = 0.0f;
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) {
#pragma unroll
        for (int i = 0; i < RB; i++) {
          const float x = fmaf(acc[ct][r0 + i], -2.0f, 2.0f);
          float e = __builtin_amdgcn_exp2f(UPD ? fmaf(x, CE(ct), lpen[ct]) : x * CE(ct));
          if (ct >= first_partial_ct(NCT)) e = (kcol(NCT, ct, c) < K) ? e : 0.0f;
          acc[ct][r0 + i] = e;
          se[i] += e;
          sx[i] = fmaf(e, x, sx[i]);
          if constexpr (UPD) sp[i] = fmaf(e, USIG ? lpen[ct] : clp[USIG ? 0 : ct], sp[i]);
          if constexpr (!USIG) sc[i] = fmaf(e, CL(ct), sc[i]);
        }
      }
      if constexpr (USIG) {
#pragma unroll
        for (int i = 0; i < RB; i++) sc[i] = se[i];   // this lane's own sum, before the row reduction
      }
      rowsum16xN<RB>(se);
      float inv[RB], pd = 0.0f, pe = 0.0f;
#pragma unroll
      for (int i = 0; i < RB; i++) {
        float i2 = __builtin_amdgcn_rcpf(se[i]);
        i2 = i2 * fmaf(-se[i], i2, 2.0f);               // one Newton step: <= 1 ulp
        i2 = (se[i] == 0.0f) ? 1.0f : i2;
        const float linv = __builtin_amdgcn_logf(i2);
        i2 = cv[i] ? i2 : 0.0f;                          // padding rows contribute exactly nothing
        inv[i] = i2;
        pd = fmaf(i2, sx[i], pd);
        if constexpr (USIG) pe = fmaf(i2 * CL(0), fmaf(CE(0), sx[i], fmaf(linv, sc[i], sp[i])), pe);
        else pe = fmaf(i2, fmaf(linv, sc[i], sp[i] - sx[i]), pe);
      }
      od += (double)pd; oe += (double)pe;
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) {
        unsigned s32 = 0u;                 // RB <= 4 values of at most 2^29 each: no carry out of 32 bits
#pragma unroll
        for (int i = 0; i < RB; i++) {
          const float rn = acc[ct][r0 + i] * inv[i];
          acc[ct][r0 + i] = rn;            // the normalised value replaces the distance (stored below, or later by store_rows)
          s32 += fx32_of(rn);
        }
        oacc[ct] += (unsigned long long)s32;
      }
      if constexpr (!DEFER) {
#ifdef HMX_TRACE
        if (!(D.upd_debug & 4))   // timing experiment: no R stores
#endif
        if (!NOSTORE && D.r_store) {    // (a pass whose R rows nobody will read leaves them in the registers: see Dev::r_store)
#pragma unroll
          for (int i = 0; i < RB; i++) put_row(Rrow[i], acc, r0 + i);
        }
      }
    }
  };
  // the deferred stores of epi_rows<DEFER>: `acc` holds the normalised rows
  auto store_rows = [&](const int cellA, const f32x4 (&acc)[NCT]) __attribute__((always_inline)) {
#ifdef HMX_TRACE
    if (D.upd_debug & 4) return;
#endif
    if (NOSTORE || !D.r_store) return;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int cell = __shfl(cellA, 4 * g + i, 64);
      put_row(D.R + (size_t)(cell >= 0 ? cell : D.n) * K, acc, i);
    }
  };
  // epilogue of a tile whose distances are in `acc`
  auto epilogue = [&](const int cellA, const int q0, f32x4 (&acc)[NCT]) __attribute__((always_inline)) {
    if constexpr (MODE == 2) {
      // nearest centre of every cell of the tile: argmin_k ||y_k||^2 - 2 x.y_k ; ties -> smallest k.  All four rows
      // together (ILP), the 16-lane reductions by DPP: first the minimum score, then the smallest k that attains it.
      float bs[4]; int bk[4];
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        bs[reg] = INFINITY; bk[reg] = 0x7fffffff;
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) {
          const float sc = (ct < first_partial_ct(NCT) || kcol(NCT, ct, c) < K) ? fmaf(acc[ct][reg], -2.0f, ce[ct]) : INFINITY;
          const bool lt = sc < bs[reg];               // strict: the smaller k (ct ascending) wins a tie
          bs[reg] = lt ? sc : bs[reg];
          bk[reg] = lt ? kcol(NCT, ct, c) : bk[reg];
        }
      }
      float m[4];
#pragma unroll
      for (int reg = 0; reg < 4; reg++) m[reg] = bs[reg];
#pragma unroll
      for (int st = 0; st < 4; st++) {
        float t[4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) t[reg] = (st == 0) ? dpp_f(m[reg], 0) : (st == 1) ? dpp_f(m[reg], 1) : (st == 2) ? dpp_f(m[reg], 2) : dpp_f(m[reg], 3);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) m[reg] = fminf(m[reg], t[reg]);
      }
      int kb[4];
#pragma unroll
      for (int reg = 0; reg < 4; reg++) kb[reg] = (bs[reg] == m[reg]) ? bk[reg] : 0x7fffffff;
#pragma unroll
      for (int st = 0; st < 4; st++) {
        int t[4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const float kf = __int_as_float(kb[reg]);
          t[reg] = __float_as_int((st == 0) ? dpp_f(kf, 0) : (st == 1) ? dpp_f(kf, 1) : (st == 2) ? dpp_f(kf, 2) : dpp_f(kf, 3));
        }
#pragma unroll
        for (int reg = 0; reg < 4; reg++) kb[reg] = min(kb[reg], t[reg]);
      }
      // the 16 lanes of a row group add the row's PCs (2^30 fixed point) to the LDS table of its centre: ALL loads of the
      // tile are issued before the first is used (they hit L1/L2 -- the rows were just read as MFMA operands)
      const int dd = D.d;          // locals: after the first LDS atomic hipcc would re-read D's fields from a spilled copy
      const float* const Zcp = D.Zc;
      if (erows) {
        // the rows are still in the A-operand registers (lane = cell l & 15, k-slot l >> 4): no second read of the tile.  The
        // centre of THIS lane's cell sits in the row group (l & 15) >> 2, register (l & 15) & 3.
        int t4[4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) t4[reg] = __shfl(kb[reg], 16 * (c >> 2), 64);
        const int kc = ((c & 3) == 0) ? t4[0] : ((c & 3) == 1) ? t4[1] : ((c & 3) == 2) ? t4[2] : t4[3];
        if (cellA >= 0) {
          long long* row = ltab + (size_t)kc * dd;
          if constexpr (BF) {      // split-bf16 row layout: v[2 s + h][e] = PC 32 s + 8 g + 4 h + e
#pragma unroll
            for (int t = 0; t < 4; t++) {
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const int jj = 32 * (t >> 1) + 8 * g + 4 * (t & 1) + e;
                if (t < 2 * D.NS2 && jj < dd) atomicAdd((unsigned long long*)&row[jj], (unsigned long long)(long long)__float2int_rn(erows->v[t][e] * 1073741824.0f));
              }
            }
            if (g == 0) atomicAdd((unsigned long long*)&ltab[K * dd + kc], 1ull);
            return;
          }
#pragma unroll
          for (int t = 0; t < 4; t++) {
            if (t < D.NT4) {
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const int jj = 16 * t + 4 * g + e;
                if (jj < dd) atomicAdd((unsigned long long*)&row[jj], (unsigned long long)(long long)__float2int_rn(erows->v[t][e] * 1073741824.0f));
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 3; u++) {
            const int jj = 16 * D.NT4 + 4 * u + g;
            if (u < D.tail && jj < dd) atomicAdd((unsigned long long*)&row[jj], (unsigned long long)(long long)__float2int_rn(erows->t[u] * 1073741824.0f));
          }
          if (g == 0) atomicAdd((unsigned long long*)&ltab[K * dd + kc], 1ull);
        }
        return;
      }
      int cellr[4];
#pragma unroll
      for (int reg = 0; reg < 4; reg++) cellr[reg] = __shfl(cellA, 4 * g + reg, 64);
      for (int j0 = 0; j0 < dd; j0 += 64) {
        float z[4][4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const float* zr = Zcp + (size_t)max(cellr[reg], 0) * zs;
#pragma unroll
          for (int u = 0; u < 4; u++) z[reg][u] = zr[min(j0 + 16 * u + c, dd - 1)];
        }
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int jj = j0 + 16 * u + c;
            if (cellr[reg] >= 0 && jj < dd) {
              // |z| <= 1 (normalised rows): the 2^30 fixed-point value fits int32 -> rndne + cvt + sign extension
              const unsigned long long v = (unsigned long long)(long long)__float2int_rn(z[reg][u] * 1073741824.0f);
              atomicAdd((unsigned long long*)&ltab[kb[reg] * dd + jj], v);
            }
          }
        }
      }
#pragma unroll
      for (int reg = 0; reg < 4; reg++)
        if (c == 0 && cellr[reg] >= 0) atomicAdd((unsigned long long*)&ltab[K * dd + kb[reg]], 1ull);
    } else if constexpr (MODE == 3) {
      // same arithmetic per (cell, anchor) as k_seed_probe: u from splitmix64(anchor key + global cell), key = -log(u) / dist
      int gc[4]; bool ok[4];
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int cell = __shfl(cellA, 4 * g + reg, 64);
        ok[reg] = cell >= 0;
        gc[reg] = D.perm[max(cell, 0)];
      }
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const uint64_t gg = D.seed_goff + (uint64_t)gc[reg];
        bool skip = !ok[reg];
        for (int x = 0; x < D.seed_nexcl; x++) skip |= ((uint64_t)D.seed_excl[x] == gg);   // re-probe passes only
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) {
          const float dis = fabsf(2.0f * (1.0f - acc[ct][reg]));
          const uint64_t h = splitmix64(sk[ct] + gg);
          const float u = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
          const float key = -logf(u) / dis;  // >= 0 (or +inf / nan when dis == 0)
          unsigned kb = __float_as_uint(key);
          if (!(key >= 0.0f)) kb = 0x7f800000u;
          kb &= 0x7fffffffu;   // -0.0 (u == 1: -log(u) = -0) must order as zero, not as the largest unsigned pattern  // nan -> +inf: never the minimum
          const unsigned long long pk = ((unsigned long long)kb << 32) | (unsigned long long)(uint32_t)gg;
          const bool take = !skip && (ct < first_partial_ct(NCT) || kcol(NCT, ct, c) < K) && pk < best[ct];
          best[ct] = take ? pk : best[ct];
        }
      }
    } else {
      epi_begin(q0);
      epi_rows(cellA, acc, std::false_type{});
    }
  };
  constexpr bool DUAL = !LEAN && NCT <= 7;  // two accumulator sets fit the 256-VGPR budget (2 waves/SIMD) only up to K = 112
  auto next_rows = [&](const int2 nxt, const int2 cur) {  // row address of the NEXT tile's lane (any valid row if padding)
    return D.Zc + (size_t)(nxt.x >= 0 ? nxt.x : (cur.x >= 0 ? cur.x : 0)) * zs;
  };
  if constexpr (CHAIN) {
    // ======================= persistent block chain of one round =======================
    // Cross-workgroup traffic (L2s of different XCDs are not coherent, so nothing here relies on plain loads of data another
    // workgroup wrote during this launch):
    //   Snew replicas   workers: device-scope atomic adds;  folder: atomic exchange with 0 (read + reset in one RMW)
    //   ctl[2 + j]      arrivals of block j (atomic add, after the wave's vmcnt(0) drained its contribution atomics)
    //   pen_g[i]        8-byte granules { tag << 32 | penalty bits } written through with ONE sc1 store each and read
    //                   with sc1 loads until the tag is this block's (self-validating: no ordering assumed)
    //   ctl[0]          block flag (tag), stored after the granules drained; one lane per workgroup polls it
    // Every spin is bounded: on a timeout ctl[1] is raised and the launch runs out with garbage instead of hanging the GPU.
    int* const ctl = D.chain_ctl;
    unsigned long long* const peng = D.pen_g;
    const unsigned tag0 = D.chain_tag;
    const int nbk = D.nb, nworkWG = (int)gridDim.x - 1, bd = blockDim.x, tid = threadIdx.x;
    constexpr int SPIN_LIMIT = 1 << 20;     // ~1 s of polling; once ANY spin has timed out (ctl[1] != 0) the others give up at once
    auto dead = [&](int spins) -> bool { return (spins & 255) == 255 && __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; };
    if ((int)blockIdx.x == nworkWG) {
      // ---------------- the folder: O' = O + new(block j-1) - old(block j), E, penalty table of block j (:312-322,:329-330)
      for (int i = tid; i < nBK; i += bd) ldsO[i] = D.O_fx[i];
      __syncthreads();
      // cluster masses rs[k] = sum over the first covariate's levels of O[.,k] (E = rs Pr_b^T), kept up to date entry by entry in
      // the fold instead of re-summed B0 times per entry in the publish step.  (The folder never reads the centroid image: its
      // LDS region holds the masses.)
      long long* const ldsRS = reinterpret_cast<long long*>(lds4);
      for (int k = tid; k < K; k += bd) {
        long long rs = 0;
        for (int b0 = 0; b0 < D.B0; b0++) rs += ldsO[b0 * K + k];
        ldsRS[k] = rs;
      }
      __syncthreads();
      const int nRS = D.B0 * K;
      constexpr int FE = 4;
      unsigned long long tw = 0, tf = 0, tp = 0, t_prev = wall_clock64();   // diagnostics: wait / fold / publish time of the folder
      for (int jj = 0; jj <= nbk; jj++) {
        long long sv[FE];
#pragma unroll
        for (int e = 0; e < FE; e++) sv[e] = (jj < nbk) ? D.Sold_fx[(size_t)jj * nBK + min(tid + e * bd, nBK - 1)] : 0;   // before the wait
        if (jj > 0) {
          if (tid == 0) {
            int spins = 0;
            // arrivals are sharded over 8 counters (workgroup b -> counter b & 7: one per XCD under the observed placement);
            // 255 increments of ONE word serialise at ~12 ns each
            const int* arr = &ctl[8 + 8 * (jj - 1)];
            auto arrived = [&]() { int t = 0; for (int x = 0; x < 8; x++) t += __hip_atomic_load(&arr[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return t; };
            while (arrived() < nworkWG) {
              __builtin_amdgcn_s_sleep(1);
              if (++spins > SPIN_LIMIT) { atomicExch(&ctl[1], 1); break; }
              if (dead(spins)) break;
            }
          }
          __syncthreads();
        }
        { const unsigned long long t = wall_clock64(); tw += t - t_prev; t_prev = t; }
        if (D.p2p_world > 1) {
          // Sharded run: the new contributions of block jj-1 are summed over the GPUs INSIDE the launch.  One-shot exchange over
          // xGMI: every folder writes its local sums straight into every peer's inbox (two self-validating 8-byte granules
          // {tag, half of the int64} per entry, system-scope write-through stores), then sums what the peers wrote into its own.
          // Integer sums: every rank folds exactly the same O.  Two parities suffice: a rank can run at most one exchange ahead
          // of a peer (it needs that peer's contribution of step jj to finish step jj).
          const int G = D.p2p_world, me = D.p2p_rank;
          const unsigned tagx = tag0 + (unsigned)jj;
          // (round 4) EVERY step exchanges, jj = 0 included, and what travels is  new(jj - 1) - old_local(jj): the old contributions are filed
          // rank-locally by the previous round's tile kernels and are summed over the ranks right here -- the per-round all-reduce of the
          // nb x K x B table (24 host-launched collectives per run) is gone.  The plane alternates with the exchange NUMBER, carried across
          // rounds (nb + 1 exchanges per round: with jj & 1 two consecutive exchanges of neighbouring rounds would share a plane).
          const size_t par = (size_t)((D.chain_xseq + (unsigned)jj) & 1u) * 8;
          for (int base = 0; base < nBK; base += FE * bd) {
            long long dl[FE];
#pragma unroll
            for (int e = 0; e < FE; e++) {
              const int i = base + tid + e * bd;
              dl[e] = 0;
              if (i < nBK) {
                long long so = 0;
                if (jj < nbk) so = D.Sold_fx[(size_t)jj * nBK + i];
                if (jj > 0) {
                  unsigned long long a[8];
#pragma unroll
                  for (int r = 0; r < 8; r++) a[r] = (r < D.nrep) ? atomicExch((unsigned long long*)&D.Snew_fx[(size_t)r * nBK + i], 0ull) : 0ull;
#pragma unroll
                  for (int r = 0; r < 8; r++) dl[e] += (long long)a[r];
                }
                dl[e] -= so;
                p2p_send(D, par, i, tagx, dl[e]);
              }
            }
#pragma unroll
            for (int e = 0; e < FE; e++) {
              const int i = base + tid + e * bd;
              if (i < nBK) {
                long long o = ldsO[i] + dl[e];
                {
                  unsigned long long lo[8], hi[8];
#pragma unroll
                  for (int gq = 0; gq < 8; gq++) if (gq < G && gq != me) {
                    const unsigned long long* src = D.p2p_inbox_self() + ((par + gq) * P2P_CAP + i) * 2;
                    lo[gq] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    hi[gq] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                  }
#pragma unroll
                  for (int gq = 0; gq < 8; gq++) if (gq < G && gq != me) {
                    const unsigned long long* src = D.p2p_inbox_self() + ((par + gq) * P2P_CAP + i) * 2;
                    int spins = 0;
                    while ((unsigned)(lo[gq] >> 32) != tagx || (unsigned)(hi[gq] >> 32) != tagx) {
                      __builtin_amdgcn_s_sleep(1);
                      if (++spins > SPIN_LIMIT) { atomicExch(&ctl[1], 5); break; }
                      if (dead(spins)) break;
                      lo[gq] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                      hi[gq] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    o += (long long)((hi[gq] << 32) | (lo[gq] & 0xffffffffull));
                  }
                }
                if (i < nRS && o != ldsO[i]) atomicAdd((unsigned long long*)&ldsRS[i % K], (unsigned long long)(o - ldsO[i]));
                ldsO[i] = o;
                if (jj == nbk) D.O_fx[i] = o;
              }
            }
          }
        }
        auto fold_entry = [&](int i, long long soldv) {
          long long o = ldsO[i];
          // every memory operation of the entry in flight before the first is consumed: the new contributions (exchange = read + reset)
          unsigned long long a[8];
#pragma unroll
          for (int r = 0; r < 8; r++) a[r] = (jj > 0 && r < D.nrep) ? atomicExch((unsigned long long*)&D.Snew_fx[(size_t)r * nBK + i], 0ull) : 0ull;
#pragma unroll
          for (int r = 0; r < 8; r++) o += (long long)a[r];
          if (jj < nbk) o -= soldv;
          if (i < nRS && o != ldsO[i]) atomicAdd((unsigned long long*)&ldsRS[i % K], (unsigned long long)(o - ldsO[i]));
          ldsO[i] = o;
          if (jj == nbk) D.O_fx[i] = o;      // the round's final O (read by the kernels that follow this launch)
        };
        if (D.p2p_world <= 1) {
#pragma unroll
          for (int e = 0; e < FE; e++) { const int i = tid + e * bd; if (i < nBK) fold_entry(i, sv[e]); }
          for (int i = tid + FE * bd; i < nBK; i += bd) fold_entry(i, jj < nbk ? D.Sold_fx[(size_t)jj * nBK + i] : 0);
        }
        if (jj == nbk) break;
        __syncthreads();
        { const unsigned long long t = wall_clock64(); tf += t - t_prev; t_prev = t; }
        const unsigned long long tagbits = (unsigned long long)(tag0 + (unsigned)jj) << 32;
        for (int i = tid; i < nBK; i += bd) {      // same arithmetic as k_foldpen / the fused prologue: identical tables
          const int b = i / K, k = i - b * K;
          const long long rs = ldsRS[k];
          const float of = (float)((double)ldsO[i] * FX_INV);
          const float ef = (float)(((double)rs * FX_INV) * (double)D.Pr_b[b]);
          const float pv = pen_pow((2.0f * ef) + 1.0f, of + ef + 1.0f, D.theta[b]);
          __hip_atomic_store(&peng[i], tagbits | (unsigned long long)__float_as_uint(pv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the flag needs no ordering against the granules (they validate themselves by tag): raise it at once
        if (tid == 0) __hip_atomic_store(&ctl[0], (int)(tag0 + (unsigned)jj), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        { const unsigned long long t = wall_clock64(); tp += t - t_prev; t_prev = t; }
      }
      if (tid == 0 && D.chain_dbg) { atomicAdd(&D.chain_dbg[0], tw); atomicAdd(&D.chain_dbg[1], tf); atomicAdd(&D.chain_dbg[2], tp); atomicAdd(&D.chain_dbg[3], 1ull); }
      if (D.chain_tail) {
        // ---- the round's tail, by the folder (compute_objective, src/harmony.cpp:158-170): the workers' per-wave objective sums, the
        // cross-entropy term from the O / E tables this workgroup holds in LDS anyway, the snapshot straight into the pinned host slot;
        // then the tables this round consumed are cleared and the control words reset.  (k_round_tail does the same after the
        // launch-per-step paths; here it would be one more launch + ~60 us of gap behind every round.)
        if (tid == 0) {
          int spins = 0;
          const int* arr = &ctl[16 + 8 * nbk];
          auto arrived = [&]() { int t = 0; for (int x = 0; x < 8; x++) t += __hip_atomic_load(&arr[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return t; };
          while (arrived() < nworkWG) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { atomicExch(&ctl[1], 6); break; }
            if (dead(spins)) break;
          }
        }
        __syncthreads();
        double* const red = reinterpret_cast<double*>(lds4) + ((K + 1) & ~1);      // [waves][3], behind the cluster masses (the centroid image's space: the folder never stages it)
        const int nwv = nworkWG * (bd >> 6);
        double pa = 0.0, pb = 0.0;
        for (int w2 = tid; w2 < nwv; w2 += bd) {       // fixed order per thread, fixed order of the reduction below: deterministic
          pa += __hip_atomic_load(&D.objpart[2 * w2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pb += __hip_atomic_load(&D.objpart[2 * w2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&D.objpart[2 * w2], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&D.objpart[2 * w2 + 1], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double cr = 0.0;
        for (int k = tid; k < K; k += bd) {              // (same arithmetic as k_objective_tables)
          const double rsd = (double)ldsRS[k] * FX_INV;
          double ck = 0.0;
          for (int bb = 0; bb < D.B; bb++) {
            const double od = (double)ldsO[bb * K + k] * FX_INV;
            const float o = (float)od, e = (float)(rsd * (double)D.Pr_b[bb]);
            const float m = D.theta[bb] * logf((o + e + 1.0f) / ((2.0f * e) + 1.0f));
            ck += od * (double)m;
          }
          cr += ck * (double)D.sigma[k];
        }
        pa = wsumd(pa); pb = wsumd(pb); cr = wsumd(cr);
        __syncthreads();                                  // (every wave is done reading the masses next to `red`)
        if (lane == 0) { red[3 * (tid >> 6)] = pa; red[3 * (tid >> 6) + 1] = pb; red[3 * (tid >> 6) + 2] = cr; }
        __syncthreads();
        if (tid == 0) {
          double sa = 0.0, sb = 0.0, sc = 0.0;
          for (int w2 = 0; w2 < (bd >> 6); w2++) { sa += red[3 * w2]; sb += red[3 * w2 + 1]; sc += red[3 * w2 + 2]; }
          if (D.p2p_world > 1) {
            // sharded: the two per-cell sums of every rank meet in the inboxes as well (one more exchange of the round, entries behind the
            // K x B table) and are added in RANK ORDER -- identical objective values on every rank, no host-launched collective per round
            // (the cross-entropy term comes from the global O / E tables: the same on every rank already)
            const int G = D.p2p_world, me = D.p2p_rank;
            const unsigned tagt = tag0 + (unsigned)nbk + 1u;
            const size_t part = (size_t)((D.chain_xseq + (unsigned)nbk + 1u) & 1u) * 8;
            p2p_send(D, part, nBK, tagt, __double_as_longlong(sa));
            p2p_send(D, part, nBK + 1, tagt, __double_as_longlong(sb));
            double va[8], vb[8];
#pragma unroll
            for (int gq = 0; gq < 8; gq++) {
              va[gq] = sa; vb[gq] = sb;
              if (gq < G && gq != me) {
                for (int which = 0; which < 2; which++) {
                  const unsigned long long* src = D.p2p_inbox_self() + ((part + gq) * P2P_CAP + nBK + which) * 2;
                  unsigned long long lo = 0, hi = 0;
                  int spins = 0;
                  for (;;) {
                    lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((unsigned)(lo >> 32) == tagt && (unsigned)(hi >> 32) == tagt) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) { atomicExch(&ctl[1], 5); break; }
                    if (dead(spins)) break;
                  }
                  const double x = __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
                  if (which == 0) va[gq] = x; else vb[gq] = x;
                }
              }
            }
            sa = 0.0; sb = 0.0;
            for (int gq = 0; gq < G; gq++) { sa += va[gq]; sb += vb[gq]; }
          }
          const double err = (double)__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + ((D.solve_err && *D.solve_err) ? 16.0 : 0.0);
          D.obj[0] = sa; D.obj[1] = sb; D.obj[2] = sa; D.obj[3] = sb; D.obj[4] = sc; D.obj[5] = err;
          if (D.tail_host_slot) {     // pinned host memory, mapped into the device: visible to the host once the event behind this launch completed
            __hip_atomic_store(&D.tail_host_slot[0], sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&D.tail_host_slot[1], sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&D.tail_host_slot[2], sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&D.tail_host_slot[3], err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        for (unsigned long long i = tid; i < D.tail_n0; i += bd) D.tail_z0[i] = 0;
        for (unsigned long long i = tid; i < D.tail_n1; i += bd) D.tail_z1[i] = 0;
        __syncthreads();
        for (int i = tid; i < 8 * nbk + 24; i += bd) ctl[i] = 0;
      }
      return;
    }
    // ---------------- the workers
    {
    static_assert(!LEAN, "the chain runs two waves per SIMD (the 3- and 4-wave variants lost rounds 2 and 3 and were removed in round 5)");
    f32x4 accC[NCT], accS[NCT];
    int2 cellC = make_int2(-1, -1), cellS = make_int2(-1, -1);
    bool have = ts < te, have2 = false;
    // MODE 5 (no R stores: the accumulators are dead once a tile's contributions are summed): the rows of the wave's SECOND tile of the next
    // block are requested into a second register buffer in front of the contribution flush -- their latency passes during the flush, the
    // barrier and the arrival instead of between the two tiles' MFMAs behind it (the workers' phase in the folder's shadow: 5.0 us, the
    // folder needs 3-4).  With the stores the normalised rows of both tiles occupy the accumulators until behind the arrival: no room
    // (a second buffer there: 28 spilled VGPRs).
    RowRegs rowsN2; bool rows2_ok = false;
    // BOTH accumulator sets are filled ahead of the flag: the MFMAs of this wave's first tile of the current block (rows were
    // requested earlier) and, if it owns a second one, of that too -- after the flag only epilogues remain for up to two tiles
    auto first_tile_a = [&]() __attribute__((always_inline)) {
      cellC = cellN;
      const RowRegs rowsA = rowsN;
      cellN = cellNN;
      cellNN = tile_cell(ts + 2 * tstep);
      if constexpr (NOSTORE) { if (rows2_ok) rowsN = rowsN2; else ld_rows(next_rows(cellN, cellC), rowsN); }
      else ld_rows(next_rows(cellN, cellC), rowsN);
      dots_regs(rowsA, cellC.x >= 0, accC);
      have2 = HMX_CHAIN_PRE2 && USIG && ts + tstep < te;    // (the general-sigma variant has no registers to spare: 95 spills)
    };
    auto first_tile_b = [&]() __attribute__((always_inline)) {
      if (have2) {
        cellS = cellN;
        const RowRegs rowsB = rowsN;
        cellN = cellNN;
        cellNN = tile_cell(ts + 3 * tstep);
        ld_rows(next_rows(cellN, cellS), rowsN);
        dots_regs(rowsB, cellS.x >= 0, accS);
      }
    };
    // both hoisted tiles against one read of the centroid image, when the second tile's rows are in registers already (MODE 5: rowsN2)
    auto first_tiles = [&]() __attribute__((always_inline)) {
      if constexpr (NOSTORE && BF) {
        if (rows2_ok && HMX_CHAIN_PRE2 && USIG && ts + tstep < te) {
          cellC = cellN; cellS = cellNN;
          const RowRegs rowsA = rowsN, rowsB = rowsN2;
          cellN = tile_cell(ts + 2 * tstep); cellNN = tile_cell(ts + 3 * tstep);
          tile_dots_bf_regs2<NCT>(ldsB, rowsA, cellC.x >= 0, rowsB, cellS.x >= 0, rowmask, lane, D.NS2, accC, accS);
          ld_rows(next_rows(cellN, cellS), rowsN);
          have2 = true;
          return;
        }
      }
      first_tile_a(); first_tile_b();
    };
    if (have) first_tiles();
    // Geometry of the block AFTER the current one and the (cell, combination) pairs of this wave's first two tiles in it.  Requested
    // in the slack behind an arrival, a whole block ahead of their use: at the top of the epilogue phase the two dependent round trips
    // (scalar load of the block offsets, then the pairs) were 1.6 us of every block step's critical path.
    int p0n = 0, ten = 0, tsn = ts; bool haveN = false;
    int2 cN1 = make_int2(-1, -1), cNN1 = make_int2(-1, -1);
    auto fetch_next = [&](const int jb) __attribute__((always_inline)) {
      p0n = 0; ten = 0; haveN = false; cN1 = make_int2(-1, -1); cNN1 = make_int2(-1, -1);
      if (jb < nbk) {
        p0n = D.boff[jb];
        ten = (D.boff[jb + 1] - p0n) >> 4;
        if constexpr (CHAIN_CONTIG) chain_range(ten, tsn, ten);      // (ten: from here on the end of THIS wave's range)
        haveN = tsn < ten;
        if (haveN) { cN1 = D.lpair[p0n + 16 * tsn + c]; if (tsn + tstep < ten) cNN1 = D.lpair[p0n + 16 * (tsn + tstep) + c]; }
      }
    };
    fetch_next(1);
    unsigned long long wq = 0, wg = 0, ww = 0, wd = 0, wm = 0, w1 = 0, w2 = 0, w3 = 0, w_prev = wall_clock64();   // diagnostics (workgroup 0, wave 0)
    unsigned long long wv_busy = 0, wv_tiles = 0;    // per wave: table in LDS -> own work done (before the barrier), tiles owned
    auto lap = [&](unsigned long long& acc) { const unsigned long long t = wall_clock64(); acc += t - w_prev; w_prev = t; };
    for (int jj = 0; jj < nbk; jj++) {
      const unsigned tag = tag0 + (unsigned)jj;
      if (tid == 0) {                       // one lane per workgroup polls the block flag
        int spins = 0;
        while ((int)((unsigned)__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tag) < 0) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_LIMIT) { atomicExch(&ctl[1], 2); break; }
          if (dead(spins)) break;
        }
      }
      __syncthreads();
      lap(wq);
      for (int i = tid; i < nBK; i += bd) {   // the block's penalty table -> LDS (granules validate themselves)
        unsigned long long gv = __hip_atomic_load(&peng[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while ((unsigned)(gv >> 32) != tag && ++spins < SPIN_LIMIT && !dead(spins)) { __builtin_amdgcn_s_sleep(1); gv = __hip_atomic_load(&peng[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if ((unsigned)(gv >> 32) != tag) atomicExch(&ctl[1], 3);
        ldsPen[i] = __uint_as_float((unsigned)(gv & 0xffffffffu));
      }
      __syncthreads();
      lap(wg);
      const unsigned long long t_tab = w_prev;
      curq = -1;                            // the table changed: the penalty row of the first tile must be re-read
      // (od / oe: this wave's objective partial sums run through ALL blocks of the round in registers, one wave reduction and one
      //  slot store at the end of the launch instead of two 6-step 64-bit reductions per block on the critical path)
      // (geometry of the next block and the pairs of this wave's first two tiles in it: fetch_next, requested one block ahead)
      const bool more = jj + 1 < nbk;
      // The R rows of a wave's LAST TWO tiles of the block leave BEHIND the arrival.  Nobody reads a block's R rows before the launch
      // ends, but their stores -- 20 MB per block step from all workgroups at once, an HBM write burst the issuing waves sit behind --
      // were 3.2 of the 6.7 us a two-tile wave needed between "table in LDS" and its arrival (tools/chain_probe.py).  The normalised
      // rows stay in the two accumulator sets (epi_rows<DEFER>) through the contribution atomics, the barrier and the arrival, and are
      // stored in the folder's shadow, interleaved with the next block's MFMAs (which overwrite the same registers).
      bool two = false;                     // both hoisted tiles are the wave's last ones of this block: both epilogues deferred
      if (have) {
        int tile0 = ts + tstep;
        if (have2) two = ts + 2 * tstep >= te;
        // (waves that enter the tile loop: every load has landed before it, see the two-accumulator loop below; the others must not wait
        //  here -- the oldest operations in flight are the R stores they issued behind the previous arrival)
        if (!two) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if (have2) {
          if (!two) {                         // more than two tiles: the first one's epilogue now, the second becomes the pending tile
            epilogue(cellC.x, tile_q(cellC), accC);
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) accC[ct] = accS[ct];
            cellC = cellS;
            tile0 = ts + 2 * tstep;
          }
        }
        if (!two) {
          for (int tile = tile0; tile < te; tile += tstep) {
            const int2 cellT = cellN;
            const RowRegs rowsA = rowsN;
            cellN = cellNN;
            cellNN = tile_cell(tile + 2 * tstep);
            ld_rows(next_rows(cellN, cellT), rowsN);
            f32x4 accT[NCT];
            dots_regs(rowsA, cellT.x >= 0, accT);   // MFMA pipe: tile i+1
            epilogue(cellC.x, tile_q(cellC), accC);                                           // VALU pipe: tile i
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) accC[ct] = accT[ct];
            cellC = cellT;
          }
        }
        // next block's first rows requested first (older than everything below), then the epilogue(s) with DEFERRED stores and the
        // contribution atomics
        lap(w1);
        if (haveN) ld_rows(D.Zc + (size_t)(cN1.x >= 0 ? cN1.x : 0) * zs, rowsN);
        epi_begin(tile_q(cellC));
        epi_rows(cellC.x, accC, std::true_type{});
        if (two) {
          epi_begin(tile_q(cellS));
          epi_rows(cellS.x, accS, std::true_type{});
        }
        lap(w2);
        if constexpr (NOSTORE) {
          rows2_ok = haveN && tsn + tstep < ten;          // (uniform: the wave owns a second tile in the next block)
          if (rows2_ok) ld_rows(next_rows(cNN1, cN1), rowsN2);
        }
        if (curq >= 0) flush_run();
        lap(w3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the wave's atomics have been performed (no stores queued behind them)
      } else {
        rows2_ok = false;
        if (haveN) ld_rows(D.Zc + (size_t)(cN1.x >= 0 ? cN1.x : 0) * zs, rowsN);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      lap(ww);
      wv_busy += w_prev - t_tab; wv_tiles += have ? (unsigned long long)((te - ts + tstep - 1) / tstep) : 0ull;
      // a BARE s_barrier (not __syncthreads(): nothing another wave of this workgroup reads is published here -- the barrier only
      // says "every wave's contribution atomics have been performed", each wave waited for its own above).
      __builtin_amdgcn_s_barrier();
      if (tid == 0) atomicAdd(&ctl[8 + 8 * jj + ((int)blockIdx.x & 7)], 1);          // arrival of this workgroup
      const bool st1 = have, st2 = have && two;
      const int sc1 = cellC.x, sc2 = cellS.x;
      if (more) { p0 = p0n; ts = tsn; te = ten; cellN = cN1; cellNN = cNN1; }
      lap(wd);
      have = haveN;
      // off the critical path, in the folder's shadow: rows of the finished tiles out, MFMAs of the next block's tiles in
      if constexpr (NOSTORE) { if (have) first_tiles(); else have2 = false; }
      else {
        if (st1) store_rows(sc1, accC);
        if (have) first_tile_a();
        if (st2) store_rows(sc2, accS);
        if (have) first_tile_b(); else have2 = false;
      }
      fetch_next(jj + 2);
      lap(wm);
    }
    od = wsumd(od); oe = wsumd(oe);
    if (D.chain_tail) {                     // the folder closes the round: the wave's sums go out write-through, then the workgroup arrives once more
      if (lane == 0) {
        double* slot = D.objpart + (size_t)wave * 2;
        __hip_atomic_store(&slot[0], od, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&slot[1], oe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid == 0) atomicAdd(&ctl[16 + 8 * nbk + ((int)blockIdx.x & 7)], 1);
    } else if (lane == 0) {                 // slot row 0 (the other rows stay zero: k_round_tail sums and clears all of them)
      double* slot = D.objpart + (size_t)wave * 2;
      slot[0] += od; slot[1] += oe;
    }
    if (blockIdx.x == 0 && tid == 0 && D.chain_dbg) {
      atomicAdd(&D.chain_dbg[4], wq); atomicAdd(&D.chain_dbg[5], wg); atomicAdd(&D.chain_dbg[6], ww); atomicAdd(&D.chain_dbg[7], wd); atomicAdd(&D.chain_dbg[8], wm);
      atomicAdd(&D.chain_dbg[9], w1); atomicAdd(&D.chain_dbg[10], w2); atomicAdd(&D.chain_dbg[11], w3);
    }
    if ((blockIdx.x == 0 || blockIdx.x == 100) && lane == 0 && D.chain_dbg) {
      const int o = (blockIdx.x == 0 ? 16 : 32) + (tid >> 6);
      atomicAdd(&D.chain_dbg[o], wv_busy); atomicAdd(&D.chain_dbg[o + 8], wv_tiles);
    }
    if (blockIdx.x == 0 && (tid == 256 || tid == 320) && D.chain_dbg) {      // the same phase clocks for a SIMD's YOUNGER wave: wave 4 (two tiles), wave 5 (one)
      unsigned long long* const o = D.chain_dbg + (tid == 256 ? 48 : 56);
      atomicAdd(&o[0], wq); atomicAdd(&o[1], wg); atomicAdd(&o[2], ww); atomicAdd(&o[3], wd); atomicAdd(&o[4], wm); atomicAdd(&o[5], w1); atomicAdd(&o[6], w2); atomicAdd(&o[7], w3);
    }
    return;
    }
  }
  if (pre && (!DUAL || MODE == 2)) {   // (Lloyd: one accumulator set, its epilogue wants the tile's rows still in registers)
    for (int tile = ts; tile < te; tile += tstep) {
      const int2 cellA = cellN;
      const RowRegs rowsA = rowsN;
      cellN = cellNN;
      cellNN = tile_cell(tile + 2 * tstep);
      ld_rows(next_rows(cellN, cellA), rowsN);
      f32x4 acc[NCT];
      dots_regs(rowsA, cellA.x >= 0, acc);
      if constexpr (MODE == 2) erows = &rowsA;
      epilogue(cellA.x, tile_q(cellA), acc);
    }
  } else if (pre) {
    // two accumulator sets: the MFMAs of tile i+1 are issued before the (VALU / transcendental / store) epilogue of tile i
    // Head of cluster_cpp (MODE 1, D.head_norm): Z_corr <- normalise(Z_corr) (src/harmony.cpp:220) happens HERE, on the A-operand
    // registers of the tile -- the four lanes that hold a cell's row (k-slots 0..3) add up their squares, scale, and write the
    // normalised pieces back where they came from: no separate pass over Z_corr (it was 67 us of 256 per head at 1M cells).
    auto norm_rows = [&](RowRegs& r, const int cell) __attribute__((always_inline)) {
      if constexpr (BF) {     // split-bf16 row layout: the groups beyond the row were loaded clamped -- zero them, then as below
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        float ss = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          r.v[t] = (rowmask >> t) & 1 ? r.v[t] : zero4;
          ss += r.v[t][0] * r.v[t][0] + r.v[t][1] * r.v[t][1] + r.v[t][2] * r.v[t][2] + r.v[t][3] * r.v[t][3];
        }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        float nrm = sqrtf(ss);
        nrm = (nrm == 0.0f) ? 1.0f : nrm;
        const float inv = 1.0f / nrm;
        float* zrow = D.Zc + (size_t)(cell >= 0 ? cell : 0) * zs;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          r.v[t] = r.v[t] * inv;
          if (cell >= 0 && ((rowmask >> t) & 1)) *reinterpret_cast<f32x4*>(zrow + 32 * (t >> 1) + 8 * g + 4 * (t & 1)) = r.v[t];
        }
        return;
      }
      float ss = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; t++) if (t < D.NT4) ss += r.v[t][0] * r.v[t][0] + r.v[t][1] * r.v[t][1] + r.v[t][2] * r.v[t][2] + r.v[t][3] * r.v[t][3];
#pragma unroll
      for (int u = 0; u < 3; u++) if (u < D.tail) ss += r.t[u] * r.t[u];
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      float nrm = sqrtf(ss);
      nrm = (nrm == 0.0f) ? 1.0f : nrm;            // arma::normalise leaves a zero column alone
      const float inv = 1.0f / nrm;
      float* zrow = D.Zc + (size_t)(cell >= 0 ? cell : 0) * zs;
#pragma unroll
      for (int t = 0; t < 4; t++) if (t < D.NT4) {
        r.v[t] = r.v[t] * inv;
        if (cell >= 0) *reinterpret_cast<f32x4*>(zrow + 16 * t + 4 * g) = r.v[t];
      }
#pragma unroll
      for (int u = 0; u < 3; u++) if (u < D.tail) {
        r.t[u] *= inv;
        if (cell >= 0) zrow[16 * D.NT4 + 4 * u + g] = r.t[u];
      }
    };
    if (ts < te) {
      f32x4 accC[NCT];
      int2 cellC = cellN;
      {
        RowRegs rowsA = rowsN;
        cellN = cellNN;
        cellNN = tile_cell(ts + 2 * tstep);
        ld_rows(next_rows(cellN, cellC), rowsN);
        if constexpr (MODE == 1) { if (D.head_norm) norm_rows(rowsA, cellC.x); }
        dots_regs(rowsA, cellC.x >= 0, accC);
      }
      stamp(3);
      // every load of the prologue has landed before the loop is entered: hipcc merges the wait state of the two loop
      // entries conservatively, and row loads still pending on THIS edge would turn the loop-top waits into vmcnt(0..2),
      // which on the back edge means draining the 28 R stores of the tile just finished
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      for (int tile = ts + tstep; tile < te; tile += tstep) {
        stamp(8);
        const int2 cellT = cellN;
        RowRegs rowsA = rowsN;
        cellN = cellNN;
        cellNN = tile_cell(tile + 2 * tstep);
        ld_rows(next_rows(cellN, cellT), rowsN);
        if constexpr (MODE == 1) { if (D.head_norm) norm_rows(rowsA, cellT.x); }
        stamp(9);
        f32x4 accT[NCT];
#ifdef HMX_TRACE
        const int dbg = D.upd_debug;   // 1 = no epilogue work, 2 = no MFMAs (WRONG RESULTS; timing experiments only)
        if (dbg & 2) {
#pragma unroll
          for (int ct = 0; ct < NCT; ct++) accT[ct] = rowsA.v[0];
        } else dots_regs(rowsA, cellT.x >= 0, accT);
        stamp(10);
        if (!(dbg & 1)) epilogue(cellC.x, tile_q(cellC), accC);
        else {
#pragma unroll
          for (int ct = 0; ct < NCT; ct++) od += (double)(accC[ct][0] + accC[ct][1] + accC[ct][2] + accC[ct][3]);
        }
        stamp(11);
#else
        dots_regs(rowsA, cellT.x >= 0, accT);   // MFMA pipe: tile i+1
        epilogue(cellC.x, tile_q(cellC), accC);                                           // VALU pipe: tile i
#endif
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) accC[ct] = accT[ct];
        cellC = cellT;
      }
      stamp(4);
      epilogue(cellC.x, tile_q(cellC), accC);
      stamp(5);
    }
  } else {
    for (int tile = ts; tile < te; tile += tstep) {
      const int2 cellA = cellN;
      cellN = cellNN;
      cellNN = tile_cell(tile + 2 * tstep);
      f32x4 acc[NCT];
      dots_stream(D.Zc + (size_t)(cellA.x >= 0 ? cellA.x : 0) * zs, cellA.x >= 0, acc);
      epilogue(cellA.x, tile_q(cellA), acc);
    }
  }
  if constexpr (MODE == 2) {
    __syncthreads();
    for (int i = threadIdx.x; i < K * D.d; i += blockDim.x)
      if (ltab[i]) atomicAdd((unsigned long long*)&D.lsum[i], (unsigned long long)ltab[i]);
    for (int i = threadIdx.x; i < K; i += blockDim.x)
      if (ltab[K * D.d + i]) atomicAdd(&D.lcnt[i], (unsigned long long)ltab[K * D.d + i]);
  } else if constexpr (MODE == 3) {
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {   // the four row groups hold candidates for the same anchors
      unsigned long long v = best[ct];
      unsigned long long o = shfl_xor_u64(v, 16); v = o < v ? o : v;
      o = shfl_xor_u64(v, 32); v = o < v ? o : v;
      if (g == 0 && kcol(NCT, ct, c) < K && v != ~0ull) atomicMin(&D.seedmin[kcol(NCT, ct, c)], v);
    }
  } else {
    if (ts >= te) return;
    if (curq >= 0) flush_run();
    od = wsumd(od); oe = wsumd(oe);
    if (lane == 0) {
      const int slotrow = (MODE == 0) ? (j % D.objslots) : 0;
      double* slot = D.objpart + ((size_t)slotrow * D.nwmax + wave) * 2;
      if (MODE == 0 && D.nb <= D.objslots) { slot[0] = od; slot[1] = oe; }  // written once per round: plain store
      else { slot[0] += od; slot[1] += oe; }
    }
    stamp(6);
    stamp(7);
  }
}



#if !HMX_TILE_BF
// cross-entropy term of the objective from the K x B tables alone (src/harmony.cpp:162):
//   sum_k sigma_k sum_b theta_b log((O+E+1)/(2E+1)) * O[k,b]     (O[k,b] = sum_{i in b} R_ki)
// single workgroup; obj[2..4] = {dist, entropy, cross} snapshot, obj[0..1] reset.
__global__ __launch_bounds__(TPB) void k_objective_tables(Dev D) {
  __shared__ double red[TPB];
  const int K = D.K, B = D.B;
  double cross = 0.0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    long long rs = 0;
    for (int b = 0; b < B; b++) {
      if (b < D.B0) rs += D.O_fx[(size_t)b * K + k];
    }
    const double rsd = (double)rs * FX_INV;
    double ck = 0.0;
    for (int b = 0; b < B; b++) {
      const double od = (double)D.O_fx[(size_t)b * K + k] * FX_INV;
      const float o = (float)od, e = (float)(rsd * (double)D.Pr_b[b]);
      const float m = D.theta[b] * logf((o + e + 1.0f) / ((2.0f * e) + 1.0f));
      ck += od * (double)m;
    }
    cross += ck * (double)D.sigma[k];
  }
  red[threadIdx.x] = cross;
  __syncthreads();
  for (int off = TPB / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    D.obj[2] = D.obj[0]; D.obj[3] = D.obj[1]; D.obj[4] = red[0];
    D.obj[5] = (D.chain_ctl ? (double)D.chain_ctl[1] : 0.0) + ((D.solve_err && *D.solve_err) ? 16.0 : 0.0);      // the chain's error word (+ 16: singular ridge system) rides along with the objective snapshot
  }
  // (this single-workgroup kernel closes every round: it also resets the chain's control block for the next one -- one memset
  //  launch less per round; the shuffle kernels cannot do it, they may run on the side stream while a chain is in flight)
  __syncthreads();
  if (D.chain_ctl) for (int i = threadIdx.x; i < 8 * D.nb + 24; i += blockDim.x) D.chain_ctl[i] = 0;
}

// obj_arith: the objective's three K x N term matrices (src/harmony.cpp:160-162, see hmx_seq.hip k_obj_terms) with the distances from the
// matrix cores: static 16-cell tiles like the head, the R row and the term rows as 16-byte accesses (a lane's clusters are consecutive,
// kcol).  Round 3's cluster-lane VALU version took 1.44 ms per evaluation at 1M cells, a quarter of the reference-arithmetic run.
// T[0] = R % dist, T[1] = (R % log R) % sigma, T[2] = (R % sigma) % (M Phi); rows at the cells' ORIGINAL positions, k fastest.
// Round 5: (1) tiles run over the ORIGINAL cell order (16 consecutive original cells per tile, rows gathered through invperm): a tile's 16 rows
// of each term array are 16 K contiguous floats -- whole 128-byte lines at K = 100; the cells of a tile no longer share a combination, so the
// M rows are fetched per row.  (2) A software pipeline like k_tile's: cell ids two tiles ahead, and ALL loads of tile i + 1 (embedding rows in
// registers, R rows, M rows) are issued before the stores of tile i -- vector memory operations retire in issue order on gfx9, so a load queued
// behind a tile's 84 stores waits for them to drain; the round-4 kernel did that four times per tile (1.02 ms per evaluation at 1M cells, 21
// evaluations per run).  (3) log R through v_log_f32 (1 ulp; the term enters a sum of K N values): 3 instructions instead of logf's ~30.
template <int NCT>
__global__ __launch_bounds__(256, (NCT <= 8 ? 2 : 1)) void k_obj_terms_mfma(Dev D, const float* __restrict__ M, float* __restrict__ T, long long stride) {      // (two waves per SIMD: with 268 registers there was one, and nothing hid the tile's memory latencies)
  extern __shared__ __attribute__((aligned(16))) f32x4 ldsI[];
  constexpr int NFULL = NCT >> 2, RT = NCT & 3;
  const int K = D.K, C = D.C, zs = D.zs, n = D.n;
  const int nY4 = D.NQ * D.NS * 64;
  { const f32x4* src = reinterpret_cast<const f32x4*>(D.Yimg);
    for (int i = threadIdx.x; i < nY4; i += blockDim.x) ldsI[i] = src[i]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nw = (int)((gridDim.x * blockDim.x) >> 6);
  const float lmin = __builtin_amdgcn_logf(FLT_MIN) * 0.69314718055994530942f;
  const int ntiles = (n + 15) >> 4;
  const int per = (ntiles + nw - 1) / nw;           // contiguous tile ranges per wave: consecutive tiles write consecutive lines
  const int t_lo = min(wave * per, ntiles), t_hi = min(t_lo + per, ntiles);
  if (t_lo >= t_hi) return;
  const float* __restrict__ Rp = D.R;
  const float* __restrict__ Zp = D.Zc;
  const int* __restrict__ qlevp = D.qlev;
  struct Ids { int cell, q; };
  auto ids_of = [&](const int tile) -> Ids {         // this lane's A-operand cell of `tile` (clamped: any valid cell beyond the end)
    Ids I; I.cell = D.invperm[min(16 * min(tile, ntiles - 1) + c, n - 1)]; I.q = D.combo[I.cell]; return I;
  };
  // per tile and lane: the R and M values of rows 4g .. 4g + 3 at this lane's clusters -- the layout of the accumulators
  struct Vals { f32x4 r[NCT], m[NCT]; };
  auto vals_of = [&](const Ids& I, Vals& V) __attribute__((always_inline)) {
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int cell = __shfl(I.cell, 4 * g + reg, 64), q = __shfl(I.q, 4 * g + reg, 64);
      const float* __restrict__ rrow = Rp + (size_t)cell * K;
      int lev[4];
#pragma unroll
      for (int cc = 0; cc < 4; cc++) lev[cc] = qlevp[q * C + min(cc, C - 1)];
#pragma unroll
      for (int qd = 0; qd < NFULL; qd++) {
        const int k0 = min(64 * qd + 4 * c, K - 4);
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(rrow + k0);
        f32x4 m4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
          if (cc < C) {
            const f32x4 mm = *reinterpret_cast<const f32x4*>(M + (size_t)lev[cc] * K + k0);
#pragma unroll
            for (int i = 0; i < 4; i++) m4[i] = __fadd_rn(m4[i], mm[i]);
          }
        }
        for (int cc = 4; cc < C; cc++) {
          const f32x4 mm = *reinterpret_cast<const f32x4*>(M + (size_t)qlevp[q * C + cc] * K + k0);
#pragma unroll
          for (int i = 0; i < 4; i++) m4[i] = __fadd_rn(m4[i], mm[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { V.r[4 * qd + i][reg] = r4[i]; V.m[4 * qd + i][reg] = m4[i]; }
      }
#pragma unroll
      for (int jj = 0; jj < RT; jj++) {
        const int k = min(64 * NFULL + RT * c + jj, K - 1);
        float m = 0.0f;
        for (int cc = 0; cc < C; cc++) m = __fadd_rn(m, M[(size_t)qlevp[q * C + cc] * K + k]);
        V.r[4 * NFULL + jj][reg] = rrow[k]; V.m[4 * NFULL + jj][reg] = m;
      }
    }
  };
  float sg[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) sg[ct] = D.sigma[min(kcol(NCT, ct, c), K - 1)];
  Ids idN = ids_of(t_lo), idNN = ids_of(t_lo + 1);
  RowRegs rowsN;
  Vals VN;
  load_rows(Zp + (size_t)idN.cell * zs, g, D.NT4, D.tail, rowsN);
  vals_of(idN, VN);
  for (int tile = t_lo; tile < t_hi; tile++) {
    const int o0 = 16 * tile;
    const bool av = o0 + c < n;
    const RowRegs rowsA = rowsN;
    const Vals V = VN;
    f32x4 acc[NCT];
    tile_dots_regs<NCT>(ldsI, rowsA, av, lane, D.NS, D.NT4, D.tail, acc);
    // everything tile + 1 needs, requested BEFORE this tile's stores
    idN = idNN; idNN = ids_of(tile + 2);
    load_rows(Zp + (size_t)idN.cell * zs, g, D.NT4, D.tail, rowsN);
    vals_of(idN, VN);
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int cl = 4 * g + reg;
      const bool cv = o0 + cl < n;
      float* __restrict__ t0p = T + (size_t)(o0 + (cv ? cl : 0)) * K;
      float* __restrict__ t1p = t0p + (size_t)stride;
      float* __restrict__ t2p = t1p + (size_t)stride;
      float a0[NCT], a1[NCT], a2[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) {
        const float r = V.r[ct][reg];
        const float dist = __fmul_rn(2.0f, __fsub_rn(1.0f, acc[ct][reg]));
        const float lg = (r > 0.0f) ? __fmul_rn(__builtin_amdgcn_logf(r), 0.69314718055994530942f) : lmin;      // arma::trunc_log
        a0[ct] = __fmul_rn(r, dist);
        a1[ct] = __fmul_rn(__fmul_rn(r, lg), sg[ct]);
        a2[ct] = __fmul_rn(__fmul_rn(r, sg[ct]), V.m[ct][reg]);
      }
      if (cv) {
#pragma unroll
        for (int qd = 0; qd < NFULL; qd++) {
          const int k0 = 64 * qd + 4 * c;
          if (4 * qd < first_partial_ct(NCT) || k0 < K) {
            const f32x4 v0 = {a0[4 * qd], a0[4 * qd + 1], a0[4 * qd + 2], a0[4 * qd + 3]}, v1 = {a1[4 * qd], a1[4 * qd + 1], a1[4 * qd + 2], a1[4 * qd + 3]},
                        v2 = {a2[4 * qd], a2[4 * qd + 1], a2[4 * qd + 2], a2[4 * qd + 3]};
            // (streaming stores: the arrays are read back by the passes only after the whole evaluation has been written -- 1.2 GB, nothing to keep in L2)
            __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(t0p + k0)); __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(t1p + k0));
            __builtin_nontemporal_store(v2, reinterpret_cast<f32x4*>(t2p + k0));
          }
        }
#pragma unroll
        for (int jj = 0; jj < RT; jj++) {
          const int k = 64 * NFULL + RT * c + jj;
          if (k < K) { t0p[k] = a0[4 * NFULL + jj]; t1p[k] = a1[4 * NFULL + jj]; t2p[k] = a2[4 * NFULL + jj]; }
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------
// MoE ridge correction (src/harmony.cpp:345-638)
//   k_moe_stats : per combination q and cluster k:  nq = sum_i R_ki,  Sq = sum_i R_ki z_i
//                 (the sufficient statistics of Phi* diag(R_k) Phi*^T and Phi* diag(R_k) Z^T)
//   host        : K small ridge solves in fp64 -> correction table Wq[q][k][:]
//   k_moe_apply : Z_corr_i = Z_orig_i - sum_k R_ki Wq[q(i)][k][:]
// --------------------------------------------------------------------------------------
// grid.y = cluster chunks of 128, grid.z = PC chunks of DP (DP = 4..32 in steps of 4, chosen so that the
// chunks cover d with little padding: d=50 -> 2 chunks of 28)
template <int DP>
__global__ __launch_bounds__(TPB) void k_moe_stats(Dev D) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const int d = D.d, K = D.K;
  const int k0 = blockIdx.y * 128 + lane, k1 = k0 + 64;
  const int zoff = blockIdx.z * DP;
  const int dch = min(DP, d - zoff);
  for (int it = wave; it < D.nitems; it += nw) {
    const Item item = D.items[it];
    float a0[DP], a1[DP];
#pragma unroll
    for (int j = 0; j < DP; j++) { a0[j] = 0.0f; a1[j] = 0.0f; }
    double n0 = 0.0, n1 = 0.0;
    size_t cell = (size_t)item.start;
    const int ls = min(lane, dch - 1), k0s = min(k0, K - 1), k1s = min(k1, K - 1);
    float zn = ld_or(D.Zo, cell * D.zs + zoff + ls, lane < dch, 0.0f);
    float r0n = ld_or(D.R, cell * K + k0s, k0 < K, 0.0f);
    float r1n = ld_or(D.R, cell * K + k1s, k1 < K, 0.0f);
    for (int p = 0; p < item.cnt; p++) {
      const float zr = zn, r0 = r0n, r1 = r1n;
      if (p + 1 < item.cnt) {  // software prefetch of the next cell's rows
        cell = (size_t)(item.start + p + 1);
        zn = ld_or(D.Zo, cell * D.zs + zoff + ls, lane < dch, 0.0f);
        r0n = ld_or(D.R, cell * K + k0s, k0 < K, 0.0f);
        r1n = ld_or(D.R, cell * K + k1s, k1 < K, 0.0f);
      }
      n0 += (double)r0; n1 += (double)r1;
#pragma unroll
      for (int j = 0; j < DP; j++) {
        const float zj = rlane(zr, j);
        a0[j] = fmaf(r0, zj, a0[j]);
        a1[j] = fmaf(r1, zj, a1[j]);
      }
    }
    double* S = D.Sq + (size_t)item.q * d * K;
#pragma unroll
    for (int j = 0; j < DP; j++) {
      if (j < dch) {
        if (k0 < K) atomicAdd(&S[(size_t)k0 * d + zoff + j], (double)a0[j]);
        if (k1 < K) atomicAdd(&S[(size_t)k1 * d + zoff + j], (double)a1[j]);
      }
    }
    if (blockIdx.z == 0) {
      if (k0 < K) atomicAdd(&D.nq[(size_t)item.q * K + k0], n0);
      if (k1 < K) atomicAdd(&D.nq[(size_t)item.q * K + k1], n1);
    }
  }
}

// one workgroup per apply item (<= APPLY_CELLS cells of one combination); Wq[q] staged in LDS
// as [K][DS] (DS = 64*DPL); lane = PC.
template <int KPL, int DPL>
__global__ __launch_bounds__(TPB) void k_moe_apply(Dev D) {
  extern __shared__ __attribute__((aligned(16))) float ldsW[];
  constexpr int CB = 4;
  constexpr int DS = 64 * DPL;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int d = D.d, K = D.K;
  for (int it = blockIdx.x; it < D.naitems; it += gridDim.x) {
    const Item item = D.aitems[it];
    __syncthreads();
    const float* W = D.Wq + (size_t)item.q * K * d;
    for (int i = threadIdx.x; i < K * DS; i += blockDim.x) {
      const int k = i / DS, jj = i - k * DS;
      ldsW[i] = ld_or(W, (size_t)k * d + min(jj, d - 1), jj < d, 0.0f);
    }
    __syncthreads();
    const int per = (item.cnt + 3) / 4;
    const int s = item.start + wib * per, e = min(item.start + item.cnt, s + per);
    for (int p = s; p < e; p += CB) {
      const int nc = min(CB, e - p);
      float rr[CB][KPL], corr[CB][DPL];
#pragma unroll
      for (int c = 0; c < CB; c++) {
#pragma unroll
        for (int q = 0; q < KPL; q++) {
          const int k = lane + 64 * q;
          rr[c][q] = ld_or(D.R, (size_t)min(p + c, e - 1) * K + min(k, K - 1), c < nc && k < K, 0.0f);
        }
#pragma unroll
        for (int t = 0; t < DPL; t++) corr[c][t] = 0.0f;
      }
#pragma unroll
      for (int q = 0; q < KPL; q++) {
        const int kend = min(64, K - 64 * q);
        for (int kk = 0; kk < kend; ++kk) {
          float w[DPL];
#pragma unroll
          for (int t = 0; t < DPL; t++) w[t] = ldsW[(64 * q + kk) * DS + 64 * t + lane];
#pragma unroll
          for (int c = 0; c < CB; c++) {
            const float rk = rlane(rr[c][q], kk);
#pragma unroll
            for (int t = 0; t < DPL; t++) corr[c][t] = fmaf(rk, w[t], corr[c][t]);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CB; c++) {
        if (c < nc) {
          const size_t cell = (size_t)(p + c);
#pragma unroll
          for (int t = 0; t < DPL; t++) {
            const int jj = 64 * t + lane;
            if (jj < d) D.Zc[cell * D.zs + jj] = D.Zo[cell * D.zs + jj] - corr[c][t];
          }
        }
      }
    }
  }
}

// index of centroid entry (PC j, cluster k) in the MFMA B-operand image (inverse of the host builder in upload_Y)
__device__ __forceinline__ size_t yimg_index(const Dev& D, int j, int k) {
  int s_, p_;
  if (j < 16 * D.NT4) { const int t = j >> 4, r = j & 15; p_ = r >> 2; s_ = 4 * t + (r & 3); }
  else { const int r = j - 16 * D.NT4; s_ = 4 * D.NT4 + (r >> 2); p_ = r & 3; }
  int qd, i, c;
  kcol_inv(D.NCT, k, qd, i, c);
  return ((((size_t)qd * D.NS + s_) * 4 + p_) * 16 + c) * 4 + i;
}
// k_moe_solve: the K ridge systems of moe_correct_ridge_cpp ON THE DEVICE (src/harmony.cpp:358-611), fp64, one workgroup
// per cluster -- no D2H of the statistics, no host solve, no H2D of the correction table: the whole correction is a
// chain of kernels with no host synchronisation.  Per cluster k:
//   kept levels (O[k,b] / N_b > cutoff and >= 2 such levels in the covariate, :368-402), lambda_k (fixed or alpha * E, :434-439),
//   cov = Phi* diag(R_k) Phi*^T + Lambda and rhs = Phi* diag(R_k) Z^T assembled from the per-combination statistics
//   (subset path = masks, :440-547), blocked Cholesky -- Phi* diag(R_k) Phi*^T + Lambda is symmetric positive definite whenever the
//   reference's arma::inv succeeds; a non-positive pivot raises flag bit 2 (HMX_ERR_SOLVE at the next objective read or getter; the
//   host solve path, HMX_MOE_SOLVE=host, retries with LU) --, W = cov^-1 rhs, Y[:,k] = W[0,:], W[0,:] = 0 (:610-611), correction table Wq[q][k][:] = sum of the kept
//   levels' rows of combination q (+ its MFMA image), Y <- normalise (:633).
// flags[k]: bit 0 subset path, bit 1 skipped (no covariate with two kept levels), bit 2 singular system.
__global__ __launch_bounds__(1024) void k_moe_solve(Dev D, SolveArgs A) {
  extern __shared__ int sm_[];
  const int K = D.K, B = D.B, C = D.C, d = D.d, Q = D.Q, M = B + 1;
  const int k = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  int* row_of = sm_;                 // [B]
  int* keepl = sm_ + B;              // [B] kept levels in order
  int* okb = sm_ + 2 * B;            // [B]
  int* misc = sm_ + 3 * B;           // [0] m, [1] active, [2] full, [3] fail, [4..4+C) cov_levels
  int* prow = sm_ + 3 * B + 4 + C;   // [B + 1] design row (reference order: intercept, kept levels ascending) -> row of the system as it is solved
  const int PAN_OFF = (4 * B + 6 + C) & ~1;      // ints in front of the fp64 panel space
  __shared__ int sch_nd;             // levels of the covariate whose (diagonal) block is eliminated first, 0: none
  double* cov = A.cov + (size_t)k * M * M;
  double* rhs = A.rhs + (size_t)k * d * M;
  for (int b = tid; b < B; b += nt) {
    const float o = A.Of ? A.Of[(size_t)b * K + k] : (float)((double)D.O_fx[(size_t)b * K + k] * FX_INV);
    okb[b] = (o / D.sizes[b]) > A.cutoff ? 1 : 0;
  }
  __syncthreads();
  if (tid == 0) {
    for (int c = 0; c < C; c++) misc[4 + c] = 0;
    for (int b = 0, cv = 0; b < B; b++) { if (!(b < A.cov_bounds[cv])) cv++; if (okb[b]) misc[4 + cv]++; }
    int nk = 0;
    for (int b = 0, cv = 0; b < B; b++) {
      if (cv < C && !(b < A.cov_bounds[cv])) cv++;
      if (okb[b] && misc[4 + cv] > 1) { keepl[nk] = b; row_of[b] = nk + 1; nk++; } else row_of[b] = -1;
    }
    int act = 0; for (int c = 0; c < C; c++) if (misc[4 + c] > 1) act++;
    misc[0] = nk + 1; misc[1] = act; misc[2] = (nk == B) ? 1 : 0; misc[3] = 0;
    // Every covariate's own block of Phi* diag(R_k) Phi*^T + Lambda is DIAGONAL (a cell has one level per covariate).  The rows of
    // the covariate with the most kept levels go LAST and are eliminated in closed form (Schur complement): the dense Cholesky
    // shrinks from m to m - nd rows -- 201 -> 73 at configs[4]'s 8 > 64 > 128 levels, 1 + B -> 1 for a single covariate.
    int cstar = -1, nd = 0;
    for (int c = 0; c < C; c++) if (misc[4 + c] > 1 && misc[4 + c] > nd) { nd = misc[4 + c]; cstar = c; }
    if (nd < 8 || A.solve_f32) { nd = 0; cstar = -1; }
    sch_nd = nd;
    prow[0] = 0;
    int nx = 1, nl = nk + 1 - nd;
    for (int a = 1; a <= nk; a++) {
      const int b = keepl[a - 1];
      int cv = 0; while (cv < C - 1 && !(b < A.cov_bounds[cv])) cv++;
      prow[a] = (cv == cstar) ? nl++ : nx++;
    }
  }
  __syncthreads();
  const int m = misc[0];
  const bool full = misc[2] != 0, skipped = !full && misc[1] == 0;
  // correction rows of this cluster start from zero (also the result for a skipped cluster, :449-452)
  for (int i = tid; i < Q * d; i += nt) { const int q = i / d, j = i - q * d; D.Wq[((size_t)q * K + k) * d + j] = 0.0f; }
  // Combination -> design rows table in LDS (the panel space, free until the Cholesky): with it every entry of the system is summed by ONE
  // thread over the combinations in ascending order -- the same order as the combination-by-combination loop below (bit-identical
  // results), without its Q workgroup barriers and read-modify-write round trips (0.26 ms of a 1.2 ms solve at configs[4]).
  const int nd_ = sch_nd, ms_ = m - nd_;
  const bool qtab = !skipped && ((size_t)2 * Q * (C + 1) + M + 1) * sizeof(int) <= A.lds_body_bytes;
  unsigned long long* const dmask = (A.lds_mask_off && qtab && nd_ > 0) ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(sm_) + A.lds_mask_off) : nullptr;
  const int mw = (nd_ + 63) >> 6;         // mask words per row: bit (r - ms) of dmask[a] <=> B[r][a] != 0 (level r meets row a in some combination)
  if (qtab) {
    int* const qr = sm_ + PAN_OFF;          // [Q][C + 1]: qr[q][0] = 0 (intercept) if any level of q is kept, else -1; qr[q][1 + c] = row of covariate c's level or -1
    if (dmask) for (int i = tid; i < ms_ * mw; i += nt) dmask[i] = 0ull;
    __syncthreads();
    for (int q = tid; q < Q; q += nt) {
      int any = -1, rD = -1;
      for (int c = 0; c < C; c++) {
        const int ro = row_of[D.qlev[q * C + c]];
        const int r = ro >= 0 ? prow[ro] : -1;
        qr[q * (C + 1) + 1 + c] = r;
        if (r >= 0) any = 0;
        if (r >= ms_) rD = r;
      }
      qr[q * (C + 1)] = any;
      if (dmask && rD >= 0) {
        atomicOr(&dmask[0 * mw + ((rD - ms_) >> 6)], 1ull << ((rD - ms_) & 63));
        for (int c = 0; c < C; c++) { const int r = qr[q * (C + 1) + 1 + c]; if (r >= 0 && r < ms_) atomicOr(&dmask[(size_t)r * mw + ((rD - ms_) >> 6)], 1ull << ((rD - ms_) & 63)); }
      }
    }
    __syncthreads();
    auto in_q = [&](const int* r, const int a) { bool in = (a == 0); for (int c = 0; c < C; c++) in |= (r[1 + c] == a); return in; };
    // per design row: the combinations that contain it, ascending (row 0: every combination with a kept level)
    int* const qoff = qr + Q * (C + 1);     // [m + 1]
    int* const qlst = qoff + M + 1;         // [<= Q (C + 1)]
    for (int a = tid; a < m; a += nt) {
      int cnt = 0;
      for (int q = 0; q < Q; q++) { const int* r = qr + q * (C + 1); if (r[0] == 0 && in_q(r, a)) cnt++; }
      qoff[a + 1] = cnt;
    }
    __syncthreads();
    if (tid == 0) { qoff[0] = 0; for (int a = 0; a < m; a++) qoff[a + 1] += qoff[a]; }
    __syncthreads();
    for (int a = tid; a < m; a += nt) {
      int o = qoff[a];
      for (int q = 0; q < Q; q++) { const int* r = qr + q * (C + 1); if (r[0] == 0 && in_q(r, a)) qlst[o++] = q; }
    }
    __syncthreads();
    if (!A.ref_tot) {
    for (int i = tid; i < m * d; i += nt) {
      const int j = i / m, a = i - j * m;
      double sacc = 0.0;
      for (int x = qoff[a]; x < qoff[a + 1]; x++) sacc += D.Sq[((size_t)qlst[x] * K + k) * d + j];
      rhs[i] = sacc;
    }
    for (int i = tid; i < m * m; i += nt) {
      const int cb = i / m, ra = i - cb * m;
      if (ra < cb) continue;                              // lower triangle, mirrored below
      double sacc = 0.0;
      if (!(cb >= ms_ && ra != cb)) {                    // (the eliminated covariate's own block is diagonal)
        // (the lists of the rows ra > 0 are short -- a level's combinations --; every one of them contains row 0)
        for (int x = qoff[ra]; x < qoff[ra + 1]; x++) { const int q = qlst[x]; if (cb == 0 || cb == ra || in_q(qr + q * (C + 1), cb)) sacc += D.nq[(size_t)q * K + k]; }
      }
      cov[(size_t)cb * m + ra] = sacc;
      cov[(size_t)ra * m + cb] = sacc;
    }
    }   // (!A.ref_tot)
    __syncthreads();
  }
  if (!skipped) {
    if (A.ref_tot) {
      // ridge_arith with several covariates: every entry of the system IS one of the reference's sequential fp32 accumulators
      // (hmx_seq.hip): Phi_Rk * Phi_moe_t entry (a, a2) = the sum of R_k over the kept cells that carry both design rows, in original order
      // (src/harmony.cpp:561-568; the intercept row is in every cell, a level row in its level's cells, two levels of different covariates
      // in their pair's cells, two levels of one covariate never meet); the right-hand sides = the same chains over fl(z_j R_k) (:592-608).
      auto chain_of = [&](const int a) { return a == 0 ? 0 : 1 + keepl[a - 1]; };
      for (int i = tid; i < m * d; i += nt) {
        const int j = i / m, a = i - j * m;
        rhs[(size_t)j * m + prow[a]] = (double)A.ref_tot[((size_t)chain_of(a) * K + k) * 64 + j];
      }
      for (int i = tid; i < m * m; i += nt) {
        const int a2 = i / m, a = i - a2 * m;
        float v;
        if (a == 0 || a2 == 0 || a == a2) v = A.ref_tot[((size_t)chain_of(a == 0 ? a2 : a) * K + k) * 64 + 63];
        else {
          const int b = keepl[a - 1], b2 = keepl[a2 - 1];
          int cv = 0, cv2 = 0;
          while (cv < C - 1 && !(b < A.cov_bounds[cv])) cv++;
          while (cv2 < C - 1 && !(b2 < A.cov_bounds[cv2])) cv2++;
          const int pi = (cv == cv2) ? -1 : A.pair_idx[(size_t)min(b, b2) * B + max(b, b2)];
          v = pi >= 0 ? A.pair_tot[(size_t)pi * K + k] : 0.0f;
        }
        cov[(size_t)prow[a2] * m + prow[a]] = (double)v;
      }
      __syncthreads();
    }
    if (!qtab && !A.ref_tot) {
    for (int i = tid; i < m * m; i += nt) cov[i] = 0.0;
    for (int i = tid; i < m * d; i += nt) rhs[i] = 0.0;
    __syncthreads();
    for (int q = 0; q < Q; q++) {   // sequential over combinations (fixed order), parallel inside
      int rows[17]; int nr = 1; rows[0] = 0;
      for (int c = 0; c < C; c++) { const int ro = row_of[D.qlev[q * C + c]]; if (ro >= 0) rows[nr++] = prow[ro]; }
      if (nr > 1) {   // (none of its levels kept: the combination's cells do not enter, :400,456-460)
        const double n = D.nq[(size_t)q * K + k];
        for (int i = tid; i < nr * nr; i += nt) { const int a = i / nr, b2 = i - a * nr; cov[(size_t)rows[b2] * m + rows[a]] += n; }
        const double* sq = D.Sq + ((size_t)q * K + k) * d;
        for (int i = tid; i < nr * d; i += nt) { const int j = i / nr, a = i - j * nr; rhs[(size_t)j * m + rows[a]] += sq[j]; }
      }
      __syncthreads();
    }
    }   // (!qtab)
    if (A.use_s0 && !A.ref_tot) {   // ridge_arith = 1, one covariate: the intercept row's own sequential fp32 totals (k_seq_ridge_store)
      if (tid == 0) cov[0] = D.n0[k];
      for (int j = tid; j < d; j += nt) rhs[(size_t)j * m] = D.S0[(size_t)k * d + j];
    }
    // lambda on the diagonal (intercept 0): estimation lambda = alpha * E[k,b]
    long long rs = 0;
    for (int b0 = 0; b0 < D.B0; b0++) rs += D.O_fx[(size_t)b0 * K + k];
    const double rsd = (double)rs * FX_INV;
    for (int a = 1 + tid; a < m; a += nt) {
      const int b = keepl[a - 1];
      const float lam = A.lambda ? A.lambda[b + 1] : (A.Ef ? A.Ef[(size_t)b * K + k] : (float)(rsd * (double)D.Pr_b[b])) * A.alpha;
      const int pa = prow[a];
      if (A.solve_f32 || A.use_s0) cov[(size_t)pa * m + pa] = (double)__fadd_rn((float)cov[(size_t)pa * m + pa], lam);     // an fp32 matrix in the reference
      else cov[(size_t)pa * m + pa] += (double)lam;
    }
    __syncthreads();
    if (A.solve_f32 && C == 1) {
      // The reference's own inverse for one covariate (src/harmony.cpp:575-586): Phi_cov is an arrowhead matrix, inverted in closed
      // form in fp32, and W = inv_cov * (the right-hand sides) accumulates in fp32, column of inv_cov after column (:599-608).
      float* const ac = reinterpret_cast<float*>(sm_ + PAN_OFF);     // [m] each (the Cholesky's panel space; solve_f32: no elimination, prow is the identity)
      float* const bb = ac + m; float* const acb = bb + m; float* const uu = acb + m;
      for (int a = tid; a < m; a += nt) {
        ac[a] = (a == 0) ? 1.0f : -(float)cov[(size_t)a * m];
        bb[a] = (a == 0) ? 0.0f : 1.0f / (float)cov[(size_t)a * m + a];
      }
      __syncthreads();
      if (tid == 0) {
        // accu(square(ac) % b) as Armadillo's linear accumulate walks it: two accumulators over the even / odd elements, added at the end
        // (arma::accu on an expression, src/harmony.cpp:581; the oracle restates the same order)
        float u1 = 0.0f, u2 = 0.0f;
        int a = 0;
        for (; a + 1 < m; a += 2) {
          u1 = __fadd_rn(u1, __fmul_rn(__fmul_rn(ac[a], ac[a]), bb[a]));
          u2 = __fadd_rn(u2, __fmul_rn(__fmul_rn(ac[a + 1], ac[a + 1]), bb[a + 1]));
        }
        if (a < m) u1 = __fadd_rn(u1, __fmul_rn(__fmul_rn(ac[a], ac[a]), bb[a]));
        const float u = __fadd_rn(u1, u2);
        uu[0] = 1.0f / __fsub_rn((float)cov[0], u);
        if (!(__fsub_rn((float)cov[0], u) != 0.0f)) misc[3] = 1;
      }
      for (int a = tid; a < m; a += nt) acb[a] = (a == 0) ? 1.0f : __fmul_rn(ac[a], bb[a]);
      __syncthreads();
      const float iu = uu[0];
      // W[r2][j] = sum_c2 inv[r2][c2] rhs[c2][j], c2 ascending; the result replaces rhs only after every entry has been computed
      float* const Wtmp = A.Wall + (size_t)k * d * M;
      for (int i = tid; i < m * d; i += nt) {
        const int j = i / m, r2 = i - j * m;
        float sacc = 0.0f;
        for (int c2 = 0; c2 < m; c2++) {
          float iv = __fmul_rn(iu, __fmul_rn(acb[r2], acb[c2]));
          if (c2 == r2) iv = __fadd_rn(iv, bb[r2]);
          sacc = __fadd_rn(sacc, __fmul_rn(iv, (float)rhs[(size_t)j * m + c2]));
        }
        Wtmp[i] = sacc;
      }
      __syncthreads();
      for (int i = tid; i < m * d; i += nt) rhs[i] = (double)Wtmp[i];
      __syncthreads();
    } else if (A.solve_f32 && C > 1) {
      // Several covariates: the reference calls arma::inv (src/harmony.cpp:573), i.e. LAPACK's LU of the fp32 matrix.  LAPACK's blocked
      // rounding order cannot be pinned without the reference's BLAS; what is reproduced bit for bit is the oracle's restatement: unblocked
      // fp32 LU with partial pivoting (first largest pivot) applied to the identity, back substitution column by column, then
      // W = inv_cov * rhs with sequential fp32 accumulation.  One workgroup per cluster; the m x m fp32 matrix and its inverse live in the
      // cluster's fp64 scratch (2 m^2 floats = m^2 doubles).
      float* const ac = reinterpret_cast<float*>(sm_ + PAN_OFF);      // [m] multipliers of the current column (the panel space)
      __shared__ unsigned long long pivkey;
      // the fp64 scratch holds the assembled fp32 values: repack them as floats (through registers: the two layouts overlap)
      {
        const int per = (m * m + nt - 1) / nt;
        float* const Af = reinterpret_cast<float*>(cov);
        for (int base = 0; base < per; base += 16) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; u++) { const int i = (base + u) * nt + tid; v[u] = (base + u < per && i < m * m) ? (float)cov[i] : 0.0f; }
          __syncthreads();
#pragma unroll
          for (int u = 0; u < 16; u++) { const int i = (base + u) * nt + tid; if (base + u < per && i < m * m) Af[i] = v[u]; }
          __syncthreads();
        }
      }
      float* const Af = reinterpret_cast<float*>(cov);
      float* const If = Af + (size_t)m * m;
      for (int i = tid; i < m * m; i += nt) If[i] = (i / m == i % m) ? 1.0f : 0.0f;
      __syncthreads();
      for (int cc = 0; cc < m && !misc[3]; cc++) {
        if (tid == 0) pivkey = 0ull;
        __syncthreads();
        unsigned long long best = 0ull;
        for (int r = cc + tid; r < m; r += nt) {
          const unsigned long long key = ((unsigned long long)(__float_as_uint(fabsf(Af[(size_t)cc * m + r]))) << 32) | (unsigned)(0x7fffffff - r);   // largest |a|, then the smallest row
          best = key > best ? key : best;
        }
        if (best) atomicMax(&pivkey, best);
        __syncthreads();
        const unsigned long long pk = pivkey;
        if ((pk >> 32) == 0ull) { if (tid == 0) misc[3] = 1; __syncthreads(); break; }
        const int pr = 0x7fffffff - (int)(pk & 0xffffffffull);
        if (pr != cc) {
          for (int j = tid; j < 2 * m; j += nt) {
            float* M2 = j < m ? Af + (size_t)j * m : If + (size_t)(j - m) * m;
            const float t = M2[cc]; M2[cc] = M2[pr]; M2[pr] = t;
          }
        }
        __syncthreads();
        const float inv = 1.0f / Af[(size_t)cc * m + cc];
        for (int r = cc + 1 + tid; r < m; r += nt) { const float f = __fmul_rn(Af[(size_t)cc * m + r], inv); ac[r] = f; }
        __syncthreads();
        for (int r = cc + 1 + tid; r < m; r += nt) if (ac[r] != 0.0f) Af[(size_t)cc * m + r] = ac[r];
        const int h = m - cc - 1;              // rows below the pivot
        // A[j][r] -= f_r A[j][cc] (j > cc), I[j][r] -= f_r I[j][cc] (all j): element-wise, one rounding for the product, one for the difference
        for (int i = tid; i < (h + m) * h; i += nt) {
          const int jj = i / h, r = cc + 1 + (i - jj * h);
          float* col = jj < h ? Af + (size_t)(cc + 1 + jj) * m : If + (size_t)(jj - h) * m;
          const float f = ac[r];
          if (f != 0.0f) col[r] = __fsub_rn(col[r], __fmul_rn(f, col[cc]));
        }
        __syncthreads();
      }
      if (!misc[3]) {
        // back substitution of every column of the identity: s -= A[c][r] * x[c], c ascending, then / A[r][r]
        for (int j = tid; j < m; j += nt) {
          float* x = If + (size_t)j * m;
          for (int r = m - 1; r >= 0; r--) {
            float sacc = x[r];
            for (int c2 = r + 1; c2 < m; c2++) sacc = __fsub_rn(sacc, __fmul_rn(Af[(size_t)c2 * m + r], x[c2]));
            x[r] = sacc / Af[(size_t)r * m + r];
          }
        }
        __syncthreads();
        float* const Wtmp = A.Wall + (size_t)k * d * M;
        for (int i = tid; i < m * d; i += nt) {
          const int j = i / m, r2 = i - j * m;
          float sacc = 0.0f;
          for (int c2 = 0; c2 < m; c2++) sacc = __fadd_rn(sacc, __fmul_rn(If[(size_t)c2 * m + r2], (float)rhs[(size_t)j * m + c2]));
          Wtmp[i] = sacc;
        }
        __syncthreads();
        for (int i = tid; i < m * d; i += nt) rhs[i] = (double)Wtmp[i];
      }
      __syncthreads();
    } else {
    // ---- blocked right-looking Cholesky (lower, column-major, in place in the L2-resident scratch).  Panels of NBW columns
    // are factored in LDS; the trailing matrix then receives ONE rank-NBW update per panel (16 fused multiply-adds per global
    // read-modify-write) instead of one rank-1 update per column -- at m = 201 (configs[4]: 200 levels) 13 passes over the
    // trailing matrix and ~40 workgroup barriers instead of 201 passes and 600 barriers (8.3 ms -> well under 1 ms per correction).
    constexpr int NBW = 16;
    double* const P = reinterpret_cast<double*>(sm_ + PAN_OFF);     // [rows of the panel][NBW], LDS
    const int nd = sch_nd, ms = m - nd;      // rows [ms, m): the diagonal block D; [0, ms): everything else
    if (nd > 0) {
      // S = A - B^T D^-1 B, rhs_A -= B^T D^-1 rhs_D   (cov = [[A, B^T], [B, D]], B = rows >= ms of the first ms columns)
      for (int r = ms + tid; r < m; r += nt) { const double dv = cov[(size_t)r * m + r]; if (!(dv > 0.0)) misc[3] = 1; cov[(size_t)r * m + r] = 1.0 / dv; }
      __syncthreads();
      // (with the coupling masks only the levels that really meet a row are visited -- the skipped terms are exact zeros, the sums
      //  keep their order: B is sparse, at configs[4]'s nested covariates a level of the eliminated one meets 3 of the 73 other rows)
      for (int i = tid; i < ms * ms; i += nt) {
        const int cb = i / ms, ra = i - cb * ms;
        if (ra < cb) continue;                      // the Cholesky reads the lower triangle only
        const double* ca = cov + (size_t)ra * m; const double* cc = cov + (size_t)cb * m;
        double sacc = 0.0;
        if (dmask) {
          for (int w2 = 0; w2 < mw; w2++) {
            unsigned long long bits = dmask[(size_t)ra * mw + w2] & dmask[(size_t)cb * mw + w2];
            while (bits) { const int r = ms + 64 * w2 + __builtin_ctzll(bits); bits &= bits - 1; sacc += ca[r] * cov[(size_t)r * m + r] * cc[r]; }
          }
        } else {
          for (int r = ms; r < m; r++) sacc += ca[r] * cov[(size_t)r * m + r] * cc[r];
        }
        cov[(size_t)cb * m + ra] -= sacc;
      }
      for (int i = tid; i < ms * d; i += nt) {
        const int j = i / ms, ra = i - j * ms;
        const double* ca = cov + (size_t)ra * m;
        double sacc = 0.0;
        if (dmask) {
          for (int w2 = 0; w2 < mw; w2++) {
            unsigned long long bits = dmask[(size_t)ra * mw + w2];
            while (bits) { const int r = ms + 64 * w2 + __builtin_ctzll(bits); bits &= bits - 1; sacc += ca[r] * cov[(size_t)r * m + r] * rhs[(size_t)j * m + r]; }
          }
        } else {
          for (int r = ms; r < m; r++) sacc += ca[r] * cov[(size_t)r * m + r] * rhs[(size_t)j * m + r];
        }
        rhs[(size_t)j * m + ra] -= sacc;
      }
      __syncthreads();
    }
    for (int c0 = 0; c0 < ms && !misc[3]; c0 += NBW) {
      const int nbw = min(NBW, ms - c0), h = ms - c0;
      for (int i = tid; i < h * nbw; i += nt) { const int kk = i / h, r = i - kk * h; P[r * NBW + kk] = cov[(size_t)(c0 + kk) * m + c0 + r]; }
      __syncthreads();
      for (int kk = 0; kk < nbw; kk++) {
        if (tid == 0) { const double sdiag = P[kk * NBW + kk]; if (!(sdiag > 0.0)) misc[3] = 1; else P[kk * NBW + kk] = sqrt(sdiag); }
        __syncthreads();
        if (misc[3]) break;
        const double linv = 1.0 / P[kk * NBW + kk];
        for (int r = kk + 1 + tid; r < h; r += nt) P[r * NBW + kk] *= linv;
        __syncthreads();
        const int rest = nbw - kk - 1;
        for (int i = tid; i < rest * h; i += nt) {
          const int k2 = kk + 1 + i / h, r = i % h;
          if (r >= k2) P[r * NBW + k2] -= P[r * NBW + kk] * P[k2 * NBW + kk];
        }
        __syncthreads();
      }
      if (misc[3]) break;
      for (int i = tid; i < h * nbw; i += nt) { const int kk = i / h, r = i - kk * h; if (r >= kk) cov[(size_t)(c0 + kk) * m + c0 + r] = P[r * NBW + kk]; }
      // trailing update: 32 consecutive rows per column and thread row (coalesced), the column's panel row cached in registers
      const int tx = tid & 31, ty = tid >> 5, TY = nt >> 5;
      for (int c2 = c0 + nbw + ty; c2 < ms; c2 += TY) {
        double pc[NBW];
#pragma unroll
        for (int kk = 0; kk < NBW; kk++) pc[kk] = (kk < nbw) ? P[(c2 - c0) * NBW + kk] : 0.0;
        double* colp = cov + (size_t)c2 * m;
        for (int r = c2 + tx; r < ms; r += 32) {
          const double* pr = P + (r - c0) * NBW;
          double sacc = 0.0;
#pragma unroll
          for (int kk = 0; kk < NBW; kk++) sacc += pr[kk] * pc[kk];
          colp[r] -= sacc;
        }
      }
      __syncthreads();
    }
    if (!misc[3]) {
      // ---- forward / back substitution for the d right-hand sides, 16 lanes per right-hand side, b in LDS when it fits
      const bool blds = (size_t)d * m * sizeof(double) <= A.lds_b_bytes;
      double* const bl = blds ? P : rhs;     // (the panel space is free now)
      if (blds) { for (int i = tid; i < d * m; i += nt) bl[i] = rhs[i]; }
      __syncthreads();
      const int ln = tid & 15;
      for (int j = tid >> 4; j < d; j += nt >> 4) {
        double* b = bl + (size_t)j * m;
        // L y = b, column by column: y_r = b_r / L_rr, then b_k -= L_kr y_r (k > r): L's column r is contiguous in k
        for (int r = 0; r < ms; r++) {
          const double* Lc = cov + (size_t)r * m;
          const double y = b[r] / Lc[r];
          if (ln == 0) b[r] = y;
          for (int kk = r + 1 + ln; kk < ms; kk += 16) b[kk] -= Lc[kk] * y;
        }
        // L^T x = y, row by row from the bottom: x_r = (y_r - sum_{k>r} L_kr x_k) / L_rr
        for (int r = ms - 1; r >= 0; r--) {
          const double* Lc = cov + (size_t)r * m;
          double t = 0.0;
          for (int kk = r + 1 + ln; kk < ms; kk += 16) t += Lc[kk] * b[kk];
          t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 1, 64);
          const double x = (b[r] - t) / Lc[r];
          if (ln == 0) b[r] = x;
        }
      }
      __syncthreads();
      if (blds) { for (int i = tid; i < d * m; i += nt) rhs[i] = bl[i]; }
      if (nd > 0) {      // x_D = D^-1 (rhs_D - B x_A): B's entries (rows >= ms of the first ms columns) were never touched
        __syncthreads();
        for (int i = tid; i < nd * d; i += nt) {
          const int j = i / nd, r = ms + (i - j * nd);
          double sacc = rhs[(size_t)j * m + r];
          if (dmask) { for (int a = 0; a < ms; a++) if ((dmask[(size_t)a * mw + ((r - ms) >> 6)] >> ((r - ms) & 63)) & 1ull) sacc -= cov[(size_t)a * m + r] * rhs[(size_t)j * m + a]; }
          else for (int a = 0; a < ms; a++) sacc -= cov[(size_t)a * m + r] * rhs[(size_t)j * m + a];
          rhs[(size_t)j * m + r] = sacc * cov[(size_t)r * m + r];
        }
      }
    }
    }   // (fp64 Cholesky branch)
    __syncthreads();
  }
  if (tid == 0) {
    A.flags[k] = (full ? 0 : 1) | (skipped ? 2 : 0) | (misc[3] ? 4 : 0);
    if (misc[3] && A.err) atomicOr(A.err, 1);
    A.mrows[k] = skipped ? 0 : m;
  }
  const bool solved = !skipped && !misc[3];
  float* Wk = A.Wall + (size_t)k * d * M;
  __shared__ float ynew[128];
  for (int j = tid; j < d; j += nt) {
    float y = D.Ycur[(size_t)k * d + j];                      // a skipped cluster keeps its centroid
    if (solved) { y = (float)rhs[(size_t)j * m]; rhs[(size_t)j * m] = 0.0; }   // :610-611
    ynew[j] = y;
  }
  __syncthreads();
  if (solved) {
    for (int i = tid; i < m * d; i += nt) { const int j = i / m, a = i - j * m; Wk[i] = (float)rhs[(size_t)j * m + prow[a]]; }     // reference row order
    __syncthreads();
    for (int i = tid; i < Q * d; i += nt) {
      const int q = i / d, j = i - q * d;
      float w = 0.0f;
      for (int c = 0; c < C; c++) { const int ro = row_of[D.qlev[q * C + c]]; if (ro >= 0) w += Wk[(size_t)j * m + ro]; }
      D.Wq[((size_t)q * K + k) * d + j] = w;
    }
  }
  if (D.moe_mfma) {   // MFMA B-operand image of the correction table (clusters = reduction dim), as the host builder lays it out
    int s_, p_;
    if (k < 16 * D.wNT4) { const int t = k >> 4, r = k & 15; p_ = r >> 2; s_ = 4 * t + (r & 3); }
    else { const int r = k - 16 * D.wNT4; s_ = 4 * D.wNT4 + (r >> 2); p_ = r & 3; }
    __syncthreads();
    for (int i = tid; i < Q * d; i += nt) {
      const int q = i / d, j = i - q * d;
      const int qd = j >> 6, ii = (j & 63) >> 4, cc = j & 15;
      D.Wimg[((((size_t)q * D.wNQ + qd) * D.wNS + s_) * 4 + p_) * 64 + cc * 4 + ii] = D.Wq[((size_t)q * K + k) * d + j];
    }
  }
  // Y[:,k] <- normalise (:633): sequential fp32 sum of squares, as the host-side normalise does
  __shared__ float nrm_;
  if (tid == 0) { float sacc = 0.0f; for (int j = 0; j < d; j++) sacc += ynew[j] * ynew[j]; float nn = sqrtf(sacc); if (nn == 0.0f) nn = 1.0f; nrm_ = nn; }
  __syncthreads();
  for (int j = tid; j < d; j += nt) {
    const float y = ynew[j] / nrm_;
    D.Ycur[(size_t)k * d + j] = y;
    D.Yt[(size_t)j * K + k] = y;
    D.Yimg[yimg_index(D, j, k)] = y;
    bfimg_store(D.Yimg3, D.NCT, D.NS2, j, k, y);
  }
}

// ---- MFMA variants of the two MoE passes (static 16-cell tiles, rows of a tile are contiguous in HBM) ----
// k_moe_stats_mfma: Sq[q] (K x d) += R_tile^T (K x 16) * Zo_tile (16 x d): the 16 cells are the MFMA reduction dim.
//   A[i = cluster 16ct+(l&15)][slot l>>4] = R[cell 4s+(l>>4)][cluster],  B[slot][j = PC 16pt+(l&15)] = Zo[cell][PC]
//   D: lane l holds clusters 16ct+4(l>>4)+reg x PC 16pt+(l&15); accumulated over a run of tiles, flushed with fp64 atomics.
// Work split: a workgroup streams a contiguous range of tiles; its wave w owns PC tile w (blockDim = 64*ceil(d/16)),
// so a wave carries only NCT fp32 MFMA accumulators (folded into fp64 shadows every 4 tiles = 64 cells) and the
// K x d result of a run is flushed ONCE per workgroup, not once per wave (the fp64 atomics dominated otherwise).
// CTS > 1 (K > 128): the cluster tiles are split over CTS workgroups per tile range (blockIdx % CTS = cluster-tile group),
// each wave carrying NCTT / CTS accumulators + fp64 shadows.  (For K <= 128 the split was measured SLOWER -- 597 vs 355 us: the
// operand loads, not the registers, limit this kernel -- so it is only used where one wave cannot hold all cluster tiles.)
// (launch bounds 256 = one wave per SIMD with the whole 512-entry register file: with a 256-register budget hipcc serialises
//  the 32 operand loads of a tile with a wait after each, 355 -> 800 us -- hence the split across workgroups, not waves)
template <int NCTT, int CTS>
__global__ __launch_bounds__(256) void k_moe_stats_mfma(Dev D, int tiles_per_wg, int npt) {
  constexpr int NCT = (NCTT + CTS - 1) / CTS;   // cluster tiles of this wave
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int pt = threadIdx.x >> 6, grp = (CTS > 1) ? (int)(blockIdx.x % CTS) : 0, ct0 = grp * NCT;   // first cluster tile of this workgroup
  const int K = D.K, d = D.d, zs = D.zs;
  const int ts = (int)(blockIdx.x / CTS) * tiles_per_wg, te = min(D.ntitems, ts + tiles_per_wg);
  if (ts >= te) return;
  const int jj = 16 * pt + c;           // this lane's PC
  const bool jv = jj < d;
  f32x4 acc[NCT];
  double sh[NCT][4], nsh[NCT];
  float nacc[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) {
    acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; nacc[ct] = 0.0f; nsh[ct] = 0.0;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) sh[ct][reg] = 0.0;
  }
  auto fold = [&]() {
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) sh[ct][reg] += (double)acc[ct][reg];
      acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      nsh[ct] += (double)nacc[ct]; nacc[ct] = 0.0f;
    }
  };
  auto flush = [&](int q) {
    double* S = D.Sq + (size_t)q * d * K;
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int k = 16 * (ct0 + ct) + 4 * g + reg;
        if (jv && k < K && sh[ct][reg] != 0.0) atomicAdd(&S[(size_t)k * d + jj], sh[ct][reg]);
        sh[ct][reg] = 0.0;
      }
      if (pt == 0) {                            // sum_i R_ki of cluster 16ct+c: add the four cell slots
        double v = nsh[ct];
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (g == 0 && 16 * (ct0 + ct) + c < K) atomicAdd(&D.nq[(size_t)q * K + 16 * (ct0 + ct) + c], v);
      }
      nsh[ct] = 0.0;
    }
  };
  // tile descriptors one tile ahead and as per-lane loads (a uniform load becomes load + readfirstlane + vmcnt(0): one more
  // serial memory latency per tile in front of the operand loads)
  auto item_at = [&](int tile) -> Item {
    const Item* tp = D.titems + min(tile, te - 1);
    asm volatile("" : "+v"(tp));
    return *tp;
  };
  Item itN = item_at(ts);
  int curq = __builtin_amdgcn_readfirstlane(itN.q);
  for (int tile = ts; tile < te; ++tile) {
    const Item it = itN;
    itN = item_at(tile + 1);
    const int tq = __builtin_amdgcn_readfirstlane(it.q);
    if (tq != curq) { fold(); flush(curq); curq = tq; }
    const size_t c0 = (size_t)it.start;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int cell = 4 * st + g;                       // cell slot of this lane in this step
      const bool cv = cell < it.cnt;
      const size_t row = c0 + (cv ? cell : 0);
      float a[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) a[ct] = ld_or(D.R, row * K + min(16 * (ct0 + ct) + c, K - 1), cv && 16 * (ct0 + ct) + c < K, 0.0f);
      const float b = ld_or(D.Zo, row * zs + min(jj, zs - 1), cv && jv, 0.0f);
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) {
        nacc[ct] += a[ct];
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ct], b, acc[ct], 0, 0, 0);
      }
    }
    if (((tile - ts) & 3) == 3) fold();
  }
  fold(); flush(curq);
}

// k_moe_stats_q: k_moe_stats_mfma with (a) 16-byte operand loads and (b) a bit-reproducible reduction (K <= 128).
//  (a) Which cluster an MFMA row stands for is free: inside a quad of cluster tiles (64 clusters) row m of tile i stands for
//      cluster 64 qd + nt m + i (nt = tiles of the quad: 4, or what is left in the last quad), so the nt A operands a lane
//      needs for one cell are CONSECUTIVE floats of its R row -- one dwordx4 (x3 / x2 / x1) load instead of nt dword gathers:
//      9 load instructions per tile and lane at K = 100 instead of 32.  Only the flush has to know the mapping.
//  (b) no floating-point atomics: every workgroup writes the K x (d+1) partial of each combination run it meets to its own slot
//      (slot order = workgroup order = cell order); k_moe_stats_reduce adds a combination's slots in ascending order -> Sq, nq
//      (and with them Z_corr) are identical from run to run.
template <int NCT>
__global__ __launch_bounds__(320) void k_moe_stats_q(Dev D, int tiles_per_wg) {
  constexpr int NQD = (NCT + 3) / 4, NTL = NCT - 4 * (NQD - 1);     // quads; tiles in the last quad
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int pt = threadIdx.x >> 6;
  const int K = D.K, d = D.d, zs = D.zs;
  const int ts = (int)blockIdx.x * tiles_per_wg, te = min(D.ntitems, ts + tiles_per_wg);
  if (ts >= te) return;
  const int jj = 16 * pt + c;           // this lane's PC
  const bool jv = jj < d;
  const bool bcol1 = jj == d;           // column d of the B operand is all ones: its output column is sum_i R_ki (the launcher
                                        // adds a wave when d is a multiple of 16)
  f32x4 acc[NCT];
  double sh[NCT][4];
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) {
    acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int reg = 0; reg < 4; reg++) sh[ct][reg] = 0.0;
  }
  auto kmap = [&](int ct, int m) -> int { const int qd = ct >> 2, nt = (qd == NQD - 1) ? NTL : 4; return 64 * qd + nt * m + (ct & 3); };
  auto fold = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) sh[ct][reg] += (double)acc[ct][reg];
      acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  int slot = D.st_slot0[blockIdx.x];
  auto flush = [&]() {
    double* S = D.st_part + (size_t)slot * ((size_t)K * d + K);
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int k = kmap(ct, 4 * g + reg);
        if (jv && k < K) S[(size_t)k * d + jj] = sh[ct][reg];
        if (bcol1 && k < K) S[(size_t)K * d + k] = sh[ct][reg];       // the ones column: sum_i R_ki
        sh[ct][reg] = 0.0;
      }
    }
    slot++;
  };
  auto item_at = [&](int tile) -> Item {       // per-lane load (a uniform one costs a vmcnt(0) per tile, see k_tile)
    const Item* tp = D.titems + min(tile, te - 1);
    asm volatile("" : "+v"(tp));
    return *tp;
  };
  // this lane's first cluster in every quad (clamped so that the whole load stays inside the row) and which of them exist
  int koff[NQD];
#pragma unroll
  for (int qd = 0; qd < NQD; qd++) {
    const int nt = (qd == NQD - 1) ? NTL : 4;
    // (a lane whose first cluster exists loads at its exact offset: the tail of a partially valid load runs into the next row --
    //  R has a dummy row behind the last cell -- and is masked by kval; only fully invalid lanes are clamped)
    koff[qd] = (64 * qd + nt * c < K) ? 64 * qd + nt * c : K - nt;
  }
  // (a software pipeline over tiles -- operands of tile t+1 requested before tile t's MFMAs -- was measured SLOWER, 413 vs 388 us:
  //  290 VGPRs leave one wave per SIMD; the pass is not bound by the loads' latency)
  Item itN = item_at(ts);
  int curq = __builtin_amdgcn_readfirstlane(itN.q);
  for (int tile = ts; tile < te; ++tile) {
    const Item it = itN;
    itN = item_at(tile + 1);
    const int tq = __builtin_amdgcn_readfirstlane(it.q);
    if (tq != curq) { fold(); flush(); curq = tq; }
    const size_t c0 = (size_t)it.start;
    float a[4][NCT], b[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {           // all of the tile's operand loads first
      const int cell = 4 * st + g;
      const size_t row = c0 + (cell < it.cnt ? cell : 0);
      const float* rr = D.R + row * K;
#pragma unroll
      for (int qd = 0; qd < NQD; qd++) {
        if (qd < NQD - 1 || NTL == 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(rr + koff[qd]);
#pragma unroll
          for (int i = 0; i < 4; i++) if (4 * qd + i < NCT) a[st][4 * qd + i] = v[i];
        } else if (NTL == 3) {
          struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
          const F3 v = *reinterpret_cast<const F3*>(rr + koff[qd]);
          a[st][4 * qd] = v.x; a[st][4 * qd + 1] = v.y; a[st][4 * qd + 2] = v.z;
        } else if (NTL == 2) {
          const float2 v = *reinterpret_cast<const float2*>(rr + koff[qd]);
          a[st][4 * qd] = v.x; a[st][4 * qd + 1] = v.y;
        } else a[st][4 * qd] = rr[koff[qd]];
      }
      b[st] = D.Zo[row * zs + min(jj, zs - 1)];
    }
    if (it.cnt == 16) {
      // full tile (all but the last tile of a combination): NO masking at all.  Rows of the result that stand for clusters >= K
      // and columns that stand for PCs >= d hold finite garbage (clamped loads) and are never flushed.  sum_i R_ki comes out of
      // the same MFMAs: the PC tile that has room for it feeds a column of ones (bcol1).
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const float bb = bcol1 ? 1.0f : b[st];
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st][ct], bb, acc[ct], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const bool cv = 4 * st + g < it.cnt;
        const float bb = cv ? (bcol1 ? 1.0f : b[st]) : 0.0f;
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv ? a[st][ct] : 0.0f, bb, acc[ct], 0, 0, 0);
      }
    }
    if (((tile - ts) & 3) == 3) fold();
  }
  fold(); flush();
}
// Sq[q], nq[q] = sum of the combination's partial slots in ascending slot (= cell) order: fixed order, no atomics
__global__ __launch_bounds__(256) void k_moe_stats_reduce(Dev D) {
  const int q = blockIdx.x;
  const size_t per = (size_t)D.K * D.d + D.K;
  const int s0 = D.st_qptr[q], s1 = D.st_qptr[q + 1];
  const size_t e = (size_t)blockIdx.y * blockDim.x + threadIdx.x;    // one entry per thread, eight slots in flight
  if (e >= per) return;
  double v = 0.0;
  int sidx = s0;
  for (; sidx + 8 <= s1; sidx += 8) {
    double t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = D.st_part[(size_t)D.st_qslots[sidx + i] * per + e];
#pragma unroll
    for (int i = 0; i < 8; i++) v += t[i];                            // ascending slot order: the sum is reproducible
  }
  for (; sidx < s1; sidx++) v += D.st_part[(size_t)D.st_qslots[sidx] * per + e];
  if (e < (size_t)D.K * D.d) D.Sq[(size_t)q * D.K * D.d + e] = v;
  else D.nq[(size_t)q * D.K + (e - (size_t)D.K * D.d)] = v;
}

// k_moe_apply_mfma: Z_corr tile (16 x d) = Z_orig tile - R tile (16 x K) * Wq[q] (K x d); clusters are the MFMA reduction
// dim, so this is tile_dots with (rows = R rows, "centroid image" = Wimg[q] staged in LDS).  One workgroup per apply item.
template <int NPT>
__global__ __launch_bounds__(256) void k_moe_apply_mfma(Dev D) {
  extern __shared__ __attribute__((aligned(16))) f32x4 ldsW4[];
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wib = threadIdx.x >> 6;
  const int K = D.K, d = D.d, zs = D.zs;
  const int nW4 = D.wNQ * D.wNS * 64;
  for (int itx = blockIdx.x; itx < D.naitems; itx += gridDim.x) {
    const Item item = D.aitems[itx];
    __syncthreads();
    const f32x4* src = reinterpret_cast<const f32x4*>(D.Wimg) + (size_t)item.q * nW4;
    for (int i = threadIdx.x; i < nW4; i += blockDim.x) ldsW4[i] = src[i];
    __syncthreads();
    const int ntl = (item.cnt + 15) >> 4;
    for (int tl = wib; tl < ntl; tl += 4) {
      const int cell0 = item.start + 16 * tl;
      const int nvalid = min(16, item.start + item.cnt - cell0);
      const bool av = c < nvalid;
      // Z_orig of the tile is requested BEFORE the MFMA chain (its latency hides behind the 100 MFMAs instead of following them)
      float zo[4][NPT];
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int cl = 4 * g + reg;
        const size_t row = (size_t)(cell0 + (cl < nvalid ? cl : 0)) * zs;
#pragma unroll
        for (int pt = 0; pt < NPT; pt++) zo[reg][pt] = D.Zo[row + min(16 * pt + c, zs - 1)];
      }
      f32x4 acc[NPT];
      tile_dots<NPT>(ldsW4, D.R + (size_t)(cell0 + (av ? c : 0)) * K, av, g, lane, D.wNS, D.wNT4, D.wtail, acc);
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int cl = 4 * g + reg;
        const bool cv = cl < nvalid;
        const size_t row = (size_t)(cell0 + (cv ? cl : 0)) * zs;
#pragma unroll
        for (int pt = 0; pt < NPT; pt++) {
          const int jj = 16 * pt + c;
          if (cv && jj < d) D.Zc[row + jj] = zo[reg][pt] - acc[pt][reg];
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------
// kmeans_centers (src/utils.cpp:10-64)
//   k_seed_probe: for every anchor i (= cluster lane) sample a cell with P ~ |2(1 - y_i.x)| via the
//                 exponential race  argmin_n  -log(u_{i,n}) / dist_{i,n}   (:27-34), all K anchors
//                 in ONE pass; result = packed (key bits, global cell) min per cluster.
//   k_lloyd     : one Lloyd iteration: nearest centre (Euclidean), sums and counts.
// --------------------------------------------------------------------------------------
template <int KPL, int DPL>
__global__ __launch_bounds__(TPB) void k_seed_probe(Dev D, uint64_t seed, uint64_t goff, const unsigned* __restrict__ excl,
                                                    int nexcl) {
  extern __shared__ __attribute__((aligned(16))) float ldsY[];
  stage_Y(ldsY, D.Yt, D.d, D.K, D.KP);
  constexpr int CB = 4;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const int d = D.d, K = D.K;
  uint64_t sk[KPL]; unsigned long long best[KPL];
#pragma unroll
  for (int q = 0; q < KPL; q++) {
    sk[q] = splitmix64(seed ^ ((uint64_t)(1 + lane + 64 * q) * 0xD1342543DE82EF95ull));
    best[q] = ~0ull;
  }
  for (int it = wave; it < D.nitems; it += nw) {
    const Item item = D.items[it];
    for (int p = 0; p < item.cnt; p += CB) {
      const int nc = min(CB, item.cnt - p);
      float z[CB][DPL];
#pragma unroll
      for (int c = 0; c < CB; c++) {
        if (c < nc) load_row<DPL>(D.Zc, (size_t)(item.start + p + c), D.zs, d, lane, z[c]);
        else {
#pragma unroll
          for (int t = 0; t < DPL; t++) z[c][t] = 0.0f;
        }
      }
      float acc[CB][KPL];
      group_dots<KPL, DPL, CB>(ldsY, d, D.KP, lane, z, acc);
#pragma unroll
      for (int c = 0; c < CB; c++) {
        if (c < nc) {
          const int cell = item.start + p + c;
          const uint64_t g = goff + (uint64_t)D.perm[cell];
          bool skip = false;
          for (int x = 0; x < nexcl; x++) skip |= ((uint64_t)excl[x] == g);
          if (!skip) {
#pragma unroll
            for (int q = 0; q < KPL; q++) {
              if (lane + 64 * q < K) {
                const float dis = fabsf(2.0f * (1.0f - acc[c][q]));
                const uint64_t h = splitmix64(sk[q] + g);
                const float u = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
                const float key = -logf(u) / dis;  // >= 0 (or +inf / nan when dis == 0)
                unsigned kb = __float_as_uint(key);
                if (!(key >= 0.0f)) kb = 0x7f800000u;
          kb &= 0x7fffffffu;   // -0.0 (u == 1: -log(u) = -0) must order as zero, not as the largest unsigned pattern  // nan -> +inf: never the minimum
                const unsigned long long pk = ((unsigned long long)kb << 32) | (unsigned long long)(uint32_t)g;
                best[q] = pk < best[q] ? pk : best[q];
              }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < KPL; q++)
    if (lane + 64 * q < K && best[q] != ~0ull) atomicMin(&D.seedmin[lane + 64 * q], best[q]);
}

// R-compatible seeding race (hmx_set_int "rng" = 1): the uniforms come from the HOST's stream (MT19937 seeded like set.seed, or
// the host's unif_rand), u[a][local original cell] for the anchors a0 .. a0+na-1; per anchor the race of src/utils.cpp:24-34
// over this shard's cells: key = -log(u) / |2(1 - y_a.x)|, index_min (ties -> the smaller global cell index).
__global__ __launch_bounds__(256) void k_seed_race_u(Dev D, const float* __restrict__ u, int a0, int na, int a_lo, int a_hi,
                                                     uint64_t goff, const unsigned* __restrict__ excl, int nexcl) {
  extern __shared__ float ys_[];   // [na][d] anchor rows
  const int d = D.d, zs = D.zs, n = D.n;
  for (int i = threadIdx.x; i < na * d; i += blockDim.x) ys_[i] = D.Ycur[(size_t)a0 * d + i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (int a = a_lo; a < a_hi; a++) {
    unsigned long long best = ~0ull;
    const float* y = ys_ + a * d;
    for (size_t p = gt; p < (size_t)n; p += stride) {
      const float* z = D.Zc + p * zs;
      float dot = 0.0f;
      for (int j = 0; j < d; j++) dot = fmaf(y[j], z[j], dot);
      const int loc = D.perm[p];
      const uint64_t gg = goff + (uint64_t)loc;
      bool skip = false;
      for (int x = 0; x < nexcl; x++) skip |= ((uint64_t)excl[x] == gg);
      const float dis = fabsf(2.0f * (1.0f - dot));
      const float key = -logf(u[(size_t)a * n + loc]) / dis;
      unsigned kb = __float_as_uint(key);
      if (!(key >= 0.0f)) kb = 0x7f800000u;
          kb &= 0x7fffffffu;   // -0.0 (u == 1: -log(u) = -0) must order as zero, not as the largest unsigned pattern
      const unsigned long long pk = ((unsigned long long)kb << 32) | (unsigned long long)(uint32_t)gg;
      if (!skip && pk < best) best = pk;
    }
    best = wmin64(best);
    if (lane == 0 && best != ~0ull) atomicMin(&D.seedmin[a0 + a], best);
  }
}

// rows[k][:] = Z_corr row of global cell gcells[k] if it lives on this shard, else 0 (summed across ranks)
__global__ void k_gather_rows(Dev D, const long long* __restrict__ gcells, uint64_t goff, double* __restrict__ rows) {
  const int k = blockIdx.x;
  const long long g = gcells[k];
  const long long loc = g - (long long)goff;
  const bool mine = (loc >= 0 && loc < (long long)D.n);
  for (int j = threadIdx.x; j < D.d; j += blockDim.x)
    rows[(size_t)k * D.d + j] = mine ? (double)D.Zc[(size_t)D.invperm[loc] * D.zs + j] : 0.0;
}

template <int KPL, int DPL>
__global__ __launch_bounds__(TPB) void k_lloyd(Dev D) {
  // LDS: [ centroids d*KP floats | (D.lloyd_lds) K*d 64-bit fixed-point sums + K counts ]
  extern __shared__ __attribute__((aligned(16))) float ldsY[];
  long long* ltab = reinterpret_cast<long long*>(ldsY + (((size_t)D.d * D.KP + 1) & ~(size_t)1));
  if (D.lloyd_lds) for (int i = threadIdx.x; i < D.K * D.d + D.K; i += blockDim.x) ltab[i] = 0;
  stage_Y(ldsY, D.Yt, D.d, D.K, D.KP);
  constexpr int CB = 4;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const int d = D.d, K = D.K;
  float yn[KPL];
#pragma unroll
  for (int q = 0; q < KPL; q++) yn[q] = ld_or(D.ynorm, (size_t)min(lane + 64 * q, K - 1), lane + 64 * q < K, 0.0f);
  for (int it = wave; it < D.nitems; it += nw) {
    const Item item = D.items[it];
    for (int p = 0; p < item.cnt; p += CB) {
      const int nc = min(CB, item.cnt - p);
      float z[CB][DPL];
#pragma unroll
      for (int c = 0; c < CB; c++) {
        if (c < nc) load_row<DPL>(D.Zc, (size_t)(item.start + p + c), D.zs, d, lane, z[c]);
        else {
#pragma unroll
          for (int t = 0; t < DPL; t++) z[c][t] = 0.0f;
        }
      }
      float acc[CB][KPL];
      group_dots<KPL, DPL, CB>(ldsY, d, D.KP, lane, z, acc);
#pragma unroll
      for (int c = 0; c < CB; c++) {
        if (c < nc) {
          unsigned long long best = ~0ull;
#pragma unroll
          for (int q = 0; q < KPL; q++) {
            const int k = lane + 64 * q;
            if (k < K) {
              const float sc = yn[q] - 2.0f * acc[c][q];  // ||x-y||^2 - ||x||^2
              unsigned ub = __float_as_uint(sc);
              ub = (ub & 0x80000000u) ? ~ub : (ub | 0x80000000u);  // order-preserving map
              const unsigned long long pk = ((unsigned long long)ub << 32) | (unsigned)k;
              best = pk < best ? pk : best;
            }
          }
          best = wmin64(best);
          const int kb = (int)(best & 0xffffffffu);
#pragma unroll
          // unit-vector components as 2^30 fixed point: exact, order-independent 64-bit sums (LDS-private per
          // workgroup when the table fits, then one global atomic per entry)
          for (int t = 0; t < DPL; t++) {
            const int jj = 64 * t + lane;
            if (jj < d) {
              const unsigned long long v = (unsigned long long)__float2ll_rn(z[c][t] * 1073741824.0f);
              if (D.lloyd_lds) atomicAdd((unsigned long long*)&ltab[kb * d + jj], v);
              else atomicAdd((unsigned long long*)&D.lsum[(size_t)kb * d + jj], v);
            }
          }
          if (lane == 0) {
            if (D.lloyd_lds) atomicAdd((unsigned long long*)&ltab[K * d + kb], 1ull);
            else atomicAdd(&D.lcnt[kb], 1ull);
          }
        }
      }
    }
  }
  if (D.lloyd_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < K * d; i += blockDim.x)
      if (ltab[i]) atomicAdd((unsigned long long*)&D.lsum[i], (unsigned long long)ltab[i]);
    for (int i = threadIdx.x; i < K; i += blockDim.x)
      if (ltab[K * d + i]) atomicAdd(&D.lcnt[i], (unsigned long long)ltab[K * d + i]);
  }
}

// Lloyd centre update on the device (src/utils.cpp:56-61): mean of the members (2^30 fixed-point sums / counts), an
// empty cluster keeps its centre; refreshes Ycur [K][d], Yt [d][K], the MFMA image and ||y||^2.  One workgroup per cluster.
__global__ __launch_bounds__(64) void k_lloyd_finish(Dev D) {
  const int k = blockIdx.x, lane = threadIdx.x, d = D.d, K = D.K;
  const unsigned long long cnt = D.lcnt[k];
  float s2 = 0.0f;
  for (int j = lane; j < d; j += 64) {
    float y = D.Ycur[(size_t)k * d + j];
    if (cnt > 0) y = (float)(((double)D.lsum[(size_t)k * d + j] * (1.0 / 1073741824.0)) / (double)cnt);
    D.Ycur[(size_t)k * d + j] = y;
    D.Yt[(size_t)j * K + k] = y;
    D.Yimg[yimg_index(D, j, k)] = y;
    bfimg_store(D.Yimg3, D.NCT, D.NS2, j, k, y);
    s2 += y * y;
  }
  // ||y_k||^2 as the host computes it: sequential fp32 sum over j (d <= 128: lane-serial is fine for K blocks)
  __shared__ float ys[128];
  for (int j = lane; j < d; j += 64) ys[j] = D.Ycur[(size_t)k * d + j];
  __syncthreads();
  if (lane == 0) { float s = 0.0f; for (int j = 0; j < d; j++) s += ys[j] * ys[j]; D.ynorm[k] = s; }
  (void)s2;
}

// Ycur [K][d] (or, rows != nullptr, the fp64 rows gathered from the seed cells) -> Ycur, Yt [d][K], the fp32 MFMA image, the split-bf16 image and
// ||y_k||^2 -- on the device: the centroids never travel to the host and back between the stages of kmeans_centers / init_cluster_cpp
// (round 3: four synchronous uploads per stage).  normalise: Y[:,k] <- Y[:,k] / ||Y[:,k]|| first (arma::normalise(Y, 2, 0),
// src/harmony.cpp:136: sequential fp32 sum of squares, a zero column is left alone).  One workgroup per cluster.
__global__ __launch_bounds__(64) void k_y_images(Dev D, const double* __restrict__ rows, int normalise) {
  const int k = blockIdx.x, lane = threadIdx.x, d = D.d, K = D.K;
  __shared__ float ys[128];
  __shared__ float nrm_;
  for (int j = lane; j < d; j += 64) ys[j] = rows ? (float)rows[(size_t)k * d + j] : D.Ycur[(size_t)k * d + j];
  __syncthreads();
  if (lane == 0) {
    float s = 0.0f;
    for (int j = 0; j < d; j++) s = __fadd_rn(s, __fmul_rn(ys[j], ys[j]));
    float nn = sqrtf(s);
    if (nn == 0.0f) nn = 1.0f;
    nrm_ = normalise ? nn : 1.0f;
  }
  __syncthreads();
  for (int j = lane; j < d; j += 64) {
    const float y = normalise ? ys[j] / nrm_ : ys[j];
    ys[j] = y;
    D.Ycur[(size_t)k * d + j] = y;
    D.Yt[(size_t)j * K + k] = y;
    D.Yimg[yimg_index(D, j, k)] = y;
    bfimg_store(D.Yimg3, D.NCT, D.NS2, j, k, y);
  }
  __syncthreads();
  if (lane == 0) { float s = 0.0f; for (int j = 0; j < d; j++) s = __fadd_rn(s, __fmul_rn(ys[j], ys[j])); D.ynorm[k] = s; }
}
void l_y_images(const Launch& L, const Dev& D, const double* rows, int normalise) {
  hipLaunchKernelGGL(k_y_images, dim3(D.K), dim3(64), 0, L.stream, D, rows, normalise);
}
#endif  // !HMX_TILE_BF
// --------------------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------------------
#if HMX_TILE_BF
#define HMX_LNAME(x) x##_bf
#else
#define HMX_LNAME(x) x
size_t lds_bytes_y(const Dev& D) { return (size_t)D.d * D.KP * sizeof(float); }
#endif
// bytes of the centroid image the tile kernels of THIS translation unit stage in LDS, and of the split-bf16 one
static inline size_t bf_image_bytes(const Dev& D) { return (size_t)D.NCT * D.NS2 * 3 * 1024; }
static inline size_t tile_image_bytes(const Dev& D) { return HMX_TILE_BF ? bf_image_bytes(D) : (size_t)D.NQ * D.NS * 64 * sizeof(f32x4); }
constexpr size_t LDS_PER_CU = 160 * 1024;
// the split-bf16 build of a launch is taken when the workgroups that are to share a CU still fit its LDS with the larger image
static inline bool bf_fits(const Dev& D, size_t rest, long long blocks) {
  const long long per_cu = (blocks + 255) / 256;
  return D.dot_bf && D.Yimg3 && (bf_image_bytes(D) + rest) * (size_t)(per_cu < 1 ? 1 : per_cu) <= LDS_PER_CU;
}

#if !HMX_TILE_BF
#define HMX_DISPATCH_KD(KERNEL, EXTRA, GRID, LDS, ...)                                         \
  do {                                                                                         \
    const int kpl_ = D.KP / 64, dpl_ = D.d > 64 ? 2 : 1;                                       \
    if (dpl_ == 1) {                                                                           \
      switch (kpl_) {                                                                          \
        case 1: hipLaunchKernelGGL((KERNEL<1, 1 EXTRA>), GRID, dim3(TPB), LDS, L.stream, __VA_ARGS__); break; \
        case 2: hipLaunchKernelGGL((KERNEL<2, 1 EXTRA>), GRID, dim3(TPB), LDS, L.stream, __VA_ARGS__); break; \
        case 3: hipLaunchKernelGGL((KERNEL<3, 1 EXTRA>), GRID, dim3(TPB), LDS, L.stream, __VA_ARGS__); break; \
        default: hipLaunchKernelGGL((KERNEL<4, 1 EXTRA>), GRID, dim3(TPB), LDS, L.stream, __VA_ARGS__); break; \
      }                                                                                        \
    } else {                                                                                   \
      switch (kpl_) {                                                                          \
        case 1: hipLaunchKernelGGL((KERNEL<1, 2 EXTRA>), GRID, dim3(TPB), LDS, L.stream, __VA_ARGS__); break; \
        case 2: hipLaunchKernelGGL((KERNEL<2, 2 EXTRA>), GRID, dim3(TPB), LDS, L.stream, __VA_ARGS__); break; \
        case 3: hipLaunchKernelGGL((KERNEL<3, 2 EXTRA>), GRID, dim3(TPB), LDS, L.stream, __VA_ARGS__); break; \
        default: hipLaunchKernelGGL((KERNEL<4, 2 EXTRA>), GRID, dim3(TPB), LDS, L.stream, __VA_ARGS__); break; \
      }                                                                                        \
    }                                                                                          \
  } while (0)
#define HMX_COMMA ,

static int stream_grid(const Launch& L, long long work_waves) {
  long long blocks = (work_waves + 3) / 4;
  if (blocks < 1) blocks = 1;
  if (blocks > L.grid) blocks = L.grid;
  return (int)blocks;
}

void l_convert_in(const Launch& L, const void* src, int f32, float* dst, const int* invperm, int n, int d, int zs) {
  if (f32) hipLaunchKernelGGL(k_convert_in<float>, dim3(2048), dim3(256), 0, L.stream, (const float*)src, dst, invperm, n, d, zs);
  else hipLaunchKernelGGL(k_convert_in<double>, dim3(2048), dim3(256), 0, L.stream, (const double*)src, dst, invperm, n, d, zs);
}
void l_convert_out(const Launch& L, const float* src, void* dst, int f32, const int* invperm, int n, int w, int ws) {
  if (f32) hipLaunchKernelGGL(k_convert_out<float>, dim3(2048), dim3(256), 0, L.stream, src, (float*)dst, invperm, n, w, ws);
  else hipLaunchKernelGGL(k_convert_out<double>, dim3(2048), dim3(256), 0, L.stream, src, (double*)dst, invperm, n, w, ws);
}
void l_copy(const Launch& L, const float* src, float* dst, size_t count) {
  hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, L.stream, src, dst, count);
}
void l_normalize(const Launch& L, float* Z, int n, int d, int zs) {
  const int nq = zs / 4;
  if (nq <= 16) hipLaunchKernelGGL(k_normalize4<1>, dim3(stream_grid(L, (n + 15) / 16)), dim3(TPB), 0, L.stream, Z, Z, n, nq);
  else if (nq <= 32) hipLaunchKernelGGL(k_normalize4<2>, dim3(stream_grid(L, (n + 15) / 16)), dim3(TPB), 0, L.stream, Z, Z, n, nq);
  else hipLaunchKernelGGL(k_normalize, dim3(stream_grid(L, n)), dim3(TPB), 0, L.stream, Z, n, d, zs);
}
// dst = normalise(src) in one pass (restart: Z_corr = normalise(Z_orig), src/harmony.cpp:42)
void l_normalize_from(const Launch& L, const float* src, float* dst, int n, int d, int zs) {
  const int nq = zs / 4;
  if (nq <= 16) hipLaunchKernelGGL(k_normalize4<1>, dim3(stream_grid(L, (n + 15) / 16)), dim3(TPB), 0, L.stream, src, dst, n, nq);
  else if (nq <= 32) hipLaunchKernelGGL(k_normalize4<2>, dim3(stream_grid(L, (n + 15) / 16)), dim3(TPB), 0, L.stream, src, dst, n, nq);
  else { l_copy(L, src, dst, (size_t)n * zs); hipLaunchKernelGGL(k_normalize, dim3(stream_grid(L, n)), dim3(TPB), 0, L.stream, dst, n, d, zs); }
}
#endif  // !HMX_TILE_BF
// MFMA tile passes over the static 16-cell tiles: mode 1 = head, mode 2 = Lloyd, mode 3 = seeding race
void HMX_LNAME(l_tile_static)(const Launch& L, const Dev& D, int mode) {
  const int wpb = tile_threads(D.NCT) / 64;
  long long blocks = (((long long)D.ntitems + D.upd_tpw - 1) / D.upd_tpw + wpb - 1) / wpb;
  if (blocks > D.nwmax / wpb) blocks = D.nwmax / wpb;
  if (D.static_maxblocks > 0 && blocks > D.static_maxblocks) blocks = D.static_maxblocks;
  const size_t rest = mode == 2 ? ((size_t)D.K * D.d + D.K) * sizeof(long long) : 0;
  if (mode == 2 && blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  int thr = tile_threads(D.NCT);
#if !HMX_TILE_BF
  if (bf_fits(D, rest, blocks) || (mode == 2 && bf_fits(D, rest, 256))) { l_tile_static_bf(L, D, mode); return; }
#else
  // Lloyd with the larger image: two 256-thread workgroups per CU no longer fit next to their K x d sum tables -> one of 512 threads
  if (mode == 2 && !bf_fits(D, rest, blocks)) { thr = 512; blocks = (blocks + 1) / 2; if (blocks > 256) blocks = 256; }
#endif
  const size_t lds = tile_image_bytes(D) + rest;
  const dim3 grid((unsigned)blocks);
  // head / Lloyd.  Head variants: general sigma | uniform sigma (D.usig) | uniform sigma at 4 waves per SIMD (K <= 64)
#define HMX_TS(N) case N: if (mode == 3) hipLaunchKernelGGL((k_tile<N, 3>), grid, dim3(thr), lds, L.stream, D, 0); \
                          else if (mode == 2) hipLaunchKernelGGL((k_tile<N, 2>), grid, dim3(thr), lds, L.stream, D, 0); \
                          else if (D.usig) hipLaunchKernelGGL((k_tile<N, 1, 2, true>), grid, dim3(thr), lds, L.stream, D, 0); \
                          else hipLaunchKernelGGL((k_tile<N, 1>), grid, dim3(thr), lds, L.stream, D, 0); break;
#define HMX_TSL(N) case N: hipLaunchKernelGGL((k_tile<N, 1, 4, true>), grid, dim3(thr), lds, L.stream, D, 0); break;
  if (mode == 1 && D.upd_wps == 4) {
    switch (D.NCT) {
      HMX_TSL(1) HMX_TSL(2) HMX_TSL(3) HMX_TSL(4)
      default: break;
    }
    return;
  }
  switch (D.NCT) {
    HMX_TS(1) HMX_TS(2) HMX_TS(3) HMX_TS(4) HMX_TS(5) HMX_TS(6) HMX_TS(7) HMX_TS(8)
    HMX_TS(10) HMX_TS(12) HMX_TS(13) HMX_TS(14) HMX_TS(16)
    default: break;
  }
#undef HMX_TS
#undef HMX_TSL
}
#if !HMX_TILE_BF
void l_head(const Launch& L, const Dev& D, int mode) {
  const dim3 grid(stream_grid(L, D.nitems));
  const size_t lds = lds_bytes_y(D);
  if (mode == 0) HMX_DISPATCH_KD(k_head, HMX_COMMA 0, grid, lds, D);
  else HMX_DISPATCH_KD(k_head, HMX_COMMA 1, grid, lds, D);
}
// fused = true: D.blk is produced by the histogram kernel from (seed, round); false: the host uploaded D.blk (injected shuffle)
static SortPtrs sort_ptrs_of(const Dev& D) { return SortPtrs{D.blk, D.blkv, D.counts, D.offs, D.binoff, D.bincnt, D.boff, D.lorder, D.lcombo, D.lpair}; }
static BlockIdArgs block_id_args(uint64_t seed, uint64_t round, uint64_t Nglob, uint64_t goff, uint64_t cells_per_block) {
  BlockIdArgs A;
  A.fk = make_keys(seed, round, Nglob); A.fk2 = make_keys(seed, round + 1, Nglob); A.Nglob = Nglob; A.goff = goff; A.cpb = cells_per_block;
  A.inv_cpb = 1.0f / (float)cells_per_block;
  return A;
}
// the histogram half (block ids from the Feistel bijection + per-chunk counts): depends on nothing but (seed, round)
void l_sort_hist(const Launch& L, const Dev& D, bool fused, uint64_t seed, uint64_t round, uint64_t Nglob, uint64_t goff, uint64_t cells_per_block) {
  const int nV = D.nxt ? D.nb * D.nb : D.nb;
  const size_t lds = (size_t)nV * sizeof(int);
  BlockIdBatch AB{}; AB.a[0] = block_id_args(seed, round, Nglob, goff, cells_per_block);
  SortBatch S{}; S.p[0] = sort_ptrs_of(D);
  if (fused) hipLaunchKernelGGL(k_sort_hist<true>, dim3(D.nchunks), dim3(WAVE), lds, L.stream, D, AB, S);
  else hipLaunchKernelGGL(k_sort_hist<false>, dim3(D.nchunks), dim3(WAVE), lds, L.stream, D, AB, S);
}
// the dependent half: bin offsets from the counts, then the placement (padding slots = -1: written by k_sort_scatter, bin by bin)
void l_sort_tail(const Launch& L, const Dev& D) {
  const int nV = D.nxt ? D.nb * D.nb : D.nb;
  const size_t lds = (size_t)nV * sizeof(int);
  SortBatch S{}; S.p[0] = sort_ptrs_of(D);
  hipLaunchKernelGGL(k_sort_binscan, dim3(nV * D.Q), dim3(WAVE), 0, L.stream, D, S);
  hipLaunchKernelGGL(k_sort_binoff, dim3(1), dim3(1024), 0, L.stream, D, S);
  hipLaunchKernelGGL(k_sort_scatter, dim3(D.nchunks), dim3(WAVE), lds, L.stream, D, S);
}
void l_sort_blocks(const Launch& L, const Dev& D, bool fused, uint64_t seed, uint64_t round, uint64_t Nglob, uint64_t goff,
                   uint64_t cells_per_block) {
  l_sort_hist(L, D, fused, seed, round, Nglob, goff, cells_per_block);
  l_sort_tail(L, D);
}
// The shuffles of `nr` consecutive rounds (first: `round`) in ONE set of four launches, blockIdx.y = the round: the four kernels of a
// sort are latency-bound chains of small dependent steps, so four rounds cost little more than one -- and inside a cluster_cpp call
// no sort is left between two block chains (round 3: 4 x ~70 us of sort tail, event hand-over and dispatch per call).
void l_sort_batch(const Launch& L, const Dev& D, const SortBatch& S, int nr, uint64_t seed, uint64_t round, uint64_t Nglob, uint64_t goff,
                  uint64_t cells_per_block) {
  const int nV = D.nxt ? D.nb * D.nb : D.nb;
  const size_t lds = (size_t)nV * sizeof(int);
  BlockIdBatch AB{};
  for (int r = 0; r < nr; r++) AB.a[r] = block_id_args(seed, round + (uint64_t)r, Nglob, goff, cells_per_block);
  hipLaunchKernelGGL(k_sort_hist<true>, dim3(D.nchunks, nr), dim3(WAVE), lds, L.stream, D, AB, S);
  hipLaunchKernelGGL(k_sort_binscan, dim3(nV * D.Q, nr), dim3(WAVE), 0, L.stream, D, S);
  hipLaunchKernelGGL(k_sort_binoff, dim3(nr), dim3(1024), 0, L.stream, D, S);
  hipLaunchKernelGGL(k_sort_scatter, dim3(D.nchunks, nr), dim3(WAVE), lds, L.stream, D, S);
}
// the padded orders of rounds round..round + nr - 1 from the inverse of the shuffle (k_shuf_*)
int shuffle_parts(uint64_t Nglob, int nb, uint64_t cells_per_block) {
  const uint64_t last = Nglob > (uint64_t)(nb - 1) * cells_per_block ? Nglob - (uint64_t)(nb - 1) * cells_per_block : 0;
  return (int)((std::max<uint64_t>(std::max(cells_per_block, last), 1) + SHUF_PART - 1) / SHUF_PART);
}
void l_shuffle_inv(const Launch& L, const Dev& D, const ShufSets& T, int nr, uint64_t seed, uint64_t round, uint64_t Nglob, uint64_t goff,
                   uint64_t cells_per_block) {
  ShufBatch S{};
  for (int r = 0; r < nr + 1; r++) S.fk[r] = make_keys(seed, round + (uint64_t)r, Nglob);
  for (int r = 0; r < nr; r++) {
    S.posr[r] = T.posr[r]; S.lpair[r] = T.lpair[r]; S.lorder[r] = T.lorder[r]; S.lcombo[r] = T.lcombo[r]; S.boff[r] = T.boff[r];
    S.partcnt[r] = T.partcnt[r]; S.binbase[r] = T.binbase[r]; S.bincnt[r] = T.bincnt[r]; S.binacc[r] = T.binacc[r];
  }
  S.Nglob = Nglob; S.goff = goff; S.cpb = cells_per_block; S.inv_cpb = 1.0f / (float)cells_per_block; S.nr = nr;
  S.P = shuffle_parts(Nglob, D.nb, cells_per_block);
  const int nbin = (D.nxt ? D.nb : 1) * D.Q;
  hipLaunchKernelGGL(k_shuf_count, dim3((unsigned)(S.P * D.nb), nr), dim3(SHUF_THREADS), ((size_t)nbin + D.Q + 1 + SHUF_PART) * sizeof(int) + SHUF_PART, L.stream, D, S);
  hipLaunchKernelGGL(k_shuf_scan, dim3(nr), dim3(1024), 0, L.stream, D, S);
  hipLaunchKernelGGL(k_shuf_place, dim3((unsigned)(S.P * D.nb), nr), dim3(1024), (size_t)nbin * sizeof(int), L.stream, D, S);
}
// D.blk of one round (the sort-free shuffle does not need it; the passes that sum a round's old contributions from R do)
void l_shuffle_blocks(const Launch& L, const Dev& D, uint64_t seed, uint64_t round, uint64_t Nglob, uint64_t goff, uint64_t cells_per_block) {
  hipLaunchKernelGGL(k_shuf_blocks, dim3((unsigned)((D.n + 255) / 256)), dim3(256), 0, L.stream, D, block_id_args(seed, round, Nglob, goff, cells_per_block));
}
// oe_arith: the round's shuffled order itself, posord[position] = internal cell id (arma::shuffle's update_order, src/harmony.cpp:272-273,
// for the documented generator: cell g sits at position feistel(seed, round, g))
__global__ void k_ref_posord(Dev D, FeistelKeys fk, uint64_t Nglob, int* __restrict__ posord, int* __restrict__ poslev) {
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < D.n; g += gridDim.x * blockDim.x) {
    const int cell = D.invperm[g];
    const size_t pos = (size_t)feistel_apply(fk, Nglob, (uint64_t)g);
    posord[pos] = cell;
    if (poslev) {       // the position's level codes, [c][n]: the sequential-sum kernels then need no combo / qlev lookups
      const int q = D.combo[cell];
      for (int c = 0; c < D.C && c < 4; c++) poslev[(size_t)c * D.n + pos] = D.qlev[q * D.C + c];
    }
  }
}
void l_ref_posord(const Launch& L, const Dev& D, uint64_t seed, uint64_t round, uint64_t Nglob, int* posord, int* poslev) {
  hipLaunchKernelGGL(k_ref_posord, dim3(1024), dim3(256), 0, L.stream, D, make_keys(seed, round, Nglob), Nglob, posord, poslev);
}
void l_oldsum(const Launch& L, const Dev& D) {
  const size_t tab = (size_t)D.nb * D.K * sizeof(unsigned long long);
  if (D.oldsum_stream && tab <= 64 * 1024) {   // sequential pass over R with LDS accumulators
    const int wgs = tab <= 20 * 1024 ? 2048 : 512;   // 8 workgroups (32 waves) per CU while the LDS tables allow it
    const dim3 sg((unsigned)std::min(wgs, D.nchunks));
    if (D.K % 4 == 0 && D.oldsum_stream == 1) {
      const dim3 g4((unsigned)std::min(tab <= 20 * 1024 ? 512 : 256, D.nchunks));
      hipLaunchKernelGGL(k_oldsum_stream4, g4, dim3(1024), tab, L.stream, D);
      return;
    }
    switch (D.KP / 64) {
      case 1: hipLaunchKernelGGL(k_oldsum_stream<1>, sg, dim3(256), tab, L.stream, D); break;
      case 2: hipLaunchKernelGGL(k_oldsum_stream<2>, sg, dim3(256), tab, L.stream, D); break;
      case 3: hipLaunchKernelGGL(k_oldsum_stream<3>, sg, dim3(256), tab, L.stream, D); break;
      default: hipLaunchKernelGGL(k_oldsum_stream<4>, sg, dim3(256), tab, L.stream, D); break;
    }
    return;
  }
  const dim3 grid(stream_grid(L, (D.n + 63) / 64));
  switch (D.KP / 64) {
    case 1: hipLaunchKernelGGL(k_oldsum<1>, grid, dim3(TPB), 0, L.stream, D); break;
    case 2: hipLaunchKernelGGL(k_oldsum<2>, grid, dim3(TPB), 0, L.stream, D); break;
    case 3: hipLaunchKernelGGL(k_oldsum<3>, grid, dim3(TPB), 0, L.stream, D); break;
    default: hipLaunchKernelGGL(k_oldsum<4>, grid, dim3(TPB), 0, L.stream, D); break;
  }
}
void l_fold(const Launch& L, const Dev& D, int j, int mode) {
  const int n = D.B * D.K;
  hipLaunchKernelGGL(k_fold, dim3((n + 255) / 256), dim3(256), 0, L.stream, D, j, mode);
}
void l_foldpen(const Launch& L, const Dev& D, int j, const long long* Oin, long long* Oout, const long long* Sin,
               long long* Szero) {
  hipLaunchKernelGGL(k_foldpen, dim3((D.K + 15) / 16), dim3(256), (size_t)D.B * 16 * sizeof(long long), L.stream, D, j, Oin,
                     Oout, Sin, Szero);
}
void l_penalty(const Launch& L, const Dev& D) {
  const int n = D.B * D.K;
  hipLaunchKernelGGL(k_penalty, dim3((n + 255) / 256), dim3(256), 0, L.stream, D);
}
// One GPU: the three kernels that close a clustering round (slot rows -> obj[0..1] -> cross-entropy term, snapshot, chain control
// reset) as ONE launch: every workgroup reduces its slot row, the last one to finish (ticket) does the rest.  Same fixed-order sums.
__global__ __launch_bounds__(1024) void k_round_tail(Dev D, double* __restrict__ host_slot, long long* __restrict__ z0, size_t n0,
                                                      long long* __restrict__ z1, size_t n1) {
  __shared__ double ra[1024], rb[1024];
  __shared__ int last;
  const int tid = threadIdx.x;
  {  // the old-contribution table this round consumed and the replica sets start the next rounds from zero: cleared here instead
     // of by memset launches
    for (size_t i = (size_t)blockIdx.x * 1024 + tid; i < n0; i += (size_t)gridDim.x * 1024) z0[i] = 0;
    for (size_t i = (size_t)blockIdx.x * 1024 + tid; i < n1; i += (size_t)gridDim.x * 1024) z1[i] = 0;
  }
  double a = 0.0, b = 0.0;
  double* row = D.objpart + (size_t)blockIdx.x * D.nwmax * 2;
  for (int i = tid; i < D.nwmax; i += 1024) { a += row[2 * i]; b += row[2 * i + 1]; row[2 * i] = 0.0; row[2 * i + 1] = 0.0; }
  ra[tid] = a; rb[tid] = b;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (tid < off) { ra[tid] += ra[tid + off]; rb[tid] += rb[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) {
    // No fences: an agent-scope release here would write back everything the round left dirty in this XCD's L2 (tens of MB of R
    // rows).  The two sums go out as write-through device-scope stores, the ticket follows once they are acknowledged, and the
    // last workgroup reads them with device-scope loads (the scheme of the block chain).
    __hip_atomic_store(&D.objrow[2 * blockIdx.x], ra[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&D.objrow[2 * blockIdx.x + 1], rb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = (atomicAdd(D.tail_ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!last) return;
  const int K = D.K, B = D.B;
  double cross = 0.0;
  for (int k = tid; k < K; k += 1024) {      // (same arithmetic as k_objective_tables)
    long long rs = 0;
    for (int b0 = 0; b0 < D.B0; b0++) rs += D.O_fx[(size_t)b0 * K + k];
    const double rsd = (double)rs * FX_INV;
    double ck = 0.0;
    for (int bb = 0; bb < B; bb++) {
      const double od = (double)D.O_fx[(size_t)bb * K + k] * FX_INV;
      const float o = (float)od, e = (float)(rsd * (double)D.Pr_b[bb]);
      const float m = D.theta[bb] * logf((o + e + 1.0f) / ((2.0f * e) + 1.0f));
      ck += od * (double)m;
    }
    cross += ck * (double)D.sigma[k];
  }
  __syncthreads();
  ra[tid] = cross;
  __syncthreads();
  // (K <= 256: every thread holds at most one cluster, the entries from 256 on are zero -- the same pairwise tree as
  //  k_objective_tables' 256-entry one, preceded by two levels that add zeros: identical bits)
  for (int off = 512; off > 0; off >>= 1) {
    if (tid < off) ra[tid] += ra[tid + off];
    __syncthreads();
  }
  // the slot rows' sums: fetched by 2 x objslots threads at once (objslots <= 64), added in slot order by one -- the 40 device-scope
  // loads used to be ONE thread's dependent chain, ~20 us of every round
  if (tid < 2 * D.objslots) rb[tid] = __hip_atomic_load(&D.objrow[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (tid == 0) {
    double sa = 0.0, sb = 0.0;
    for (int sl = 0; sl < D.objslots; sl++) { sa += rb[2 * sl]; sb += rb[2 * sl + 1]; }
    D.obj[0] = sa; D.obj[1] = sb;
    D.obj[2] = sa; D.obj[3] = sb; D.obj[4] = ra[0];
    // error word of the snapshot: the chain's code (< 16) + 16 if a ridge system of the correction before this round was singular
    const double err = (D.chain_ctl ? (double)D.chain_ctl[1] : 0.0) + ((D.solve_err && *D.solve_err) ? 16.0 : 0.0);
    D.obj[5] = err;
    if (host_slot) {     // pinned host memory, mapped into the device: visible to the host once the event behind this launch completed
      __hip_atomic_store(&host_slot[0], sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&host_slot[1], sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&host_slot[2], ra[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&host_slot[3], err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *D.tail_ticket = 0;
  }
  __syncthreads();
  if (D.chain_ctl) for (int i = tid; i < 8 * D.nb + 24; i += 1024) D.chain_ctl[i] = 0;
}
void l_round_tail(const Launch& L, const Dev& D, double* host_slot, long long* z0, size_t n0, long long* z1, size_t n1) {
  hipLaunchKernelGGL(k_round_tail, dim3(D.objslots), dim3(1024), 0, L.stream, D, host_slot, z0, n0, z1, n1);
}
void l_obj_reduce(const Launch& L, const Dev& D) {
  hipLaunchKernelGGL(k_obj_reduce, dim3(D.objslots), dim3(1024), 0, L.stream, D);
  hipLaunchKernelGGL(k_obj_final, dim3(1), dim3(1), 0, L.stream, D);
}
#endif  // !HMX_TILE_BF
// launch with the start / stop events of profile mode attached to the dispatch (no barrier packets around the launch), or plainly
#define HMX_LAUNCH_EV(KERNEL, GRID, BLOCK, LDS, ...)                                                                      \
  do {                                                                                                                     \
    if (L.ev0) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, (std::uint32_t)(LDS), L.stream, L.ev0, L.ev1, 0, __VA_ARGS__);   \
    else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, L.stream, __VA_ARGS__);                                              \
  } while (0)
void HMX_LNAME(l_update)(const Launch& L, const Dev& D, int j) {
#if !HMX_TILE_BF
  if (D.upd_impl == 1) {
    // a block holds ~n/nb cells; D.upd_cpw cells per wave (tunable: HMX_UPD_CPW)
    const long long waves = ((long long)D.n / (D.nb > 0 ? D.nb : 1) + D.upd_cpw - 1) / D.upd_cpw + 1;
    const dim3 grid(stream_grid(L, waves));
    if (L.ev0) (void)hipEventRecord(L.ev0, L.stream);       // (profile mode: the first-generation kernel is launched plainly, the pair is recorded around it)
    HMX_DISPATCH_KD(k_update, , grid, lds_bytes_y(D), D, j);
    if (L.ev1) (void)hipEventRecord(L.ev1, L.stream);
    return;
  }
#endif
  const long long tiles = ((long long)D.n / (D.nb > 0 ? D.nb : 1) + 15) / 16 + (long long)D.Q + 1;
  const int wpb = D.upd_threads / 64;
  long long blocks = ((tiles + D.upd_tpw - 1) / D.upd_tpw + wpb - 1) / wpb;
  if (blocks > D.upd_maxblocks) blocks = D.upd_maxblocks;   // resident capacity (workgroups per CU x CUs)
  if (blocks > D.nwmax / wpb) blocks = D.nwmax / wpb;
  if (blocks < 1) blocks = 1;
  const dim3 grid((unsigned)blocks);
  const size_t rest = (D.fused_fold ? (size_t)D.B * D.K * 8 : 0) +
                      ((D.pen_lds || D.fused_fold) ? ((size_t)((D.B * D.K + 3) & ~3) + (size_t)D.Q * D.C) * 4 : 0);
#if !HMX_TILE_BF
  if (bf_fits(D, rest, blocks)) { l_update_bf(L, D, j); return; }
#endif
  const size_t lds = tile_image_bytes(D) + rest;
#define HMX_UPDL(N) case N: HMX_LAUNCH_EV((k_tile<N, 0, 4, true>), grid, dim3(1024), lds, D, j); break;
  if (D.upd_wps == 4) {   // hmx_setup: upd_threads == 1024, uniform sigma, K <= 64
    switch (D.NCT) {
      HMX_UPDL(1) HMX_UPDL(2) HMX_UPDL(3) HMX_UPDL(4)
      default: break;
    }
    return;
  }
#undef HMX_UPDL
#define HMX_UPD(N) case N: if (D.usig) HMX_LAUNCH_EV((k_tile<N, 0, 2, true>), grid, dim3(D.upd_threads), lds, D, j); \
                           else HMX_LAUNCH_EV((k_tile<N, 0>), grid, dim3(D.upd_threads), lds, D, j); break;
  switch (D.NCT) {
    HMX_UPD(1) HMX_UPD(2) HMX_UPD(3) HMX_UPD(4) HMX_UPD(5) HMX_UPD(6) HMX_UPD(7) HMX_UPD(8)
    HMX_UPD(10) HMX_UPD(12) HMX_UPD(13) HMX_UPD(14) HMX_UPD(16)
    default: break;
  }
#undef HMX_UPD
}
void HMX_LNAME(l_chain)(const Launch& L, const Dev& D, int workgroups) {
  const size_t rest = (size_t)D.B * D.K * 8 + ((size_t)((D.B * D.K + 3) & ~3) + (size_t)D.Q * D.C) * 4;
#if !HMX_TILE_BF
  if (bf_image_bytes(D) + rest + 64 <= 150 * 1024 && bf_fits(D, rest, 1)) { l_chain_bf(L, D, workgroups); return; }
#endif
  const size_t lds = tile_image_bytes(D) + rest;
  const dim3 grid((unsigned)workgroups);
  // (two waves per SIMD, two accumulator sets; MODE 5 = the variant without R stores, split-bf16 build only)
#if HMX_TILE_BF
#define HMX_CH(N) case N: if (D.usig && !D.r_store) HMX_LAUNCH_EV((k_tile<N, 5, 2, true>), grid, dim3(512), lds, D, 0); \
                          else if (D.usig) HMX_LAUNCH_EV((k_tile<N, 4, 2, true>), grid, dim3(512), lds, D, 0); \
                          else HMX_LAUNCH_EV((k_tile<N, 4>), grid, dim3(512), lds, D, 0); break;
#else
#define HMX_CH(N) case N: if (D.usig) HMX_LAUNCH_EV((k_tile<N, 4, 2, true>), grid, dim3(512), lds, D, 0); \
                          else HMX_LAUNCH_EV((k_tile<N, 4>), grid, dim3(512), lds, D, 0); break;
#endif
  switch (D.NCT) {
    HMX_CH(1) HMX_CH(2) HMX_CH(3) HMX_CH(4) HMX_CH(5) HMX_CH(6) HMX_CH(7)
    default: break;
  }
#undef HMX_CH
}
#if !HMX_TILE_BF
// Self-test of the peer-to-peer inboxes, run by every rank at the same time before the chain may use them: P2P_TEST_STEPS exchanges of
// a 2048-entry table with known contents through exactly the chain's code path (p2p_send, the same slots, parities and polls),
// every received value checked.  result[0] = wrong or missing values (0 = pass), result[1] = 100 MHz ticks of the steps after the
// first (the first absorbs the launch skew between the ranks; bounded at ~3 s).
constexpr int P2P_TEST_STEPS = 64;
__device__ __forceinline__ long long p2p_test_value(int rank, int step, int i) {
  const long long v = (long long)(rank + 1) * 0x100000001ll * (long long)(i + 1) + (long long)step * 7919;
  return ((i + step) & 1) ? -v : v;
}
__global__ void __launch_bounds__(512) k_p2p_selftest(Dev D, unsigned tag, int* result) {
  const int tid = threadIdx.x, G = D.p2p_world, me = D.p2p_rank;
  __shared__ int gave_up, bad;
  if (tid == 0) { gave_up = 0; bad = 0; }
  __syncthreads();
  int wrong = 0;
  unsigned long long t1 = 0;
  for (int step = 0; step < P2P_TEST_STEPS; step++) {
    if (step == 1) t1 = wall_clock64();
    const unsigned tagx = tag + (unsigned)step;
    const size_t par = (size_t)(step & 1) * 8;      // (the chain's two planes)
#pragma unroll
    for (int e = 0; e < 4; e++) p2p_send(D, par, tid + e * 512, tagx, p2p_test_value(me, step, tid + e * 512));
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int i = tid + e * 512;
#pragma unroll
      for (int gq = 0; gq < 8; gq++) if (gq < G && gq != me) {
        const unsigned long long* src = D.p2p_inbox_self() + ((par + gq) * P2P_CAP + i) * 2;
        unsigned long long lo = 0, hi = 0;
        bool got = false;
        for (int spins = 0; spins < (1 << 20); spins++) {
          lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if ((unsigned)(lo >> 32) == tagx && (unsigned)(hi >> 32) == tagx) { got = true; break; }
          if ((spins & 63) == 63 && __hip_atomic_load(&gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
          if (step == 0) __builtin_amdgcn_s_sleep(100); else __builtin_amdgcn_s_sleep(2);
        }
        if (!got) { __hip_atomic_store(&gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); wrong++; }
        else if ((long long)((hi << 32) | (lo & 0xffffffffull)) != p2p_test_value(gq, step, i)) wrong++;
      }
    }
  }
  const unsigned long long t2 = wall_clock64();
  if (wrong) atomicAdd(&bad, wrong);
  __syncthreads();
  if (tid == 0) { result[0] = bad; result[1] = (int)(t2 - t1); }
}
void l_p2p_selftest(const Launch& L, const Dev& D, unsigned tag, int* result) {
  hipLaunchKernelGGL(k_p2p_selftest, dim3(1), dim3(512), 0, L.stream, D, tag, result);
}
// Generic all-reduce of a small buffer through the peers' inboxes (planes 2 / 3): the collectives of a run that are NOT block steps --
// O after a head, the objective's two sums, the Lloyd sums and counts, the seeding minima, small ridge statistics -- are a few KB each
// and latency-bound: as host-launched ncclAllReduce calls they cost a launch + a ring each (~50 per run).  Here: one workgroup, every
// rank writes its values straight into every peer's inbox (self-validating {tag, half} granules, as the chain does) and adds up what
// arrived in its own, in RANK ORDER (fp64 sums are then identical on every rank).  A rank can be at most one call ahead of a peer (it
// needs the peer's values of call n to finish call n), so two planes alternate.  Every spin is bounded; a timeout raises *err (the
// chain's error word: it reaches the host with the next objective snapshot).
__global__ void __launch_bounds__(1024) k_p2p_allreduce(Dev D, unsigned long long* __restrict__ buf, int n, int dtype, unsigned seq, int* err) {
  const int tid = threadIdx.x, G = D.p2p_world, me = D.p2p_rank;
  const unsigned tag = 0x40000000u + (seq & 0x3fffffffu);
  const size_t par = (size_t)(2 + (seq & 1u)) * 8;
  for (int base = 0; base < n; base += 1024 * 4) {
    unsigned long long mine[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int i = base + tid + e * 1024;
      mine[e] = (i < n) ? buf[i] : 0ull;
      if (i < n) p2p_send(D, par, i, tag, (long long)mine[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int i = base + tid + e * 1024;
      if (i >= n) continue;
      unsigned long long val[8];
#pragma unroll
      for (int gq = 0; gq < 8; gq++) {
        val[gq] = mine[e];
        if (gq < G && gq != me) {
          const unsigned long long* src = D.p2p_inbox_self() + ((par + gq) * P2P_CAP + i) * 2;
          unsigned long long lo = 0, hi = 0;
          bool got = false;
          for (int spins = 0; spins < (1 << 20); spins++) {
            lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) { got = true; break; }
            if ((spins & 255) == 255 && err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            __builtin_amdgcn_s_sleep(2);
          }
          if (!got && err) atomicExch(err, 7);
          val[gq] = (hi << 32) | (lo & 0xffffffffull);
        }
      }
      unsigned long long out;
      if (dtype == 1) { double a = 0.0; for (int gq = 0; gq < G; gq++) a += __longlong_as_double((long long)val[gq]); out = (unsigned long long)__double_as_longlong(a); }
      else if (dtype == 2) { long long a = (long long)val[0]; for (int gq = 1; gq < G; gq++) a = min(a, (long long)val[gq]); out = (unsigned long long)a; }
      else { long long a = 0; for (int gq = 0; gq < G; gq++) a += (long long)val[gq]; out = (unsigned long long)a; }
      buf[i] = out;
    }
  }
}
void l_p2p_allreduce(const Launch& L, const Dev& D, void* buf, int n, int dtype, unsigned seq, int* err) {
  hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(1024), 0, L.stream, D, (unsigned long long*)buf, n, dtype, seq, err);
}
// LARGE buffers (the ridge statistics of many-level designs: Q K (d + 1) doubles, 10 MB at BASELINE configs[4]) through the same inboxes as
// reduce-scatter + all-gather: the one-shot form above would push every rank's WHOLE buffer over each of its links; here entry e of a
// window belongs to rank e / S (S = P2P_CAP / 2 entries per rank and window): (1) every rank sends its value of e to the owner only, (2) the owner
// adds the G values in RANK ORDER (fp64 sums identical on every rank) and sends the result to everybody, (3) the others pick it up -- 2 (G - 1) / G
// of the buffer per link instead of (G - 1) times it, many workgroups wide.  Inbox layout per (plane, source): entries [0, S) carry the
// scattered values, [S, 2 S) the gathered results; planes and tags as k_p2p_allreduce (one call = one window = one `seq`).  Three separate
// sweeps, so no thread waits while a peer still needs one of its sends; every spin is bounded (err = 7).
__device__ __forceinline__ void p2p_send_to(const Dev& D, int peer, size_t par, int i, unsigned tag, unsigned long long v) {
  const unsigned long long tb = (unsigned long long)tag << 32;
  const unsigned long long lo = tb | (v & 0xffffffffull), hi = tb | (v >> 32);
#pragma unroll
  for (int gq = 0; gq < 8; gq++) if (gq == peer) {       // (static indices only: a dynamic one would spill the kernarg copy)
    unsigned long long* dst = D.p2p_inbox[gq] + ((par + D.p2p_rank) * P2P_CAP + i) * 2;
    __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__device__ __forceinline__ unsigned long long p2p_wait(const Dev& D, size_t par, int src_rank, int i, unsigned tag, int* err) {
  const unsigned long long* src = D.p2p_inbox_self() + ((par + src_rank) * P2P_CAP + i) * 2;
  unsigned long long lo = 0, hi = 0;
  bool got = false;
  for (int spins = 0; spins < (1 << 20); spins++) {
    lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) { got = true; break; }
    if ((spins & 255) == 255 && err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    __builtin_amdgcn_s_sleep(2);
  }
  if (!got && err) atomicExch(err, 7);
  return (hi << 32) | (lo & 0xffffffffull);
}
__global__ void __launch_bounds__(1024) k_p2p_allreduce_big(Dev D, unsigned long long* __restrict__ buf, int n, int dtype, unsigned seq, int* err) {
  const int G = D.p2p_world, me = D.p2p_rank;
  constexpr int S = P2P_CAP / 2;
  const unsigned tag = 0x40000000u + (seq & 0x3fffffffu);
  const size_t par = (size_t)(2 + (seq & 1u)) * 8;
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int e = t0; e < n; e += nt) {                       // (1) scatter: my value of every entry another rank owns
    const int owner = e / S;
    if (owner != me) p2p_send_to(D, owner, par, e - owner * S, tag, buf[e]);
  }
  for (int l = t0; l < S; l += nt) {                       // (2) my slice: reduce in rank order, gather out
    const int e = me * S + l;
    if (e >= n) break;
    unsigned long long val[8];
#pragma unroll
    for (int gq = 0; gq < 8; gq++) { val[gq] = buf[e]; if (gq < G && gq != me) val[gq] = p2p_wait(D, par, gq, l, tag, err); }
    unsigned long long out;
    if (dtype == 1) { double a = 0.0; for (int gq = 0; gq < G; gq++) a += __longlong_as_double((long long)val[gq]); out = (unsigned long long)__double_as_longlong(a); }
    else if (dtype == 2) { long long a = (long long)val[0]; for (int gq = 1; gq < G; gq++) a = min(a, (long long)val[gq]); out = (unsigned long long)a; }
    else { long long a = 0; for (int gq = 0; gq < G; gq++) a += (long long)val[gq]; out = (unsigned long long)a; }
    buf[e] = out;
#pragma unroll
    for (int gq = 0; gq < 8; gq++) if (gq < G && gq != me) p2p_send_to(D, gq, par, S + l, tag, out);
  }
  for (int e = t0; e < n; e += nt) {                       // (3) the other ranks' slices
    const int owner = e / S;
    if (owner != me) buf[e] = p2p_wait(D, par, owner, S + (e - owner * S), tag, err);
  }
}
void l_p2p_allreduce_big(const Launch& L, const Dev& D, void* buf, int n, int dtype, unsigned seq, int* err) {
  int blocks = (n + 4095) / 4096; if (blocks > 128) blocks = 128; if (blocks < 1) blocks = 1;        // (all resident at once: the sweeps wait on peers, never on each other)
  hipLaunchKernelGGL(k_p2p_allreduce_big, dim3(blocks), dim3(1024), 0, L.stream, D, (unsigned long long*)buf, n, dtype, seq, err);
}
void l_objective_tables(const Launch& L, const Dev& D) {
  hipLaunchKernelGGL(k_objective_tables, dim3(1), dim3(TPB), 0, L.stream, D);
}
// false: shape outside this kernel's envelope (the caller falls back to the cluster-lane version)
bool l_obj_terms_mfma(const Launch& L, const Dev& D, const float* M, float* T, long long stride) {
  const size_t lds = (size_t)D.NQ * D.NS * 64 * sizeof(f32x4);
  if (!D.tile_impl || D.K % 4 != 0 || lds > 64 * 1024 || !D.Yimg || D.obj_stale || D.NT4 > 4) return false;      // (rows of <= 64 + 3 PCs in registers)
  int blocks = ((D.n + 15) / 16 + 3) / 4; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
#define HMX_OT(N) case N: hipLaunchKernelGGL((k_obj_terms_mfma<N>), dim3(blocks), dim3(256), lds, L.stream, D, M, T, stride); break;
  switch (D.NCT) {
    HMX_OT(1) HMX_OT(2) HMX_OT(3) HMX_OT(4) HMX_OT(5) HMX_OT(6) HMX_OT(7) HMX_OT(8) HMX_OT(10) HMX_OT(12) HMX_OT(13) HMX_OT(14) HMX_OT(16)
    default: return false;
  }
#undef HMX_OT
  return true;
}
void l_moe_stats(const Launch& L, const Dev& D) {
  const int zch = (D.d + 31) / 32;
  int dp = (D.d + zch - 1) / zch;
  dp = (dp + 3) / 4 * 4;
  const dim3 grid(stream_grid(L, D.nitems), (D.K + 127) / 128, (D.d + dp - 1) / dp);
  switch (dp) {
    case 4: hipLaunchKernelGGL(k_moe_stats<4>, grid, dim3(TPB), 0, L.stream, D); break;
    case 8: hipLaunchKernelGGL(k_moe_stats<8>, grid, dim3(TPB), 0, L.stream, D); break;
    case 12: hipLaunchKernelGGL(k_moe_stats<12>, grid, dim3(TPB), 0, L.stream, D); break;
    case 16: hipLaunchKernelGGL(k_moe_stats<16>, grid, dim3(TPB), 0, L.stream, D); break;
    case 20: hipLaunchKernelGGL(k_moe_stats<20>, grid, dim3(TPB), 0, L.stream, D); break;
    case 24: hipLaunchKernelGGL(k_moe_stats<24>, grid, dim3(TPB), 0, L.stream, D); break;
    case 28: hipLaunchKernelGGL(k_moe_stats<28>, grid, dim3(TPB), 0, L.stream, D); break;
    default: hipLaunchKernelGGL(k_moe_stats<32>, grid, dim3(TPB), 0, L.stream, D); break;
  }
}
void l_moe_apply(const Launch& L, const Dev& D) {
  const int dpl = D.d > 64 ? 2 : 1;
  const size_t lds = (size_t)D.K * 64 * dpl * sizeof(float);
  int g = D.naitems < 4 * L.grid ? D.naitems : 4 * L.grid;
  if (g < 1) g = 1;
  const dim3 grid(g);
  HMX_DISPATCH_KD(k_moe_apply, , grid, lds, D);
}
void l_moe_solve(const Launch& L, const Dev& D, const SolveArgs& A0) {
  SolveArgs A = A0;
  const size_t M = (size_t)D.B + 1;
  const size_t ints = (((size_t)4 * D.B + 6 + D.C) & ~(size_t)1) * sizeof(int);
  const size_t panel = M * 16 * sizeof(double), ball = M * D.d * sizeof(double);
  size_t body = panel;
  if (ints + ball <= 150 * 1024) body = std::max(panel, ball);     // the d right-hand sides live in LDS during the substitution
  if (ints + body > 158 * 1024) body = 0;                           // (B + 1 > ~1200 levels: outside the device-solve envelope, see hmx_setup)
  A.lds_b_bytes = (body >= ball) ? ball : 0;
  A.lds_body_bytes = body;
  // coupling masks of the Schur complement: one bit per (row, eliminated level) -- only while they are small
  const size_t maskb = M * ((M + 63) / 64) * sizeof(unsigned long long);
  const size_t moff = (ints + body + 7) & ~(size_t)7;
  A.lds_mask_off = (D.C > 1 && body > 0 && moff + maskb <= 159 * 1024) ? moff : 0;
  const int threads = M > 48 ? 1024 : 256;
  hipLaunchKernelGGL(k_moe_solve, dim3(D.K), dim3(threads), A.lds_mask_off ? moff + maskb : ints + body, L.stream, D, A);
}
void l_moe_stats_mfma(const Launch& L, const Dev& D) {
  const int npt = (D.d + 15) / 16;
  if (D.st_dma) {   // 16-byte operand loads + deterministic slot reduction (K <= 128)
    const dim3 grid((unsigned)D.st_nwg), block(64 * ((D.d + 16) / 16));   // PC tiles incl. the ones column at index d
#define HMX_MSQ(N) case N: hipLaunchKernelGGL((k_moe_stats_q<N>), grid, block, 0, L.stream, D, D.st_cpw); break;
    switch (D.NCT) { HMX_MSQ(1) HMX_MSQ(2) HMX_MSQ(3) HMX_MSQ(4) HMX_MSQ(5) HMX_MSQ(6) HMX_MSQ(7) HMX_MSQ(8) default: break; }
#undef HMX_MSQ
    const size_t per = (size_t)D.K * D.d + D.K;
    hipLaunchKernelGGL(k_moe_stats_reduce, dim3((unsigned)D.Q, (unsigned)((per + 255) / 256)), dim3(256), 0, L.stream, D);
    return;
  }
  int tpw = (D.ntitems + 2 * 256 - 1) / (2 * 256);   // ~2 workgroups per CU
  if (tpw < 16) tpw = 16;
  const bool split = D.NCT > 8;                       // K > 128: two workgroups (cluster-tile halves) per tile range
  const dim3 grid(((D.ntitems + tpw - 1) / tpw) * (split ? 2 : 1)), block(64 * npt);
#define HMX_MS(N) case N: hipLaunchKernelGGL((k_moe_stats_mfma<N, 1>), grid, block, 0, L.stream, D, tpw, npt); break;
#define HMX_MS2(N) case N: hipLaunchKernelGGL((k_moe_stats_mfma<N, 2>), grid, block, 0, L.stream, D, tpw, npt); break;
  switch (D.NCT) {
    HMX_MS(1) HMX_MS(2) HMX_MS(3) HMX_MS(4) HMX_MS(5) HMX_MS(6) HMX_MS(7) HMX_MS(8)
    HMX_MS2(10) HMX_MS2(12) HMX_MS2(13) HMX_MS2(14) HMX_MS2(16)
    default: break;
  }
#undef HMX_MS
#undef HMX_MS2
}
void l_moe_apply_mfma(const Launch& L, const Dev& D) {
  int g = D.naitems < 4 * L.grid ? D.naitems : 4 * L.grid;
  if (g < 1) g = 1;
  const dim3 grid(g);
  const size_t lds = (size_t)D.wNQ * D.wNS * 64 * sizeof(f32x4);
  switch ((D.d + 15) / 16) {
    case 1: hipLaunchKernelGGL(k_moe_apply_mfma<1>, grid, dim3(256), lds, L.stream, D); break;
    case 2: hipLaunchKernelGGL(k_moe_apply_mfma<2>, grid, dim3(256), lds, L.stream, D); break;
    case 3: hipLaunchKernelGGL(k_moe_apply_mfma<3>, grid, dim3(256), lds, L.stream, D); break;
    default: hipLaunchKernelGGL(k_moe_apply_mfma<4>, grid, dim3(256), lds, L.stream, D); break;
  }
}
void l_seed_probe(const Launch& L, const Dev& D, uint64_t seed, uint64_t goff, const unsigned* excl, int nexcl) {
  const dim3 grid(stream_grid(L, D.nitems));
  HMX_DISPATCH_KD(k_seed_probe, , grid, lds_bytes_y(D), D, seed, goff, excl, nexcl);
}
void l_seed_race_u(const Launch& L, const Dev& D, const float* u, int a0, int na, int only, uint64_t goff, const unsigned* excl,
                   int nexcl) {
  int blocks = (D.n + 255) / 256; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
  const int lo = excl ? only : 0, hi = excl ? only + 1 : na;
  hipLaunchKernelGGL(k_seed_race_u, dim3(blocks), dim3(256), (size_t)na * D.d * sizeof(float), L.stream, D, u, a0, na, lo, hi, goff,
                     excl, nexcl);
}
void l_gather_rows(const Launch& L, const Dev& D, const long long* gcells, uint64_t goff, double* rows) {
  hipLaunchKernelGGL(k_gather_rows, dim3(D.K), dim3(64), 0, L.stream, D, gcells, goff, rows);
}
void l_lloyd_finish(const Launch& L, const Dev& D) {
  hipLaunchKernelGGL(k_lloyd_finish, dim3(D.K), dim3(64), 0, L.stream, D);
}
void l_lloyd(const Launch& L, const Dev& D) {
  int blocks = stream_grid(L, D.nitems);
  if (D.lloyd_lds && blocks > 512) blocks = 512;  // every workgroup flushes a K x d table: keep them few and fat
  const dim3 grid(blocks);
  const size_t lds = (((size_t)D.d * D.KP + 1) & ~(size_t)1) * sizeof(float) +
                     (D.lloyd_lds ? ((size_t)D.K * D.d + D.K) * sizeof(long long) : 0);
  HMX_DISPATCH_KD(k_lloyd, , grid, lds, D);
}

#endif  // !HMX_TILE_BF
}  // namespace hmx
