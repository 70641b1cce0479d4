// hmx_seq.hip -- the reference's SEQUENTIAL fp32 accumulators, reproduced on the GPU ("reference arithmetic").
//
// The reference sums in fp32, one term after the other, in a fixed order:
//   my_accu                       src/utils.cpp:67-75      objective terms, K*N terms per chain (cell-major, k fastest)
//   sum(R, 1), R * Phi_t          src/harmony.cpp:149-150  E / O of the head: per cluster over all cells / the cells of a level
//   sum(Rcells, 1), Rcells*Phi_t  src/harmony.cpp:312-313,329-330  per block, in the round's shuffled order
//   sum(Z_tmp, 1), sum(Z_tmp.cols(index[b]), 1), Phi_Rk * Phi_moe_t   src/harmony.cpp:567,599-608   ridge statistics
// Once such an accumulator has grown, terms below half an ulp of it are dropped: a systematic, N-dependent bias that a parity
// claim "against the reference's own arithmetic" has to reproduce.  A chain of 10^6..10^8 dependent adds cannot run as ONE
// dependency chain on a GPU, and it need not:
//
// RESTARTED SEQUENTIAL SUMS.  Cut a chain into segments.  Given the accumulator value at the START of a segment, the segment is an
// ordinary sequential fp32 loop and all segments run in parallel.  The starts are found by fixed-point iteration:
//   pass 1: every segment starts from 0            -> deltas (end - start) ~ the exact segment sums
//   scan  : start[s] = sum of the deltas before s  (fp64 arithmetic on fp32-representable values: exact)
//   pass 2: segments restart from these values     -> deltas carry the rounding of an accumulator of the right magnitude
//   scan, pass 3, scan ...
// A segment's delta depends on its start only through the binade the running sum is in (inside one binade every term is rounded to
// the same grid, whatever multiple of the grid the start is) and through exact ties, so every pass shrinks the distance to the fixed
// point by about the relative size of the rounding bias itself (1e-2 .. 1e-4): three passes (the default) leave starts that are
// ~1e-8 from the fixed point -- far below what the bias being reproduced amounts to -- and at the fixed point, where every
// segment's start equals the end of the segment before it bit for bit, the concatenated segment loops ARE the sequential loop, i.e.
// the chain total is bit-identical to the one-after-the-other accumulator (tests: equality for enough passes).  The last scan
// reports how far the starts still moved: hmx_get "seq:mismatch" (segments), "seq:residual" (largest move of a start relative to the
// largest start of its chain).  Long chains are iterated until that residual is below 2^-22 (at most "seq_max_passes" passes): when an
// accumulator SATURATES -- my_accu over 10^9 terms at 10M cells stalls 25 % below the exact sum -- the iteration needs more than three.
#include "hmx_internal.h"
#include <float.h>

namespace hmx {

// ---- pass kernels ----------------------------------------------------------------------------------------------------------------
// (a) O / E sums: a chain set runs over a LIST of cells (the round's shuffled order, or the original order for the head) and carries
//     (1 + B) * K lane-chains: row 0 = sum(Rcells, 1) (every cell), row 1 + b = the cells of level b only (Rcells * Phi_tcells: Armadillo
//     walks the sparse operand column by column, i.e. per level in ascending position).  One wave per (segment, 64 clusters), lane =
//     cluster: row 0 lives in a register, the level rows in LDS (one column per lane: no cross-lane traffic, the read-modify-write of
//     a lane's own slot is its sequential chain).  The 64 cell ids / level codes of a batch are fetched with one load each, the R rows
//     32 at a time (a segment is latency-bound: what counts is the number of memory round trips).
// ROWS = false: row 0 only -- a plain sequential sum of R_k over a list of cells, W = K lane-chains per segment (the level-pair sums of
// Phi_Rk * Phi_moe_t for several covariates, src/harmony.cpp:561-568).
// Round 4: all 64 R loads of a batch are in flight together (a segment is latency-bound: one memory round trip per 64 cells instead of two).
// poslev (or nullptr): the level codes of the list's positions, [c][nlist] -- without them a wave walks list -> combo -> qlev -> R, four
// dependent memory round trips before the first add of a 64-cell batch; with them two.
// Round 5: NLV > 0 -- the level rows live in REGISTERS (B <= NLV): the level of a cell is wave-uniform, so "add to row b" is a scalar
// branch into one v_add_f32 (a uniform switch), ~10 cycles per cell, where the LDS row's read-modify-write -- which must wait for the
// write of the cell before, the compiler cannot prove the rows differ -- cost 150-200: a 128-cell segment took 20 us, 10 of them there.
// (vector-typed rows: a dynamic but UNIFORM element index is register-indexed addressing -- s_set_gpr_idx / v_movrel --, three instructions;
//  a switch over 32 cases inside the 64-cell unrolled loop kept hipcc from unrolling it and sent the R values to scratch)
// (round 6: 32 rows as ONE 32-element vector -- two 16-element vectors behind `if (b < 16)` made hipcc copy the whole second vector around every indexed add)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
template <int NLV> struct LvVec { typedef f32x16 type; };
template <> struct LvVec<32> { typedef f32x32 type; };
template <int NLV> struct LvRows {
  typename LvVec<NLV>::type w;
};
template <int NLV> __device__ __forceinline__ void lv_add(LvRows<NLV>& lv, const int b, const float r) { lv.w[b] = __fadd_rn(lv.w[b], r); }
template <bool ROWS, int NLV = 0>
__global__ __launch_bounds__(256) void k_seq_oe_pass(const float* __restrict__ R, int K, int B, int C, const int* __restrict__ list,
                                                     const int* __restrict__ poslev, int nlist,
                                                     const int* __restrict__ combo, const int* __restrict__ qlev,
                                                     const SeqSeg* __restrict__ segs, int seg0, int nsegs,
                                                     const float* __restrict__ start, float* __restrict__ end, int zero_start, unsigned* __restrict__ conv_zero) {
  extern __shared__ float acc_[];                        // [waves][B][64]
  if (conv_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) conv_zero[threadIdx.x] = 0u;      // the statistics words of the scan that follows (no memset launch)
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int sl = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + wib));      // wave-uniform: scalar loop control below
  if (sl >= nsegs) return;
  const int seg = seg0 + sl;
  const int k = blockIdx.y * 64 + lane, ks = min(k, K - 1);
  const SeqSeg sg = segs[seg];
  float* const acc = acc_ + (size_t)wib * B * 64 + lane;
  const int NR = ROWS ? 1 + B : 1;
  const size_t so = (size_t)seg * NR * K + ks;
  float s0 = (zero_start || k >= K) ? 0.0f : start[so];
  LvRows<NLV> lv;
  if constexpr (ROWS && NLV > 0) {
#pragma unroll
    for (int b = 0; b < NLV; b++) lv.w[b] = (zero_start || k >= K || b >= B) ? 0.0f : start[so + (size_t)(1 + min(b, B - 1)) * K];
  } else if constexpr (ROWS) for (int b = 0; b < B; b++) acc[b * 64] = (zero_start || k >= K) ? 0.0f : start[so + (size_t)(1 + b) * K];
  // software pipeline over the batches of 64 cells: the ids (cell, level codes) of batch b + 2 and the 64 R loads of batch b + 1 are in
  // flight while batch b is added up -- a segment of 128 cells costs two memory round trips instead of four
  struct Ids { int myc, myq, lev[4]; };
  auto fetch_ids = [&](const int base) __attribute__((always_inline)) {
    Ids I; I.myq = 0; I.lev[0] = I.lev[1] = I.lev[2] = I.lev[3] = 0;
    const int ci = sg.off + min(base + lane, sg.cnt - 1);
    I.myc = list ? list[min(ci, sg.off + sg.cnt - 1)] : ci;
    if constexpr (ROWS) {
      if (poslev && C <= 4) {
#pragma unroll
        for (int c = 0; c < 4; c++) I.lev[c] = poslev[(size_t)min(c, C - 1) * nlist + ci];
      } else {
        I.myq = combo[I.myc];
#pragma unroll
        for (int c = 0; c < 4; c++) I.lev[c] = qlev[I.myq * C + min(c, C - 1)];
      }
    }
    return I;
  };
  auto fetch_r = [&](const Ids& I, float (&r)[64]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 64; u++) {
      const int cell = __builtin_amdgcn_readlane(I.myc, u);       // (lanes past the end hold the segment's last cell: a valid row)
      r[u] = R[(size_t)cell * K + ks];
    }
  };
  Ids idc = fetch_ids(0), idn = fetch_ids(64);
  float rc[64], rn[64];
  fetch_r(idc, rc);
  for (int base = 0; base < sg.cnt; base += 64) {
    const int nc = min(64, sg.cnt - base);
    const bool more = base + 64 < sg.cnt;
    if (more) fetch_r(idn, rn);
    const Ids idnn = fetch_ids(base + 128);                     // (clamped into the segment: harmless when there is no such batch)
#pragma unroll
    for (int u = 0; u < 64; u++) {
      if (u < nc) {
        s0 = __fadd_rn(s0, rc[u]);
        if constexpr (ROWS) {
#pragma unroll
          for (int c = 0; c < 4; c++) {
            if (c < C) {
              const int b = __builtin_amdgcn_readlane(idc.lev[c], u);
              if constexpr (NLV > 0) lv_add<NLV>(lv, b, rc[u]);
              else acc[b * 64] = __fadd_rn(acc[b * 64], rc[u]);
            }
          }
          for (int c = 4; c < C; c++) {       // (more than four covariates: level codes straight from the table)
            const int b = __builtin_amdgcn_readfirstlane(qlev[__builtin_amdgcn_readlane(idc.myq, u) * C + c]);
            if constexpr (NLV > 0) lv_add<NLV>(lv, b, rc[u]);
            else acc[b * 64] = __fadd_rn(acc[b * 64], rc[u]);
          }
        }
      }
    }
    idc = idn; idn = idnn;
#pragma unroll
    for (int u = 0; u < 64; u++) rc[u] = rn[u];
  }
  if (k < K) {
    end[so] = s0;
    if constexpr (ROWS && NLV > 0) {
#pragma unroll
      for (int b = 0; b < NLV; b++) if (b < B) end[so + (size_t)(1 + b) * K] = lv.w[b];
    } else if constexpr (ROWS) for (int b = 0; b < B; b++) end[so + (size_t)(1 + b) * K] = acc[b * 64];
  }
}

// (a'') round 6: the same sums LEVEL BY LEVEL.  In the loop above every (cell, covariate) costs a read-modify-write of a register chosen at run time --
//      v_readlane, s_set_gpr_idx_on, v_mov, off, v_add, on, v_mov, off, between scalar branches on u < nc and c < C: ~190 cycles per cell, 10 of the 18 us
//      of a pass over a 50k-cell block (phase clocks of a traced wave: ids 0.6, first 64 R loads 2.9, issue of the next batch 1.6, the adds 10.2 us).  A row's
//      chain only fixes the order of ITS OWN cells, so a batch of 64 cells is taken row by row instead: the lanes hold the cells' level codes, one
//      v_cmp per level gives the 64-bit mask of the row's cells (wave-uniform), and the row -- a register chosen at COMPILE time -- adds them in ascending
//      position; what is chosen at run time is only the R value READ (one indexed v_mov).  Same chains, same order inside every chain: bit-identical totals.
template <int NLV>
__global__ __launch_bounds__(256) void k_seq_oe_pass_bl(const float* __restrict__ R, int K, int B, int C, const int* __restrict__ list,
                                                        const int* __restrict__ poslev, int nlist, const int* __restrict__ combo, const int* __restrict__ qlev,
                                                        const SeqSeg* __restrict__ segs, int seg0, int nsegs,
                                                        const float* __restrict__ start, float* __restrict__ end, int zero_start, unsigned* __restrict__ conv_zero) {
  if (conv_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) conv_zero[threadIdx.x] = 0u;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int sl = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + wib));
  if (sl >= nsegs) return;
  const int seg = seg0 + sl;
  const int k = blockIdx.y * 64 + lane, ks = min(k, K - 1);
  const SeqSeg sg = segs[seg];
  const size_t so = (size_t)seg * (1 + B) * K + ks;
  float s0 = (zero_start || k >= K) ? 0.0f : start[so];
  LvRows<NLV> lv;
#pragma unroll
  for (int b = 0; b < NLV; b++) lv.w[b] = (zero_start || k >= K || b >= B) ? 0.0f : start[so + (size_t)(1 + min(b, B - 1)) * K];
  struct Ids { int myc, lev[4]; };
  auto fetch_ids = [&](const int base) __attribute__((always_inline)) {
    Ids I;
    const int ci = sg.off + min(base + lane, sg.cnt - 1);
    I.myc = list ? list[min(ci, sg.off + sg.cnt - 1)] : ci;
    if (poslev) {
#pragma unroll
      for (int c = 0; c < 4; c++) I.lev[c] = poslev[(size_t)min(c, C - 1) * nlist + ci];
    } else {
      const int myq = combo[I.myc];
#pragma unroll
      for (int c = 0; c < 4; c++) I.lev[c] = qlev[myq * C + min(c, C - 1)];
    }
    return I;
  };
  auto fetch_r = [&](const Ids& I, f32x32& lo, f32x32& hi) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 32; u++) {
      lo[u] = R[(size_t)__builtin_amdgcn_readlane(I.myc, u) * K + ks];            // (lanes past the end hold the segment's last cell: a valid row)
      hi[u] = R[(size_t)__builtin_amdgcn_readlane(I.myc, 32 + u) * K + ks];
    }
  };
  Ids idc = fetch_ids(0), idn = fetch_ids(64);
  f32x32 cl, chh, nl, nh;
  fetch_r(idc, cl, chh);
  for (int base = 0; base < sg.cnt; base += 64) {
    const int nc = min(64, sg.cnt - base);
    const unsigned long long vm = nc >= 64 ? ~0ull : ((1ull << nc) - 1ull);
    const bool more = base + 64 < sg.cnt;
    if (more) fetch_r(idn, nl, nh);
    const Ids idnn = fetch_ids(base + 128);
    if (nc == 64) {
#pragma unroll
      for (int u = 0; u < 32; u++) s0 = __fadd_rn(s0, cl[u]);
#pragma unroll
      for (int u = 0; u < 32; u++) s0 = __fadd_rn(s0, chh[u]);
    } else {
#pragma unroll
      for (int u = 0; u < 32; u++) if (u < nc) s0 = __fadd_rn(s0, cl[u]);
#pragma unroll
      for (int u = 0; u < 32; u++) if (32 + u < nc) s0 = __fadd_rn(s0, chh[u]);
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (c < C) {
        const int levc = idc.lev[c];
#pragma unroll
        for (int b = 0; b < NLV; b++) {
          if (b < B) {
            const unsigned long long m = __ballot(levc == b) & vm;
            if (m) {
              unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
              float a = lv.w[b];
              while (mlo) { const int u = __builtin_ctz(mlo); mlo &= mlo - 1u; a = __fadd_rn(a, cl[u]); }
              while (mhi) { const int u = __builtin_ctz(mhi); mhi &= mhi - 1u; a = __fadd_rn(a, chh[u]); }
              lv.w[b] = a;
            }
          }
        }
      }
    }
    idc = idn; idn = idnn; cl = nl; chh = nh;
  }
  if (k < K) {
    end[so] = s0;
#pragma unroll
    for (int b = 0; b < NLV; b++) if (b < B) end[so + (size_t)(1 + b) * K] = lv.w[b];
  }
}

// (a') round 6: MANY levels (B > 32: BASELINE configs[4] has 200 in three nested covariates).  The LDS rows of the form above cost a dependent LDS
//      read-modify-write per (cell, covariate) -- 452 us per pass over a 50k-cell block, 55 % of a reference-arithmetic run at that shape -- and, with
//      (1 + B) K = 40 200 lane-chains per 128-cell segment, more workspace traffic than the block's R rows.  Here the levels are dealt to LEVEL GROUPS of 32:
//      wave g of a workgroup owns the rows of levels [32 g, 32 g + 32) in REGISTERS (the fast path above), all waves of the workgroup walk the same
//      (segment, 64 clusters) -- the R values come from L1 for all but the first -- and a cell's level is a wave-uniform compare + branch per covariate:
//      the row is touched by the one wave that owns it.  Row 0 (every cell) belongs to group 0.  Segments are long (512 cells: a quarter of the
//      workspace traffic; the adds per wave stay short because a wave only meets the cells of its own levels).
__global__ __launch_bounds__(512) void k_seq_oe_pass_lg(const float* __restrict__ R, int K, int B, int C, const int* __restrict__ list,
                                                        const int* __restrict__ poslev, int nlist, const int* __restrict__ combo, const int* __restrict__ qlev,
                                                        const SeqSeg* __restrict__ segs, int seg0, int nsegs, int4 cb,
                                                        const float* __restrict__ start, float* __restrict__ end, int zero_start, unsigned* __restrict__ conv_zero) {
  if (conv_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) conv_zero[threadIdx.x] = 0u;
  const int lane = threadIdx.x & 63;
  const int lg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));         // level group of this wave
  const int b_lo = 32 * lg;
  if (b_lo >= B) return;
  int cmask = 0;                     // covariates (of the first four) with a level in [b_lo, b_lo + 32): cb = their cumulative ends
  { int lo = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) { const int hi = c == 0 ? cb.x : c == 1 ? cb.y : c == 2 ? cb.z : cb.w; if (c < C && lo < b_lo + 32 && hi > b_lo) cmask |= 1 << c; lo = hi; } }
  cmask = __builtin_amdgcn_readfirstlane(cmask);
  const int seg = seg0 + blockIdx.x;
  const int k = blockIdx.y * 64 + lane, ks = min(k, K - 1);
  const SeqSeg sg = segs[seg];
  const size_t so = (size_t)seg * (1 + B) * K + ks;
  float s0 = (zero_start || k >= K || lg != 0) ? 0.0f : start[so];
  LvRows<32> lv;
#pragma unroll
  for (int b = 0; b < 32; b++) lv.w[b] = (zero_start || k >= K || b_lo + b >= B) ? 0.0f : start[so + (size_t)(1 + min(b_lo + b, B - 1)) * K];
  struct Ids { int myc, myq, lev[4]; };
  auto fetch_ids = [&](const int base) __attribute__((always_inline)) {
    Ids I; I.myq = 0; I.lev[0] = I.lev[1] = I.lev[2] = I.lev[3] = 0;
    const int ci = sg.off + min(base + lane, sg.cnt - 1);
    I.myc = list ? list[min(ci, sg.off + sg.cnt - 1)] : ci;
    if (poslev && C <= 4) {
#pragma unroll
      for (int c = 0; c < 4; c++) I.lev[c] = poslev[(size_t)min(c, C - 1) * nlist + ci];
    } else {
      I.myq = combo[I.myc];
#pragma unroll
      for (int c = 0; c < 4; c++) I.lev[c] = qlev[I.myq * C + min(c, C - 1)];
    }
    return I;
  };
  auto fetch_r = [&](const Ids& I, f32x32& lo, f32x32& hi) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 32; u++) {
      lo[u] = R[(size_t)__builtin_amdgcn_readlane(I.myc, u) * K + ks];
      hi[u] = R[(size_t)__builtin_amdgcn_readlane(I.myc, 32 + u) * K + ks];
    }
  };
  Ids idc = fetch_ids(0), idn = fetch_ids(64);
  f32x32 cl, chh, nl, nh;
  fetch_r(idc, cl, chh);
  for (int base = 0; base < sg.cnt; base += 64) {
    const int nc = min(64, sg.cnt - base);
    const unsigned long long vm = nc >= 64 ? ~0ull : ((1ull << nc) - 1ull);
    const bool more = base + 64 < sg.cnt;
    if (more) fetch_r(idn, nl, nh);
    const Ids idnn = fetch_ids(base + 128);
    if (lg == 0) {
#pragma unroll
      for (int u = 0; u < 32; u++) if (u < nc) s0 = __fadd_rn(s0, cl[u]);
#pragma unroll
      for (int u = 0; u < 32; u++) if (32 + u < nc) s0 = __fadd_rn(s0, chh[u]);
    }
    // row by row (k_seq_oe_pass_bl): the covariates whose levels meet this wave's 32, one v_cmp per row and covariate, the row's cells added in ascending position
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (c < C && ((cmask >> c) & 1)) {
        const int levc = idc.lev[c] - b_lo;
#pragma unroll
        for (int b = 0; b < 32; b++) {
          const unsigned long long m = __ballot(levc == b) & vm;
          if (m) {
            unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
            float a = lv.w[b];
            while (mlo) { const int u = __builtin_ctz(mlo); mlo &= mlo - 1u; a = __fadd_rn(a, cl[u]); }
            while (mhi) { const int u = __builtin_ctz(mhi); mhi &= mhi - 1u; a = __fadd_rn(a, chh[u]); }
            lv.w[b] = a;
          }
        }
      }
    }
    for (int c = 4; c < C; c++) {       // (more than four covariates: level codes straight from the table, cell by cell)
#pragma unroll
      for (int u = 0; u < 64; u++) {
        if (u < nc) {
          const int b = __builtin_amdgcn_readfirstlane(qlev[__builtin_amdgcn_readlane(idc.myq, u) * C + c]) - b_lo;
          if (b >= 0 && b < 32) lv_add<32>(lv, b, u < 32 ? cl[u & 31] : chh[u & 31]);
        }
      }
    }
    idc = idn; idn = idnn; cl = nl; chh = nh;
  }
  if (k < K) {
    if (lg == 0) end[so] = s0;
#pragma unroll
    for (int b = 0; b < 32; b++) if (b_lo + b < B) end[so + (size_t)(1 + b_lo + b) * K] = lv.w[b];
  }
}

// (b) ridge statistics of KPW = 8 clusters per wave: W = K * 64 lane-chains per segment.  lane j < d: sum_i fl(z_ij * R_ki)
//     (Z_tmp = Z_orig % R_k is rounded to fp32 first, src/harmony.cpp:592); lane 63: sum_i R_ki (the matching entry of
//     Phi* diag(R_k) Phi*^T, :567).  A cell enters cluster k's regression only if one of its levels is kept for k (:400,456-460):
//     inset[combination][k] (bytes, row stride KP8); a cell outside contributes the term +0, which leaves an fp32 accumulator untouched.
//     Round 4: ONE workgroup per segment, one wave per 8 clusters -- all waves of a workgroup walk the same cells, so a cell's embedding
//     row comes from HBM once (round 3: one single-wave workgroup per (segment, 8 clusters): 13 x the Z traffic at K = 100, 3.4 ms per
//     pass at 1M cells).  The cell's 8 R values are wave-uniform: they travel through the SCALAR cache (s_load_dwordx8) and feed
//     v_pk_mul_f32 / v_pk_add_f32 straight from SGPR pairs -- two clusters per VALU instruction, each component rounded on its own
//     (no contraction: the reference multiplies, rounds, then adds).  10 VALU instructions per (cell, 8 clusters) instead of ~30.
//     (Round 5 tried a software pipeline over the batches -- ids two ahead through a static combination list `listq`, raw R / flags one ahead, all 64
//      rows of a batch in flight together: 128 VGPRs, 1.20 -> 1.46 ms per pass.  The form below stays; `listq` is accepted and unused.  Two workgroups
//      per CU at 72 VGPRs: 21 -> 35 ms per run.  Both measured on the final code of round 5.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KPW>
__global__ __launch_bounds__(1024) void k_seq_ridge_pass(const float* __restrict__ R, const float* __restrict__ Zo, const int* __restrict__ combo,
                                                         int K, int d, int zs, int KP8, const int* __restrict__ list, const int* __restrict__ listq,
                                                         const SeqSeg* __restrict__ segs, int seg0, const unsigned char* __restrict__ inset,
                                                         const float* __restrict__ start, float* __restrict__ end, int zero_start, unsigned* __restrict__ conv_zero) {
#pragma clang fp contract(off)
  static_assert(KPW == 8, "one flag byte per cluster, eight per load");
  if (conv_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) conv_zero[threadIdx.x] = 0u;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int seg = seg0 + blockIdx.x, k0 = (blockIdx.y * (int)(blockDim.x >> 6) + wv) * KPW;
  if (k0 >= K) return;
  const SeqSeg sg = segs[seg];
  f32x2 s[KPW / 2];
#pragma unroll
  for (int kk = 0; kk < KPW; kk++) s[kk >> 1][kk & 1] = (zero_start || k0 + kk >= K) ? 0.0f : start[((size_t)seg * K + k0 + kk) * 64 + lane];
  const int js = min(lane, d - 1);
  // lane j < d: z_j; lane 63: 1 (the mass chain); the others 0 -- as z * zmask + one63 (exact: x * 1 + 0), NOT as a select around the load:
  // hipcc sinks the load into the select's branch and waits for it on the spot (one exposed memory latency per cell)
  const float zmask = lane < d ? 1.0f : 0.0f, one63 = lane == 63 ? 1.0f : 0.0f;
  for (int base = 0; base < sg.cnt; base += 64) {
    const int nc = min(64, sg.cnt - base);
    const int ci = sg.off + min(base + lane, sg.cnt - 1);
    const int myc = list ? list[ci] : ci;
    const unsigned long long fl = *reinterpret_cast<const unsigned long long*>(inset + (size_t)combo[myc] * KP8 + k0);     // 8 flag bytes
    unsigned mym = 0;
#pragma unroll
    for (int kk = 0; kk < KPW; kk++) mym |= ((fl >> (8 * kk)) & 0xffull) ? (1u << kk) : 0u;
    // R of the whole batch with EIGHT vector loads -- lane l of load i holds cluster k0 + (l & 7) of cell 8 i + (l >> 3) -- all in flight
    // together (vector loads retire in order: counted waits, unlike scalar loads); a cell's 8 values then reach the SGPRs by v_readlane and
    // feed the packed multiplies from there.  The in-set flags (and the end of a partial batch) are applied to the loaded value, lane by
    // lane: a masked term is +0 and the inner loop has no branches.  (Clusters >= K of the last group read into the next row -- R has a
    // dummy row behind the last cell -- and are never stored.)  Z rows 16 cells at a time.
    float rq[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int src = 8 * i + (lane >> 3);
      const int cellv = __shfl(myc, src, 64);
      const unsigned mv = (unsigned)__shfl((int)mym, src, 64);
      const float rl = R[(size_t)cellv * K + k0 + (lane & 7)];
      rq[i] = rl * ((src < nc && ((mv >> (lane & 7)) & 1u)) ? 1.0f : 0.0f);       // (a product, not a select around the load: see zmask)
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (16 * i >= nc) break;
      float z[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int cell = __builtin_amdgcn_readlane(myc, 16 * i + u);       // (lanes past the end hold the segment's last cell: a valid row)
        z[u] = __builtin_fmaf(Zo[(size_t)cell * zs + js], zmask, one63);
      }
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const f32x2 zz = {z[u], z[u]};
        const float rsrc = rq[2 * i + (u >> 3)];
#pragma unroll
        for (int h = 0; h < KPW / 2; h++) {
          const f32x2 r2 = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(rsrc), 8 * (u & 7) + 2 * h)),
                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rsrc), 8 * (u & 7) + 2 * h + 1))};
          s[h] = s[h] + zz * r2;
        }
      }
    }
  }
#pragma unroll
  for (int kk = 0; kk < KPW; kk++)
    if (k0 + kk < K) end[((size_t)seg * K + k0 + kk) * 64 + lane] = s[kk >> 1][kk & 1];
}

// (b') round 6: the same chains with lane = CLUSTER.  k_seq_ridge_pass (lane = PC) pays one v_readlane per (cell, cluster) to bring R_ki into a scalar
//      operand and runs 13 waves of 8 clusters over every cell: 538 M wave-instructions per pass at 1M cells, 0.87 of its 1.1 ms pure VALU issue
//      (profiles/r5_ref_arith_pmc_sq.txt).  Here a lane owns cluster k and holds the d + 1 accumulators of ITS (k, j) chains in registers (j < d: sum_i
//      fl(z_ij R_ki); j = d: the mass sum_i R_ki, as fl(1 * R_ki)): R_ki is the lane's own coalesced load, the in-set flag its own byte, and the cell's
//      embedding row is the wave-UNIFORM operand -- it comes through the scalar cache (s_load) and feeds v_pk_mul_f32 / v_pk_add_f32 from SGPR pairs, two
//      PCs per instruction, each product rounded before it is added (no contraction: :592).  2 x NP VALU instructions per (cell, 64 clusters) and no
//      cross-lane traffic at all.  Same chains, same order, same roundings as k_seq_ridge_pass: the totals are bit-identical.
//      start / end: [segment][d + 1][K] (row j contiguous over the clusters: coalesced); k_seq_ridge_rows2lanes hands the chain totals to the consumers in
//      k_seq_ridge_pass's [chain][K][64] layout.
template <int ZS4>      // the embedding rows hold exactly 4 ZS4 floats (zs; zero beyond d)
__global__ __launch_bounds__(256) void k_seq_ridge_pass_kl(const float* __restrict__ R, const float* __restrict__ Zo, const int* __restrict__ combo, int K, int d, int KP8,
                                                           const int* __restrict__ list, const SeqSeg* __restrict__ segs, int seg0, int nsegs, const unsigned char* __restrict__ inset,
                                                           const float* __restrict__ start, float* __restrict__ end, int zero_start, unsigned* __restrict__ conv_zero) {
#pragma clang fp contract(off)
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  constexpr int ZS = 4 * ZS4, NP = 2 * ZS4;
  if (conv_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) conv_zero[threadIdx.x] = 0u;
  // the waves that walk the SAME cells (the cluster groups of a segment) sit in one workgroup: the second one finds the rows in the CU's scalar cache
  // (the scalar miss path is what bounds this kernel: one group per workgroup 0.92 ms per pass at 1M cells)
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int ngr = (K + 63) >> 6, gpw = ngr >= 4 ? 4 : ngr >= 2 ? 2 : 1, spw = 4 / gpw;       // cluster groups / segments per workgroup
  const int sl = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * spw + wib / gpw));
  const int grp = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * gpw + wib % gpw));
  if (sl >= nsegs || grp >= ngr) return;
  const int seg = seg0 + sl;
  const int k = grp * 64 + lane, ks = min(k, K - 1);
  const SeqSeg sg = segs[seg];
  const size_t so = (size_t)seg * (d + 1) * K + ks;
  const bool warm = !zero_start && k < K;
  // accumulators: pair p = PCs 2 p, 2 p + 1 (slots beyond d: the rows' zero padding -- they add +0 and are never stored); the mass on its own
  f32x2 acc[NP];
  float mass = warm ? start[so + (size_t)d * K] : 0.0f;
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const float a0 = start[so + (size_t)min(2 * p, d) * K], a1 = start[so + (size_t)min(2 * p + 1, d) * K];      // (unconditional, clamped loads: no branch per row)
    acc[p][0] = (warm && 2 * p < d) ? a0 : 0.0f; acc[p][1] = (warm && 2 * p + 1 < d) ? a1 : 0.0f;
  }
  for (int base = 0; base < sg.cnt; base += 64) {
    const int nc = min(64, sg.cnt - base);
    const int ci = sg.off + min(base + lane, sg.cnt - 1);
    const int myc = list ? list[ci] : ci;
    const int myq = combo[myc];
    // the lane's own R value and in-set flag of the NEXT cell are requested while this cell's 2 NP packed operations run
    int cell = __builtin_amdgcn_readlane(myc, 0), q = __builtin_amdgcn_readlane(myq, 0);
    float rn = R[(size_t)cell * K + ks];
    unsigned char fn = inset[(size_t)q * KP8 + ks];
    for (int u = 0; u < nc; u++) {
      const float r = rn * (fn ? 1.0f : 0.0f);                          // (a product, not a select around the load)
      const f32x4_* __restrict__ zp = reinterpret_cast<const f32x4_*>(Zo + (size_t)cell * ZS);      // wave-uniform: the row comes through the scalar cache
      const int un = min(u + 1, nc - 1);
      cell = __builtin_amdgcn_readlane(myc, un); q = __builtin_amdgcn_readlane(myq, un);
      rn = R[(size_t)cell * K + ks]; fn = inset[(size_t)q * KP8 + ks];
      const f32x2 rr = {r, r};
      f32x4_ zq[ZS4];
#pragma unroll
      for (int i = 0; i < ZS4; i++) zq[i] = zp[i];
#pragma unroll
      for (int i = 0; i < ZS4; i++) {
        const f32x2 za = {zq[i][0], zq[i][1]}, zb = {zq[i][2], zq[i][3]};
        acc[2 * i] = acc[2 * i] + za * rr;
        acc[2 * i + 1] = acc[2 * i + 1] + zb * rr;
      }
      mass = __fadd_rn(mass, r);
    }
  }
  if (k < K) {
#pragma unroll
    for (int p = 0; p < NP; p++) {
      if (2 * p < d) end[so + (size_t)(2 * p) * K] = acc[p][0];
      if (2 * p + 1 < d) end[so + (size_t)(2 * p + 1) * K] = acc[p][1];
    }
    end[so + (size_t)d * K] = mass;
  }
  (void)ZS;
}
// chain totals [chain][d + 1][K] -> [chain][K][64] (lane j < d: the PCs, lane 63: the mass; the lanes between: 0)
__global__ void k_seq_ridge_rows2lanes(const float* __restrict__ in, float* __restrict__ out, int nchains, int K, int d) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)nchains * K * 64) return;
  const int j = (int)(i & 63); const size_t ck = i >> 6; const int k = (int)(ck % K); const size_t chain = ck / K;
  const int row = j < d ? j : (j == 63 ? d : -1);
  out[i] = row >= 0 ? in[(chain * (size_t)(d + 1) + row) * K + k] : 0.0f;
}

// (c) a contiguous array of terms (the objective's three K*N-term chains, one array each): thread = segment of L terms.
//     Round 4: a thread reads one whole 128-byte line (32 terms) per step, the next line already in flight while the 32 dependent adds of
//     this one run (round 3: 64 bytes per step and no prefetch -- with one wave per SIMD every step exposed a full memory latency).
// Round 5: every workgroup also leaves the fp64 sum of its 256 segments' deltas (end - start) in `partial[array][workgroup]`, added in a fixed
// order: the scan that follows (k_seq_scan1) is then ONE wide launch -- a workgroup per 256 segments takes its base from the partials in
// front of it -- instead of one 1024-thread workgroup per array walking 49k segments (131 us per scan at 1M cells, 77 scans per run).
__global__ __launch_bounds__(256) void k_seq_arr_pass(const float* __restrict__ T, long long n, long long stride, int L, int nsegs,
                                                      const float* __restrict__ start, float* __restrict__ end, int zero_start,
                                                      double* __restrict__ partial, unsigned* __restrict__ conv_zero) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ double bsum[256];
  if (conv_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) conv_zero[threadIdx.x] = 0u;
  const int seg = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = seg < nsegs;
  const float* __restrict__ t = T + (size_t)blockIdx.y * (size_t)stride;
  const size_t so = (size_t)blockIdx.y * nsegs + min(seg, nsegs - 1);
  const long long off = (long long)min(seg, nsegs - 1) * L;
  const int cnt = live ? (int)min((long long)L, n - off) : 0;
  const float s_in = (zero_start || !live) ? 0.0f : start[so];
  float s = s_in;
  int i = 0;
  if ((((uintptr_t)(t + off)) & 15) == 0 && cnt >= 32) {
    const f4* __restrict__ t4 = reinterpret_cast<const f4*>(t + off);
    const int nl = cnt >> 5;                          // whole lines
    f4 c[8], nx[8];
#pragma unroll
    for (int q = 0; q < 8; q++) c[q] = t4[q];
    for (int l = 0; l < nl; l++) {
      const int ln = min(l + 1, nl - 1);              // (the last step re-reads its own line: no branch around the loads)
#pragma unroll
      for (int q = 0; q < 8; q++) nx[q] = t4[8 * ln + q];
#pragma unroll
      for (int q = 0; q < 8; q++) { s = __fadd_rn(s, c[q][0]); s = __fadd_rn(s, c[q][1]); s = __fadd_rn(s, c[q][2]); s = __fadd_rn(s, c[q][3]); }
#pragma unroll
      for (int q = 0; q < 8; q++) c[q] = nx[q];
    }
    i = nl << 5;
  }
  for (; i < cnt; i++) s = __fadd_rn(s, t[off + i]);
  if (live) end[so] = s;
  bsum[threadIdx.x] = live ? (double)s - (double)s_in : 0.0;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) bsum[threadIdx.x] += bsum[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = bsum[0];
}

// (c') Round 5: the objective's entropy and cross-entropy chains WITHOUT their term arrays.  T[1] = (R % log R) % sigma and T[2] = (R % sigma) % (M Phi)
//     (src/harmony.cpp:161-162) are functions of R alone: thread = the same segment of L consecutive terms (original cell order, k fastest) as in
//     k_seq_arr_pass, walking the cells' R rows through invperm and forming both terms on the fly with exactly the roundings k_obj_terms_mfma used
//     (v_log_f32 * ln 2; product, round, product, round; M rows added in covariate order) -- two dependent chains per thread instead of one, 400 MB read per
//     pass instead of 800 MB, and 800 MB less written per evaluation.  Needs K % 4 == 0 and L % 4 == 0 (16-byte loads stay inside a row).
//     LDSTAB (counters of round 5, profiles/r5_ref_arith_pmc_sq.txt: the waves of this pass wait on memory 78 % of their life -- every lane of a load addresses its
//     own cache line, the texture-address unit serves them one line per cycle, and two of the three loads of a term are table lookups): the M table (B x K) and
//     sigma are staged in LDS once per workgroup and read from there (ds_read_b128); only the R row stays a global load.  Tables beyond 32 KB: the global form.
template <bool LDSTAB>
__global__ __launch_bounds__(256) void k_seq_objr_pass(Dev D, const float* __restrict__ M, long long n, int L, int nsegs,
                                                       const float* __restrict__ start, float* __restrict__ end, int zero_start, double* __restrict__ partial) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ double bsum[2][256];
  extern __shared__ __attribute__((aligned(16))) float ltab[];      // [B][K] M, then [K] sigma (LDSTAB)
  const int K = D.K, C = D.C;
  if constexpr (LDSTAB) {
    for (int i = threadIdx.x; i < D.B * K; i += blockDim.x) ltab[i] = M[i];
    for (int i = threadIdx.x; i < K; i += blockDim.x) ltab[D.B * K + i] = D.sigma[i];
    __syncthreads();
  }
  const float* const lsig = ltab + D.B * K;
  const int seg = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = seg < nsegs;
  const long long off = (long long)min(seg, nsegs - 1) * L;
  int cnt = live ? (int)min((long long)L, n - off) : 0;
  const size_t so1 = (size_t)nsegs + min(seg, nsegs - 1), so2 = (size_t)2 * nsegs + min(seg, nsegs - 1);
  const float s1_in = (zero_start || !live) ? 0.0f : start[so1], s2_in = (zero_start || !live) ? 0.0f : start[so2];
  float s1 = s1_in, s2 = s2_in;
  const float lmin = __builtin_amdgcn_logf(FLT_MIN) * 0.69314718055994530942f;
  int cell = (int)(off / K), k = (int)(off - (long long)cell * K);
  const float* __restrict__ Rp = D.R;
  while (cnt > 0) {
    const int icell = D.invperm[min(cell, D.n - 1)];
    const int q = D.combo[icell];
    const float* __restrict__ rrow = Rp + (size_t)icell * K;
    int lev[4];
#pragma unroll
    for (int cc = 0; cc < 4; cc++) lev[cc] = D.qlev[q * C + min(cc, C - 1)];
    const int kend = min(K, k + cnt);
    constexpr int NU = 4;                                  // 16 terms of this row per step, all loads of the step in flight together (few registers: the
    for (int k0 = k; k0 < kend; k0 += 4 * NU) {            // latency of a step is hidden by the other waves of the SIMD, segments are short -- L = 512 -- and many)
      const int nq = min(NU, (kend - k0 + 3) >> 2);
      f4 r4[NU], g4[NU], m4[NU];
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const int kk = min(k0 + 4 * u, K - 4);
        r4[u] = *reinterpret_cast<const f4*>(rrow + kk);
        if constexpr (LDSTAB) g4[u] = *reinterpret_cast<const f4*>(lsig + kk); else g4[u] = *reinterpret_cast<const f4*>(D.sigma + kk);
        f4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
          if (cc < C) {
            f4 mm;
            if constexpr (LDSTAB) mm = *reinterpret_cast<const f4*>(ltab + lev[cc] * K + kk); else mm = *reinterpret_cast<const f4*>(M + (size_t)lev[cc] * K + kk);
#pragma unroll
            for (int i = 0; i < 4; i++) m[i] = __fadd_rn(m[i], mm[i]); }
        }
        for (int cc = 4; cc < C; cc++) { const f4 mm = *reinterpret_cast<const f4*>(M + (size_t)D.qlev[q * C + cc] * K + kk);
#pragma unroll
          for (int i = 0; i < 4; i++) m[i] = __fadd_rn(m[i], mm[i]); }
        m4[u] = m;
      }
#pragma unroll
      for (int u = 0; u < NU; u++) {
        if (u < nq) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            if (k0 + 4 * u + i < kend) {
              const float r = r4[u][i], sg = g4[u][i];
              const float lg = (r > 0.0f) ? __fmul_rn(__builtin_amdgcn_logf(r), 0.69314718055994530942f) : lmin;      // arma::trunc_log
              s1 = __fadd_rn(s1, __fmul_rn(__fmul_rn(r, lg), sg));
              s2 = __fadd_rn(s2, __fmul_rn(__fmul_rn(r, sg), m4[u][i]));
            }
          }
        }
      }
    }
    cnt -= kend - k; cell++; k = 0;
  }
  if (live) { end[so1] = s1; end[so2] = s2; }
  bsum[0][threadIdx.x] = live ? (double)s1 - (double)s1_in : 0.0;
  bsum[1][threadIdx.x] = live ? (double)s2 - (double)s2_in : 0.0;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) { bsum[0][threadIdx.x] += bsum[0][threadIdx.x + o]; bsum[1][threadIdx.x] += bsum[1][threadIdx.x + o]; } __syncthreads(); }
  if (threadIdx.x == 0) { partial[(size_t)gridDim.x + blockIdx.x] = bsum[0][0]; partial[(size_t)2 * gridDim.x + blockIdx.x] = bsum[1][0]; }
}

// ---- (d) round 6: the objective's three chains in ONE launch -----------------------------------------------------------------------
// The passes above re-read their 400 MB term arrays for every pass of the fixed-point iteration (2.7 passes x 0.33 ms per evaluation at 1M
// cells, 21 evaluations per run).  Here a thread keeps its segment -- 32 consecutive terms of each of the three chains -- in REGISTERS, so a
// further pass costs 96 adds instead of a sweep over HBM; what the scan kernels did between the passes happens inside the launch:
//   * workgroup: the deltas (end - start, fp64 on fp32-representable values) are scanned over the wave (shuffles) and the 8 waves (LDS);
//   * grid: every workgroup takes a TICKET (its rank in start order: whoever holds a lower ticket is running or done -- no assumption about
//     dispatch order) and publishes its three aggregates as self-validating 8-byte granules {tag, 32 bits} (written through, read around L1 /
//     the other XCDs' L2s: no fence, no flag -- MI355X_MICROARCH.md, the transport the block chain and the peer inboxes use); tag = (launch
//     epoch, stage), so nothing is ever reset.  Two levels: a workgroup sums the aggregates of the lower tickets of its group of 64 (one
//     lane each) and the totals of the groups in front of it (published by each group's last member) -- it never waits for a higher ticket,
//     so the wait graph is acyclic whatever is resident.  One stage per pass; every stage has its own slots (a slow reader of stage p must
//     not find stage p + 2 there).  Every spin is bounded by the wall clock and raises X.err instead of hanging the GPU.
// MODE 0: three term arrays T[c][nt] (the probe hmx_debug_seq_arr, and the fallback when all three arrays are materialised);
// MODE 1: T[0] = R % dist from k_obj_terms_mfma; the entropy and cross-entropy terms are formed from the cells' R rows (through invperm; level
//         codes in original cell order, `olev`) with exactly the roundings of k_obj_terms / k_seq_objr_pass.  K % 4 == 0.
// stats: [0] starts that still moved in the last stage, [1 + c] the largest move of a start of chain c (float bits), [4 + c] the chain totals (float
//        bits), [7] error word, [8] the ticket counter -- zeroed by the host in front of the launch (one 64-byte memset).
struct SeqXchg { unsigned long long* slotA; unsigned long long* slotG; unsigned long long* slotS; unsigned epoch; int ngroups, nsuper; };
__device__ __forceinline__ void xg_put3(unsigned long long* slot, const int lane, const unsigned tag, const double a0, const double a1, const double a2) {
  if (lane < 6) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(lane < 2 ? a0 : lane < 4 ? a1 : a2);
    const unsigned half = (lane & 1) ? (unsigned)(bits & 0xffffffffull) : (unsigned)(bits >> 32);
    __hip_atomic_store(slot + lane, ((unsigned long long)tag << 32) | (unsigned long long)half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void xg_get3(const unsigned long long* slot, const unsigned tag, double& a0, double& a1, double& a2, unsigned* err) {
  unsigned long long g[6];
  const unsigned long long t_in = wall_clock64();
  bool ok = false;
  for (int spins = 0; !ok; spins++) {
    ok = true;
#pragma unroll
    for (int q = 0; q < 6; q++) { g[q] = __hip_atomic_load(slot + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = ok && (unsigned)(g[q] >> 32) == tag; }
    if (ok) break;
    if ((spins & 63) == 63 && (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || wall_clock64() - t_in > 200000000ull)) { atomicExch(err, 9u); break; }
    __builtin_amdgcn_s_sleep(1);
  }
  a0 = __longlong_as_double((long long)(((g[0] & 0xffffffffull) << 32) | (g[1] & 0xffffffffull)));
  a1 = __longlong_as_double((long long)(((g[2] & 0xffffffffull) << 32) | (g[3] & 0xffffffffull)));
  a2 = __longlong_as_double((long long)(((g[4] & 0xffffffffull) << 32) | (g[5] & 0xffffffffull)));
}
__device__ __forceinline__ double wave_sum_d(double v) {      // butterfly: both partners form the same sum at every step -> all lanes end bit-identical
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
// the two halves of a stage's exchange for workgroup `w` (ticket order), run by two different waves at the same time:
//   mates: publish the own aggregate; sum the aggregates of the lower tickets of the group of 64 (one lane each); the group's last member also publishes
//          the group's total;   groups: sum the totals of the groups in front (one lane each, 64 at a time)
__device__ __forceinline__ void xg_mates3(const SeqXchg& X, const int stage, const int w, const int nwg, const int lane, const double A0, const double A1, const double A2,
                                          double& p0, double& p1, double& p2, unsigned* err) {
  const unsigned tag = (X.epoch << 4) | (unsigned)stage;
  const int g = w >> 6, j = w & 63;
  xg_put3(X.slotA + ((size_t)stage * nwg + w) * 8, lane, tag, A0, A1, A2);
  p0 = 0.0; p1 = 0.0; p2 = 0.0;
  if (lane < j) xg_get3(X.slotA + ((size_t)stage * nwg + (size_t)g * 64 + lane) * 8, tag, p0, p1, p2, err);
  p0 = wave_sum_d(p0); p1 = wave_sum_d(p1); p2 = wave_sum_d(p2);
  if (j == 63 || w == nwg - 1) xg_put3(X.slotG + ((size_t)stage * X.ngroups + g) * 8, lane, tag, p0 + A0, p1 + A1, p2 + A2);
}
// (three levels since the 10M-cell runs: groups of 64 workgroups, super-groups of 64 groups.  qi: the totals of the groups in front INSIDE this
//  workgroup's super-group; qs: the totals of the super-groups in front -- published by each super-group's last workgroup, 4096 tickets ago or more)
__device__ __forceinline__ void xg_groups3(const SeqXchg& X, const int stage, const int w, const int lane, double& qi0, double& qi1, double& qi2,
                                           double& qs0, double& qs1, double& qs2, unsigned* err) {
  const unsigned tag = (X.epoch << 4) | (unsigned)stage;
  const int g = w >> 6, sg = g >> 6, g_lo = sg << 6;
  qi0 = 0.0; qi1 = 0.0; qi2 = 0.0; qs0 = 0.0; qs1 = 0.0; qs2 = 0.0;
  if (g_lo + lane < g) xg_get3(X.slotG + ((size_t)stage * X.ngroups + g_lo + lane) * 8, tag, qi0, qi1, qi2, err);
  for (int s0 = 0; s0 < sg; s0 += 64) {
    if (s0 + lane < sg) { double t0, t1, t2; xg_get3(X.slotS + ((size_t)stage * X.nsuper + s0 + lane) * 8, tag, t0, t1, t2, err); qs0 += t0; qs1 += t1; qs2 += t2; }
  }
  qi0 = wave_sum_d(qi0); qi1 = wave_sum_d(qi1); qi2 = wave_sum_d(qi2);
  qs0 = wave_sum_d(qs0); qs1 = wave_sum_d(qs1); qs2 = wave_sum_d(qs2);
}
// inclusive scan of a double over the wave without LDS traffic: DPP row shifts inside the rows of 16, then row_bcast15 / row_bcast31 across them
// (an invalid source lane or a masked row contributes the `old` operand, +0.0)
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_d(const double v) {
  const long long bts = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(bts & 0xffffffffll), CTRL, ROWMASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(bts >> 32), CTRL, ROWMASK, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_scan_d(double v) {
  v += dpp_d<0x111, 0xF>(v); v += dpp_d<0x112, 0xF>(v); v += dpp_d<0x114, 0xF>(v); v += dpp_d<0x118, 0xF>(v);
  v += dpp_d<0x142, 0xA>(v); v += dpp_d<0x143, 0xC>(v);
  return v;
}
constexpr int OBJF_TPT = 32, OBJF_THREADS = 256, OBJF_WAVES = OBJF_THREADS / 64, OBJF_MAXSTAGE = 8;
// a wave's 64 x 32 consecutive terms, loaded COALESCED -- load q of lane l fetches the 16 bytes at term 4 (64 q + l) of the wave's window, a kilobyte per
// instruction -- and dealt to the lanes that own them through LDS (an 8 x 8 transpose of 16-byte items per group of 8 lanes; item (row L, slot r) lives at
// slot r ^ (L & 7) of its row: writes and reads are conflict-free).  (The first version let every lane read its own 128-byte line: eight waves of such
// loads thrash the L1, 1.1 ms per evaluation against 0.22 ms for the same bytes in k_seq_arr_pass.)
template <class LOAD> __device__ __forceinline__ void objf_deal(float* __restrict__ lbuf, const int lane, LOAD load16, float (&out)[OBJF_TPT]) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 v[8];
#pragma unroll
  for (int q = 0; q < 8; q++) v[q] = load16(q * 64 + lane);
  f4* const L4 = reinterpret_cast<f4*>(lbuf);
#pragma unroll
  for (int q = 0; q < 8; q++) { const int i = q * 64 + lane, row = i >> 3, r = i & 7; L4[row * 8 + (r ^ (row & 7))] = v[q]; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int r = 0; r < 8; r++) { const f4 x = L4[lane * 8 + (r ^ (lane & 7))]; out[4 * r] = x[0]; out[4 * r + 1] = x[1]; out[4 * r + 2] = x[2]; out[4 * r + 3] = x[3]; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int MODE, bool LDSTAB, bool DIST = false>      // DIST (MODE 1): T holds dist_mat, the first chain's terms R % dist are formed here (one rounding, as k_obj_terms_mfma forms them)
__global__ __launch_bounds__(OBJF_THREADS, 3) void k_seq_obj_fused(Dev D, const float* __restrict__ T, long long stride, const float* __restrict__ M, const int* __restrict__ olev,
                                                                   long long nt, int nsegs, int npass, int zero_start, float* __restrict__ starts,
                                                                   unsigned* __restrict__ stats, double* __restrict__ wgagg, SeqXchg X) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ double wtot[OBJF_WAVES][3];
  __shared__ double xin[3], xgr[3], xsu[3];
  __shared__ unsigned wgid;
  extern __shared__ __attribute__((aligned(16))) float lds_[];      // [waves][64 x 32] deal buffers, then (MODE 1, LDSTAB) [B][K] M and [K] sigma
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* const lbuf = lds_ + (size_t)wave * 64 * OBJF_TPT;
  float* const ltab = lds_ + (size_t)OBJF_WAVES * 64 * OBJF_TPT;
  if (tid == 0) wgid = atomicAdd(stats + 8, 1u);
  if constexpr (MODE == 1 && LDSTAB) {
    for (int i = tid; i < D.B * D.K; i += OBJF_THREADS) ltab[i] = M[i];
    for (int i = tid; i < D.K; i += OBJF_THREADS) ltab[D.B * D.K + i] = D.sigma[i];
  }
  __syncthreads();
  const int w = (int)wgid, nwg = (int)gridDim.x;
  const int seg = w * OBJF_THREADS + tid;
  const bool live = seg < nsegs;
  const long long tw = ((long long)w * OBJF_THREADS + wave * 64) * OBJF_TPT;      // first term of this wave's window
  const long long t0 = (long long)seg * OBJF_TPT;                                   // first term of this lane's segment
  float a[OBJF_TPT], b[OBJF_TPT], c[OBJF_TPT];
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto plain = [&](const float* __restrict__ arr) {      // 16 bytes of a term array at item i of the window; beyond the end: zeros (a ragged end term by term)
    return [=](const int i) -> f4 {
      const long long t = tw + 4 * (long long)i;
      if (t + 3 < nt) return *reinterpret_cast<const f4*>(arr + t);
      f4 x = zero4;
#pragma unroll
      for (int u = 0; u < 4; u++) if (t + u < nt) x[u] = arr[t + u];
      return x;
    };
  };
  if constexpr (MODE == 0) {
    objf_deal(lbuf, lane, plain(T), a);
    objf_deal(lbuf, lane, plain(T + (size_t)stride), b);
    objf_deal(lbuf, lane, plain(T + 2 * (size_t)stride), c);
  } else {
    const int K = D.K, C = D.C, n = D.n;
    objf_deal(lbuf, lane, plain(T), a);
    // the same items of R: term t belongs to original cell t / K, cluster t % K -> row invperm[cell] of R (K % 4 == 0: an item never straddles a row)
    objf_deal(lbuf, lane, [&](const int i) -> f4 {
      const long long t = tw + 4 * (long long)i;
      if (t >= nt) return zero4;
      const int cell = (int)(t / K), k = (int)(t - (long long)cell * K);
      return *reinterpret_cast<const f4*>(D.R + (size_t)D.invperm[cell] * K + k); }, b);
    const float lmin = __builtin_amdgcn_logf(FLT_MIN) * 0.69314718055994530942f;
    const float* const lsig = ltab + D.B * K;
    int cell = (int)(t0 / K), k = (int)(t0 - (long long)cell * K);
#pragma unroll
    for (int j = 0; j < OBJF_TPT / 4; j++) {
      const bool ok = live && t0 + 4 * j < nt;
      const int cs = min(cell, n - 1), ks = min(k, K - 4);
      f4 g4, m4 = zero4;
      if constexpr (LDSTAB) g4 = *reinterpret_cast<const f4*>(lsig + ks); else g4 = *reinterpret_cast<const f4*>(D.sigma + ks);
#pragma unroll
      for (int cc = 0; cc < 4; cc++) {
        if (cc < C) {
          const int lev = olev[(size_t)cc * n + cs];
          f4 mm;
          if constexpr (LDSTAB) mm = *reinterpret_cast<const f4*>(ltab + lev * K + ks); else mm = *reinterpret_cast<const f4*>(M + (size_t)lev * K + ks);
#pragma unroll
          for (int i = 0; i < 4; i++) m4[i] = __fadd_rn(m4[i], mm[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float r = b[4 * j + i], sg = g4[i];
        const float lg = (r > 0.0f) ? __fmul_rn(__builtin_amdgcn_logf(r), 0.69314718055994530942f) : lmin;      // arma::trunc_log
        if constexpr (DIST) a[4 * j + i] = ok ? __fmul_rn(r, a[4 * j + i]) : 0.0f;
        b[4 * j + i] = ok ? __fmul_rn(__fmul_rn(r, lg), sg) : 0.0f;
        c[4 * j + i] = ok ? __fmul_rn(__fmul_rn(r, sg), m4[i]) : 0.0f;
      }
      k += 4; if (k >= K) { k -= K; cell++; }
    }
  }
  float st0 = 0.0f, st1 = 0.0f, st2 = 0.0f, pv0 = 0.0f, pv1 = 0.0f, pv2 = 0.0f;
  if (!zero_start && live) { st0 = starts[seg]; st1 = starts[(size_t)nsegs + seg]; st2 = starts[2 * (size_t)nsegs + seg]; }
  for (int p = 0; p < npass; p++) {
    float s0 = st0, s1 = st1, s2 = st2;
#pragma unroll
    for (int i = 0; i < OBJF_TPT; i++) { s0 = __fadd_rn(s0, a[i]); s1 = __fadd_rn(s1, b[i]); s2 = __fadd_rn(s2, c[i]); }      // (a masked term is +0: it leaves the accumulator as it is)
    const double d0 = live ? (double)s0 - (double)st0 : 0.0, d1 = live ? (double)s1 - (double)st1 : 0.0, d2 = live ? (double)s2 - (double)st2 : 0.0;
    const double i0 = wave_scan_d(d0), i1 = wave_scan_d(d1), i2 = wave_scan_d(d2);
    if (lane == 63) { wtot[wave][0] = i0; wtot[wave][1] = i1; wtot[wave][2] = i2; }
    double e0 = dpp_d<0x138, 0xF>(i0), e1 = dpp_d<0x138, 0xF>(i1), e2 = dpp_d<0x138, 0xF>(i2);      // wave_shr:1 -> the exclusive prefix (lane 0: +0)
    __syncthreads();
    double A0 = 0.0, A1 = 0.0, A2 = 0.0;
#pragma unroll
    for (int u = 0; u < OBJF_WAVES; u++) {
      const double v0 = wtot[u][0], v1 = wtot[u][1], v2 = wtot[u][2];
      if (u == wave) { e0 += A0; e1 += A1; e2 += A2; }      // (the waves in front of this one)
      A0 += v0; A1 += v1; A2 += v2;
    }
    if (p == npass - 1) {
      // the LAST pass needs no exchange: nobody restarts from its result inside this launch.  The workgroup leaves its aggregate and the start its first
      // segment ran from (this pass and the one before); k_seq_objf_close sums the aggregates in ticket order -> chain totals, and sees how far
      // the workgroups' starts would still move.  The segment starts of this pass are what the next evaluation starts warm from.
      if (live) { starts[seg] = st0; starts[(size_t)nsegs + seg] = st1; starts[2 * (size_t)nsegs + seg] = st2; }
      if (tid == 0) {      // component-major records [7][nwg] of 8 bytes: the closing kernel reads them coalesced
        const size_t nw_ = (size_t)nwg;
        wgagg[w] = A0; wgagg[nw_ + w] = A1; wgagg[2 * nw_ + w] = A2;
        auto pack = [](const float x, const float y) { return __longlong_as_double((long long)(((unsigned long long)__float_as_uint(y) << 32) | (unsigned long long)__float_as_uint(x))); };
        wgagg[3 * nw_ + w] = pack(st0, pv0); wgagg[4 * nw_ + w] = pack(st1, pv1); wgagg[5 * nw_ + w] = pack(st2, pv2);
        // flags: the starts of this pass were iterated ones (not zeros by construction) | so were the ones of the pass before
        wgagg[6 * nw_ + w] = __longlong_as_double((long long)(((npass > 1 || !zero_start) ? 1ull : 0ull) | (npass > 1 ? 2ull : 0ull)));
      }
      break;
    }
    if (wave == 0) { double p0, p1, p2; xg_mates3(X, p, w, nwg, lane, A0, A1, A2, p0, p1, p2, stats + 7); if (lane == 0) { xin[0] = p0; xin[1] = p1; xin[2] = p2; } }
    else if (wave == 1) { double q0, q1, q2, u0, u1, u2; xg_groups3(X, p, w, lane, q0, q1, q2, u0, u1, u2, stats + 7);
                          if (lane == 0) { xgr[0] = q0; xgr[1] = q1; xgr[2] = q2; xsu[0] = u0; xsu[1] = u1; xsu[2] = u2; } }
    __syncthreads();
    if (wave == 0 && ((w & 4095) == 4095 || w == nwg - 1) && X.nsuper > 1)      // the last workgroup of a super-group: its total = the groups in front inside it + this group
      xg_put3(X.slotS + ((size_t)p * X.nsuper + (w >> 12)) * 8, lane, (X.epoch << 4) | (unsigned)p, xgr[0] + xin[0] + A0, xgr[1] + xin[1] + A1, xgr[2] + xin[2] + A2);
    const double b0 = xsu[0] + xgr[0] + xin[0], b1 = xsu[1] + xgr[1] + xin[1], b2 = xsu[2] + xgr[2] + xin[2];
    pv0 = st0; pv1 = st1; pv2 = st2;
    st0 = (float)(b0 + e0); st1 = (float)(b1 + e1); st2 = (float)(b2 + e2);
    __syncthreads();          // (wtot / xin / xgr are rewritten by the next pass)
  }
}
// closes a fused launch: the workgroups' aggregates of the last pass, in ticket order -> the chain totals; and, per workgroup, the start its first segment
// WOULD take next (the sum of the aggregates in front of it) against the one it ran from in the last pass and in the pass before:
//   stats[0] workgroups whose start still moves, stats[1 + c] the largest such move of chain c, stats[9 + c] the largest move one pass earlier (float bits;
//   0 when that pass started from zeros), stats[4 + c] the chain totals (float bits)
// A grid of workgroups of 1024 records each, one record per thread (coalesced component-major reads; one workgroup walking 12k records needed 57 us on its
// one CU's address unit): a workgroup first sums the aggregates in FRONT of its records -- thread-strided, then a fixed tree -- which makes the
// few workgroups independent of each other; stats[0..3], [9..11] are combined with atomics (zeroed by the launcher's memset).
__global__ __launch_bounds__(1024) void k_seq_objf_close(const double* __restrict__ wgagg, int nwg, float* __restrict__ total, unsigned* __restrict__ stats) {
  __shared__ double part[3][1024];
  __shared__ double wsum[3][16];
  __shared__ unsigned smm[16]; __shared__ float sx[6][16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const size_t nw_ = (size_t)nwg;
  const int lo = blockIdx.x * 1024, i = lo + t;
  const bool last_wg = blockIdx.x == gridDim.x - 1;
  double f0 = 0.0, f1 = 0.0, f2 = 0.0;                  // records in front of this workgroup's (the last workgroup: ALL the others -> the totals)
  for (int j = t; j < lo; j += 1024) { f0 += wgagg[j]; f1 += wgagg[nw_ + j]; f2 += wgagg[2 * nw_ + j]; }
  part[0][t] = f0; part[1][t] = f1; part[2][t] = f2;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if (t < o) { part[0][t] += part[0][t + o]; part[1][t] += part[1][t + o]; part[2][t] += part[2][t + o]; } __syncthreads(); }
  const double base0 = part[0][0], base1 = part[1][0], base2 = part[2][0];
  const bool live = i < nwg;
  const double v0 = live ? wgagg[i] : 0.0, v1 = live ? wgagg[nw_ + i] : 0.0, v2 = live ? wgagg[2 * nw_ + i] : 0.0;
  const double i0 = wave_scan_d(v0), i1 = wave_scan_d(v1), i2 = wave_scan_d(v2);
  if (lane == 63) { wsum[0][wv] = i0; wsum[1][wv] = i1; wsum[2][wv] = i2; }
  __syncthreads();
  double w0 = 0.0, w1 = 0.0, w2 = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (int u = 0; u < 16; u++) { if (u == wv) { w0 = a0; w1 = a1; w2 = a2; } a0 += wsum[0][u]; a1 += wsum[1][u]; a2 += wsum[2][u]; }
  const double r[3] = {base0 + w0 + (i0 - v0), base1 + w1 + (i1 - v1), base2 + w2 + (i2 - v2)};      // the sum in front of record i
  unsigned mm = 0; float x[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (live) {
    const unsigned long long fl = (unsigned long long)__double_as_longlong(wgagg[6 * nw_ + i]);
    if (fl & 1ull) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const unsigned long long pk = (unsigned long long)__double_as_longlong(wgagg[(3 + c) * nw_ + i]);
        const float nn = (float)r[c], last = __uint_as_float((unsigned)(pk & 0xffffffffull)), prev = __uint_as_float((unsigned)(pk >> 32));
        if (__float_as_uint(nn) != __float_as_uint(last)) { mm++; x[c] = fabsf(nn - last); }
        if (fl & 2ull) x[3 + c] = fabsf(last - prev);
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { mm += __shfl_xor(mm, m, 64); for (int c = 0; c < 6; c++) x[c] = fmaxf(x[c], __shfl_xor(x[c], m, 64)); }
  if (lane == 0) { smm[wv] = mm; for (int c = 0; c < 6; c++) sx[c][wv] = x[c]; }
  __syncthreads();
  if (t == 0) {
    unsigned sm = 0; float y[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int u = 0; u < 16; u++) { sm += smm[u]; for (int c = 0; c < 6; c++) y[c] = fmaxf(y[c], sx[c][u]); }
    if (sm) atomicAdd(stats, sm);
    for (int c = 0; c < 3; c++) { if (y[c] > 0.f) atomicMax(stats + 1 + c, __float_as_uint(y[c])); if (y[3 + c] > 0.f) atomicMax(stats + 9 + c, __float_as_uint(y[3 + c])); }
    if (last_wg) {
      const float t0 = (float)(base0 + a0), t1 = (float)(base1 + a1), t2 = (float)(base2 + a2);
      total[0] = t0; total[1] = t1; total[2] = t2;
      stats[4] = __float_as_uint(t0); stats[5] = __float_as_uint(t1); stats[6] = __float_as_uint(t2);
    }
  }
}

// ---- scans: start[s] <- sum of (end - start) over the chain's segments before s ------------------------------------------------
// lanes = lane-chains (w), 16 waves split the chain's segments; the differences and their partial sums are fp64 operations on values
// that fp32 can hold: exact.  mismatch counts the (segment, lane-chain) pairs whose new start differs from the one the pass used.
__global__ __launch_bounds__(1024) void k_seq_scan(const SeqChain* __restrict__ chains, int chain0, int W, const float* start_in,
                                                   const float* __restrict__ end, float* start_out, float* __restrict__ total,
                                                   unsigned* __restrict__ mismatch, int zero_start, int reduce_only) {
  // reduce_only (round 5): the scan behind the LAST pass of a sum nobody iterates further -- only the chain totals are wanted: one sweep, no starts written
  __shared__ double tot[16][64];
  __shared__ float dmx[16][64], smx[16][64];
  const int lane = threadIdx.x & 63, v = threadIdx.x >> 6;
  const int chain = chain0 + blockIdx.x, w = blockIdx.y * 64 + lane, ws = min(w, W - 1);
  const SeqChain c = chains[chain];
  if (c.nseg == 0) { if (v == 15 && w < W) total[(size_t)chain * W + w] = 0.0f; return; }      // (an empty chain: a level without cells in this block)
  const int per = (c.nseg + 15) / 16;
  const int s0 = c.seg0 + min(v * per, c.nseg), s1 = c.seg0 + min((v + 1) * per, c.nseg);
  unsigned mm = 0; float dmax = 0.0f, smax = 0.0f;       // how far this lane-chain's starts moved, against the largest start of the chain
  double run = 0.0;
  constexpr int PC = 28;
  if (per <= PC) {
    // short chains (a block's O / E sums: 391 segments): a wave's whole share of the chain is loaded at once and stays in registers -- ONE
    // memory round trip per scan (round 3: two sweeps of eight loads each: 18 us for a block, as long as the pass it followed)
    float e8[PC], o8[PC];
    const int slast = c.seg0 + c.nseg - 1;              // (unconditional loads from clamped, always valid segments; masked by s0 + u < s1 below)
#pragma unroll
    for (int u = 0; u < PC; u++) {
      const size_t i = (size_t)min(s0 + u, slast) * W + ws;
      e8[u] = end[i]; o8[u] = zero_start ? 0.0f : start_in[i];
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < PC; u++) if (s0 + u < s1) acc += (double)e8[u] - (double)o8[u];
    tot[v][lane] = acc;
    __syncthreads();
    for (int u = 0; u < v; u++) run += tot[u][lane];
    if (reduce_only) { if (v == 15 && w < W) total[(size_t)chain * W + w] = (float)(run + acc); return; }
#pragma unroll
    for (int u = 0; u < PC; u++) {
      if (s0 + u < s1) {
        const float ns = (float)run;
        if (w < W) {
          if (!zero_start && __float_as_uint(ns) != __float_as_uint(o8[u])) { mm++; dmax = fmaxf(dmax, fabsf(ns - o8[u])); }
          smax = fmaxf(smax, fabsf(ns));
          start_out[(size_t)(s0 + u) * W + w] = ns;
        }
        run += (double)e8[u] - (double)o8[u];
      }
    }
  } else {
  // (sixteen segments per step: the loads of a step are independent of each other and in flight together -- a step costs one memory
  //  latency, not sixteen)
  double acc = 0.0;
  for (int sb = s0; sb < s1; sb += 16) {
    float e8[16], o8[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const size_t i = (size_t)min(sb + u, s1 - 1) * W + ws;
      e8[u] = end[i]; o8[u] = zero_start ? 0.0f : start_in[i];
    }
#pragma unroll
    for (int u = 0; u < 16; u++) if (sb + u < s1) acc += (double)e8[u] - (double)o8[u];
  }
  tot[v][lane] = acc;
  __syncthreads();
  for (int u = 0; u < v; u++) run += tot[u][lane];
  if (reduce_only) { if (v == 15 && w < W) total[(size_t)chain * W + w] = (float)(run + acc); return; }
  for (int sb = s0; sb < s1; sb += 16) {
    float e8[16], o8[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const size_t i = (size_t)min(sb + u, s1 - 1) * W + ws;
      e8[u] = end[i]; o8[u] = zero_start ? 0.0f : start_in[i];
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (sb + u < s1) {
        const float ns = (float)run;
        if (w < W) {
          if (!zero_start && __float_as_uint(ns) != __float_as_uint(o8[u])) { mm++; dmax = fmaxf(dmax, fabsf(ns - o8[u])); }
          smax = fmaxf(smax, fabsf(ns));
          start_out[(size_t)(sb + u) * W + w] = ns;
        }
        run += (double)e8[u] - (double)o8[u];
      }
    }
  }
  }
  if (v == 15 && w < W) total[(size_t)chain * W + w] = (float)run;     // (empty chunks: run = the sum of all chunks before)
  if (mismatch && !zero_start) {      // [0] segments whose start moved in this scan, [1] the largest move of a start relative to its chain's largest start (float bits)
    dmx[v][lane] = dmax; smx[v][lane] = fmaxf(smax, fabsf((float)run));
    __syncthreads();
    // (relative to the largest start among the 64 lane-chains of this block -- the 64 clusters of a level row, or the PCs of one
    //  cluster together with its mass: a chain that hovers around zero next to chains of size 1e4 has not "moved by 100 %")
    float dd = 0.0f, ss = 0.0f;
    if (v == 0 && w < W) for (int u = 0; u < 16; u++) { dd = fmaxf(dd, dmx[u][lane]); ss = fmaxf(ss, smx[u][lane]); }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { mm += __shfl_xor(mm, m, 64); dd = fmaxf(dd, __shfl_xor(dd, m, 64)); ss = fmaxf(ss, __shfl_xor(ss, m, 64)); }
    if (lane == 0 && mm) atomicAdd(mismatch, mm);
    if (lane == 0 && v == 0 && ss > 0.0f && dd > 0.0f) atomicMax(mismatch + 1, __float_as_uint(dd / ss));
  }
}
// one lane-chain per chain (the objective's arrays): a workgroup per 256 segments (the pass's own workgroups: blockIdx.x alike), its base =
// the partial sums the pass left for the workgroups in front of it.  Sums and differences are fp64 operations on values fp32 can hold.
__global__ __launch_bounds__(256) void k_seq_scan1(int nsegs, const float* start_in, const float* __restrict__ end, float* start_out,
                                                   float* __restrict__ total, unsigned* __restrict__ mismatch, int zero_start,
                                                   const double* __restrict__ partial) {
  __shared__ double sh[256];
  __shared__ float dm1[256], sm1[256];
  const int t = threadIdx.x, nblk = gridDim.x, bx = blockIdx.x;
  const size_t base = (size_t)blockIdx.y * nsegs;
  const double* __restrict__ pp = partial + (size_t)blockIdx.y * nblk;
  double before = 0.0, all = 0.0;            // partials in front of this workgroup / of the whole array, added in workgroup order per thread, then a fixed tree
  for (int b = t; b < nblk; b += 256) { const double v = pp[b]; all += v; before += (b < bx) ? v : 0.0; }
  const int seg = bx * 256 + t;
  const bool live = seg < nsegs;
  const float e = live ? end[base + seg] : 0.0f, o = (live && !zero_start) ? start_in[base + seg] : 0.0f;
  const double delta = live ? (double)e - (double)o : 0.0;
  sh[t] = before; __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (t < off) sh[t] += sh[t + off]; __syncthreads(); }
  const double b0 = sh[0]; __syncthreads();
  sh[t] = all; __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (t < off) sh[t] += sh[t + off]; __syncthreads(); }
  const double tot = sh[0]; __syncthreads();
  sh[t] = delta; __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const double vv = (t >= off) ? sh[t - off] : 0.0;
    __syncthreads();
    sh[t] += vv;
    __syncthreads();
  }
  const double run = b0 + sh[t] - delta;     // exclusive prefix: the accumulator value this segment starts from
  unsigned mm = 0; float dmax = 0.0f, smax = 0.0f;
  if (live) {
    const float ns = (float)run;
    if (!zero_start && __float_as_uint(ns) != __float_as_uint(o)) { mm = 1; dmax = fabsf(ns - o); }
    smax = fabsf(ns);
    start_out[base + seg] = ns;
  }
  if (bx == 0 && t == 0) total[blockIdx.y] = (float)tot;
  if (mismatch && !zero_start) {
    dm1[t] = dmax; sm1[t] = smax; sh[t] = (double)mm;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (t < off) { dm1[t] = fmaxf(dm1[t], dm1[t + off]); sm1[t] = fmaxf(sm1[t], sm1[t + off]); sh[t] += sh[t + off]; }
      __syncthreads();
    }
    // (relative to the largest start of the whole chain: for these one-signed sums that is the total, which every workgroup knows)
    const float sref = fmaxf(sm1[0], fabsf((float)tot));
    if (t == 0 && sh[0] > 0.0) atomicAdd(mismatch, (unsigned)sh[0]);
    if (t == 0 && sref > 0.0f && dm1[0] > 0.0f) atomicMax(mismatch + 1, __float_as_uint(dm1[0] / sref));
  }
}

// ---- O / E tables in the reference's fp32 arithmetic (oe_arith) ----------------------------------------------------------------
// tot_*: [(1 + B)][K] chain totals of one block (or of the head): row 0 = sum(Rcells, 1), row 1 + b = Rcells * Phi_tcells column b.
//   head      E = rs Pr_b^T, O = tmp                                              (src/harmony.cpp:149-150, 226-227; tot_sub holds the sums)
//   tot_add   put back: E += rs Pr_b^T, O += tmp   (the block just updated)         (:329-330)
//   tot_sub   remove:   E -= rs Pr_b^T, O -= tmp, then this block's penalty table   (:312-313, :322)
// One launch does "put block j-1 back, take block j out" (round 3: two launches per block step) -- the same two roundings per entry in the
// same order.
__global__ void k_oe_fold(float* __restrict__ Of, float* __restrict__ Ef, const float* __restrict__ tot_add, const float* __restrict__ tot_sub,
                          const float* __restrict__ Pr_b, const float* __restrict__ theta, float* __restrict__ pen, int B, int K, int head) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * K) return;
  const int b = i / K, k = i - b * K;
  float e, o;
  if (head) { e = __fmul_rn(tot_sub[k], Pr_b[b]); o = tot_sub[(size_t)(1 + b) * K + k]; }
  else {
    e = Ef[i]; o = Of[i];
    if (tot_add) { e = __fadd_rn(e, __fmul_rn(tot_add[k], Pr_b[b])); o = __fadd_rn(o, tot_add[(size_t)(1 + b) * K + k]); }
    if (tot_sub) { e = __fsub_rn(e, __fmul_rn(tot_sub[k], Pr_b[b])); o = __fsub_rn(o, tot_sub[(size_t)(1 + b) * K + k]); }
  }
  Ef[i] = e; Of[i] = o;
  if (!head && tot_sub && pen) {
    const float num = __fadd_rn(__fmul_rn(2.0f, e), 1.0f), den = __fadd_rn(__fadd_rn(o, e), 1.0f);
    pen[i] = powf(num / den, theta[b]);      // harmony_pow(((2E+1)/(O+E+1)), theta) :322 (not on any critical path here: the accurate powf)
  }
}

// ---- objective terms (obj_arith): the three K x N matrices my_accu walks, in the reference's memory order ------------------------
// T[0] = R % dist_mat, T[1] = safe_entropy(R).each_col() % sigma, T[2] = (R.each_col() % sigma) % ((theta^T (x) 1) % log((O+E+1)/(2E+1))) * Phi
// (src/harmony.cpp:160-162), each stored cell-major in ORIGINAL cell order, k fastest -- the order of arma's column-major K x N memory.
// One wave per cell, lane = cluster (k, k + 64, ...); distances recomputed from the embedding row (cluster-lane dot products, Y in LDS).
// M: [B][K] table theta_b * log((O+E+1)/(2E+1)), fp32.
__global__ __launch_bounds__(256) void k_obj_terms(Dev D, const float* __restrict__ M, float* __restrict__ T, long long stride) {
  extern __shared__ __attribute__((aligned(16))) float ldsY[];      // [d][KP]
  const int d = D.d, K = D.K, KP = D.KP, C = D.C;
  for (int i = threadIdx.x; i < d * KP; i += blockDim.x) {
    const int j = i / KP, k = i - j * KP;
    ldsY[i] = (k < K) ? D.Yt[(size_t)j * K + k] : 0.0f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int cell = wave; cell < D.n; cell += nw) {          // internal order; the row goes to its original position
    const int orig = D.perm[cell], q = D.combo[cell];
    const float* __restrict__ z = D.Zc + (size_t)cell * D.zs;
    const float za = z[min(lane, d - 1)], zb = z[min(lane + 64, d - 1)];
    for (int kq = 0; kq < KP; kq += 64) {
      const int k = kq + lane;
      float dot = 0.0f;
      for (int j = 0; j < d; j++) {
        const float zj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(j < 64 ? za : zb), j & 63));
        dot = fmaf(zj, ldsY[j * KP + k], dot);
      }
      if (k < K) {
        const float dist = 2.0f * (1.0f - dot);
        const float r = D.R[(size_t)cell * K + k], sg = D.sigma[k];
        float m = 0.0f;
        for (int c = 0; c < C; c++) m = __fadd_rn(m, M[(size_t)D.qlev[q * C + c] * K + k]);
        const float lg = (r > 0.0f) ? logf(r) : logf(FLT_MIN);      // arma::trunc_log
        const size_t o = (size_t)orig * K + k;
        T[o] = __fmul_rn(r, dist);
        T[o + (size_t)stride] = __fmul_rn(__fmul_rn(r, lg), sg);
        T[o + 2 * (size_t)stride] = __fmul_rn(__fmul_rn(r, sg), m);
      }
    }
  }
}
// M[b][k] = theta_b * log((O + E + 1) / (2E + 1))   (:162), from the fp32 tables (oe_arith) or from the exact fixed-point O
__global__ void k_obj_mtable(Dev D, const float* __restrict__ Of, const float* __restrict__ Ef, float* __restrict__ M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.B * D.K) return;
  const int b = i / D.K, k = i - b * D.K;
  float o, e;
  if (Of) { o = Of[i]; e = Ef[i]; }
  else {
    long long rs = 0;
    for (int b0 = 0; b0 < D.B0; b0++) rs += D.O_fx[(size_t)b0 * D.K + k];
    o = (float)((double)D.O_fx[i] * FX_INV); e = (float)(((double)rs * FX_INV) * (double)D.Pr_b[b]);
  }
  M[i] = D.theta[b] * logf((o + e + 1.0f) / ((2.0f * e) + 1.0f));
}
// the three totals -> the objective snapshot obj[2..4]
__global__ void k_obj_store(const float* __restrict__ total, double* __restrict__ obj, const unsigned* __restrict__ xerr) {
  if (threadIdx.x == 0) { obj[2] = (double)total[0]; obj[3] = (double)total[1]; obj[4] = (double)total[2];      // (obj[5]: the error word of in-launch exchanges)
                          if (xerr && *xerr) obj[5] = (double)(*xerr & 15u); }
}
// cross-entropy term from the fp32 tables when only the tables follow the reference (oe_arith without obj_arith): obj[0..1] hold the
// exact per-cell sums, the cross term is sum_kb sigma_k M[b][k] O[b][k]
__global__ __launch_bounds__(256) void k_obj_cross_f32(Dev D, const float* __restrict__ Of, const float* __restrict__ M) {
  __shared__ double red[256];
  double cross = 0.0;
  for (int i = threadIdx.x; i < D.B * D.K; i += 256) cross += (double)D.sigma[i % D.K] * (double)M[i] * (double)Of[i];
  red[threadIdx.x] = cross;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) { D.obj[2] = D.obj[0]; D.obj[3] = D.obj[1]; D.obj[4] = red[0]; }
}

// ---- ridge statistics: which combinations enter cluster k's regression, and the hand-over of the chain totals ---------------------
// inset[q][k] = 1 if a cell of combination q enters the regression of cluster k: one of its levels is kept, i.e. O[k,b] / N_b > cutoff
// and its covariate has at least two such levels (src/harmony.cpp:368-402).  One workgroup per cluster.
__global__ __launch_bounds__(256) void k_seq_inset(Dev D, const float* __restrict__ Of, const int* __restrict__ cov_bounds, float cutoff,
                                                   unsigned char* __restrict__ inset, int KP8) {
  extern __shared__ int sm[];      // [B] ok, [B] keep, [C] levels per covariate
  const int k = blockIdx.x, B = D.B, C = D.C, K = D.K, Q = D.Q;
  int* ok = sm; int* keep = sm + B; int* lev = sm + 2 * B;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float o = Of ? Of[(size_t)b * K + k] : (float)((double)D.O_fx[(size_t)b * K + k] * FX_INV);
    ok[b] = (o / D.sizes[b]) > cutoff ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int c = 0; c < C; c++) lev[c] = 0;
    for (int b = 0, cv = 0; b < B; b++) { if (!(b < cov_bounds[cv])) cv++; if (ok[b]) lev[cv]++; }
    for (int b = 0, cv = 0; b < B; b++) { if (cv < C && !(b < cov_bounds[cv])) cv++; keep[b] = (ok[b] && lev[cv] > 1) ? 1 : 0; }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < Q; q += blockDim.x) {
    int in = 0;
    for (int c = 0; c < C; c++) in |= keep[D.qlev[q * C + c]];
    inset[(size_t)q * KP8 + k] = (unsigned char)in;
    if (k == K - 1) for (int kp = K; kp < KP8; kp++) inset[(size_t)q * KP8 + kp] = 1;      // (the pad columns of the last 8-cluster group: defined, never used)
  }
}
// chain totals [1 + Q][K][64] -> S0 / n0 (chain 0: the intercept row) and Sq / nq (chain 1 + q: combination = level q's row)
__global__ void k_seq_ridge_store(Dev D, const float* __restrict__ total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)D.K * 64;
  if (i >= (size_t)(1 + D.Q) * per) return;
  const int chain = (int)(i / per), k = (int)((i - (size_t)chain * per) >> 6), j = (int)(i & 63);
  const double v = (double)total[i];
  if (j < D.d) { if (chain == 0) D.S0[(size_t)k * D.d + j] = v; else D.Sq[((size_t)(chain - 1) * D.K + k) * D.d + j] = v; }
  else if (j == 63) { if (chain == 0) D.n0[k] = v; else D.nq[(size_t)(chain - 1) * D.K + k] = v; }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------------
void l_seq_oe_pass(const Launch& L, const Dev& D, const int* list, const int* poslev, int nlist, const SeqSeg* segs, int seg0, int nsegs, const float* start,
                   float* end, int zero_start, unsigned* conv_zero) {
  if (nsegs <= 0) { if (conv_zero) (void)hipMemsetAsync(conv_zero, 0, 2 * sizeof(unsigned), L.stream); return; }      // (no pass: the statistics words its scan adds to are still zeroed)
  if (D.B <= 32 && D.C <= 4) {             // level rows in registers, a batch taken level by level (round 6)
    const dim3 grid((nsegs + 3) / 4, (D.K + 63) / 64);
    if (D.B <= 16) hipLaunchKernelGGL((k_seq_oe_pass_bl<16>), grid, dim3(256), 0, L.stream, D.R, D.K, D.B, D.C, list, poslev, nlist, D.combo, D.qlev, segs, seg0, nsegs, start, end, zero_start, conv_zero);
    else hipLaunchKernelGGL((k_seq_oe_pass_bl<32>), grid, dim3(256), 0, L.stream, D.R, D.K, D.B, D.C, list, poslev, nlist, D.combo, D.qlev, segs, seg0, nsegs, start, end, zero_start, conv_zero);
    return;
  }
  if (D.B <= 32) {                         // level rows in registers, cell by cell (more than four covariates; more levels: in LDS, the round-3 form)
    const dim3 grid((nsegs + 3) / 4, (D.K + 63) / 64);
    if (D.B <= 16) hipLaunchKernelGGL((k_seq_oe_pass<true, 16>), grid, dim3(256), 0, L.stream, D.R, D.K, D.B, D.C, list, poslev, nlist, D.combo, D.qlev, segs, seg0, nsegs, start, end, zero_start, conv_zero);
    else hipLaunchKernelGGL((k_seq_oe_pass<true, 32>), grid, dim3(256), 0, L.stream, D.R, D.K, D.B, D.C, list, poslev, nlist, D.combo, D.qlev, segs, seg0, nsegs, start, end, zero_start, conv_zero);
    return;
  }
  if (D.B <= 256) {                        // level groups of 32, rows in registers (round 6; B <= 8 x 32: one wave per group)
    int4 cb = {D.B, D.B, D.B, D.B};                 // cumulative level counts of the first four covariates (from the combinations' level codes: D.cov_end)
    cb.x = D.cov_end[0]; cb.y = D.cov_end[1]; cb.z = D.cov_end[2]; cb.w = D.cov_end[3];
    hipLaunchKernelGGL(k_seq_oe_pass_lg, dim3(nsegs, (D.K + 63) / 64), dim3(64 * ((D.B + 31) / 32)), 0, L.stream, D.R, D.K, D.B, D.C, list, poslev, nlist, D.combo, D.qlev, segs, seg0,
                       nsegs, cb, start, end, zero_start, conv_zero);
    return;
  }
  int wpb = 4;                                           // waves per workgroup, limited by the level rows in LDS
  while (wpb > 1 && (size_t)wpb * D.B * 256 > 60 * 1024) wpb >>= 1;
  hipLaunchKernelGGL((k_seq_oe_pass<true, 0>), dim3((nsegs + wpb - 1) / wpb, (D.K + 63) / 64), dim3(64 * wpb), (size_t)wpb * D.B * 256, L.stream, D.R, D.K, D.B,
                     D.C, list, poslev, nlist, D.combo, D.qlev, segs, seg0, nsegs, start, end, zero_start, conv_zero);
}
// plain list sums: W = K lane-chains per segment (row 0 of the kernel above only)
void l_seq_sum_pass(const Launch& L, const Dev& D, const int* list, const SeqSeg* segs, int seg0, int nsegs, const float* start, float* end,
                    int zero_start, unsigned* conv_zero) {
  if (nsegs <= 0) { if (conv_zero) (void)hipMemsetAsync(conv_zero, 0, 2 * sizeof(unsigned), L.stream); return; }      // (no pass: the statistics words its scan adds to are still zeroed)
  hipLaunchKernelGGL((k_seq_oe_pass<false, 0>), dim3((nsegs + 3) / 4, (D.K + 63) / 64), dim3(256), 0, L.stream, D.R, D.K, 0, 0, list, nullptr, 0, D.combo, D.qlev, segs,
                     seg0, nsegs, start, end, zero_start, conv_zero);
}
void l_seq_ridge_pass(const Launch& L, const Dev& D, const int* list, const int* listq, const SeqSeg* segs, int seg0, int nsegs, const unsigned char* inset,
                      const float* start, float* end, int zero_start, unsigned* conv_zero) {
  if (nsegs <= 0) { if (conv_zero) (void)hipMemsetAsync(conv_zero, 0, 2 * sizeof(unsigned), L.stream); return; }      // (no pass: the statistics words its scan adds to are still zeroed)
  // one wave per 8 clusters.  Round 5: at most TWELVE waves per workgroup -- at 80 VGPRs a CU holds 24 waves, i.e. two 12-wave workgroups where a
  // 13-wave one (K = 100) left the rest of the CU empty; the clusters beyond 96 go to a second, small workgroup of the same segment (grid.y)
  // (measured at K = 100, 1M cells: ridge statistics 20.9 -> 19.7 ms per run)
  const int kg = (D.K + 7) / 8, wpg = kg < 12 ? kg : 12;
  hipLaunchKernelGGL(k_seq_ridge_pass<8>, dim3(nsegs, (kg + wpg - 1) / wpg), dim3(64 * wpg), 0, L.stream, D.R, D.Zo, D.combo, D.K, D.d, D.zs, (D.K + 7) / 8 * 8, list, listq, segs,
                     seg0, inset, start, end, zero_start, conv_zero);
}
// lane = cluster form (round 6): start / end hold [segment][d + 1][K]; false for embedding rows of more than 64 floats
bool l_seq_ridge_pass_kl(const Launch& L, const Dev& D, const int* list, const SeqSeg* segs, int seg0, int nsegs, const unsigned char* inset,
                         const float* start, float* end, int zero_start, unsigned* conv_zero) {
  if (D.zs > 64 || D.zs % 4 != 0 || D.d > D.zs) return false;
  if (nsegs <= 0) { if (conv_zero) (void)hipMemsetAsync(conv_zero, 0, 2 * sizeof(unsigned), L.stream); return true; }
  const int ngr = (D.K + 63) / 64, gpw = ngr >= 4 ? 4 : ngr >= 2 ? 2 : 1, spw = 4 / gpw;
  const dim3 grid((nsegs + spw - 1) / spw, (ngr + gpw - 1) / gpw);
  const int KP8 = (D.K + 7) / 8 * 8;
#define HMX_RPKL(Z4) case Z4: hipLaunchKernelGGL((k_seq_ridge_pass_kl<Z4>), grid, dim3(256), 0, L.stream, D.R, D.Zo, D.combo, D.K, D.d, KP8, list, segs, seg0, nsegs, inset, start, end, zero_start, conv_zero); break;
  switch (D.zs / 4) {
    HMX_RPKL(1) HMX_RPKL(2) HMX_RPKL(3) HMX_RPKL(4) HMX_RPKL(5) HMX_RPKL(6) HMX_RPKL(7) HMX_RPKL(8) HMX_RPKL(9) HMX_RPKL(10) HMX_RPKL(11) HMX_RPKL(12) HMX_RPKL(13) HMX_RPKL(14)
    HMX_RPKL(15) HMX_RPKL(16)
    default: return false;
  }
#undef HMX_RPKL
  return true;
}
void l_seq_ridge_rows2lanes(const Launch& L, const float* in, float* out, int nchains, int K, int d) {
  const size_t n = (size_t)nchains * K * 64;
  hipLaunchKernelGGL(k_seq_ridge_rows2lanes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, L.stream, in, out, nchains, K, d);
}
void l_seq_arr_pass(const Launch& L, const float* T, long long n, long long stride, int narr, int Lseg, int nsegs, const float* start, float* end,
                    int zero_start, double* partial, unsigned* conv_zero) {
  hipLaunchKernelGGL(k_seq_arr_pass, dim3((nsegs + 255) / 256, narr), dim3(256), 0, L.stream, T, n, stride, Lseg, nsegs, start, end, zero_start, partial, conv_zero);
}
void l_seq_scan(const Launch& L, const SeqChain* chains, int chain0, int nchains, int W, const float* start_in, const float* end, float* start_out,
                float* total, unsigned* mismatch, int zero_start, int reduce_only) {
  if (nchains <= 0) return;
  hipLaunchKernelGGL(k_seq_scan, dim3(nchains, (W + 63) / 64), dim3(1024), 0, L.stream, chains, chain0, W, start_in, end, start_out, total, mismatch,
                     zero_start, (reduce_only && !mismatch) ? 1 : 0);
}
void l_seq_scan1(const Launch& L, int narr, int nsegs, const float* start_in, const float* end, float* start_out, float* total,
                 unsigned* mismatch, int zero_start, const double* partial) {
  hipLaunchKernelGGL(k_seq_scan1, dim3((nsegs + 255) / 256, narr), dim3(256), 0, L.stream, nsegs, start_in, end, start_out, total, mismatch, zero_start, partial);
}
void l_oe_fold(const Launch& L, const Dev& D, float* Of, float* Ef, const float* tot_add, const float* tot_sub, float* pen, int head) {
  const int n = D.B * D.K;
  hipLaunchKernelGGL(k_oe_fold, dim3((n + 255) / 256), dim3(256), 0, L.stream, Of, Ef, tot_add, tot_sub, D.Pr_b, D.theta, pen, D.B, D.K, head);
}
int l_obj_terms(const Launch& L, const Dev& D, const float* Of, const float* Ef, float* M, float* T, long long stride, int dist_mode) {
  // dist_mode (round 6): 1 = T[0 .. stride) <- dist_mat itself (returns 2; 0 and nothing written where the MFMA kernel does not apply), 2 = the M table only
  // (the caller holds this call's dist_mat already: returns 2)
  const int n = D.B * D.K;
  hipLaunchKernelGGL(k_obj_mtable, dim3((n + 255) / 256), dim3(256), 0, L.stream, D, Of, Ef, M);
  if (dist_mode == 2) return 2;
  if (dist_mode == 1) return l_obj_terms_mfma(L, D, M, T, stride, 2) ? 2 : 0;
  if (l_obj_terms_mfma(L, D, M, T, stride, 0)) return 1;          // distances on the matrix cores, 16-byte rows (hmx_k_correct.inc): T[0] only
  int blocks = (D.n + 3) / 4; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_obj_terms, dim3(blocks), dim3(256), (size_t)D.d * D.KP * sizeof(float), L.stream, D, M, T, stride);
  return 3;
}
void l_seq_objr_pass(const Launch& L, const Dev& D, const float* M, long long nterms, int Lseg, int nsegs, const float* start, float* end, int zero_start, double* partial) {
  const size_t tab = ((size_t)D.B * D.K + D.K) * sizeof(float);
  if (tab <= 32 * 1024 && D.K % 4 == 0)            // (measured at K = 100, B = 10, 1M cells: objective 26.6 -> 23.9 ms per run)
    hipLaunchKernelGGL(k_seq_objr_pass<true>, dim3((nsegs + 255) / 256), dim3(256), tab, L.stream, D, M, nterms, Lseg, nsegs, start, end, zero_start, partial);
  else
    hipLaunchKernelGGL(k_seq_objr_pass<false>, dim3((nsegs + 255) / 256), dim3(256), 0, L.stream, D, M, nterms, Lseg, nsegs, start, end, zero_start, partial);
}
// the objective's three chains in one launch (k_seq_obj_fused).  mode 0: T = three term arrays at `stride`; mode 1: T = R % dist, the other two chains from R.
// slots: [OBJF_MAXSTAGE][nwg + ngroups][8] granules, zeroed once at allocation; `epoch` must differ from launch to launch.  Returns false when the shape is
// outside the kernel's envelope (K % 4, more than four covariates, more passes than slot sets): the caller falls back to the pass / scan launches.
size_t seq_obj_fused_slot_words(long long nt) {      // granule slots [OBJF_MAXSTAGE][nwg + ngroups][8] + the workgroups' closing records [nwg][8]
  const int nsegs = (int)((nt + OBJF_TPT - 1) / OBJF_TPT), nwg = (nsegs + OBJF_THREADS - 1) / OBJF_THREADS, ng = (nwg + 63) / 64;
  return (size_t)OBJF_MAXSTAGE * ((size_t)nwg + ng + (ng + 63) / 64) * 8 + (size_t)nwg * 8;
}
int seq_obj_fused_nsegs(long long nt) { return (int)((nt + OBJF_TPT - 1) / OBJF_TPT); }
bool l_seq_obj_fused(const Launch& L, const Dev& D, int mode, const float* T, long long stride, const float* M, const int* olev, long long nt, int npass, int zero_start,
                     float* starts, float* total, unsigned* stats, unsigned long long* slots, unsigned epoch) {
  if (npass < 1 || npass > OBJF_MAXSTAGE || nt < 4) return false;
  if (mode >= 1 && (D.K % 4 != 0 || D.C > 4 || !olev)) return false;      // (mode 2: T = dist_mat, see k_seq_obj_fused<., ., DIST>)
  const int nsegs = seq_obj_fused_nsegs(nt), nwg = (nsegs + OBJF_THREADS - 1) / OBJF_THREADS, ng = (nwg + 63) / 64;
  const int nsup = (ng + 63) / 64;
  SeqXchg X; X.slotA = slots; X.slotG = slots + (size_t)OBJF_MAXSTAGE * nwg * 8; X.slotS = X.slotG + (size_t)OBJF_MAXSTAGE * ng * 8; X.epoch = epoch & 0x0fffffffu; X.ngroups = ng; X.nsuper = nsup;
  double* const wgagg = reinterpret_cast<double*>(slots + (size_t)OBJF_MAXSTAGE * ((size_t)nwg + ng + nsup) * 8);
  (void)hipMemsetAsync(stats, 0, 16 * sizeof(unsigned), L.stream);
  const size_t deal = (size_t)OBJF_WAVES * 64 * OBJF_TPT * sizeof(float);
  if (mode == 0) hipLaunchKernelGGL((k_seq_obj_fused<0, false>), dim3(nwg), dim3(OBJF_THREADS), deal, L.stream, D, T, stride, M, olev, nt, nsegs, npass, zero_start, starts, stats, wgagg, X);
  else {
    const size_t tab = ((size_t)D.B * D.K + D.K) * sizeof(float);
    if (mode == 2) {
      if (tab <= 24 * 1024) hipLaunchKernelGGL((k_seq_obj_fused<1, true, true>), dim3(nwg), dim3(OBJF_THREADS), deal + tab, L.stream, D, T, stride, M, olev, nt, nsegs, npass, zero_start, starts, stats, wgagg, X);
      else hipLaunchKernelGGL((k_seq_obj_fused<1, false, true>), dim3(nwg), dim3(OBJF_THREADS), deal, L.stream, D, T, stride, M, olev, nt, nsegs, npass, zero_start, starts, stats, wgagg, X);
    } else
    if (tab <= 24 * 1024) hipLaunchKernelGGL((k_seq_obj_fused<1, true>), dim3(nwg), dim3(OBJF_THREADS), deal + tab, L.stream, D, T, stride, M, olev, nt, nsegs, npass, zero_start, starts, stats, wgagg, X);
    else hipLaunchKernelGGL((k_seq_obj_fused<1, false>), dim3(nwg), dim3(OBJF_THREADS), deal, L.stream, D, T, stride, M, olev, nt, nsegs, npass, zero_start, starts, stats, wgagg, X);
  }
  hipLaunchKernelGGL(k_seq_objf_close, dim3((nwg + 1023) / 1024), dim3(1024), 0, L.stream, wgagg, nwg, total, stats);
  return true;
}
void l_obj_store(const Launch& L, const float* total, double* obj, const unsigned* xerr) { hipLaunchKernelGGL(k_obj_store, dim3(1), dim3(64), 0, L.stream, total, obj, xerr); }
void l_obj_cross_f32(const Launch& L, const Dev& D, const float* Of, const float* Ef, float* M) {
  const int n = D.B * D.K;
  hipLaunchKernelGGL(k_obj_mtable, dim3((n + 255) / 256), dim3(256), 0, L.stream, D, Of, Ef, M);
  hipLaunchKernelGGL(k_obj_cross_f32, dim3(1), dim3(256), 0, L.stream, D, Of, M);
}
void l_seq_inset(const Launch& L, const Dev& D, const float* Of, const int* cov_bounds, float cutoff, unsigned char* inset) {
  hipLaunchKernelGGL(k_seq_inset, dim3(D.K), dim3(256), (size_t)(2 * D.B + D.C) * sizeof(int), L.stream, D, Of, cov_bounds, cutoff, inset, (D.K + 7) / 8 * 8);
}
void l_seq_ridge_store(const Launch& L, const Dev& D, const float* total) {
  const size_t n = (size_t)(1 + D.Q) * D.K * 64;
  hipLaunchKernelGGL(k_seq_ridge_store, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, L.stream, D, total);
}

}  // namespace hmx
