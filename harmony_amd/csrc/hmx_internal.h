// hmx_internal.h -- shared between the host orchestration (hmx_api.cpp) and the gfx950
// kernels (hmx_kernels.hip).  Not part of the public ABI (see include/harmony_mi355x.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hmx {

constexpr int WAVE = 64;
constexpr int TPB = 256;            // threads per workgroup (4 waves, one per SIMD)
constexpr int ITEM_CELLS = 256;     // cells per wave work item in streaming passes
constexpr int APPLY_CELLS = 1024;   // cells per workgroup work item in the apply pass
constexpr int SORT_CHUNK = 512;     // cells per wave in the per-round block counting sort
// R in [0,1] -> fixed point, exact int64 sums.  Round 5: 29 fractional bits (was 31) -- the four rows a lane holds of one column then add up in
// ONE 32-bit register (4 x 2^29 < 2^32) before a single widening add per column and tile; with 31 bits every value was zero-extended and added
// in 64 bits on its own (44 64-bit moves + 33 64-bit adds of the ~540 instructions of a tile's epilogue).  The quantum, 1.9e-9, is
// far below anything the tables are compared at (the sums are exact sums of the quantised values either way).
constexpr float FX_SCALE = 536870912.0f;
constexpr double FX_INV = 1.0 / 536870912.0;

// A run of cells (internal order) that share one covariate-level combination.
struct Item { int q; int start; int cnt; };

// Everything the kernels need; plain pointers, filled by the host side.
struct Dev {
  int n;            // local cells
  int d, K, B, C, Q;
  int B0;           // number of levels of covariate 0 (rowsum(R) = sum of its O columns)
  int cov_end[4];   // cumulative level counts of the first four covariates (B beyond the last one): which covariates a range of levels belongs to
  int upd_cpw;      // cells per wave in the update kernel
  int KP;           // K rounded up to a multiple of 64
  int zs;           // row stride of Zo/Zc in floats: d rounded up to a multiple of 4 (16-byte rows), pads are 0
  // MFMA tile path (16 cells x 16 clusters x 4 PCs per v_mfma_f32_16x16x4_f32)
  int NCT;          // cluster tiles of 16 (template instantiation, >= ceil(K/16))
  int NQ;           // quads of cluster tiles (ceil(NCT/4))
  int NT4, tail, NS;  // PC steps: NT4 float4 groups of 4 steps + tail single steps; NS = 4*NT4 + tail
  float* Yimg;      // [NQ][NS][4][16][4] LDS image of the centroids in MFMA B-operand order
  // split-bf16 form of the distance GEMM (v_mfma_f32_16x16x32_bf16, see bf3_split): NS2 steps of 32 PCs, the centroids as three
  // bf16 parts in B-operand order [NCT][NS2][part][lane = 16 g + c][8 bf16] (PC j: step j >> 5, lane group g = (j & 31) >> 3, slot j & 7)
  int NS2;
  unsigned short* Yimg3;
  int dot_bf;       // 1: tile kernels built for the split-bf16 form are launched where their LDS image fits (HMX_DOT=f32 turns it off)
  int upd_impl;     // 0: MFMA tile kernel, 1: cluster-lane VALU kernel (v1)
  int upd_tpw;      // tiles per wave target of the MFMA update kernel
  int upd_wps;      // waves per SIMD the update/head kernels are built for: 2 | 4 (lean: uniform sigma, K <= 64)
  int usig;         // all clusters share one sigma (scalar-constant kernel variants)
  int static_maxblocks; // cap of the static-tile launches (head / Lloyd / seeding); 0 = nwmax / waves per workgroup
  // seeding race (k_tile mode 3): hash key, first global cell of this shard, cells already chosen (re-probe passes)
  unsigned long long seed_key, seed_goff; const unsigned* seed_excl; int seed_nexcl;
  int oldsum_stream; // 1: k_oldsum_stream (sequential R pass, LDS accumulators) | 0: k_oldsum (gather through the sorted order)
  int upd_debug;    // diagnostics only (see k_tile)
  int upd_threads;   // workgroup size of the update kernel (256 or 512)
  int upd_maxblocks; // grid cap of the update kernel = workgroups resident at once (HMX_UPD_MAXBLOCKS)
  int nb;           // blocks per clustering round
  // cell data, internal (combo-sorted) order, cell-major rows
  float* Zo;        // [n][d]  Z_orig
  float* Zc;        // [n][d]  Z_corr (cosine-normalised while clustering)
  float* R;         // [n][K]
  int* perm;        // [n] internal -> local original index
  int* invperm;     // [n] local original -> internal
  int* combo;       // [n] combination id (non-decreasing)
  int* qlev;        // [Q][C] global level index of each covariate for combination q
  // small tables
  float* Yt;        // [d][K]   centroids, k fastest
  float* Ycur;      // [K][d]   centroids in the reference's layout (device copy used by the on-device Lloyd update)
  float* sigma;     // [K]
  float* ce;        // [K] -log2(e)/sigma_k
  float* cl;        // [K] sigma_k * ln 2
  float* theta;     // [B]
  float* Pr_b;      // [B]
  long long* O_fx;    // [B][K] fixed-point O (exact sum of quantised R)
  long long* Snew_fx; // [nrep][B][K] contribution of the block being updated
  long long* Sold_fx; // [nb][B][K] old contribution of every block of this round
  int fused_fold;      // 1: k_tile<.,0> rebuilds O' and the penalty table in its prologue (no k_foldpen launch per step)
  const long long* fold_prev;  // replica set written by the previous block update (read in the prologue)
  long long* fold_zero;        // replica set of the next block update (zeroed by workgroup 0)
  long long* Snew_set[3];      // the three rotating replica sets of the fused path
  long long* O_alt;   // ping-pong partners of O_fx / Snew_fx for the single-launch fold+penalty (host swaps)
  long long* Snew_alt;
  // persistent block chain (k_tile MODE 4): penalty granules, control block, tag of the round's first block
  unsigned long long* pen_g;   // [B][K] { tag << 32 | penalty bits }
  int* chain_ctl;              // [0] block flag (tag), [1] error, [2 + j] arrivals of block j
  unsigned chain_tag;
  unsigned chain_xseq;         // sharded chain: number of this round's first inbox exchange (plane = exchange number & 1, carried across rounds)
  int* tail_ticket;            // k_round_tail: workgroups done (the last one finishes the round's objective)
  // round tail inside the persistent chain (one GPU): the folder closes the round itself -- objective snapshot into the pinned host
  // slot, consumed tables cleared, control words reset -- instead of a k_round_tail launch behind every chain launch
  int chain_tail; double* tail_host_slot; long long* tail_z0; unsigned long long tail_n0; long long* tail_z1; unsigned long long tail_n1;
  int* solve_err;              // set by k_moe_solve when a ridge system is singular; rides to the host with the next objective snapshot
  int nxt;                     // this shuffle keys the cells by (block, block of the NEXT round): nb * nb sort keys, lpair.y carries the next block
  int* bincnt;                 // [keys * Q] cells of a (key, combination) bin before padding
  int* blkv;                   // [n] composite sort key of a cell (nxt)
  long long* Sold_next;        // [nb][B][K] old contributions of the NEXT round's blocks, filled by this round's tile kernels (or nullptr)
  int need_lorder;             // the first-generation kernels (k_update, k_oldsum's gather variant) read lorder / lcombo; the tile kernels read lpair only
  int upd_contig;              // k_tile MODE 0: a wave owns a contiguous range of the block's tiles (run-length flush) instead of every nw-th
  int qmask;                   // mask of the combination in a tile's combination word: 0x7FFFF if shuffles may be keyed by blocks, else 0x7FFFFFFF
  int head_gather;             // k_tile MODE 1 runs over the padded order of the upcoming round (lpair) instead of the static tiles
  long long* Sold_head;        // [nb][B][K] the head files its R sums here as the old contributions of that round's blocks (or nullptr)
  int head_norm;               // k_tile MODE 1: normalise the tile's Z_corr rows in registers and write them back (fused head of cluster_cpp)
  // 0: the block updates of this round do NOT write their R rows.  Inside a cluster_cpp call every round rewrites every cell's row, the
  // distances are recomputed from Z, and with the round-to-round carry (Sold_next) the next round takes its old contributions from the
  // sums this round files -- so the rows of every round but the call's last are never read by anything (src/harmony.cpp:285-339 shuffles
  // and un-shuffles R every round because its update reads R; ours does not).  20 MB per block step that never leave the registers.
  int r_store;
  int obj_stale;               // compute_objective on the stale snapshot (stale_dist): Yt / Zc are not the ones the MFMA images were built from
  int rvec;                    // K % 4 == 0: R rows are 16-byte aligned, the tile kernels store them with vector stores
  int chain_wps;               // waves per SIMD of the chain kernel: 2 (two accumulator sets)
  // wave-pair chain (k_tile MODE 6; 112 < K <= 224, BASELINE configs[4]: K = 200): the clusters are split in two halves [0, KH) and [KH, K), the two
  // waves of a pair run the K <= 112 worker on one half each and exchange the halves of a row's normalisation sum through LDS; the table is folded by
  // chain_folders workgroups (chain_kw clusters each), the penalty table is read per level from `pen` in memory (write-through stores, drained flag)
  int chain_pair, KH, chain_folders, chain_kw;
  unsigned short* Yimg3p;      // [2 halves][NCTP][NS2][part][lane][8 bf16]: the split-bf16 centroid image of each half in the layout of NCTP = ceil(KH / 16) cluster tiles
  // peer-to-peer block chain (sharded runs, one process per GPU on a node): p2p_inbox[g] = rank g's inbox as mapped into THIS
  // process (fine-grained device memory shared through HIP IPC), [2 parities][8 sources][P2P_CAP entries][2 granules]
  int p2p_world, p2p_rank;
  unsigned long long* p2p_inbox[8];
  __host__ __device__ unsigned long long* p2p_inbox_self() const {   // (static indices only: a dynamic one would spill the kernarg copy)
    unsigned long long* p = p2p_inbox[0];
#pragma unroll
    for (int q = 1; q < 8; q++) if (q == p2p_rank) p = p2p_inbox[q];
    return p;
  }
  unsigned long long* chain_dbg;   // diagnostics: accumulated 100 MHz ticks [0..2] folder wait / fold / publish, [3] launches,
                                   //              [4..8] worker (workgroup 0) flag wait / table copy / tiles / drain+arrive / next block's MFMAs
  float* pen;         // [B][K] ((2E+1)/(O+E+1))^theta
  double* obj;        // [0..1] reduced sums: sum R*dist, sum sigma*R*log R ; [2..4] snapshot incl. cross term
  double* objpart;    // [objslots][nwmax][2] per-(block,wave) partial sums: private slots, no atomics
  int objslots, nwmax;
  unsigned long long* trace;  // diagnostics (HMX_TRACE=1): [nwmax][16] per-wave phase stamps of the last k_tile<.,0> launch
  double* objrow;     // [objslots][2] per-slot-row sums
  int pen_lds;        // 1: the penalty table and qlev are staged in LDS by the update kernel
  int nrep;           // replicas of Snew_fx (power of two) to spread atomic contention
  // per-round block order
  int* blk;         // [n] block id of each internal cell
  int* lorder;      // [npad] internal cell ids grouped by block (stable); every (block, combination) bin is
                    //        padded to a multiple of 16 with -1 so that MFMA tiles are combination-pure
  int* lcombo;      // [npad] combination of position p (valid where lorder[p] >= 0)
  int2* lpair;      // [npad] (lorder, lcombo) interleaved: one 8-byte load per lane in the MFMA update kernel
  int npad;         // n + nb*Q*16 (upper bound of the padded length)
  int* boff;        // [nb+1] padded start of every block (multiples of 16)
  int* counts;      // [nb][nchunks] histogram of the counting sort
  int* offs;        // [nb][nchunks] destination offset of every (block, chunk)
  int* binoff;      // [nb*Q+1] padded start of every (block, combination) bin
  Item* schunks;    // [nchunks] static sort chunks: <= SORT_CHUNK cells of one combination
  int* qchunk;      // [Q+1] first sort chunk of every combination
  int nchunks;
  // static work lists
  Item* items; int nitems;        // <= ITEM_CELLS cells each
  Item* aitems; int naitems;      // <= APPLY_CELLS cells each
  Item* titems; int ntitems;      // <= 16 cells each: static MFMA tiles (head, Lloyd)
  int tile_impl;                  // 1: head / Lloyd run on the MFMA tile kernel, 0: cluster-lane VALU kernels
  // MoE
  double* Sq;       // [Q][K][d]  sum_i R_ki z_ij over cells of combination q
  double* nq;       // [Q][K]     sum_i R_ki
  // deterministic statistics pass (k_moe_stats_q): per-(workgroup, combination run) partial slots, reduced in fixed order
  int st_dma, st_nwg, st_cpw;       // enabled, workgroups, 16-cell tiles per workgroup
  int st_halves, st_KH;             // 2: K in (128, 224] -- the pass runs once per half of the clusters ([0, st_KH) / [st_KH, K), blockIdx.y), each with <= 8 cluster tiles
  double* st_part;                  // [slots][K*d + K]
  int* st_slot0;                    // [st_nwg] first slot of every workgroup
  int* st_qptr; int* st_qslots;     // CSR: slots of every combination in ascending (= cell) order
  double* S0;       // [K][d]     ridge_arith = 1: the intercept row's own sequential sum over all kept cells
  double* n0;       // [K]
  int* qstart;      // [Q+1] first internal cell of every combination
  float* sizes;     // [B] N_b
  float* Wq;        // [Q][K][d]  correction table
  float* Wimg;      // [Q][wNQ][wNS][4][16][4] the same table as MFMA B-operand image (clusters = reduction dim)
  int wNQ, wNT4, wtail, wNS;  // image geometry for k_moe_apply_mfma (valid when moe_mfma)
  int moe_mfma;     // 1: MFMA stats/apply kernels (needs K % 4 == 0, d <= 64, K <= 128)
  // kmeans init
  long long* km_gcells; double* km_rows; unsigned* km_excl;   // [K] chosen global cells, [K][d] their rows, [K] exclusion list
  unsigned long long* seedmin;  // [K] packed (key bits << 32 | global cell)
  long long* lsum;  // [K][d] 2^30 fixed-point sums of unit-vector components (exact, order-independent)
  int lloyd_lds;    // 1: the Lloyd sums are accumulated in an LDS table per workgroup first
  unsigned long long* lcnt;  // [K]
  float* ynorm;     // [K]
};

// arguments of the device-side ridge solve (k_moe_solve)
struct SolveArgs {
  double* cov; double* rhs;        // scratch: [K][M*M], [K][d*M]   (M = B + 1)
  float* Wall; int* mrows; int* flags;   // [K][d*M] W of every cluster (column-major m x d), its rows m, flags
  int* err;                              // device word: |= 1 when a system is singular
  const float* lambda;             // [B+1] fixed lambda or nullptr (estimation: alpha * E)
  const int* cov_bounds;           // [C] cumulative level counts
  float alpha, cutoff; int use_s0;
  const float* Of; const float* Ef;   // oe_arith: the fp32 O / E tables that replace the exact fixed-point O (keep rule, lambda = alpha E); else nullptr
  // ridge_arith with several covariates: the systems come straight from the sequential fp32 chain totals -- ref_tot [1 + B][K][64] (chain 0: all
  // kept cells, chain 1 + b: level b; lane j < d: sum fl(z_j R_k), lane 63: sum R_k), pair_tot [pairs][K], pair_idx [B][B] (b < b2) -> pair or -1
  const float* ref_tot; const float* pair_tot; const int* pair_idx;
  int solve_f32;                      // solve_arith: one covariate -> the reference's closed-form fp32 arrowhead inverse (src/harmony.cpp:575-586)
  size_t lds_b_bytes;              // LDS bytes available for the right-hand sides during the substitution (0: leave them in HBM)
  size_t lds_body_bytes;           // LDS bytes behind the index arrays (panel / right-hand sides; the combination-row table during the assembly)
  size_t lds_mask_off;             // byte offset of the coupling masks behind them (0: none -- dense Schur complement)
};

struct Launch {
  hipStream_t stream;
  int grid;  // workgroups for streaming kernels
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // l_update / l_chain: start / stop events attached to the launch itself (profile mode)
};

// ---- launchers (hmx_kernels.hip) -----------------------------------------------------
void l_convert_in(const Launch& L, const void* src, int f32, float* dst, const int* invperm, int n, int d, int zs);
void l_convert_out(const Launch& L, const float* src, void* dst, int f32, const int* invperm, int n, int w, int ws);
void l_copy(const Launch& L, const float* src, float* dst, size_t count);
void l_zero4(const Launch& L, void* a, size_t na, void* b, size_t nb, void* c, size_t nc, void* d, size_t nd);      // up to four buffers of 8-byte words, one launch
void l_normalize(const Launch& L, float* Z, int n, int d, int zs);
void l_normalize_from(const Launch& L, const float* src, float* dst, int n, int d, int zs);
// mode 0: head (write R, accumulate O_fx, objective partials); mode 1: objective only (read R)
void l_head(const Launch& L, const Dev& D, int mode);
void l_tile_static(const Launch& L, const Dev& D, int mode);
void l_sort_blocks(const Launch& L, const Dev& D, bool fused, uint64_t seed, uint64_t round, uint64_t Nglob, uint64_t goff,
                   uint64_t cells_per_block);
void l_sort_hist(const Launch& L, const Dev& D, bool fused, uint64_t seed, uint64_t round, uint64_t Nglob, uint64_t goff, uint64_t cells_per_block);
void l_sort_tail(const Launch& L, const Dev& D);
struct SortPtrs { int* blk; int* blkv; int* counts; int* offs; int* binoff; int* bincnt; int* boff; int* lorder; int* lcombo; int2* lpair; };
struct SortBatch { SortPtrs p[4]; };
struct ShufSets { int2* posr[4]; int2* lpair[4]; int* lorder[4]; int* lcombo[4]; int* boff[4]; int* partcnt[4]; int* binbase[4]; int* bincnt[4]; int* binacc[4]; };
int shuffle_parts(uint64_t Nglob, int nb, uint64_t cells_per_block);
void l_shuffle_blocks(const Launch& L, const Dev& D, uint64_t seed, uint64_t round, uint64_t Nglob, uint64_t goff, uint64_t cells_per_block);
void l_shuffle_inv(const Launch& L, const Dev& D, const ShufSets& T, int nr, uint64_t seed, uint64_t round, uint64_t Nglob, uint64_t goff,
                   uint64_t cells_per_block);
void l_sort_batch(const Launch& L, const Dev& D, const SortBatch& S, int nr, uint64_t seed, uint64_t round, uint64_t Nglob, uint64_t goff,
                  uint64_t cells_per_block);
void l_oldsum(const Launch& L, const Dev& D);
void l_fold(const Launch& L, const Dev& D, int j, int mode);
void l_penalty(const Launch& L, const Dev& D);
void l_foldpen(const Launch& L, const Dev& D, int j, const long long* Oin, long long* Oout, const long long* Sin,
               long long* Szero);
void l_obj_reduce(const Launch& L, const Dev& D);
void l_update(const Launch& L, const Dev& D, int j);
void l_chain(const Launch& L, const Dev& D, int workgroups);
// the same three launchers of the translation unit built with HMX_TILE_BF=1 (hmx_tile_bf.hip): k_tile with the split-bf16 distance GEMM
void l_tile_static_bf(const Launch& L, const Dev& D, int mode);
void l_update_bf(const Launch& L, const Dev& D, int j);
void l_chain_bf(const Launch& L, const Dev& D, int workgroups);
void l_round_tail(const Launch& L, const Dev& D, double* host_slot, long long* z0, size_t n0, long long* z1, size_t n1);
// Cluster <-> MFMA column mapping of the tile kernels.  Lane (g, c) of a wave holds column c of every 16-wide cluster tile ct.
// Clusters are dealt so that a lane's columns are CONSECUTIVE clusters: within a full quad of cluster tiles (4q..4q+3) the lane
// holds clusters 64q + 4c + {0,1,2,3} (one 16-byte store per R row instead of four 4-byte ones), within the remaining r = nct % 4
// tiles clusters 64 nfull + r c + {0..r-1} (one 12/8/4-byte store).  Ascending in ct for a fixed lane.
__host__ __device__ constexpr int kcol(int nct, int ct, int c) {
  return ct < 4 * (nct >> 2) ? 64 * (ct >> 2) + 4 * c + (ct & 3) : 64 * (nct >> 2) + (nct & 3) * c + (ct - 4 * (nct >> 2));
}
// inverse: cluster k -> (quad qd, component i of the quad's float4, column c) of the B-operand image
__host__ __device__ inline void kcol_inv(int nct, int k, int& qd, int& i, int& c) {
  const int nfull = nct >> 2, r = nct & 3;
  if (k < 64 * nfull) { qd = k >> 6; c = (k & 63) >> 2; i = k & 3; }
  else { qd = nfull; const int kk = k - 64 * nfull; c = kk / r; i = kk - c * r; }
}
// fp32 -> three bf16 parts by truncation: x = hi + mid + lo EXACTLY (8 + 8 + 8 mantissa bits; both differences are exact in fp32),
// so a product of two fp32 numbers is the sum of nine exact bf16 x bf16 products; the tile kernels keep the six largest
// (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi: what is dropped -- mid lo + lo mid + lo lo -- is below 2^-21 |x||y| at worst with this
// truncating split, |mid| < 2^-7 |x|, |lo| < 2^-15 |x|, and ~2^-24 typically: tests/test_abi_cpu.py) and add them in fp32 on the matrix cores.
__host__ __device__ inline void bf3_split(float x, unsigned short (&p)[3]) {
  union { float f; unsigned u; } a, b;
  a.f = x;
  p[0] = (unsigned short)(a.u >> 16);
  b.u = a.u & 0xffff0000u;
  a.f = x - b.f;
  p[1] = (unsigned short)(a.u >> 16);
  b.u = a.u & 0xffff0000u;
  a.f = a.f - b.f;
  p[2] = (unsigned short)(a.u >> 16);
}
// bf16 index of (PC j, cluster k, part) in the split image
__host__ __device__ inline size_t bfimg_index(int nct, int ns2, int j, int k, int part) {
  int qd, i, c;
  kcol_inv(nct, k, qd, i, c);
  const int ct = 4 * qd + i, s = j >> 5, g = (j & 31) >> 3;
  return ((((size_t)ct * ns2 + s) * 3 + part) * 64 + 16 * g + c) * 8 + (j & 7);
}
__host__ __device__ inline void bfimg_store(unsigned short* img, int nct, int ns2, int j, int k, float y) {
  unsigned short p[3];
  bf3_split(y, p);
  for (int part = 0; part < 3; part++) img[bfimg_index(nct, ns2, j, k, part)] = p[part];
}
// both split-bf16 images of the centroids (the second one only where the wave-pair chain runs)
__host__ __device__ inline void bfimg_store_all(const Dev& D, int j, int k, float y) {
  bfimg_store(D.Yimg3, D.NCT, D.NS2, j, k, y);
  if (D.Yimg3p) {
    const int nctp = (D.KH + 15) >> 4, h = k >= D.KH ? 1 : 0;
    bfimg_store(D.Yimg3p + (size_t)h * nctp * D.NS2 * 3 * 512, nctp, D.NS2, j, h ? k - D.KH : k, y);
  }
}
constexpr int P2P_CAP = 65536;                       // K x B entries an inbox holds per (plane, source): 200 clusters x 200 levels (BASELINE configs[4]) fit
// an inbox = [4 planes][8 sources][P2P_CAP entries][2 granules]: planes 0 / 1 = the block chain's exchanges (alternating by exchange
// number), planes 2 / 3 = the generic small all-reduces (alternating by call number); 64 granules behind them: the connection self-test
constexpr int P2P_PLANES = 4;
constexpr size_t P2P_TEST_BASE = (size_t)P2P_PLANES * 8 * P2P_CAP * 2;
constexpr size_t P2P_INBOX_GRANULES = P2P_TEST_BASE + 64;
// generic all-reduce of a small buffer through the inboxes (k_p2p_allreduce): dtype 0 int64 sum | 1 float64 sum (rank order: identical on
// every rank) | 2 int64 min; `seq` = the call's number (same on every rank: plane 2 + (seq & 1), tag 0x40000000 + seq)
void l_p2p_allreduce(const Launch& L, const Dev& D, void* buf, int n, int dtype, unsigned seq, int* err);
// the same for a window of up to (P2P_CAP / 2) * world entries as reduce-scatter + all-gather (big buffers: one call per window, own `seq` each)
void l_p2p_allreduce_big(const Launch& L, const Dev& D, void* buf, int n, int dtype, unsigned seq, int* err);
void l_p2p_selftest(const Launch& L, const Dev& D, unsigned tag, int* result);   // the whole block chain of a round: one persistent launch
void l_objective_tables(const Launch& L, const Dev& D);  // cross-entropy term only -> obj[4]
void l_moe_stats(const Launch& L, const Dev& D);
void l_moe_apply(const Launch& L, const Dev& D);
void l_moe_stats_mfma(const Launch& L, const Dev& D);
void l_moe_solve(const Launch& L, const Dev& D, const SolveArgs& A);
void l_moe_apply_mfma(const Launch& L, const Dev& D);
void l_seed_probe(const Launch& L, const Dev& D, uint64_t seed, uint64_t goff, const unsigned* excl, int nexcl);
void l_seed_race_u(const Launch& L, const Dev& D, const float* u, int a0, int na, int only, uint64_t goff, const unsigned* excl,
                   int nexcl);
void l_gather_rows(const Launch& L, const Dev& D, const long long* gcells, uint64_t goff, double* rows);
void l_lloyd(const Launch& L, const Dev& D);
void l_lloyd_finish(const Launch& L, const Dev& D);
void l_y_images(const Launch& L, const Dev& D, const double* rows, int normalise);
size_t lds_bytes_y(const Dev& D);

// ---- reference arithmetic: restarted sequential fp32 sums (hmx_seq.hip) ------------------------------------------------------------
struct SeqSeg { int off; int cnt; };       // a segment of a chain: cells list[off .. off + cnt) (or the cells off .. off + cnt - 1 themselves)
struct SeqChain { int seg0; int nseg; };   // the segments of one chain, in chain order
// (conv_zero: the two statistics words the scan behind this pass will add to -- zeroed by the pass itself, no memset launch; or nullptr)
void l_seq_oe_pass(const Launch& L, const Dev& D, const int* list, const int* poslev, int nlist, const SeqSeg* segs, int seg0, int nsegs, const float* start,
                   float* end, int zero_start, unsigned* conv_zero);
void l_seq_sum_pass(const Launch& L, const Dev& D, const int* list, const SeqSeg* segs, int seg0, int nsegs, const float* start, float* end,
                    int zero_start, unsigned* conv_zero);
void l_ref_posord(const Launch& L, const Dev& D, uint64_t seed, uint64_t round, uint64_t Nglob, int* posord, int* poslev);
// (listq: the combination of every entry of `list`, or nullptr -- the kernel then looks it up through D.combo, one more dependent load per batch)
void l_seq_ridge_pass(const Launch& L, const Dev& D, const int* list, const int* listq, const SeqSeg* segs, int seg0, int nsegs, const unsigned char* inset,
                      const float* start, float* end, int zero_start, unsigned* conv_zero);
bool l_seq_ridge_pass_kl(const Launch& L, const Dev& D, const int* list, const SeqSeg* segs, int seg0, int nsegs, const unsigned char* inset,
                         const float* start, float* end, int zero_start, unsigned* conv_zero);
void l_seq_ridge_rows2lanes(const Launch& L, const float* in, float* out, int nchains, int K, int d);
// (partial: [narr][ceil(nsegs / 256)] doubles, the deltas of every workgroup's 256 segments -- k_seq_scan1's bases)
void l_seq_arr_pass(const Launch& L, const float* T, long long n, long long stride, int narr, int Lseg, int nsegs, const float* start, float* end,
                    int zero_start, double* partial, unsigned* conv_zero);
void l_seq_scan(const Launch& L, const SeqChain* chains, int chain0, int nchains, int W, const float* start_in, const float* end, float* start_out,
                float* total, unsigned* mismatch, int zero_start, int reduce_only = 0);
void l_seq_scan1(const Launch& L, int narr, int nsegs, const float* start_in, const float* end, float* start_out, float* total,
                 unsigned* mismatch, int zero_start, const double* partial);
void l_oe_fold(const Launch& L, const Dev& D, float* Of, float* Ef, const float* tot_add, const float* tot_sub, float* pen, int head);
// returns the number of term arrays materialised in T: 1 (T[0] = R % dist only: the MFMA kernel; the passes recompute the other two from R,
// l_seq_objr_pass) or 3 (the cluster-lane fallback kernel: K % 4 != 0, rows of more than 67 PCs, the stale-distance snapshot)
int l_obj_terms(const Launch& L, const Dev& D, const float* Of, const float* Ef, float* M, float* T, long long stride, int dist_mode = 0);
// arrays 1 and 2 of the objective (entropy, cross-entropy) as sequential sums straight from R: same segments, starts / ends / partials as l_seq_arr_pass's arrays 1, 2
void l_seq_objr_pass(const Launch& L, const Dev& D, const float* M, long long nterms, int Lseg, int nsegs, const float* start, float* end, int zero_start, double* partial);
// round 6: the three chains in ONE launch, segments in registers, the scans between the passes inside the launch (k_seq_obj_fused, hmx_seq.hip)
size_t seq_obj_fused_slot_words(long long nt);
int seq_obj_fused_nsegs(long long nt);
bool l_seq_obj_fused(const Launch& L, const Dev& D, int mode, const float* T, long long stride, const float* M, const int* olev, long long nt, int npass, int zero_start,
                     float* starts, float* total, unsigned* stats, unsigned long long* slots, unsigned epoch);
void l_obj_store(const Launch& L, const float* total, double* obj, const unsigned* xerr = nullptr);      // (xerr: a fused launch's exchange error word -> obj[5])
bool l_obj_terms_mfma(const Launch& L, const Dev& D, const float* M, float* T, long long stride, int all3);
void l_obj_cross_f32(const Launch& L, const Dev& D, const float* Of, const float* Ef, float* M);
void l_seq_inset(const Launch& L, const Dev& D, const float* Of, const int* cov_bounds, float cutoff, unsigned char* inset);
void l_seq_ridge_store(const Launch& L, const Dev& D, const float* total);

}  // namespace hmx
