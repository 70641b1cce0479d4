/* harmony_mi355x_lab.h -- laboratory equipment of libharmony_mi355x.so
 *
 * NOT part of the reference's interface (include/harmony_mi355x.h is: one entry point per method / field of the Rcpp module,
 * /root/reference/src/harmony.cpp:672-709).  What is declared and documented here exists for this repository's tests, measurements and
 * fallbacks: host-side probes of the generators, device probes of the restarted sequential sums, the tuning / fallback selectors of
 * hmx_set_int and of the environment, the settings of the reference-arithmetic machinery.  A maintainer binding the library into the R
 * package needs none of it.
 */
#ifndef HARMONY_MI355X_LAB_H
#define HARMONY_MI355X_LAB_H

#include "harmony_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- further writable fields of hmx_set_int ----------------------------------------------------------------------------
 * (the public header lists "max_iter_kmeans", "seed", "device", "profile", "rng", "ref_arith", "stale_dist")
 * writable fields: "ridge_arith" / "oe_arith" / "obj_arith" / "solve_arith" (the accumulator groups of "ref_arith" one by one, before setup),
 * "seq_passes" / "seq_warm_passes" / "seq_tol_ppb" / "seq_strict" / "seq_stats" / "seq_max_passes" / "seq_fused" (see "reference arithmetic" below).
 * Tuning / fallback selectors (tests and measurements; the defaults are the measured best):
 *   fields "grid", "upd_wps" (before setup), "upd_tpw", "upd_cpw", "upd_impl", "comm_force";
 *   environment, read by hmx_setup: HMX_GRID, HMX_NREP, HMX_UPD_WPS (2|4), HMX_USIG=0 (general-sigma kernels),
 *   HMX_UPD_THREADS, HMX_UPD_MAXBLOCKS, HMX_STATIC_MAXBLOCKS, HMX_UPD_TPW, HMX_UPD_CPW, HMX_FUSED_FOLD=0, HMX_FOLD_IMPL=split,
 *   HMX_OLDSUM_IMPL=gather|stream1, HMX_UPDATE_IMPL=v1, HMX_TILE_IMPL=v1, HMX_MOE_IMPL=v1 (first-generation kernels),
 *   HMX_DOT=f32 (tile kernels: fp32-MFMA distance GEMM only; default: the split-bf16 build wherever its LDS image fits -- same
 *   fp32 accuracy);
 *   host matrices (hmx_setup / hmx_get_matrix with HMX_HOST): HMX_XFER=pin (register the caller's buffer instead of moving it
 *   through the process-wide ring of page-locked slots), HMX_XFER_THREADS (host threads that fill / drain the ring, default 8),
 *   HMX_PIN=0 (plain pageable copies).  INTEGRATION.md has the complete table of environment switches. */

/* probes of the R-compatible stream (host only, no device needed; used by the CPU tests) */
void hmx_r_runif(uint32_t seed, int32_t n, double* out);            /* set.seed(seed); runif(n)                      */
void hmx_r_shuffle(uint32_t seed, int64_t N, int64_t* out);         /* set.seed(seed); arma::shuffle(0..N-1)         */
void hmx_mt19937_by_array(const uint32_t* key, int32_t len, int32_t n, uint32_t* out);  /* MT19937 known-answer vector */
float hmx_u01(uint64_t seed, uint64_t stream, uint64_t idx);
/* probe (host only): which cluster MFMA column c of cluster tile ct holds when a launch uses nct cluster tiles (the tile kernels
 * deal a lane consecutive clusters so that its R values are adjacent in memory) */
int32_t hmx_cluster_of_column(int32_t nct, int32_t ct, int32_t c);

/* ---- reference arithmetic: groups, settings, probes ------------------------------------------------------
 * By default every cross-cell accumulator is exact (64-bit fixed point / fp64 in a fixed order).  The reference accumulates in
 * fp32, one term after the other; at 10^6 cells that is a visible, systematic bias (terms below half an ulp of a grown
 * accumulator are dropped).  Four switches (hmx_set_int, before hmx_setup; one GPU) make the library reproduce it, group by group:
 *   "ridge_arith"  Phi* diag(R_k) Phi*^T and Phi* diag(R_k) Z^T as sequential fp32 sums over the cells  (src/harmony.cpp:567,599-608)
 *   "oe_arith"     O, E as fp32 tables: block sums in the round's shuffled order, -= / += drift       (:149-150,312-313,329-330)
 *   "obj_arith"    compute_objective's three K*N-term my_accu sums                                       (src/utils.cpp:67-75)
 *   "solve_arith"  the closed-form fp32 arrowhead inverse of the one-covariate ridge system               (src/harmony.cpp:575-586)
 *   "ref_arith"    1: all four; 2: all but "oe_arith" (the groups that move the result; update_R in the persistent block chain).
 * The sequential sums are computed as RESTARTED sequential sums (segments in parallel, their starting values by fixed-point
 * iteration; at the fixed point the result is bit-identical to the one-after-the-other loop).  Settings (hmx_set_int):
 *   "seq_passes" passes of a sum that starts from zero (default 2), "seq_warm_passes" passes of a sum that starts from the starts of its
 *   last evaluation (default 2); long chains (>= 200k cells, the objective's K N terms) continue until the largest move of a start in the last
 *   scan is below "seq_tol_ppb" parts per billion of the largest start of its lane group (default 10000 = 1e-5; at most "seq_max_passes");
 *   ("seq_warm_passes" above "seq_passes" is clamped to it;)
 *   "seq_strict" = 1: EVERY sum is iterated until no start moves any more -- the bit-exact fixed point (slow: ~1.2 s per run at 1M cells).
 *   Measured at BASELINE configs[2] (profiles/r5_strict_probe_1M_*.json): the default, six passes and the strict fixed point all end
 *   1.9e-6 .. 2.1e-6 from the faithful oracle and as far from EACH OTHER -- the faithful fp32 trajectory itself moves by that much under any
 *   ulp-level change (profiles/r5_oracle_liberties.json).
 *   hmx_get "seq:mismatch" / "seq:residual" = how many segment starts still moved in the last scans and by how much, relatively (long chains
 *   always; the short per-block sums only with "seq_stats" = 1); "seq:group_passes" / "seq:group_runs" = passes / evaluations per group (O/E,
 *   objective, ridge, level pairs); "seq:unsettled" = sums that hit seq_max_passes.
 *   "seq_fused" (round 6, default 13 = all on): bit 0 the objective's three chains in ONE launch (k_seq_obj_fused: segments of 32 terms in registers, the
 *   passes and the scans between them inside the launch); bit 2 the ridge pass with lane = cluster (k_seq_ridge_pass_kl); bit 3 dist_mat of a cluster_cpp
 *   call computed by its first objective evaluation and kept for the others (R % dist formed inside the fused launch).  0: the round-5 kernels.
 * Probes of that machinery on caller-provided data (device needed; hmx_debug_seq_arr with seg_terms = 0 and three arrays: the one-launch form): */
int hmx_debug_seq_oe(const float* R, int64_t n, int32_t K, const int32_t* level, int32_t B, const int32_t* list, int64_t nlist,
                     const int32_t* chain_off, const int32_t* chain_cnt, int32_t nchains, int32_t seg_cells, int32_t passes, float* totals,
                     int64_t* mismatch, double* residual);
int hmx_debug_seq_arr(const float* T, int64_t n, int32_t narr, int32_t seg_terms, int32_t passes, float* total, int64_t* mismatch, double* residual);


#ifdef __cplusplus
}
#endif
#endif /* HARMONY_MI355X_LAB_H */
