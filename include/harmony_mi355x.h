/* harmony_mi355x.h -- C ABI of libharmony_mi355x.so
 *
 * MI355X-native (gfx950, hand-written HIP) replacement for the reference's Rcpp
 * module `harmony_module` (class_<harmony>, /root/reference/src/harmony.cpp:672-709;
 * instantiated by `new(harmony)` at R/ui.R:269 and driven by R/ui.R:271-295 and
 * R/utils.R:15-46).  One opaque handle == one `harmony` object (for multi-GPU: one
 * handle per process/GPU holding a contiguous shard of the cells).
 *
 * Conventions
 *   - all matrices cross the boundary exactly as the reference's do: double,
 *     column-major (Z is d x N, R is K x N, Y is d x K, O/E are K x B, W is (B+1) x d,
 *     Lambda is K x (B+1)); Phi is a dgCMatrix (CSC i/p/x), B x N, C ones per column,
 *     rows grouped by covariate in vars_use order (R/ui.R:210-213).
 *   - inputs are borrowed only for the duration of the call; outputs are written into
 *     caller-allocated buffers (reference: conv_to copies, src/harmony.cpp:41-45,640-655).
 *   - no exceptions cross the ABI: every call returns an int status
 *        0 ok | -1 aborted by the poll callback (reference: Progress::check_abort,
 *        src/harmony.cpp:233-234) | >0 error class (message: hmx_last_error).
 *   - state is device-resident between calls; calls are synchronous w.r.t. the host
 *     (results of getters are complete on return), re-entrant per handle, and there is no
 *     global state (several handles may coexist).
 *   - the library REQUIRES a HIP device (gfx950).  There is no CPU fallback: without a
 *     device hmx_setup fails with HMX_ERR_DEVICE.
 */
#ifndef HARMONY_MI355X_H
#define HARMONY_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hmx_ctx hmx_ctx;

enum {
  HMX_OK = 0,
  HMX_ABORTED = -1,
  HMX_ERR_ARG = 1,      /* bad argument / inconsistent shapes                                  */
  HMX_ERR_TOO_FEW = 2,  /* "Refusing to run with less than 6 cells" (src/harmony.cpp:83-85)    */
  HMX_ERR_PHI = 3,      /* Phi is not C-hot / rows not grouped by covariate                    */
  HMX_ERR_SOLVE = 4,    /* singular ridge system                                               */
  HMX_ERR_DEVICE = 5,   /* no HIP device / HIP runtime error                                   */
  HMX_ERR_STATE = 6,    /* method called before setup / init                                   */
  HMX_ERR_LIMIT = 7,    /* shape outside the supported envelope (d <= 128, K <= 256)           */
  HMX_ERR_COMM = 8      /* all-reduce callback failed                                          */
};

/* ---- life cycle ---------------------------------------------------------------------- */
/* new(harmony): constructor src/harmony.cpp:18-25, R/ui.R:269 */
hmx_ctx* hmx_create(void);
/* R external-pointer finalizer (the Rcpp module's XPtr deleter) */
void hmx_destroy(hmx_ctx* ctx);
const char* hmx_last_error(hmx_ctx* ctx);
/* last Rcpp::warning equivalent ("Too few cells. Setting block_size to 0.2", src/harmony.cpp:87); "" if none */
const char* hmx_last_warning(hmx_ctx* ctx);

/* ---- harmony::setup  (src/harmony.h:25-30, src/harmony.cpp:29-111, called R/ui.R:271-275)
 * Z: d x N doubles (this rank's cells); phi_i/phi_p/phi_x: CSC of the B x N one-hot design
 * (phi_x may be NULL = all ones); sigma[K], theta[B]; lambda: B+1 values or a single -1
 * (n_lambda == 1) meaning automatic estimation (R/ui.R:224-231, src/harmony.cpp:75-79). */
int hmx_setup(hmx_ctx* ctx, const double* Z, int64_t N, int32_t d,
              const int32_t* phi_i, const int32_t* phi_p, const double* phi_x, int32_t B,
              const double* sigma, const double* theta, const double* lambda, int32_t n_lambda,
              double alpha, int32_t max_iter_kmeans, double epsilon_kmeans, double epsilon_harmony,
              int32_t K, double block_size, const int32_t* B_vec, int32_t C,
              double batch_proportion_cutoff, int32_t verbose);

/* Device-side / single-precision ingest (the reference's seam is R/ui.R:178-183 -> conv_to<fp32>,
 * src/harmony.cpp:41).  Same as hmx_setup, but Z may be float (HMX_F32) and/or already live in HBM (HMX_DEVICE: a device
 * pointer valid on the handle's device; it is read once and may be freed after the call).  hmx_setup(...) ==
 * hmx_setup_ex(..., HMX_F64, HMX_HOST, ...).  Host input is streamed through two HBM staging slabs (copy of slab s+1
 * overlaps the conversion of slab s); hmx_get("timer:ingest_Z") reports the time of the last ingest. */
enum { HMX_F64 = 0, HMX_F32 = 1 };
enum { HMX_HOST = 0, HMX_DEVICE = 1 };
int hmx_setup_ex(hmx_ctx* ctx, const void* Z, int32_t z_dtype, int32_t z_location, int64_t N, int32_t d,
                 const int32_t* phi_i, const int32_t* phi_p, const double* phi_x, int32_t B,
                 const double* sigma, const double* theta, const double* lambda, int32_t n_lambda,
                 double alpha, int32_t max_iter_kmeans, double epsilon_kmeans, double epsilon_harmony,
                 int32_t K, double block_size, const int32_t* B_vec, int32_t C,
                 double batch_proportion_cutoff, int32_t verbose);

/* Return to the state right after hmx_setup without re-uploading the inputs (Z_corr <-
 * normalise(Z_orig), objective series cleared).  No reference counterpart; used by bench.py
 * so that every timed step starts from HBM-resident inputs. */
int hmx_restart(hmx_ctx* ctx);

/* ---- harmony::init_cluster_cpp (src/harmony.cpp:131-156, R/ui.R:281).
 * Y0 == NULL: centroids come from the on-device kmeans_centers (src/utils.cpp:10-64) using
 * the documented counter-based generator below and the handle's seed.  Y0 != NULL (d x K):
 * use these centroids instead (test hook: lets the oracle and the GPU share random choices). */
int hmx_init_cluster(hmx_ctx* ctx, const double* Y0);
/* kmeans_centers alone (.Call entry _harmony_kmeans_centers, src/RcppExports.cpp:14-25):
 * writes the d x K un-normalised centres of the current Z_corr into Y_out. */
int hmx_kmeans_centers(hmx_ctx* ctx, double* Y_out);

/* ---- harmony::cluster_cpp (src/harmony.cpp:208-262): 0 ok, -1 aborted, >0 error */
int hmx_cluster(hmx_ctx* ctx);
/* ---- harmony::moe_correct_ridge_cpp (src/harmony.cpp:345-638).  The whole correction is queued on the device and NOT waited for:
 * an error of its ridge solves (HMX_ERR_SOLVE: singular system) is DEFERRED -- it is returned by the next call that reads a result
 * of that correction (hmx_cluster's objective read, hmx_check_convergence of the following iteration, hmx_get / hmx_get_matrix,
 * which then return -1 with hmx_last_error = "singular ridge system ..."). */
int hmx_moe_correct_ridge(hmx_ctx* ctx);
/* ---- harmony::check_convergence (src/harmony.cpp:173-205): returns 1/0, or <0 on error */
int hmx_check_convergence(hmx_ctx* ctx, int32_t type);
/* ---- harmony::compute_objective (src/harmony.cpp:158-170): appends to the 4 series.
 * Deviation (documented): the reference evaluates the k-means term on its STORED dist_mat (:160), which
 * moe_correct_ridge_cpp does not refresh; this library keeps no K x N distance matrix and recomputes the distances from
 * the current Z_corr and Y.  Inside the reference's own call sequence (init_cluster_cpp / cluster_cpp, where dist_mat is
 * current) the values agree; a stand-alone call made between a correction and the next cluster_cpp sees the corrected,
 * un-normalised Z_corr and the new Y instead of the stale distances -- unless hmx_set_int(ctx, "stale_dist", 1) was set before
 * hmx_setup: every correction then keeps a snapshot of the normalised Z_corr and of Y it overwrites (N x d floats more), and such a
 * call reproduces the reference's value. */
int hmx_compute_objective(hmx_ctx* ctx);

/* ---- fields and getters (src/harmony.cpp:675-707, 640-669).
 * `field` is one of: "Z_corr" "Z_orig" "R" "Y" "O" "E" "W" "Lambda" "Pr_b" "theta" "sigma"
 * "lambda" "B_vec" "objective_kmeans" "objective_kmeans_dist" "objective_kmeans_entropy"
 * "objective_kmeans_cross" "objective_harmony" "kmeans_rounds" "N" "B" "K" "d" "alpha"
 * "max_iter_kmeans" "W_rows" (+ diagnostics: "subset_clusters" "skipped_clusters" "n_combos" "usig"
 * "upd_wps" "comm:calls" "comm:bytes" and "timer:<phase>" in ms).  Returns the number of doubles the field holds (call with
 * out == NULL to size), or -1 for an unknown field.  At most `cap` values are written. */
int64_t hmx_get(hmx_ctx* ctx, const char* field, double* out, int64_t cap);
/* getZcorr / getZorig / getR (src/harmony.cpp:640-655) with a choice of element type (HMX_F64 = the R seam, HMX_F32) and
 * destination (HMX_HOST or HMX_DEVICE pointer); `cap` counts elements.  Host destinations are filled slab by slab
 * through two HBM staging buffers -- no N x w fp64 device copy -- and a ring of page-locked slots that a few host threads drain
 * into `out` (pageable, possibly never touched: R's allocMatrix) while the next slabs are on their way.
 * hmx_get(ctx, "Z_corr", ...) is this with (F64, HOST). */
int64_t hmx_get_matrix(hmx_ctx* ctx, const char* field, void* out, int32_t dtype, int32_t location, int64_t cap);
/* writable fields: "max_iter_kmeans" (vignettes/detailedWalkthrough.Rmd:364), "seed",
 * "device" (before setup), "profile" (HIP-event timing of the update kernel, see "measurement"),
 * "rng" (0 counter-based generator | 1 R-compatible stream, see "randomness"),
 * "ref_arith" (before setup; see "arithmetic"), "stale_dist" (before setup; see hmx_compute_objective).
 * The tuning / fallback selectors and the settings of the reference-arithmetic machinery are laboratory equipment, not part of the
 * reference's interface: include/harmony_mi355x_lab.h lists them. */
int hmx_set_int(hmx_ctx* ctx, const char* field, int64_t value);

/* ---- randomness ------------------------------------------------------------------------
 * The reference draws its per-round cell shuffle (arma::shuffle, src/harmony.cpp:272-273) and
 * its centroid seeds (fill::randu, src/utils.cpp:12,29) from R's RNG, which is not available
 * to a C library.  Only block MEMBERSHIP matters mathematically, so the library derives, for
 * round r, the position of global cell g in the shuffled order from a stateless bijection
 *      pos = hmx_feistel_pos(seed, r, N_global, g)       (6-round Feistel + cycle walking)
 * and block(g) = min(pos / cells_per_block, n_blocks-1) exactly as src/harmony.cpp:280-300.
 * A host that owns an RNG stream (e.g. R's) can inject its own shuffles instead:
 * each call queues one update_order (N_global entries, update_order[p] = cell at position p,
 * consumed by the next update_R round).
 *
 * R-compatible mode: hmx_set_int(ctx, "rng", 1) makes the library draw exactly what the reference draws
 * from R: K anchors `randu`, then N uniforms per anchor in cell order (src/utils.cpp:12,29), then per clustering round N
 * `randi` values sorted into arma::shuffle's order (src/harmony.cpp:272-273) -- RcppArmadillo maps randu to
 * Rf_runif(0,1) and randi to int(Rf_runif(0,RAND_MAX)).  The uniforms come either from the built-in MT19937 seeded as
 * R's set.seed(seed) does (field "seed"; reproduces `set.seed(seed); RunHarmony(...)` of a default R session), or from
 * the host's own generator through hmx_set_uniform_source (the .Call glue passes R's unif_rand between GetRNGstate /
 * PutRNGstate).  Verified here against MT19937's published vector and R's documented set.seed(1); runif(3) stream;
 * equality with a live R run needs R.  The draws are made on the host (N per round): a compatibility mode, not the
 * fast path. */
uint64_t hmx_feistel_pos(uint64_t seed, uint64_t round, uint64_t N, uint64_t g);
/* the inverse of the same bijection: the global cell that sits at position `pos` of round `round` (the rounds run backwards, the cycle walk
 * likewise).  The cells of block b are hmx_feistel_cell(seed, r, N, p) for p in [b * cells_per_block, (b + 1) * cells_per_block): the library
 * builds a round's block order from it without sorting (round 4). */
uint64_t hmx_feistel_cell(uint64_t seed, uint64_t round, uint64_t N, uint64_t pos);
int hmx_set_uniform_source(hmx_ctx* ctx, double (*unif_rand)(void* user), void* user);
int hmx_push_update_order(hmx_ctx* ctx, const int64_t* update_order);

/* ---- multi-GPU: one handle per process/GPU, cells sharded contiguously --------------------
 * An optional host-provided all-reduce hook must SUM (dtype 0: int64, 1: float64) or MIN
 * (dtype 2: int64) `count` elements in place at device pointer `buf`, enqueued on `stream`
 * (a hipStream_t) or completed on return.  Return 0 on success. */
typedef int (*hmx_allreduce_fn)(void* user, void* buf, int64_t count, int32_t dtype, void* stream);
/* Built-in collective: RCCL over xGMI, bound lazily from the system librccl.  Rank 0 creates a 128-byte unique
 * id (hmx_comm_unique_id) and the host ships it to the other ranks by any means (MPI, torch.distributed/gloo,
 * a file, an R socket); every rank then calls hmx_comm_init before hmx_set_shard / hmx_setup.  With a
 * communicator, hmx_set_shard may pass fn == NULL. */
int hmx_comm_unique_id(uint8_t* out128);
int hmx_comm_init(hmx_ctx* ctx, int32_t rank, int32_t world, const uint8_t* unique_id128);
/* Host-side helper for hosts that bring no collective library of their own (a plain C / R host, bench.py --bootstrap file): all-reduce
 * `count` doubles in place over the ranks of the handle's communicator -- op 0 sum, 1 max, 2 min -- after everything queued on the handle's
 * stream has completed: a barrier and a device synchronisation in one call (count = 1 works as a plain barrier).  Needs hmx_comm_init. */
int hmx_comm_allreduce_host(hmx_ctx* ctx, double* inout, int32_t count, int32_t op);
/* Peer-to-peer block chain (sharded runs, up to 8 ranks of one node).  The update_R block chain (src/harmony.cpp:296-331) needs
 * the K x B contribution table of every block summed over all ranks before the next block starts: 20 dependent all-reduces per
 * round.  With the peers' inboxes connected, the persistent chain kernel does that sum INSIDE the launch (each GPU writes its
 * table straight into every peer's inbox over xGMI and adds up what arrived in its own): no collective call, no launch per block.
 * With the inboxes connected the library also sends every SMALL collective of a run through them (O after a head, the objective's sums, the
 * Lloyd sums, the seeding minima, ridge statistics up to 65 536 values: one launch and one trip over xGMI each instead of a ring; the old
 * contributions of a round's blocks travel with the chain's own exchange) -- hmx_get "p2p:allreduce_calls" counts them, "comm:calls" what is
 * left on the communicator / hook.  HMX_P2P_AR=0 keeps them on the communicator.  Round 5: LARGER buffers (the ridge statistics of many-level
 * designs -- Q K (d + 1) doubles, 1.3M at BASELINE configs[4] --, the per-round old sums of the launch-per-step path) go through the same inboxes
 * as reduce-scatter + all-gather windows of 32768 x world entries (rank e / 32768 owns entry e of a window, adds the ranks' values in rank order,
 * sends the result to everybody: 2 (G - 1) / G of the buffer per link instead of G - 1 times it) -- hmx_get "p2p:allreduce_big_windows" counts the
 * windows; HMX_P2P_AR_BIG=0 sends those buffers to the communicator / hook instead.
 * hmx_comm_init sets all of this up by itself (HMX_P2P=0 disables it).  A host that brings its own all-reduce hook can do it by
 * hand: every rank exports a handle (a hipIpcMemHandle_t, HMX_P2P_HANDLE_BYTES bytes), the host all-gathers them (rank order),
 * every rank connects, then -- after a host barrier -- every rank runs the self-test AT THE SAME TIME, and only if it passed on
 * EVERY rank (the host ANDs the results) every rank enables it.  Until then, and whenever it is off, sharded runs use one launch +
 * one all-reduce per block.  Ranks must be separate processes.  hmx_get "p2p" = 1 when on; hmx_p2p_status = why not. */
#define HMX_P2P_HANDLE_BYTES 64
int hmx_p2p_export(hmx_ctx* ctx, uint8_t* handle_out);
int hmx_p2p_connect(hmx_ctx* ctx, int32_t rank, int32_t world, const uint8_t* handles /* world x HMX_P2P_HANDLE_BYTES */);
int hmx_p2p_selftest(hmx_ctx* ctx);
int hmx_p2p_enable(hmx_ctx* ctx, int32_t on);
const char* hmx_p2p_status(hmx_ctx* ctx);
/* must be called before hmx_setup; global_offset = index of this rank's first cell.  fn != NULL overrides the
 * built-in RCCL all-reduce with a host-provided hook (tests: thread rendezvous, gloo). */
int hmx_set_shard(hmx_ctx* ctx, int32_t rank, int32_t world, int64_t global_offset,
                  int64_t N_global, hmx_allreduce_fn fn, void* user);
/* run all kernels on this hipStream_t (default: the library's own stream) */
int hmx_set_stream(hmx_ctx* ctx, void* hip_stream);
/* optional user-interrupt poll, checked once per clustering round and per correction
 * (Progress::check_abort(), src/harmony.cpp:233,355); non-zero return aborts */
int hmx_set_abort_poll(hmx_ctx* ctx, int (*poll)(void*), void* user);

/* ---- arithmetic -------------------------------------------------------------------------------------------------
 * By default every cross-cell accumulator is exact (64-bit fixed point / fp64 in a fixed order): the result is what the reference's
 * own sources compute when built with their own precision switch (-DHARMONY_SCALAR_DOUBLE, src/types.h:5-9), on one GPU or sharded
 * (identical for every shard count).  The reference as it ships accumulates in fp32, one term after the other (src/utils.cpp:67-75,
 * src/harmony.cpp:312-313,329-330,567,592-608); at 10^6 cells that is a visible, systematic bias.
 * hmx_set_int(ctx, "ref_arith", 1) before hmx_setup (one GPU) reproduces that arithmetic -- the O / E tables with their -= / += drift, the
 * objective's my_accu sums, the ridge statistics, the closed-form fp32 inverse -- as restarted sequential sums and follows
 * the CPU package's numbers to 2e-6 in Z_corr.  "ref_arith" = 2 keeps the O / E tables exact and reproduces the other three groups (the ones
 * that move the result): 4e-5 .. 6e-5 from the CPU package at 10^6 .. 2 10^6 cells, inside the 1e-4 contract, at a bit over half the time.  The
 * accumulator groups can be switched one by one and the iteration of the restarted sums tuned: include/harmony_mi355x_lab.h. */

/* ---- measurement ----------------------------------------------------------------------------
 * HIP-event timing of the dominant kernel on the library's stream: after
 * hmx_set_int(ctx,"profile",1) every launch of the E-step update kernel is bracketed by
 * events; "prof:update_ms" / "prof:update_launches" / "prof:update_cells" via hmx_get. */

#ifdef __cplusplus
}
#endif
#endif /* HARMONY_MI355X_H */
