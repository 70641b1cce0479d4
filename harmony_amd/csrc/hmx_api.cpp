// hmx_api.cpp -- host orchestration behind the C ABI of include/harmony_mi355x.h.
// Mirrors the reference's `harmony` object (src/harmony.h:20-70): same methods, same
// per-call operation order, but every pass over the cells is a gfx950
// kernel (hmx_kernels.hip) and all per-cell state stays in HBM between calls.
#include "../../include/harmony_mi355x_lab.h"      // (the reference interface + the probes / tuning declarations: the library defines both)
#include "hmx_internal.h"
#include "hmx_rrng.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <numeric>
#include <set>
#include <string>
#include <thread>
#include <vector>

using namespace hmx;

namespace {

inline uint64_t h_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
inline uint32_t h_fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
int my_ceil(float num) {  // src/utils.cpp:102-108
  int inum = (int)num;
  if (num == (float)inum) return inum;
  return inum + 1;
}
constexpr unsigned long long SEED_SENTINEL = 0x7fffffffffffffffull;

// RCCL is bound lazily (dlopen) so that single-GPU users never load the 570 MB library.  It must be the
// SYSTEM librccl (the one built against the libamdhip64 this library links), not a copy bundled elsewhere.
struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;   // optional (P2P bootstrap)
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi* rccl_api(std::string* err) {
  static RcclApi api; static bool tried = false;
  if (api.h) return &api;
  if (tried) { if (err) *err = "librccl could not be loaded"; return nullptr; }
  tried = true;
  // 1) an RCCL that is ALREADY loaded in the process (PyTorch's bundled librccl.so when the host imported torch first:
  //    it is the build that matches the HIP runtime this library then shares with torch); 2) the system library.
  api.h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!api.h) {
    const char* names[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    for (const char* n : names) { api.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (api.h) break; }
  }
  if (!api.h) { if (err) *err = std::string("dlopen(librccl): ") + dlerror(); return nullptr; }
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
  api.AllReduce = (decltype(api.AllReduce))dlsym(api.h, "ncclAllReduce");
  api.AllGather = (decltype(api.AllGather))dlsym(api.h, "ncclAllGather");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
    if (err) *err = "librccl lacks the expected symbols"; dlclose(api.h); api.h = nullptr; return nullptr;
  }
  return &api;
}

// Two persistent block chains must never be resident at once: each needs one workgroup on EVERY CU and spins on its peers, so
// two of them sharing the CUs would starve each other (bounded spins turn that into an error, not a hang -- but the round is
// lost).  Handles of one process are therefore chained through an event per device: a chain launch waits for the previous
// chain launch of any handle on that device.  (Two PROCESSES running unsharded chains on one GPU are not protected; HMX_CHAIN=0.)
struct ChainGate { std::mutex mu; std::map<int, hipEvent_t> last; std::map<int, const void*> owner; };
ChainGate& chain_gate() { static ChainGate g; return g; }

bool host_pin_enabled() { const char* e = getenv("HMX_PIN"); return !(e && atoi(e) == 0); }
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#include "hmx_api_seam.inc"
#include "hmx_api_kmeans.inc"
#include "hmx_api_refarith.inc"
#include "hmx_api_update.inc"
#include "hmx_api_ridge.inc"
}  // namespace

// =====================================================================================================
extern "C" {

hmx_ctx* hmx_create(void) { return new hmx_ctx(); }

void hmx_destroy(hmx_ctx* ctx) {
  if (!ctx) return;
  if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
  if (ctx->comm) { RcclApi* api = rccl_api(nullptr); if (api) (void)api->CommDestroy(ctx->comm); ctx->comm = nullptr; }
  for (int g = 0; g < 8; g++) if (ctx->p2p_peer[g] && ctx->p2p_peer[g] != ctx->p2p_self) (void)hipIpcCloseMemHandle(ctx->p2p_peer[g]);
  if (ctx->p2p_self) (void)hipFree(ctx->p2p_self);
  if (ctx->p2p_result) (void)hipFree(ctx->p2p_result);
  free_all(ctx);
  if (ctx->own_stream && ctx->L.stream) (void)hipStreamDestroy(ctx->L.stream);
  delete ctx;
}
const char* hmx_last_error(hmx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null handle"; }
const char* hmx_p2p_status(hmx_ctx* ctx) { return ctx ? ctx->p2p_note.c_str() : "null handle"; }
const char* hmx_last_warning(hmx_ctx* ctx) {   // one-shot: a warning is reported once (Rcpp::warning fires once, src/harmony.cpp:87)
  if (!ctx) return "";
  ctx->warn_ret.swap(ctx->warn); ctx->warn.clear();
  return ctx->warn_ret.c_str();
}

uint64_t hmx_feistel_cell(uint64_t seed, uint64_t round, uint64_t N, uint64_t pos) {
  int bits = 2;
  while (((uint64_t)1 << bits) < N) bits += 2;
  const int half = bits / 2;
  const uint32_t mask = (uint32_t)(((uint64_t)1 << half) - 1);
  uint32_t keys[6];
  for (int r = 0; r < 6; r++)
    keys[r] = (uint32_t)(h_splitmix64(seed ^ (round * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(r + 1) << 56)) >> 32);
  uint64_t x = pos;
  do {
    uint32_t L = (uint32_t)(x >> half), R = (uint32_t)(x & mask);
    for (int r = 5; r >= 0; r--) { uint32_t t = R ^ (h_fmix32(L * 0x9E3779B1u + keys[r]) & mask); R = L; L = t; }
    x = ((uint64_t)L << half) | R;
  } while (x >= N);
  return x;
}
uint64_t hmx_feistel_pos(uint64_t seed, uint64_t round, uint64_t N, uint64_t g) {
  int bits = 2;
  while (((uint64_t)1 << bits) < N) bits += 2;
  const int half = bits / 2;
  const uint32_t mask = (uint32_t)(((uint64_t)1 << half) - 1);
  uint32_t keys[6];
  for (int r = 0; r < 6; r++)
    keys[r] = (uint32_t)(h_splitmix64(seed ^ (round * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(r + 1) << 56)) >> 32);
  uint64_t x = g;
  do {
    uint32_t L = (uint32_t)(x >> half), R = (uint32_t)(x & mask);
    for (int r = 0; r < 6; r++) { uint32_t t = L ^ (h_fmix32(R * 0x9E3779B1u + keys[r]) & mask); L = R; R = t; }
    x = ((uint64_t)L << half) | R;
  } while (x >= N);
  return x;
}
int32_t hmx_cluster_of_column(int32_t nct, int32_t ct, int32_t c) {
  if (nct < 1 || nct > 16 || ct < 0 || ct >= nct || c < 0 || c > 15) return -1;
  const int k = kcol(nct, ct, c);
  int qd, i, cc; kcol_inv(nct, k, qd, i, cc);        // (the image builders use the inverse: both must agree)
  return (4 * qd + i == ct && cc == c) ? k : -2;
}
float hmx_u01(uint64_t seed, uint64_t stream, uint64_t idx) {
  const uint64_t h = h_splitmix64(h_splitmix64(seed ^ (stream * 0xD1342543DE82EF95ull)) + idx);
  return ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
}

int hmx_set_shard(hmx_ctx* ctx, int32_t rank, int32_t world, int64_t global_offset, int64_t N_global,
                  hmx_allreduce_fn fn, void* user) {
  if (!ctx) return HMX_ERR_ARG;
  if (ctx->ran_setup) return fail(ctx, HMX_ERR_STATE, "hmx_set_shard must precede hmx_setup");
  if (world < 1 || rank < 0 || rank >= world) return fail(ctx, HMX_ERR_ARG, "bad shard description");
  if (world > 1 && !fn && !ctx->comm) return fail(ctx, HMX_ERR_ARG, "a sharded handle needs hmx_comm_init or an all-reduce hook");
  ctx->rank = rank; ctx->world = world; ctx->goff = global_offset; ctx->N_global = N_global; ctx->ar = fn; ctx->ar_user = user;
  return 0;
}
int hmx_comm_unique_id(uint8_t* out) {
  if (!out) return HMX_ERR_ARG;
  RcclApi* api = rccl_api(nullptr);
  if (!api) return HMX_ERR_COMM;
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return HMX_ERR_COMM;
  std::memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}
static void p2p_auto(hmx_ctx* ctx, RcclApi* api, int rank, int world);
int hmx_comm_init(hmx_ctx* ctx, int32_t rank, int32_t world, const uint8_t* unique_id) {
  if (!ctx || !unique_id || world < 1 || rank < 0 || rank >= world) return ctx ? fail(ctx, HMX_ERR_ARG, "bad communicator description") : HMX_ERR_ARG;
  if (ctx->ran_setup) return fail(ctx, HMX_ERR_STATE, "hmx_comm_init must precede hmx_setup");
  std::string err;
  RcclApi* api = rccl_api(&err);
  if (!api) return fail(ctx, HMX_ERR_COMM, err);
  if (ctx->device < 0) { int cur = 0; (void)hipGetDevice(&cur); ctx->device = cur; }
  HIPCHK(hipSetDevice(ctx->device));
  ncclUniqueId id; std::memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = api->CommInitRank(&ctx->comm, world, id, rank);
  if (r != ncclSuccess) { ctx->comm = nullptr; return fail(ctx, HMX_ERR_COMM, std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(r) : "error")); }
  p2p_auto(ctx, api, rank, world);     // in-launch exchange of the block chain over the peers' inboxes, if the node allows it
  return 0;
}
int hmx_comm_allreduce_host(hmx_ctx* ctx, double* inout, int32_t count, int32_t op) {
  if (!ctx || !inout || count <= 0 || op < 0 || op > 2) return ctx ? fail(ctx, HMX_ERR_ARG, "bad arguments") : HMX_ERR_ARG;
  if (!ctx->comm) return fail(ctx, HMX_ERR_STATE, "hmx_comm_init first");
  RcclApi* api = rccl_api(nullptr);
  if (ctx->device >= 0) HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->L.stream) { HIPCHK(hipStreamCreateWithFlags(&ctx->L.stream, hipStreamNonBlocking)); ctx->own_stream = true; }
  if (ctx->side) HIPCHK(hipStreamSynchronize(ctx->side));
  double* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, sizeof(double) * (size_t)count));
  hipError_t e = hipMemcpyAsync(d, inout, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->L.stream);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) r = api->AllReduce(d, d, (size_t)count, ncclFloat64, op == 0 ? ncclSum : op == 1 ? ncclMax : ncclMin, ctx->comm, ctx->L.stream);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(inout, d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->L.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->L.stream);
  (void)hipFree(d);
  if (r != ncclSuccess) return fail(ctx, HMX_ERR_COMM, "ncclAllReduce failed");
  if (e != hipSuccess) return fail(ctx, HMX_ERR_DEVICE, hipGetErrorString(e));
  return 0;
}
#include "hmx_api_p2p.inc"
#include "hmx_api_setup.inc"
int hmx_restart(hmx_ctx* ctx) {
  if (!ctx || !ctx->ran_setup) return ctx ? fail(ctx, HMX_ERR_STATE, "setup first") : HMX_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  const Dev& D = ctx->D;
  l_normalize_from(ctx->L, D.Zo, D.Zc, D.n, D.d, D.zs); KCHK();  // Z_corr = normalise(Z_orig) :42 (one pass)
  for (int i = 0; i < 2; i++) if (ctx->sold_state[i] == 2) ctx->sold_state[i] = 1;
  HIPCHK(hipStreamSynchronize(ctx->L.stream));
  ctx->obj_pending = 0; ctx->obj_harmony_pending = false;
  ctx->obj_kmeans.clear(); ctx->obj_dist.clear(); ctx->obj_entropy.clear(); ctx->obj_cross.clear(); ctx->obj_harmony.clear();
  ctx->kmeans_rounds.clear(); ctx->round_counter = 0; ctx->ran_init = false; ctx->injected.clear(); ctx->rrng_seeded = false;
  ctx->y_on_device = false; ctx->solve_pending = false;
  ctx->obj_warm = ctx->rg_warm = ctx->rp_warm = false;            // (a run never depends on what the handle computed before it)
  ctx->head_is_stale = false;
  HIPCHK(hipMemsetAsync(ctx->D.solve_err, 0, sizeof(int), ctx->L.stream));
  if (ctx->side) HIPCHK(hipStreamSynchronize(ctx->side));
  for (int i = 0; i < 4; i++) ctx->sorted_round[i] = -1;      // (sorted_on_side stays: a sort still running on the side stream is waited for before its set is reused)
  return 0;
}

int hmx_kmeans_centers(hmx_ctx* ctx, double* Y_out) {
  if (!ctx || !ctx->ran_setup) return ctx ? fail(ctx, HMX_ERR_STATE, "setup first") : HMX_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  CHK(kmeans_centers(ctx));
  CHK(sync_solve_results(ctx));       // (the centres live on the device: fetch the host copy)
  if (Y_out) for (size_t i = 0; i < ctx->Y.size(); i++) Y_out[i] = (double)ctx->Y[i];
  return 0;
}

int hmx_init_cluster(hmx_ctx* ctx, const double* Y0) {  // src/harmony.cpp:131-156
  if (!ctx || !ctx->ran_setup) return ctx ? fail(ctx, HMX_ERR_STATE, "setup first") : HMX_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  const double t0 = now_ms();
  ctx->y_on_device = false;
  if (Y0) {
    ctx->Y.resize((size_t)ctx->d * ctx->K); for (size_t i = 0; i < ctx->Y.size(); i++) ctx->Y[i] = (float)Y0[i];
    normalise_cols(ctx->Y, ctx->d, ctx->K);  // :136
    CHK(upload_Y(ctx));
  } else {
    { PhaseScope ph(ctx, "kmeans_centers"); CHK(kmeans_centers(ctx)); }
    l_y_images(ctx->L, ctx->D, nullptr, 1); KCHK();       // Y <- normalise(Y) (:136) and its images, on the device
    ctx->y_on_device = true;
  }
  CHK(head_pass(ctx));
  CHK(objective_snapshot(ctx));
  CHK(push_objective(ctx));
  CHK(flush_objectives(ctx));
  ctx->obj_harmony.push_back(ctx->obj_kmeans.back());
  ctx->ran_init = true;
  ctx->timers["init_cluster"] += now_ms() - t0;
  return 0;
}

int hmx_compute_objective(hmx_ctx* ctx) {  // src/harmony.cpp:158-170 on the current R, Z_corr, Y, O, E
  if (!ctx || !ctx->ran_init) return ctx ? fail(ctx, HMX_ERR_STATE, "init_cluster first") : HMX_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->R_valid) return fail(ctx, HMX_ERR_STATE, "R is not available: the clustering call that would have stored it did not complete");
  HIPCHK(hipMemsetAsync(ctx->D.objpart, 0, sizeof(double) * 2 * (size_t)ctx->D.objslots * ctx->D.nwmax, ctx->L.stream));
  Dev Ds = ctx->D;
  const bool stale = ctx->stale_dist && ctx->head_is_stale;
  if (stale) { Ds.Zc = ctx->Zc_head; Ds.Yt = ctx->Yt_head; Ds.obj_stale = 1; }          // the reference's stored dist_mat (:160)
  l_head(ctx->L, Ds, 1); KCHK();
  l_obj_reduce(ctx->L, ctx->D); KCHK();
  CHK(allreduce(ctx, ctx->D.obj, 2, 1));
  CHK(objective_snapshot(ctx, stale ? &Ds : nullptr));     // (obj_arith re-derives the terms: from the same snapshot)
  CHK(push_objective(ctx));
  return flush_objectives(ctx);
}

int hmx_check_convergence(hmx_ctx* ctx, int32_t type) {
  if (!ctx) return -HMX_ERR_ARG;
  if (flush_objectives(ctx)) return -HMX_ERR_DEVICE;
  // (The results of the last device-side correction are NOT waited for here: the decision needs the objective only, and the
  //  host is free to queue the next clustering call while the correction kernels still run.  A singular ridge system surfaces with
  //  the next objective snapshot -- flush_objectives -- or, after the last correction, when a result is read: hmx_get / hmx_get_matrix.)
  if ((type == 0 && ctx->obj_kmeans.size() < (size_t)ctx->window_size + 1) || (type == 1 && ctx->obj_harmony.size() < 2)) {
    fail(ctx, HMX_ERR_STATE, "not enough objective values"); return -HMX_ERR_STATE;
  }
  return check_convergence_impl(ctx, type) ? 1 : 0;
}

int hmx_cluster(hmx_ctx* ctx) {  // src/harmony.cpp:208-262
  if (!ctx || !ctx->ran_init) return ctx ? fail(ctx, HMX_ERR_STATE, "init_cluster first") : HMX_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  const double t0 = now_ms();
  CHK(flush_objectives(ctx));   // (values of an earlier call that nobody asked for yet)
  if (ctx->obj_harmony.size() != 1) {  // :214-228
    PhaseScope ph(ctx, "cluster_head");
    CHK(head_pass(ctx, true));
  }
  int iter;
  ctx->objD_valid = false;
  struct RoundLoop { hmx_ctx* c; ~RoundLoop() { c->in_cluster_rounds = false; c->objD_valid = false; } } round_loop{ctx};      // (dist_mat of this call serves its rounds' objective evaluations only)
  ctx->in_cluster_rounds = true;
  for (iter = 0; iter < ctx->max_iter_kmeans; iter++) {
    if (ctx->poll && ctx->poll(ctx->poll_user)) return HMX_ABORTED;  // :233-234
    ctx->last_round_hint = (iter == ctx->max_iter_kmeans - 1);         // (nothing follows the last round that could use its R sums)
    ctx->round_may_be_last = ctx->last_round_hint || iter > ctx->window_size;      // (the windowed convergence check below can end the call after this round)
    CHK(update_R(ctx));                                                 // :241 (objective fused, :248)
    if (iter > ctx->window_size) {                                      // :250-256 (the only place a round's value is needed at once)
      CHK(flush_objectives(ctx));
      if (check_convergence_impl(ctx, 0)) { iter++; break; }
    }
  }
  ctx->kmeans_rounds.push_back(iter);
  ctx->obj_harmony_pending = true;   // objective_harmony <- objective_kmeans.back() (:260), resolved with the pending rounds
  if (ctx->obj_pending == 0) { ctx->obj_harmony.push_back(ctx->obj_kmeans.back()); ctx->obj_harmony_pending = false; }
  ctx->timers["cluster"] += now_ms() - t0;
  return 0;
}

int hmx_moe_correct_ridge(hmx_ctx* ctx) {  // src/harmony.cpp:345-638
  if (!ctx || !ctx->ran_init) return ctx ? fail(ctx, HMX_ERR_STATE, "init_cluster first") : HMX_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->poll && ctx->poll(ctx->poll_user)) return HMX_ABORTED;  // :355-356
  if (!ctx->R_valid) return fail(ctx, HMX_ERR_STATE, "R is not available: the clustering call that would have stored it did not complete");
  const double t0 = now_ms();
  const Dev& D = ctx->D;
  const int K = ctx->K, B = ctx->B, d = ctx->d, Q = ctx->Q;
  const bool seq = ctx->ridge_arith == 1;
  if (ctx->stale_dist && !ctx->head_is_stale) {      // what dist_mat was computed from (:141 / :221): kept for a stand-alone compute_objective
    l_copy(ctx->L, D.Zc, ctx->Zc_head, (size_t)ctx->N * D.zs); KCHK();
    l_copy(ctx->L, D.Yt, ctx->Yt_head, (size_t)d * K); KCHK();
    ctx->head_is_stale = true;
  }
  { PhaseScope pall(ctx, "correct_ridge_loop");
  { PhaseScope ph(ctx, "ridge_statistics");   // reference timers Phi_Rk + Phi_cov + Z_tmp + Z_intercept + batch_exprod: ONE pass here
    l_zero4(ctx->L, D.Sq, (size_t)Q * d * K, D.nq, (size_t)Q * K, nullptr, 0, nullptr, 0); KCHK();      // (one launch instead of two memsets)
    if (seq) CHK(seq_ridge_stats(ctx));
    else if (D.moe_mfma) { l_moe_stats_mfma(ctx->L, D); KCHK(); } else { l_moe_stats(ctx->L, D); KCHK(); }
    CHK(allreduce(ctx, D.Sq, (int64_t)Q * d * K, 1));
    CHK(allreduce(ctx, D.nq, (int64_t)Q * K, 1)); }
  if (ctx->solve_on_device) {
    // the whole correction stays on the device: statistics -> K fp64 solves (one workgroup per cluster) -> apply -> Y;
    // no host synchronisation (a singular system is reported by the next call that waits for the device)
    SolveArgs A;
    A.cov = ctx->sv_cov; A.rhs = ctx->sv_rhs; A.Wall = ctx->sv_Wall; A.mrows = ctx->sv_mrows; A.flags = ctx->sv_flags;
    A.lambda = ctx->lambda_estimation ? nullptr : ctx->sv_lambda; A.cov_bounds = ctx->sv_cov_bounds;
    A.alpha = ctx->alpha; A.cutoff = ctx->cutoff; A.use_s0 = seq ? 1 : 0; A.err = ctx->D.solve_err;
    A.Of = ctx->oe_arith ? ctx->Of : nullptr; A.Ef = ctx->oe_arith ? ctx->Ef : nullptr; A.solve_f32 = ctx->solve_arith;
    A.ref_tot = (seq && ctx->C > 1) ? ctx->rg_tot : nullptr; A.pair_tot = ctx->rp_tot; A.pair_idx = ctx->pair_idx;
    { PhaseScope ph(ctx, "arma_inv"); l_moe_solve(ctx->L, D, A); KCHK(); }
    { PhaseScope ph(ctx, "update_Zcorr");
      if (D.moe_mfma) { l_moe_apply_mfma(ctx->L, D); KCHK(); } else { l_moe_apply(ctx->L, D); KCHK(); } }
    ctx->y_on_device = true; ctx->solve_pending = true;
    ctx->timers["moe_correct_ridge"] += now_ms() - t0;
    return 0;
  }
  }
  std::vector<double> Sq((size_t)Q * d * K), nq((size_t)Q * K);
  std::vector<long long> ofx((size_t)B * K);
  CHK(d2h(ctx, Sq.data(), D.Sq, Sq.size())); CHK(d2h(ctx, nq.data(), D.nq, nq.size())); CHK(d2h(ctx, ofx.data(), D.O_fx, ofx.size()));
  std::vector<double> S0, n0;
  if (seq) { S0.resize((size_t)K * d); n0.resize((size_t)K); CHK(d2h(ctx, S0.data(), D.S0, S0.size())); CHK(d2h(ctx, n0.data(), D.n0, n0.size())); }
  const double t1 = now_ms();
  const std::vector<float> O = table_O(ctx, ofx), E = table_E(ctx, ofx);
  CHK(sync_solve_results(ctx));        // (host solve path: the centroids may still live on the device only)
  std::vector<float> Wq((size_t)Q * K * d), Ynew = ctx->Y;
  std::vector<SolveOut> outs(K);
  {
    unsigned nt = std::thread::hardware_concurrency(); if (nt < 1) nt = 1; if (nt > 16) nt = 16; if ((int)nt > K) nt = K;
    if ((size_t)K * (B + 1) * (B + 1) < 200000) nt = 1;
    std::vector<std::thread> th;
    auto work = [&](int t) { for (int k = t; k < K; k += (int)nt) solve_cluster(ctx, k, O, E, Sq, nq, Wq, Ynew, outs[k], seq ? &S0[(size_t)k * d] : nullptr, seq ? &n0[k] : nullptr); };
    if (nt == 1) work(0);
    else { for (unsigned t = 0; t < nt; t++) th.emplace_back(work, (int)t); for (auto& x : th) x.join(); }
  }
  ctx->subset_clusters = ctx->skipped_clusters = 0;
  for (int k = 0; k < K; k++) {
    if (outs[k].status) return fail(ctx, outs[k].status, "singular ridge system");
    if (outs[k].subset) ctx->subset_clusters++;
    if (outs[k].skipped) ctx->skipped_clusters++;
    else { ctx->W = outs[k].W; ctx->W_rows = outs[k].m; }
  }
  ctx->timers["moe_solve_host"] += now_ms() - t1;
  if (D.moe_mfma) {
    // image[q][qd][s][p][c][i] = Wq[q][cluster(s,p)][16*(4qd+i)+c], cluster(s,p) as tile_dots assigns reduction slots
    std::vector<float> img((size_t)Q * D.wNQ * D.wNS * 256, 0.f);
    for (int q = 0; q < Q; q++) for (int qd = 0; qd < D.wNQ; qd++) for (int s = 0; s < D.wNS; s++) for (int p = 0; p < 4; p++) {
      const int k = (s < 4 * D.wNT4) ? 16 * (s / 4) + 4 * p + (s % 4) : 16 * D.wNT4 + 4 * (s - 4 * D.wNT4) + p;
      if (k >= K) continue;
      const float* w = &Wq[((size_t)q * K + k) * d];
      float* o = &img[((((size_t)q * D.wNQ + qd) * D.wNS + s) * 4 + p) * 64];
      for (int c = 0; c < 16; c++) for (int i = 0; i < 4; i++) { const int jj = 16 * (4 * qd + i) + c; if (jj < d) o[c * 4 + i] = w[jj]; }
    }
    CHK(h2d(ctx, D.Wimg, img.data(), img.size()));
    l_moe_apply_mfma(ctx->L, D); KCHK();   // Z_corr = Z_orig - sum_k R_k W_k[levels]   :347,:615
  } else {
    CHK(h2d(ctx, D.Wq, Wq.data(), Wq.size()));
    l_moe_apply(ctx->L, D); KCHK();
  }
  ctx->Y = Ynew;
  normalise_cols(ctx->Y, d, K);     // :633
  CHK(upload_Y(ctx));
  HIPCHK(hipStreamSynchronize(ctx->L.stream));
  ctx->timers["moe_correct_ridge"] += now_ms() - t0;
  return 0;
}

#include "hmx_api_diag.inc"
