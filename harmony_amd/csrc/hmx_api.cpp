// hmx_api.cpp -- host orchestration behind the C ABI of include/harmony_mi355x.h.
// Mirrors the reference's `harmony` object (src/harmony.h:20-70): same methods, same
// per-call operation order, but every pass over the cells is a gfx950
// kernel (hmx_kernels.hip) and all per-cell state stays in HBM between calls.
#include "../../include/harmony_mi355x.h"
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

// ---- host side of the R seam (R/ui.R:178-183, 292-295: a pageable fp64 matrix in, a fresh pageable fp64 matrix out) ---------------------
// Page-locking the caller's 400 MB matrix for one transfer costs more than the transfer (every page is faulted in and pinned by ONE thread
// before the first byte moves).  Default path instead: a small ring of page-locked buffers, allocated once per process, and a few host
// threads that move the bytes between the caller's matrix and the ring slot by slot while the DMA engine moves the previous slots --
// on egress the threads are also the ones that first touch the fresh destination pages, in parallel.  HMX_XFER=pin: register the caller's
// buffer (the earlier path); HMX_PIN=0: plain pageable copies; HMX_XFER_THREADS: threads (default 8).
struct XferPool {
  // (the workers SPIN between the slots of one transfer -- a transfer lasts ~10-20 ms and hands out a slot every ~0.5 ms; waking sleeping
  //  threads through a condition variable for every slot cost more than the copies)
  std::vector<std::thread> th;
  const char* src = nullptr; char* dst = nullptr; size_t bytes = 0;
  std::atomic<size_t> next{0};
  std::atomic<int> gen{0}, running{0};
  std::atomic<bool> stop{false};
  static constexpr size_t CHUNK = 1 << 20;
  void work() {
    for (;;) {
      const size_t off = next.fetch_add(CHUNK);
      if (off >= bytes) break;
      memcpy(dst + off, src + off, std::min(CHUNK, bytes - off));
    }
  }
  void loop() {
    int seen = 0;
    for (;;) {
      int g;
      while ((g = gen.load(std::memory_order_acquire)) == seen) {
        if (stop.load(std::memory_order_relaxed)) return;
        __builtin_ia32_pause();
      }
      seen = g;
      work();
      running.fetch_sub(1, std::memory_order_release);
    }
  }
  void start(int T) {      // (a thread that cannot be created is simply missing: the caller works too, copy() falls back to a plain memcpy without helpers)
    for (int i = 0; i < T; i++) {
      try { th.emplace_back([this] { loop(); }); } catch (...) { break; }
    }
  }
  void copy(void* d, const void* s_, size_t n) {           // blocking; the caller works too
    if (th.empty() || n < 4 * CHUNK) { memcpy(d, s_, n); return; }
    src = (const char*)s_; dst = (char*)d; bytes = n; next.store(0);
    running.store((int)th.size(), std::memory_order_relaxed);
    gen.fetch_add(1, std::memory_order_release);
    work();
    while (running.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
  }
  ~XferPool() {
    stop.store(true);
    for (auto& t : th) t.join();
  }
};
int xfer_threads();
struct XferRing {     // page-locked staging slots + the copy stream, HBM staging slabs and events of a transfer: one set per device and
                      // process, built (and run through once) on first use, shared by all handles (guarded: one transfer at a time)
  static constexpr int NB = 4;
  static constexpr size_t SLOT = (size_t)16 << 20;
  std::mutex mu, init_mu;
  void* slot[NB] = {nullptr, nullptr, nullptr, nullptr};
  void* stage[2] = {nullptr, nullptr};                       // HBM staging slabs (SLOT bytes each)
  hipStream_t cs = nullptr;
  hipEvent_t ev_a[2] = {nullptr, nullptr}, ev_b[2] = {nullptr, nullptr}, ev_slot[NB] = {nullptr, nullptr, nullptr, nullptr};
  bool ok = false, tried = false;
  bool ensure() {
    std::lock_guard<std::mutex> lk(init_mu);
    if (tried) return ok;
    tried = true;
    bool good = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < NB && good; i++) good = hipHostMalloc(&slot[i], SLOT, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&ev_slot[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && good; i++) good = hipMalloc(&stage[i], SLOT) == hipSuccess && hipEventCreateWithFlags(&ev_a[i], hipEventDisableTiming) == hipSuccess &&
                                               hipEventCreateWithFlags(&ev_b[i], hipEventDisableTiming) == hipSuccess;
    // first touch from both sides and one pass through exactly the calls a transfer makes (the CPU faults the slots' pages in on its
    // first write, the device maps them and sets up its copy queues on the first asynchronous copy): once per process instead of inside
    // the first matrix's transfer
    for (int i = 0; i < NB && good; i++) {
      memset(slot[i], 0, SLOT);
      good = hipMemcpyAsync(stage[i & 1], slot[i], SLOT, hipMemcpyHostToDevice, cs) == hipSuccess && hipEventRecord(ev_slot[i], cs) == hipSuccess &&
             hipMemcpyAsync(slot[i], stage[i & 1], SLOT, hipMemcpyDeviceToHost, cs) == hipSuccess && hipEventRecord(ev_a[i & 1], cs) == hipSuccess &&
             hipEventSynchronize(ev_slot[i]) == hipSuccess;
    }
    if (good) good = hipStreamSynchronize(cs) == hipSuccess;
    if (good) {
      // ... and once through the helper threads (round 3 measured +16 ms on a process' FIRST threaded transfer and left it unexplained:
      // thread stacks, first scheduling of eight spinning threads, the cores' clocks): a pool of the size a transfer uses moves 4 slots'
      // worth of bytes between the slots here, once per process, instead of inside the first matrix's transfer
      XferPool warm; warm.start(xfer_threads());
      for (int r = 0; r < 2; r++) for (int i = 0; i + 1 < NB; i += 2) warm.copy(slot[i + 1], slot[i], SLOT);
    }
    if (!good) (void)hipGetLastError();
    return ok = good;
  }
};
XferRing& xfer_ring(int device) { static std::mutex m; static std::map<int, XferRing> rings; std::lock_guard<std::mutex> lk(m); return rings[device]; }
int xfer_mode() {     // 0 pageable copies, 1 register the caller's buffer, 2 ring of page-locked slots + host threads (default)
  if (!host_pin_enabled()) return 0;
  const char* e = getenv("HMX_XFER");
  if (e && std::string(e) == "pin") return 1;
  return 2;
}
int xfer_threads() {      // (the workers spin while a transfer runs: never more of them than spare cores)
  const char* e = getenv("HMX_XFER_THREADS");
  const int t = e ? atoi(e) : 8, spare = (int)std::thread::hardware_concurrency() - 1;
  return std::max(0, std::min(std::min(t, 64), std::max(spare, 0)));
}

}  // namespace

struct hmx_ctx {
  // ---- sharding -------------------------------------------------------------------
  int rank = 0, world = 1;
  int64_t goff = 0, N_global = 0;
  hmx_allreduce_fn ar = nullptr; void* ar_user = nullptr;
  ncclComm_t comm = nullptr;   // built-in all-reduce: RCCL over xGMI (hmx_comm_init)
  int64_t comm_calls = 0, comm_bytes = 0;
  bool comm_force = false;     // test hook: issue the collectives even when world == 1
  // peer-to-peer block chain (hmx_p2p_*): inboxes shared through HIP IPC; on only after the connection self-test passed everywhere
  unsigned long long* p2p_self = nullptr; unsigned long long* p2p_peer[8] = {}; int p2p_rank = 0, p2p_world = 0; bool p2p_on = false;
  unsigned p2p_tests = 0; int* p2p_result = nullptr; double p2p_exchange_us = 0.0; std::string p2p_note = "not connected";
  unsigned p2p_ar_seq = 0, p2p_xseq = 0; int64_t p2p_ar_calls = 0, p2p_ar_big_windows = 0;     // generic inbox all-reduces issued (same on every rank) / chain exchanges issued
  int (*poll)(void*) = nullptr; void* poll_user = nullptr;
  // ---- problem --------------------------------------------------------------------
  int64_t N = 0;  // local cells
  int d = 0, K = 0, B = 0, C = 0, Q = 0;
  std::vector<int> B_vec, cov_bounds;
  std::vector<float> sigma, theta, lambda, Pr_b, sizes;
  bool lambda_estimation = false;
  float alpha = 0.2f, block_size = 0.05f, eps_k = 1e-3f, eps_h = 1e-2f, cutoff = 1e-5f;
  int max_iter_kmeans = 4, window_size = 3, verbose = 0;
  uint64_t seed = 0, round_counter = 0;
  int nb = 20; uint64_t cells_per_block = 1;
  // ---- host-side state --------------------------------------------------------------
  std::vector<float> Y;  // d x K column-major (reference layout)
  std::vector<float> W; int W_rows = 0;
  std::vector<float> obj_kmeans, obj_dist, obj_entropy, obj_cross, obj_harmony;
  std::vector<int> kmeans_rounds;
  std::vector<int> qlev, perm;
  std::vector<long long> seed_cells;
  std::deque<std::vector<int64_t>> injected;
  int64_t subset_clusters = 0, skipped_clusters = 0;
  // objective snapshots are read back asynchronously: one pinned 3-double slot per clustering round, resolved into the
  // four series when a value is needed (convergence checks, getters) -- no host sync per round
  double* h_obj = nullptr; int obj_cap = 0, obj_pending = 0; hipEvent_t obj_event = nullptr; bool obj_harmony_pending = false;
  // randomness: 0 = documented counter-based generator, 1 = R-compatible (MT19937 seeded like set.seed, arma draw order)
  int rng_mode = 0; hmx::RRng rrng; bool rrng_seeded = false;
  // device-side ridge solve (default; HMX_MOE_SOLVE=host selects the synchronous host path): scratch and result buffers
  bool solve_on_device = true, y_on_device = false, solve_pending = false;
  double* sv_cov = nullptr; double* sv_rhs = nullptr; float* sv_Wall = nullptr; int* sv_mrows = nullptr; int* sv_flags = nullptr;
  float* sv_lambda = nullptr; int* sv_cov_bounds = nullptr;
  // Reference arithmetic: which accumulator groups follow the reference's fp32 operation order instead of the exact /
  // fp64 default -- the same four groups as the oracle's arithmetic mask.  ridge_arith: the ridge statistics (sequential fp32 sums
  // over the cells, src/harmony.cpp:567,599-608); oe_arith: the O / E tables (fp32, block sums in the round's shuffled order,
  // -= / += drift, :149-150,312-313,329-330); obj_arith: my_accu's K*N-term sequential fp32 sums (src/utils.cpp:67-75);
  // solve_arith: the closed-form fp32 arrowhead inverse (:575-586).  "ref_arith" sets all four.
  int ridge_arith = 0, oe_arith = 0, obj_arith = 0, solve_arith = 0;
  // "stale_dist" = 1: a stand-alone compute_objective between a correction and the next cluster_cpp evaluates the k-means term on the
  // distances of the LAST head (the reference's stored dist_mat, src/harmony.cpp:160, which moe_correct_ridge_cpp does not refresh):
  // the correction keeps a snapshot of the normalised Z_corr and of Y it is about to overwrite
  int stale_dist = 0; float* Zc_head = nullptr; float* Yt_head = nullptr; bool head_is_stale = false;
  // restarted sequential sums (hmx_seq.hip): plans (segments + chains on the device), shared workspace, cell lists
  struct SeqPlan { SeqSeg* d_segs = nullptr; SeqChain* d_chains = nullptr; size_t cap_segs = 0, cap_chains = 0; int nsegs = 0, nchains = 0;
                   std::vector<int> seg0;   /* [nchains + 1] first segment of every chain */ int seg_cells = 0; };
  SeqPlan plan_head, plan_ridge, plan_round, plan_pair;
  // ridge_arith with several covariates: Phi_Rk * Phi_moe_t has one entry per PAIR of levels that meet in a cell (src/harmony.cpp:561-568):
  // a sequential sum of R_k over the cells that carry both levels, in original order.  pairlist = those cells pair by pair, pair_idx[b][b2]
  // (b < b2) = the pair's chain or -1; rg_tot / rp_tot = the chain totals the device solve assembles its systems from.
  int* pairlist = nullptr; int* pair_idx = nullptr; int npairs = 0;
  float* rg_tot = nullptr; float* rp_tot = nullptr; float* rp_start = nullptr; bool rp_warm = false;
  float* sq_start = nullptr; float* sq_end = nullptr; size_t sq_cap = 0;
  float* sq_total = nullptr; size_t sq_total_cap = 0;
  unsigned* sq_mismatch = nullptr; int seq_passes = 2, seq_warm_passes = 2; int64_t seq_runs = 0;      // (round 5: 2 passes cold AND warm, see seq_tol / seq_iterate)
  bool seq_scan_final = false;      // set by seq_iterate around the scan behind the last pass of a non-adaptive sum (the O / E scans then only reduce)
  bool seq_stats = false;     // the last scan of NON-adaptive groups also counts the starts that still moved ("seq:mismatch" / "seq:residual" then cover every group)
  // long chains (>= seq_adaptive_cells cells) are iterated until the starts stop moving (chain-relative residual <= 2^-22) or seq_max_passes
  unsigned* sq_conv = nullptr; int seq_max_passes = 24; int64_t seq_adaptive_cells = 200000, seq_extra_passes = 0; double seq_resid_max = 0.0; uint64_t seq_mismatch_sum = 0;
  // seq_strict: EVERY group of restarted sums (short chains too) is iterated until no segment start moves any more -- the fixed point, at which
  // the concatenated segment loops are the reference's one-after-the-other loop bit for bit (tests/test_gpu_seq.py) -- instead of stopping at the
  // default pass count / the 2^-22 residual.  seq_group_passes / seq_group_runs: passes and runs per group (0 O/E, 1 objective, 2 ridge, 3 level pairs)
  // seq_tol: adaptive groups (long chains) stop when the largest move of a start in the last scan, relative to the largest start of its lane
  // group, is below it.  What is left after such a pass is the move times the iteration's contraction factor (the relative size of the
  // rounding bias itself, 1e-2 .. 1e-4; 0.25 where the accumulators saturate at 10M cells), i.e. far below the move.
  // Round 5 (tools/strict_probe.py at BASELINE configs[2], profiles/r5_strict_probe_1M_*.json): 2 passes + tol 1e-5, the round-3 default (3 passes +
  // tol 2^-22), 6 passes and the bit-for-bit fixed point (seq_strict, 1.2 s per run) all end 1.9-2.1e-6 from the oracle's faithful run and
  // 1.6-2.0e-6 from EACH OTHER -- the faithful trajectory itself moves by that much under any ulp-level change (the oracle's own liberties:
  // profiles/r5_oracle_liberties.json) -- so the cheap setting is the default; seq_passes / seq_tol_ppb / seq_strict select the others.
  double seq_tol = 1e-5;
  bool seq_strict = false; uint64_t seq_last_mismatch = 0, seq_unsettled = 0; int64_t seq_group_passes[4] = {0, 0, 0, 0}, seq_group_runs[4] = {0, 0, 0, 0};
  int* headlist = nullptr;                 // [(1 + C) n] cells in original order | cells by (level of covariate c, original order)
  int* headq = nullptr;                    // the same entries' combinations (ridge_arith: static, saves a dependent load per batch of the ridge pass)
  std::vector<int> lev_off, lev_cnt;       // [B] a level's range inside its covariate's part of headlist
  int* headlev = nullptr; int* roundlev = nullptr;     // [min(C, 4)][n] level codes of the positions of headlist's first part / of roundlist
  int* roundlist = nullptr;                // [(1 + C) n] this round's cells in shuffled order | by (block, level of covariate c), shuffled order
  std::vector<int> invperm_h, combo_h;     // host copies (internal order)
  float* Of = nullptr; float* Ef = nullptr; float* Mtab = nullptr;    // [B][K] fp32 O / E (oe_arith), theta log((O+E+1)/(2E+1))
  float* objT = nullptr; size_t objT_cap = 0;                         // the objective's three K x N term matrices (obj_arith)
  unsigned char* inset = nullptr;                                      // [Q][K] cells of combination q enter cluster k's regression
  // the objective's and the ridge statistics' segment starts live in their own buffers and survive from one evaluation to the next
  // (same chains, slowly changing terms): every evaluation after the first starts warm and needs one pass less
  float* obj_start = nullptr; size_t obj_start_cap = 0; bool obj_warm = false;
  double* obj_partial = nullptr; size_t obj_partial_cap = 0;      // per array and 256-segment workgroup of k_seq_arr_pass: the sum of its deltas (k_seq_scan1's bases)
  float* rg_start = nullptr; size_t rg_start_cap = 0; bool rg_warm = false;
  std::map<std::string, double> timers;
  // ---- device -------------------------------------------------------------------------
  int device = -1;
  Dev D{}; Launch L{};
  bool own_stream = false, ran_setup = false, ran_init = false;
  std::vector<void*> allocs;
  // profiling of the dominant kernel
  int profile = 0;             // 0 off | 1 the dominant kernel's launches carry a start / stop event pair | 2 and every phase is bracketed by events (PhaseScope)
  bool fused_ok = false;       // k_tile prologue fold usable (LDS budget) and not disabled
  // Old-contribution tables: two buffers.  `cur` is what this round's block steps subtract; the other one collects, inside this
  // round's tile kernels, the old contributions of the NEXT round's blocks (carry_ok: the shuffle keys every tile by its cells'
  // next block) -- then the next round needs no pass over R (k_oldsum).  state: 0 all zero, 1 unknown contents, 2 carried for round sold_round.
  long long* sold_buf[2] = {nullptr, nullptr}; int sold_cur = 0, sold_state[2] = {1, 1}; int64_t sold_round[2] = {-1, -1}; uint64_t sold_seed[2] = {0, 0};
  bool sets_clean = false;     // the three Snew replica sets are all zero
  bool carry_ok = false, last_round_hint = false, round_may_be_last = true; bool sorted_nxt[4] = {};
  // R rows that nobody reads are not stored (Dev::r_store = 0: the head inside cluster_cpp, rounds that cannot be a call's last).  R_valid says
  // whether D.R holds the rows of the LAST head / round: cleared when a pass starts, set when a storing pass has been queued completely.  A call
  // that fails half way leaves it false, and the getters / the correction refuse to consume stale rows.  r_store_always: HMX_R_STORE=1, read at setup.
  bool R_valid = false, r_store_always = false;
  int64_t rounds_without_R = 0;
  int64_t carried_rounds = 0;
  bool chain_ok = false; int chain_wgs = 0; uint64_t chain_rounds = 0;   // persistent block chain (one launch per round)
  int tun_impl = -1, tun_tpw = -1, tun_cpw = -1, tun_wps = -1;  // tunables set through hmx_set_int before setup
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool; size_t ev_used = 0;
  // GPU phase timers (profile mode): event pairs tagged with a phase name, named after the reference's Timer phases
  // (src/harmony.cpp:302-335,557-615) where a phase has a counterpart; resolved lazily into gpu_timers
  struct PhaseEv { hipEvent_t a, b; int name; };
  std::vector<PhaseEv> ph_pool; size_t ph_used = 0; std::vector<std::string> ph_names; std::map<std::string, double> gpu_timers;
  double prof_update_ms = 0; int64_t prof_update_launches = 0, prof_update_cells = 0, prof_update_steps = 0;
  bool chain_check = false;    // a persistent-chain launch has run since the error word was last read
  // The round's shuffle (counting sort by block) touches no algorithmic state, and with the counter-based generator it depends
  // on (seed, round) only: the sort of round r+1 runs on a SIDE stream while round r's old-sum pass streams on the main one,
  // into the second of two buffer sets.
  struct SortSet { int* blk; int* lorder; int2* lpair; int* lcombo; int* boff; int* binoff; int* counts; int* offs; int* blkv; int* bincnt; };
  SortSet sets[4] = {}; int oset_mask = 1;      // order sets: round & oset_mask (two; four with the batched shuffle, sort_sched = 3)
  hipStream_t side = nullptr; hipEvent_t ev_sorted[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
  int64_t sorted_round[4] = {-1, -1, -1, -1}; uint64_t sorted_seed[4] = {}; bool sorted_on_side[4] = {}; bool sort_overlap = true;
  int sort_sched = 3;
  // sort_sched = 3 (default): the shuffles of FOUR consecutive rounds in one set of launches on the main stream (l_sort_batch / l_shuffle_inv), into four full order
  // sets -- rounds keep their numbers across cluster_cpp calls, so a batch serves whichever calls its rounds fall into; between two block
  // chains of a batch there is no sort kernel and no event at all.  The batches are aligned groups (round >> 2).  sort_sched = 1: round 3's
  // schedule (the whole sort of round r + 1 on the side stream behind chain r; also what host-provided orders use).  (A schedule in between --
  // histogram halves rounds ahead on the side stream, the dependent tail on the main stream behind the chain -- was built and superseded: DESIGN 4.5.)
  // the sort-free form of the batched shuffle (k_shuf_*): position -> (cell, rank) per order set, the blocks of the round behind a batch,
  // the (block, bin, part) count matrix
  bool shuf_inv = false; int2* posr[4] = {}; int* shuf_partcnt[4] = {}; int* shuf_binacc[4] = {}; int64_t injected_round = -1;   // (injected_round: the round whose order the host provided -- its D.blk came with it)
  std::string err, warn, warn_ret;
};

namespace {

int fail(hmx_ctx* c, int code, const std::string& msg) { c->err = msg; return code; }

#define HIPCHK(expr)                                                                            \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(ctx, HMX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

template <class T> int dalloc(hmx_ctx* ctx, T** p, size_t count) {
  void* q = nullptr;
  if (count == 0) count = 1;
  HIPCHK(hipMalloc(&q, count * sizeof(T)));
  ctx->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}
void free_all(hmx_ctx* ctx) {
  for (void* p : ctx->allocs) (void)hipFree(p);
  ctx->allocs.clear();
  for (auto& e : ctx->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  ctx->ev_pool.clear(); ctx->ev_used = 0;
  for (auto& e : ctx->ph_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  ctx->ph_pool.clear(); ctx->ph_used = 0;
  if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); ctx->side = nullptr; }
  for (int i = 0; i < 2; i++) {
    if (ctx->ev_sorted[i]) { (void)hipEventDestroy(ctx->ev_sorted[i]); ctx->ev_sorted[i] = nullptr; }
    if (ctx->ev_free[i]) { (void)hipEventDestroy(ctx->ev_free[i]); ctx->ev_free[i] = nullptr; }
  }
  for (int i = 0; i < 4; i++) { ctx->sorted_round[i] = -1; ctx->sorted_on_side[i] = false; }
  {   // reference-arithmetic buffers (grown on demand, not in `allocs`)
    void* ps[] = {ctx->sq_start, ctx->sq_end, ctx->sq_total, ctx->sq_mismatch, ctx->headlist, ctx->roundlist, ctx->Of, ctx->Ef, ctx->Mtab, ctx->objT,
                  ctx->inset, ctx->headq, ctx->obj_start, ctx->obj_partial, ctx->rg_start, ctx->headlev, ctx->roundlev, ctx->pairlist, ctx->pair_idx, ctx->rg_tot, ctx->rp_tot, ctx->rp_start, ctx->plan_pair.d_segs, ctx->plan_pair.d_chains, ctx->plan_head.d_segs, ctx->plan_head.d_chains, ctx->plan_ridge.d_segs, ctx->plan_ridge.d_chains,
                  ctx->plan_round.d_segs, ctx->plan_round.d_chains};
    for (void* q : ps) if (q) (void)hipFree(q);
    ctx->sq_start = ctx->sq_end = ctx->sq_total = nullptr; ctx->sq_mismatch = nullptr; ctx->sq_conv = nullptr; ctx->headlist = ctx->roundlist = nullptr; ctx->headq = nullptr;
    ctx->Of = ctx->Ef = ctx->Mtab = ctx->objT = nullptr; ctx->inset = nullptr; ctx->sq_cap = ctx->sq_total_cap = ctx->objT_cap = 0;
    ctx->obj_start = ctx->rg_start = nullptr; ctx->obj_start_cap = ctx->rg_start_cap = 0; ctx->obj_warm = ctx->rg_warm = false;
    ctx->obj_partial = nullptr; ctx->obj_partial_cap = 0;
    ctx->plan_head = hmx_ctx::SeqPlan(); ctx->plan_ridge = hmx_ctx::SeqPlan(); ctx->plan_round = hmx_ctx::SeqPlan(); ctx->plan_pair = hmx_ctx::SeqPlan();
    ctx->headlev = ctx->roundlev = nullptr; ctx->pairlist = ctx->pair_idx = nullptr; ctx->rg_tot = ctx->rp_tot = ctx->rp_start = nullptr; ctx->npairs = 0; ctx->rp_warm = false;
  }
  if (ctx->h_obj) { (void)hipHostFree(ctx->h_obj); ctx->h_obj = nullptr; ctx->obj_cap = 0; }
  if (ctx->obj_event) { (void)hipEventDestroy(ctx->obj_event); ctx->obj_event = nullptr; }
  ctx->obj_pending = 0; ctx->obj_harmony_pending = false;
}
template <class T> int h2d(hmx_ctx* ctx, T* dst, const T* src, size_t count) {
  if (count) HIPCHK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->L.stream));
  HIPCHK(hipStreamSynchronize(ctx->L.stream));
  return 0;
}
template <class T> int d2h(hmx_ctx* ctx, T* dst, const T* src, size_t count) {
  if (count) HIPCHK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, ctx->L.stream));
  HIPCHK(hipStreamSynchronize(ctx->L.stream));
  return 0;
}
int allreduce(hmx_ctx* ctx, void* buf, int64_t count, int dtype) {
  if (ctx->world <= 1 && !ctx->comm_force) return 0;
  // small buffers go through the peers' inboxes when they are connected and tested (k_p2p_allreduce: one launch, one trip over xGMI, no ring):
  // everything but the big ridge statistics of many-level designs.  HMX_P2P_AR=0: always the communicator / hook.
  if (ctx->p2p_on && ctx->p2p_world == ctx->world && !ctx->comm_force && ctx->ran_setup && ctx->D.chain_ctl) {
    static const bool off = [] { const char* e = getenv("HMX_P2P_AR"); return e && atoi(e) == 0; }();
    static const bool big_off = [] { const char* e = getenv("HMX_P2P_AR_BIG"); return e && atoi(e) == 0; }();      // (0: buffers above P2P_CAP entries go to the communicator / hook as in round 4)
    if (!off && (count <= (int64_t)P2P_CAP || !big_off)) {
      Dev T = ctx->D; T.p2p_world = ctx->p2p_world; T.p2p_rank = ctx->p2p_rank;
      for (int g = 0; g < 8; g++) T.p2p_inbox[g] = ctx->p2p_peer[g];
      if (count <= (int64_t)P2P_CAP) {
        l_p2p_allreduce(ctx->L, T, buf, (int)count, dtype, ctx->p2p_ar_seq++, ctx->D.chain_ctl + 1);
        ctx->p2p_ar_calls++;
      } else {
        // big buffers (ridge statistics of many-level designs): reduce-scatter + all-gather through the inboxes, a window at a time
        const int64_t win = (int64_t)(P2P_CAP / 2) * ctx->p2p_world;
        for (int64_t off0 = 0; off0 < count; off0 += win) {
          l_p2p_allreduce_big(ctx->L, T, (char*)buf + 8 * off0, (int)std::min(win, count - off0), dtype, ctx->p2p_ar_seq++, ctx->D.chain_ctl + 1);
          ctx->p2p_ar_calls++; ctx->p2p_ar_big_windows++;
        }
      }
      if (hipGetLastError() != hipSuccess) return fail(ctx, HMX_ERR_COMM, "inbox all-reduce launch failed");
      return 0;
    }
  }
  ctx->comm_calls++; ctx->comm_bytes += count * 8;
  if (ctx->ar) {
    int st = ctx->ar(ctx->ar_user, buf, count, dtype, (void*)ctx->L.stream);
    if (st) return fail(ctx, HMX_ERR_COMM, "all-reduce callback failed");
    return 0;
  }
  // (a handle that has inboxes but neither a communicator nor a hook -- hmx_p2p_connect by hand -- gets here with what the inboxes do not take:
  //  the collectives of hmx_setup, buffers above P2P_CAP entries, everything under HMX_P2P_AR=0.  That is an error, not an RCCL call on a null communicator.)
  if (!ctx->comm) return fail(ctx, HMX_ERR_COMM, ctx->p2p_on ? "this collective does not go through the peer inboxes (before setup / HMX_P2P_AR=0): the handle also needs hmx_comm_init or an all-reduce hook"
                                                            : "sharded handle without hmx_comm_init or an all-reduce hook");
  RcclApi* api = rccl_api(nullptr);
  if (!api || !api->AllReduce) return fail(ctx, HMX_ERR_COMM, "librccl is not loadable");
  ncclResult_t r = api->AllReduce(buf, buf, (size_t)count, dtype == 1 ? ncclFloat64 : ncclInt64,
                                  dtype == 2 ? ncclMin : ncclSum, ctx->comm, ctx->L.stream);
  if (r != ncclSuccess) return fail(ctx, HMX_ERR_COMM, std::string("ncclAllReduce: ") + (api->GetErrorString ? api->GetErrorString(r) : "error"));
  return 0;
}
#define CHK(expr) do { int s_ = (expr); if (s_) return s_; } while (0)
#define KCHK() HIPCHK(hipGetLastError())

// profile mode only: bracket a group of launches with an event pair tagged `name`
struct PhaseScope {
  hmx_ctx* c; int idx = -1;
  PhaseScope(hmx_ctx* ctx, const char* name) : c(ctx) {
    if (c->profile < 2) return;      // (an event record is a barrier packet of its own: ~2-5 us each between dependent kernels -- 200 of them per run were 0.4-1 ms of a 15 ms run)
    int id = -1;
    for (size_t i = 0; i < c->ph_names.size(); i++) if (c->ph_names[i] == name) { id = (int)i; break; }
    if (id < 0) { id = (int)c->ph_names.size(); c->ph_names.push_back(name); }
    if (c->ph_used == c->ph_pool.size()) {
      hmx_ctx::PhaseEv e; e.name = id;
      if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
      c->ph_pool.push_back(e);
    }
    idx = (int)c->ph_used++;
    c->ph_pool[idx].name = id;
    (void)hipEventRecord(c->ph_pool[idx].a, c->L.stream);
  }
  ~PhaseScope() { if (idx >= 0) (void)hipEventRecord(c->ph_pool[idx].b, c->L.stream); }
};
// profile mode: the dominant kernel's launches carry their own start / stop events (hipExtLaunchKernelGGL: the timestamps of the dispatch
// packet itself, no barrier packets around it)
int launch_with_events(hmx_ctx* ctx, Launch& L) {
  L = ctx->L;
  if (!ctx->profile) return 0;
  if (ctx->ev_used == ctx->ev_pool.size()) { hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); ctx->ev_pool.emplace_back(a, b); }
  L.ev0 = ctx->ev_pool[ctx->ev_used].first; L.ev1 = ctx->ev_pool[ctx->ev_used].second;
  ctx->ev_used++;
  return 0;
}
void resolve_phases(hmx_ctx* c) {
  if (!c->ph_used) return;
  (void)hipStreamSynchronize(c->L.stream);
  for (size_t i = 0; i < c->ph_used; i++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ph_pool[i].a, c->ph_pool[i].b) == hipSuccess) c->gpu_timers[c->ph_names[c->ph_pool[i].name]] += ms;
  }
  c->ph_used = 0;
}

void normalise_cols(std::vector<float>& Y, int d, int K) {  // arma::normalise(Y, 2, 0)
  for (int k = 0; k < K; k++) {
    float s = 0.f;
    for (int j = 0; j < d; j++) s += Y[(size_t)k * d + j] * Y[(size_t)k * d + j];
    float nrm = std::sqrt(s); if (nrm == 0.f) nrm = 1.f;
    for (int j = 0; j < d; j++) Y[(size_t)k * d + j] /= nrm;
  }
}
int upload_Y(hmx_ctx* ctx) {  // host Y[k*d+j] -> device Ycur, from it (k_y_images) Yt[j*K+k], the MFMA B-operand images and ||y||^2
  CHK(h2d(ctx, ctx->D.Ycur, ctx->Y.data(), ctx->Y.size()));
  l_y_images(ctx->L, ctx->D, nullptr, 0); KCHK();
  return 0;
}

// objective snapshot obj[2..4] -> the four series (src/harmony.cpp:165-168).  The copy is enqueued into a pinned slot;
// flush_objectives() waits for the last copy only (an event, not the stream: later kernels keep running) and appends.
int flush_objectives(hmx_ctx* ctx) {
  if (!ctx->obj_pending) return 0;
  HIPCHK(hipEventSynchronize(ctx->obj_event));
  const float norm_const = 2000 / ((float)ctx->N_global);
  for (int i = 0; i < ctx->obj_pending; i++) {
    const double* o = ctx->h_obj + 4 * i;
    const int chain_err = (int)o[3] & 15, solve_err = (int)o[3] >> 4;
    if (solve_err) { ctx->obj_pending = 0; return fail(ctx, HMX_ERR_SOLVE, "singular ridge system (moe_correct_ridge_cpp; reported with the objective of the clustering round that followed it)"); }
    if (chain_err) { ctx->obj_pending = 0; return fail(ctx, HMX_ERR_DEVICE, "persistent block chain: a workgroup timed out waiting for its peers (code " + std::to_string(chain_err) + ")"); }
    if (ctx->obj_arith) {   // the three sums are the reference's fp32 accumulators: combined in fp32 as well (:165-168)
      const float a = (float)o[0], b = (float)o[1], c = (float)o[2];
      ctx->obj_kmeans.push_back(((a + b) + c) * norm_const);
      ctx->obj_dist.push_back(a * norm_const); ctx->obj_entropy.push_back(b * norm_const); ctx->obj_cross.push_back(c * norm_const);
      continue;
    }
    ctx->obj_kmeans.push_back((float)((o[0] + o[1] + o[2]) * norm_const));
    ctx->obj_dist.push_back((float)(o[0] * norm_const));
    ctx->obj_entropy.push_back((float)(o[1] * norm_const));
    ctx->obj_cross.push_back((float)(o[2] * norm_const));
  }
  ctx->obj_pending = 0;
  if (ctx->obj_harmony_pending) { ctx->obj_harmony.push_back(ctx->obj_kmeans.back()); ctx->obj_harmony_pending = false; }
  return 0;
}
int objective_slot(hmx_ctx* ctx, double** slot) {     // next pinned host slot of the objective series (4 doubles)
  if (!ctx->h_obj) {
    ctx->obj_cap = 64;
    HIPCHK(hipHostMalloc((void**)&ctx->h_obj, sizeof(double) * 4 * ctx->obj_cap, hipHostMallocDefault));
    std::memset(ctx->h_obj, 0, sizeof(double) * 4 * ctx->obj_cap);
    HIPCHK(hipEventCreateWithFlags(&ctx->obj_event, hipEventDisableTiming));
  }
  if (ctx->obj_pending == ctx->obj_cap) CHK(flush_objectives(ctx));
  *slot = ctx->h_obj + 4 * ctx->obj_pending;
  return 0;
}
int push_objective(hmx_ctx* ctx) {
  if (!ctx->h_obj) {
    ctx->obj_cap = 64;
    HIPCHK(hipHostMalloc((void**)&ctx->h_obj, sizeof(double) * 4 * ctx->obj_cap, hipHostMallocDefault));
    std::memset(ctx->h_obj, 0, sizeof(double) * 4 * ctx->obj_cap);
    HIPCHK(hipEventCreateWithFlags(&ctx->obj_event, hipEventDisableTiming));
  }
  if (ctx->obj_pending == ctx->obj_cap) CHK(flush_objectives(ctx));
  HIPCHK(hipMemcpyAsync(ctx->h_obj + 4 * ctx->obj_pending, ctx->D.obj + 2, sizeof(double) * 4, hipMemcpyDeviceToHost, ctx->L.stream));   // dist, entropy, cross, chain error word
  HIPCHK(hipEventRecord(ctx->obj_event, ctx->L.stream));
  ctx->obj_pending++;
  return 0;
}

// R, O, E from scratch (src/harmony.cpp:141-150, :221-227); leaves objective partials in obj[0..1]
int prepare_round(hmx_ctx* ctx, uint64_t round);
int oe_head(hmx_ctx* ctx);
int objective_snapshot(hmx_ctx* ctx, const Dev* Dterms = nullptr);
int head_pass(hmx_ctx* ctx, bool normalise = false) {   // normalise: Z_corr <- normalise(Z_corr) first (:220)
  // With the round-to-round carry (update_R) the head of cluster_cpp runs over the padded order of the round that FOLLOWS it and
  // files its R sums as that round's old contributions: no pass over R between the head and the first round either.
  const bool sharded_ = ctx->world > 1 || ctx->comm_force;
  // (the head of init_cluster_cpp too: the first round then finds its old contributions filed as well -- no pass over R at all)
  const bool gather = ctx->carry_ok && ctx->injected.empty() && ctx->rng_mode == 0 && ctx->D.upd_impl == 0 &&
                      ctx->D.tile_impl && (size_t)ctx->D.NQ * ctx->D.NS * 1024 <= 160 * 1024;
  if (gather) {
    PhaseScope ph(ctx, "randomize");
    CHK(prepare_round(ctx, ctx->round_counter));     // (update_R finds this round sorted and the next one in flight)
  }
  (void)sharded_;
  Dev D = ctx->D;
  ctx->head_is_stale = false;      // (dist_mat is recomputed here)
  const bool tiles = D.tile_impl && (size_t)D.NQ * D.NS * 1024 <= 160 * 1024;
  // the register-pipelined head (two accumulator sets, rows of a tile in registers) normalises the rows it has loaded anyway
  const bool fused_norm = normalise && tiles && D.NT4 <= 4 && D.NCT <= 7 && D.upd_wps != 4;
  if (normalise && !fused_norm) { l_normalize(ctx->L, D.Zc, D.n, D.d, D.zs); KCHK(); }
  D.head_norm = fused_norm ? 1 : 0;
  for (int i = 0; i < 2; i++) if (ctx->sold_state[i] == 2) ctx->sold_state[i] = 1;     // R is rewritten: carried old contributions are void
  D.head_gather = 0; D.Sold_head = nullptr;
  if (gather && ctx->sorted_nxt[ctx->round_counter & ctx->oset_mask] && ctx->sorted_round[ctx->round_counter & ctx->oset_mask] == (int64_t)ctx->round_counter) {
    const int cur = ctx->sold_cur;
    if (ctx->sold_state[cur] != 0)
      HIPCHK(hipMemsetAsync(ctx->sold_buf[cur], 0, sizeof(long long) * (size_t)D.nb * D.B * D.K, ctx->L.stream));
    D.head_gather = 1; D.Sold_head = ctx->sold_buf[cur];
    ctx->sold_state[cur] = 2; ctx->sold_round[cur] = (int64_t)ctx->round_counter; ctx->sold_seed[cur] = ctx->seed;
    // the head of cluster_cpp is followed, inside the same call, by a round that rewrites every R row and takes its old contributions from
    // the sums filed here: the head's own rows are never read (Dev::r_store) -- 4K bytes per cell less.  (init_cluster_cpp's head is followed
    // by the caller, who may read R: it stores.)
    if (normalise && ctx->max_iter_kmeans >= 1 && !ctx->poll && !ctx->r_store_always) D.r_store = 0;
  }
  ctx->R_valid = false;
  HIPCHK(hipMemsetAsync(D.O_fx, 0, sizeof(long long) * D.B * D.K, ctx->L.stream));
  HIPCHK(hipMemsetAsync(D.Snew_fx, 0, sizeof(long long) * (size_t)D.nrep * D.B * D.K, ctx->L.stream));
  HIPCHK(hipMemsetAsync(D.objpart, 0, sizeof(double) * 2 * (size_t)D.objslots * D.nwmax, ctx->L.stream));
  if (D.tile_impl && (size_t)D.NQ * D.NS * 1024 <= 160 * 1024) {
    l_tile_static(ctx->L, D, 1); KCHK();      // MFMA tiles; O contributions land in the Snew replicas
    l_fold(ctx->L, D, -1, 0); KCHK();         // O = sum of the replicas
  } else {
    l_head(ctx->L, D, 0); KCHK();
  }
  l_obj_reduce(ctx->L, D); KCHK();
  CHK(allreduce(ctx, D.O_fx, (int64_t)D.B * D.K, 0));
  CHK(allreduce(ctx, D.obj, 2, 1));
  if (ctx->oe_arith) CHK(oe_head(ctx));        // E = sum(R, 1) Pr_b^T, O = R Phi^T as the reference sums them (:149-150)
  ctx->R_valid = D.r_store != 0;
  return 0;
}

bool check_convergence_impl(hmx_ctx* c, int type) {  // src/harmony.cpp:173-205
  float obj_new, obj_old;
  if (type == 0) {
    obj_old = 0; obj_new = 0;
    for (int i = 0; i < c->window_size; i++) {
      obj_old += c->obj_kmeans[c->obj_kmeans.size() - 2 - i];
      obj_new += c->obj_kmeans[c->obj_kmeans.size() - 1 - i];
    }
    return std::fabs(obj_old - obj_new) / std::fabs(obj_old) < c->eps_k;
  } else if (type == 1) {
    obj_old = c->obj_harmony[c->obj_harmony.size() - 2];
    obj_new = c->obj_harmony[c->obj_harmony.size() - 1];
    return (obj_old - obj_new) / std::fabs(obj_old) < c->eps_h;
  }
  return true;
}

// Results of the last device-side correction that live on the host only on demand: flags (singular system -> error; subset /
// skipped counts), W of the last solved cluster, the centroids.
int sync_solve_results(hmx_ctx* ctx) {
  if (ctx->solve_pending) {
    ctx->solve_pending = false;
    const int K = ctx->K, d = ctx->d;
    std::vector<int> flags(K), mrows(K);
    CHK(d2h(ctx, flags.data(), ctx->sv_flags, (size_t)K)); CHK(d2h(ctx, mrows.data(), ctx->sv_mrows, (size_t)K));
    ctx->subset_clusters = ctx->skipped_clusters = 0;
    int last = -1;
    for (int k = 0; k < K; k++) {
      if (flags[k] & 4) return fail(ctx, HMX_ERR_SOLVE, "singular ridge system");
      if (flags[k] & 1) ctx->subset_clusters++;
      if (flags[k] & 2) ctx->skipped_clusters++; else last = k;
    }
    if (last >= 0) {   // the reference's W field holds the last cluster's coefficients (:592-611)
      const int m = mrows[last];
      ctx->W.resize((size_t)m * d); ctx->W_rows = m;
      CHK(d2h(ctx, ctx->W.data(), ctx->sv_Wall + (size_t)last * d * ((size_t)ctx->B + 1), (size_t)m * d));
    }
  }
  if (ctx->y_on_device) { ctx->y_on_device = false; CHK(d2h(ctx, ctx->Y.data(), ctx->D.Ycur, ctx->Y.size())); }
  return 0;
}

// R-compatible mode: the generator is seeded like set.seed(seed) once per run (hmx_restart re-arms it)
void ensure_rrng(hmx_ctx* ctx) {
  if (!ctx->rrng_seeded) { ctx->rrng.set_seed((uint32_t)ctx->seed); ctx->rrng_seeded = true; }
}

// ---- kmeans_centers (src/utils.cpp:10-64) ---------------------------------------------------
int gather_centres(hmx_ctx* ctx, const std::vector<long long>& gcells, long long* d_gcells, double* d_rows) {
  const int K = ctx->K, d = ctx->d;
  CHK(h2d(ctx, d_gcells, gcells.data(), (size_t)K));
  l_gather_rows(ctx->L, ctx->D, d_gcells, (uint64_t)ctx->goff, d_rows); KCHK();
  CHK(allreduce(ctx, d_rows, (int64_t)K * d, 1));
  l_y_images(ctx->L, ctx->D, d_rows, 0); KCHK();       // rows -> Ycur, Yt, the MFMA images, ||y||^2: no trip to the host
  ctx->y_on_device = true;                              // (the host copy is fetched when somebody asks for it)
  return 0;
}

int kmeans_centers(hmx_ctx* ctx) {
  const int K = ctx->K, d = ctx->d;
  const Dev& D = ctx->D;
  ctx->Y.assign((size_t)d * K, 0.f);
  long long* const d_gcells = D.km_gcells; double* const d_rows = D.km_rows; unsigned* const d_excl = D.km_excl;   // allocated once in hmx_setup
  // random anchors (:12-15): indices = floor(randu * (N-1))
  std::vector<long long> gcells(K);
  const float Nm1 = (float)((uint64_t)ctx->N_global - 1);
  const bool rmode = ctx->rng_mode == 1;
  if (rmode) ensure_rrng(ctx);
  for (int i = 0; i < K; i++) gcells[i] = (long long)std::floor((rmode ? ctx->rrng.arma_randu() : hmx_u01(ctx->seed, 0, (uint64_t)i)) * Nm1);
  CHK(gather_centres(ctx, gcells, d_gcells, d_rows));
  std::vector<unsigned long long> win(K), sentinel(K, SEED_SENTINEL);
  std::set<unsigned> sup;
  if (rmode) {
    // R-compatible stream: anchor i consumes N_global uniforms in cell order (VECTYPE random_numbers(size(distances), randu), :29).
    // The host draws them (MT19937 or the host's unif_rand callback), a batch of anchors at a time; the race itself
    // (-log(u) / |2(1 - y_i.x)|, index_min) runs on the device over this shard's cells.
    const int64_t n = ctx->N;
    int A = (int)std::max<int64_t>(1, std::min<int64_t>(K, (int64_t)(64ll << 20) / (4 * std::max<int64_t>(n, 1))));
    A = std::max(1, std::min(A, 12288 / d));   // anchor rows of a batch live in LDS (<= 48 KB)
    float* d_u; HIPCHK(hipMalloc((void**)&d_u, sizeof(float) * (size_t)A * (size_t)n));
    std::vector<float> hu((size_t)A * (size_t)n);
    int st = 0;
    for (int a0 = 0; a0 < K && !st; a0 += A) {
      const int na = std::min(A, K - a0);
      for (int a = 0; a < na; a++)
        for (int64_t g = 0; g < ctx->N_global; g++) {
          const float u = ctx->rrng.arma_randu();
          if (g >= ctx->goff && g < ctx->goff + n) hu[(size_t)a * n + (size_t)(g - ctx->goff)] = u;
        }
      st = h2d(ctx, d_u, hu.data(), (size_t)na * (size_t)n);
      if (!st) st = h2d(ctx, D.seedmin, sentinel.data(), (size_t)K);
      if (!st) { l_seed_race_u(ctx->L, D, d_u, a0, na, 0, (uint64_t)ctx->goff, nullptr, 0); if (hipGetLastError() != hipSuccess) st = fail(ctx, HMX_ERR_DEVICE, "seed race launch failed"); }
      if (!st) st = allreduce(ctx, D.seedmin, K, 2);
      if (!st) st = d2h(ctx, win.data(), D.seedmin, (size_t)K);
      for (int i = a0; i < a0 + na && !st; i++) {   // duplicates in cluster order (:38-43)
        if (win[i] == SEED_SENTINEL) { st = fail(ctx, HMX_ERR_STATE, "centroid seeding found no candidate cell"); break; }
        unsigned g = (unsigned)(win[i] & 0xffffffffu);
        if (sup.count(g)) {
          std::vector<unsigned> ex(sup.begin(), sup.end());
          st = h2d(ctx, d_excl, ex.data(), ex.size());
          if (!st) st = h2d(ctx, D.seedmin, sentinel.data(), (size_t)K);
          if (!st) { l_seed_race_u(ctx->L, D, d_u, a0, na, i - a0, (uint64_t)ctx->goff, d_excl, (int)ex.size()); if (hipGetLastError() != hipSuccess) st = fail(ctx, HMX_ERR_DEVICE, "seed race launch failed"); }
          if (!st) st = allreduce(ctx, D.seedmin, K, 2);
          std::vector<unsigned long long> w2(K);
          if (!st) st = d2h(ctx, w2.data(), D.seedmin, (size_t)K);
          if (!st && w2[i] == SEED_SENTINEL) st = fail(ctx, HMX_ERR_STATE, "centroid seeding ran out of distinct cells");
          g = (unsigned)(w2[i] & 0xffffffffu);
        }
        sup.insert(g);
        gcells[i] = (long long)g;
      }
    }
    (void)hipFree(d_u);
    if (st) return st;
  } else {
  // exponential race for every anchor in one pass (:24-34)
  CHK(h2d(ctx, D.seedmin, sentinel.data(), (size_t)K));
  // the race on the matrix cores (k_tile mode 3) when the centroid image fits; else the cluster-lane VALU kernel
  const bool seed_tile = D.tile_impl && (size_t)D.NQ * D.NS * 1024 <= 150 * 1024;
  auto seed_probe = [&](const unsigned* excl, int nexcl) -> int {
    if (seed_tile) {
      ctx->D.seed_key = ctx->seed; ctx->D.seed_goff = (unsigned long long)ctx->goff; ctx->D.seed_excl = excl; ctx->D.seed_nexcl = nexcl;
      l_tile_static(ctx->L, ctx->D, 3);
    } else l_seed_probe(ctx->L, D, ctx->seed, (uint64_t)ctx->goff, excl, nexcl);
    KCHK();
    return 0;
  };
  CHK(seed_probe(nullptr, 0));
  CHK(allreduce(ctx, D.seedmin, K, 2));
  CHK(d2h(ctx, win.data(), D.seedmin, (size_t)K));
  // duplicates are resolved in cluster order (:38-43): re-sample cluster i among cells not yet chosen
  for (int i = 0; i < K; i++) {
    unsigned g = (unsigned)(win[i] & 0xffffffffu);
    if (win[i] == SEED_SENTINEL) return fail(ctx, HMX_ERR_STATE, "centroid seeding found no candidate cell");
    if (sup.count(g)) {
      std::vector<unsigned> ex(sup.begin(), sup.end());
      CHK(h2d(ctx, d_excl, ex.data(), ex.size()));
      CHK(h2d(ctx, D.seedmin, sentinel.data(), (size_t)K));
      CHK(seed_probe(d_excl, (int)ex.size()));
      CHK(allreduce(ctx, D.seedmin, K, 2));
      std::vector<unsigned long long> w2(K);
      CHK(d2h(ctx, w2.data(), D.seedmin, (size_t)K));
      if (w2[i] == SEED_SENTINEL) return fail(ctx, HMX_ERR_STATE, "centroid seeding ran out of distinct cells");
      g = (unsigned)(w2[i] & 0xffffffffu);
    }
    sup.insert(g);
    gcells[i] = (long long)g;
  }
  }
  ctx->seed_cells.assign(gcells.begin(), gcells.end());   // diagnostics: hmx_get("seed_cells")
  CHK(gather_centres(ctx, gcells, d_gcells, d_rows));
  // 10 x one Lloyd iteration (:53-64); the centre update runs on the device, no host round trip per iteration
  const bool tile_ok = D.tile_impl && (size_t)D.NQ * D.NS * 1024 + ((size_t)K * d + K) * 8 <= 160 * 1024;
  for (int it = 0; it < 10; it++) {
    HIPCHK(hipMemsetAsync(D.lsum, 0, sizeof(long long) * ((size_t)K * d + K), ctx->L.stream));   // sums + counts: one buffer
    if (tile_ok) { l_tile_static(ctx->L, D, 2); KCHK(); }
    else { l_lloyd(ctx->L, D); KCHK(); }
    CHK(allreduce(ctx, D.lsum, (int64_t)K * d + K, 0));   // sums and counts in one collective
    l_lloyd_finish(ctx->L, D); KCHK();
  }
  ctx->y_on_device = true;       // (Ycur holds the centres; the host copy follows on demand)
  return 0;
}

// The round's block order: block id per cell (Feistel bijection of (seed, round), or a host-injected shuffle) + the padded
// counting sort.  Touches no algorithmic state (only blk / lorder / lcombo / lpair / boff).  (Enqueuing it speculatively
// for the NEXT round before the host waits for this round's objective was measured: no gain, 26.1 vs 25.9 ms per step.)
void apply_set(Dev& D, const hmx_ctx::SortSet& s) {
  D.blk = s.blk; D.lorder = s.lorder; D.lpair = s.lpair; D.lcombo = s.lcombo; D.boff = s.boff; D.binoff = s.binoff; D.counts = s.counts; D.offs = s.offs; D.blkv = s.blkv; D.bincnt = s.bincnt;
}
// ---- sort_sched = 3 -----------------------------------------------------------------------------------------------------------
// rounds first..(first | 3) in one batch of launches on the main stream, into sets round & 3.
// (Sorting a group AHEAD on the side stream was measured twice and lost twice: next to the persistent block chain every block step got
//  1 us slower (14.2 vs 14.1 ms per run); in the shadow of the correction's statistics pass that pass went from 1.47 to 2.25 ms per run for
//  0.5 ms of sort taken off the main stream.  The sort's thousands of one-wave workgroups get in the way of whatever runs beside them.)
int sort_group(hmx_ctx* ctx, uint64_t first) {
  const int nr = 4 - (int)(first & 3);
  SortBatch Sb{};
  Dev Dt = ctx->D; Dt.nxt = ctx->carry_ok ? 1 : 0;
  if (ctx->shuf_inv) {
    ShufSets T{};
    for (int r = 0; r < nr; r++) {
      const int os = (int)((first + (uint64_t)r) & 3);
      const hmx_ctx::SortSet& t = ctx->sets[os];
      T.posr[r] = ctx->posr[os]; T.lpair[r] = t.lpair; T.lorder[r] = t.lorder; T.lcombo[r] = t.lcombo; T.boff[r] = t.boff; T.partcnt[r] = ctx->shuf_partcnt[os]; T.binbase[r] = t.binoff; T.bincnt[r] = t.bincnt; T.binacc[r] = ctx->shuf_binacc[os];
    }
    l_shuffle_inv(ctx->L, Dt, T, nr, ctx->seed, first, (uint64_t)ctx->N_global, (uint64_t)ctx->goff, ctx->cells_per_block); KCHK();
  } else {
  for (int r = 0; r < nr; r++) {
    const hmx_ctx::SortSet& t = ctx->sets[(first + (uint64_t)r) & 3];
    Sb.p[r] = SortPtrs{t.blk, t.blkv, t.counts, t.offs, t.binoff, t.bincnt, t.boff, t.lorder, t.lcombo, t.lpair};
  }
  l_sort_batch(ctx->L, Dt, Sb, nr, ctx->seed, first, (uint64_t)ctx->N_global, (uint64_t)ctx->goff, ctx->cells_per_block); KCHK();
  }
  for (int r = 0; r < nr; r++) {
    const int os = (int)((first + (uint64_t)r) & 3);
    ctx->sorted_round[os] = (int64_t)first + r; ctx->sorted_seed[os] = ctx->seed; ctx->sorted_nxt[os] = Dt.nxt != 0; ctx->sorted_on_side[os] = false;
  }
  return 0;
}
int prepare_round(hmx_ctx* ctx, uint64_t round) {
  Dev& D = ctx->D;
  const int sset = (int)(round & (uint64_t)ctx->oset_mask);
  const bool host_order = !ctx->injected.empty() || ctx->rng_mode == 1;
  if (ctx->sort_sched == 3 && !host_order) {
    if (!(ctx->sorted_round[sset] == (int64_t)round && ctx->sorted_seed[sset] == ctx->seed)) CHK(sort_group(ctx, round));    // this round and the rest of its group
    apply_set(D, ctx->sets[sset]);
    D.nxt = ctx->sorted_nxt[sset] ? 1 : 0;
    return 0;
  }
  if (ctx->sorted_on_side[sset]) {   // a prefetch into this set is (or was) in flight on the side stream: order the main stream behind it
    HIPCHK(hipStreamWaitEvent(ctx->L.stream, ctx->ev_sorted[sset], 0));
    ctx->sorted_on_side[sset] = false;
  }
  apply_set(D, ctx->sets[sset]);
  const bool have = !host_order && ctx->sorted_round[sset] == (int64_t)round && ctx->sorted_seed[sset] == ctx->seed;
  auto prefetch_next = [&]() -> int {   // round + 1 into the other set, on the side stream, behind everything that still reads that set
    if (!ctx->sort_overlap || host_order || !ctx->side) return 0;
    const int t = sset ^ 1;
    if (ctx->sorted_round[t] == (int64_t)round + 1 && ctx->sorted_seed[t] == ctx->seed) return 0;     // (second call for this round)
    HIPCHK(hipEventRecord(ctx->ev_free[t], ctx->L.stream));
    HIPCHK(hipStreamWaitEvent(ctx->side, ctx->ev_free[t], 0));
    Dev Dt = D; apply_set(Dt, ctx->sets[t]);
    Dt.nxt = ctx->carry_ok ? 1 : 0;
    Launch L2 = ctx->L; L2.stream = ctx->side;
    l_sort_blocks(L2, Dt, true, ctx->seed, round + 1, (uint64_t)ctx->N_global, (uint64_t)ctx->goff, ctx->cells_per_block); KCHK();
    HIPCHK(hipEventRecord(ctx->ev_sorted[t], ctx->side));
    ctx->sorted_round[t] = (int64_t)round + 1; ctx->sorted_seed[t] = ctx->seed; ctx->sorted_on_side[t] = true; ctx->sorted_nxt[t] = Dt.nxt != 0;
    return 0;
  };
  if (have) return prefetch_next();
  ctx->sorted_round[sset] = -1;
  bool gen_blocks = false;
  if (ctx->injected.empty() && ctx->rng_mode == 1) {   // update_order = shuffle(linspace(0, N-1, N)) from R's stream (:272-273)
    ensure_rrng(ctx);
    std::vector<int64_t> order;
    ctx->rrng.arma_shuffle(ctx->N_global, order);
    ctx->injected.push_back(std::move(order));
  }
  if (!ctx->injected.empty()) {  // host-provided shuffle: block(g) from its position
    std::vector<int64_t> order = std::move(ctx->injected.front());
    ctx->injected.pop_front();
    std::vector<int> pos_blk((size_t)ctx->N);
    std::vector<int64_t> pos((size_t)ctx->N_global);
    for (int64_t p = 0; p < ctx->N_global; p++) pos[(size_t)order[p]] = p;
    for (int64_t i = 0; i < ctx->N; i++) {
      uint64_t b = (uint64_t)pos[(size_t)(ctx->goff + ctx->perm[i])] / ctx->cells_per_block;
      pos_blk[i] = (int)std::min<uint64_t>(b, (uint64_t)(ctx->nb - 1));
    }
    CHK(h2d(ctx, D.blk, pos_blk.data(), pos_blk.size()));
    ctx->injected_round = (int64_t)round;
  } else gen_blocks = true;   // block ids from the Feistel bijection, computed inside the sort's histogram kernel
  D.nxt = (gen_blocks && ctx->carry_ok) ? 1 : 0;
  l_sort_blocks(ctx->L, D, gen_blocks, ctx->seed, round, (uint64_t)ctx->N_global, (uint64_t)ctx->goff, ctx->cells_per_block); KCHK();
  ctx->sorted_nxt[sset] = D.nxt != 0;
  if (gen_blocks) { ctx->sorted_round[sset] = (int64_t)round; ctx->sorted_seed[sset] = ctx->seed; }
  return prefetch_next();
}

// ================================================================================================================================
// Reference arithmetic: the reference's sequential fp32 accumulators as restarted sequential sums (hmx_seq.hip)
// ================================================================================================================================
template <class T> int seq_grow(hmx_ctx* ctx, T*& p, size_t& cap, size_t need) {
  if (need <= cap && p) return 0;
  if (p) { HIPCHK(hipStreamSynchronize(ctx->L.stream)); (void)hipFree(p); p = nullptr; cap = 0; }
  void* q = nullptr;
  HIPCHK(hipMalloc(&q, std::max<size_t>(need, 1) * sizeof(T)));
  p = (T*)q; cap = need;
  return 0;
}
// chains = (first entry of the list, number of cells); every chain is cut into segments of L cells
int seq_plan_build(hmx_ctx* ctx, hmx_ctx::SeqPlan& P, const std::vector<std::pair<int, int>>& chains, int L) {
  std::vector<SeqSeg> segs; std::vector<SeqChain> ch(chains.size());
  P.seg0.assign(chains.size() + 1, 0);
  for (size_t c = 0; c < chains.size(); c++) {
    ch[c].seg0 = (int)segs.size(); P.seg0[c] = (int)segs.size();
    for (int o = 0; o < chains[c].second; o += L) segs.push_back({chains[c].first + o, std::min(L, chains[c].second - o)});
    ch[c].nseg = (int)segs.size() - ch[c].seg0;
  }
  P.seg0[chains.size()] = (int)segs.size();
  P.nsegs = (int)segs.size(); P.nchains = (int)chains.size(); P.seg_cells = L;
  CHK(seq_grow(ctx, P.d_segs, P.cap_segs, segs.size())); CHK(seq_grow(ctx, P.d_chains, P.cap_chains, ch.size()));
  CHK(h2d(ctx, P.d_segs, segs.data(), segs.size())); CHK(h2d(ctx, P.d_chains, ch.data(), ch.size()));
  return 0;
}
int seq_workspace(hmx_ctx* ctx, size_t seg_floats, size_t total_floats) {
  if (seg_floats > ctx->sq_cap) { size_t c1 = ctx->sq_cap, c2 = ctx->sq_cap; CHK(seq_grow(ctx, ctx->sq_start, c1, seg_floats)); CHK(seq_grow(ctx, ctx->sq_end, c2, seg_floats)); ctx->sq_cap = seg_floats; }
  CHK(seq_grow(ctx, ctx->sq_total, ctx->sq_total_cap, total_floats));
  if (!ctx->sq_mismatch) { size_t c = 0; CHK(seq_grow(ctx, ctx->sq_mismatch, c, 4)); HIPCHK(hipMemsetAsync(ctx->sq_mismatch, 0, 4 * sizeof(unsigned), ctx->L.stream)); ctx->sq_conv = ctx->sq_mismatch + 2; }
  return 0;
}
// after a checked scan (the scan wrote {segments that moved, largest chain-relative move} into sq_conv): did the starts settle?
int seq_settled(hmx_ctx* ctx, bool* settled) {
  unsigned w[2] = {0, 0};
  CHK(d2h(ctx, w, ctx->sq_conv, 2));
  float r; std::memcpy(&r, &w[1], 4);
  *settled = w[0] == 0 || (!ctx->seq_strict && (double)r <= ctx->seq_tol);
  ctx->seq_last_mismatch = w[0];
  ctx->seq_mismatch_sum += w[0]; if ((double)r > ctx->seq_resid_max) ctx->seq_resid_max = (double)r;
  return 0;
}
// one restarted-sum iteration scheme for all users: `pass(p, zero_start)` runs the segments, `scan(p, zero_start, conv)` the scan.  The first
// `seq_passes` passes always run (warm: one less); long chains continue until the starts settled.
template <class PASS, class SCAN> int seq_iterate(hmx_ctx* ctx, int group, bool warm, bool adaptive, PASS pass, SCAN scan) {
  // cold: seq_passes passes from zero starts.  warm: seq_warm_passes passes from the starts the workspace still holds -- NOT one pass less by
  // default: a warm start of a block's put-back sums is as far from the truth as R moved in the block update, and ONE pass from it carries a
  // first-order error (100k cells: Z_corr 5.8e-6 from the oracle instead of 2.2e-6, tools/strict_probe.py); two passes from zero are second order.
  int p = warm ? std::max(0, ctx->seq_passes - ctx->seq_warm_passes) : 0;
  const int p_first = p;
  ctx->seq_group_runs[group]++;
  if (ctx->seq_strict) adaptive = true;
  struct Count { hmx_ctx* c; int g; const int& p; int p0; ~Count() { c->seq_group_passes[g] += p - p0; } } count{ctx, group, p, p_first};
  for (; p < ctx->seq_passes; p++) {
    const bool last = p == ctx->seq_passes - 1 && (adaptive || ctx->seq_stats);     // (short chains: statistics only on request, "seq_stats")
    CHK(pass(p == 0 && !warm, last ? ctx->sq_conv : nullptr));       // (the pass zeroes the statistics words its scan adds to)
    ctx->seq_scan_final = (p == ctx->seq_passes - 1) && !adaptive && !last;      // nobody reads the starts this scan would write: totals only
    CHK(scan(p == 0 && !warm, last ? ctx->sq_conv : nullptr));
    ctx->seq_scan_final = false;
  }
  if (!adaptive) {      // (short chains: three passes are far inside fp32 noise; their last scan's statistics are read when a getter asks)
    return 0;
  }
  bool ok = false;
  CHK(seq_settled(ctx, &ok));
  for (; !ok && p < ctx->seq_max_passes; p++) {
    CHK(pass(false, ctx->sq_conv));
    CHK(scan(false, ctx->sq_conv));
    CHK(seq_settled(ctx, &ok));
    ctx->seq_extra_passes++;
  }
  if (!ok) ctx->seq_unsettled++;       // (seq_max_passes reached with starts still moving: reported, "seq:unsettled")
  return 0;
}
// O / E sums of the chain sets [chain0, chain0 + nchains) of plan P over `list`: per chain set (1 + B) * K sequential fp32 sums
// (row 0: all its cells, row 1 + b: its cells of level b) -> ctx->sq_total[chain][1 + B][K].  passes x (segments in parallel, then the
// scan that hands every segment its start).
// warm: the workspace still holds these segments' starts from a run over (nearly) the same terms -- the first pass starts from them
// instead of from zero, which is worth one pass.
int seq_run_oe(hmx_ctx* ctx, const hmx_ctx::SeqPlan& P, const int* list, const int* poslev, int chain0, int nchains, bool warm = false) {
  const int W = (1 + ctx->B) * ctx->K, lo = P.seg0[chain0], n = P.seg0[chain0 + nchains] - lo;
  CHK(seq_workspace(ctx, (size_t)P.nsegs * W, (size_t)P.nchains * W));
  int longest = 0;
  for (int c = chain0; c < chain0 + nchains; c++) longest = std::max(longest, P.seg0[c + 1] - P.seg0[c]);
  const bool adaptive = (int64_t)longest * P.seg_cells >= ctx->seq_adaptive_cells;
  CHK(seq_iterate(ctx, 0, warm, adaptive,
                  [&](bool zero, unsigned* cz) -> int { l_seq_oe_pass(ctx->L, ctx->D, list, poslev, (int)ctx->N, P.d_segs, lo, n, ctx->sq_start, ctx->sq_end, zero ? 1 : 0, cz); KCHK(); return 0; },
                  [&](bool zero, unsigned* conv) -> int { l_seq_scan(ctx->L, P.d_chains, chain0, nchains, W, ctx->sq_start, ctx->sq_end, ctx->sq_start, ctx->sq_total, conv, zero ? 1 : 0, ctx->seq_scan_final ? 1 : 0); KCHK(); return 0; }));
  ctx->seq_runs++;
  return 0;
}
// static lists and plans of a handle that runs (part of) the reference's arithmetic; called at the end of hmx_setup
int seq_setup_static(hmx_ctx* ctx) {
  const int n = (int)ctx->N, C = ctx->C, B = ctx->B, K = ctx->K, Q = ctx->Q;
  const bool any = ctx->ridge_arith || ctx->oe_arith || ctx->obj_arith || ctx->solve_arith;
  if (!any) return 0;
  if (ctx->world > 1 || ctx->comm_force) return fail(ctx, HMX_ERR_ARG, "the reference-arithmetic modes (ridge_arith / oe_arith / obj_arith / solve_arith) run on one GPU");
  // headlist: [ cells in original order | for every covariate: cells by (level, original order) ]  (internal cell ids)
  std::vector<int> hl((size_t)(1 + C) * n);
  for (int i = 0; i < n; i++) hl[i] = ctx->invperm_h[i];
  ctx->lev_off.assign(B, 0); ctx->lev_cnt.assign(B, 0);
  for (int c = 0; c < C; c++) {
    const int b0 = c ? ctx->cov_bounds[c - 1] : 0, nl = ctx->B_vec[c];
    std::vector<int> cnt(nl + 1, 0);
    for (int i = 0; i < n; i++) cnt[ctx->qlev[(size_t)ctx->combo_h[ctx->invperm_h[i]] * C + c] - b0 + 1]++;
    for (int l = 0; l < nl; l++) { ctx->lev_cnt[b0 + l] = cnt[l + 1]; cnt[l + 1] += cnt[l]; ctx->lev_off[b0 + l] = (1 + c) * n + cnt[l]; }
    std::vector<int> cur(cnt.begin(), cnt.end() - 1);
    for (int i = 0; i < n; i++) { const int cell = ctx->invperm_h[i]; const int l = ctx->qlev[(size_t)ctx->combo_h[cell] * C + c] - b0; hl[(size_t)(1 + c) * n + cur[l]++] = cell; }
  }
  { size_t cap = 0; CHK(seq_grow(ctx, ctx->headlist, cap, hl.size())); CHK(h2d(ctx, ctx->headlist, hl.data(), hl.size())); }
  if (ctx->ridge_arith) {      // the list's combinations (static): the ridge pass fetches (cell, combination) with two independent loads
    std::vector<int> hq(hl.size());
    for (size_t i = 0; i < hl.size(); i++) hq[i] = ctx->combo_h[hl[i]];
    size_t cap = 0; CHK(seq_grow(ctx, ctx->headq, cap, hq.size())); CHK(h2d(ctx, ctx->headq, hq.data(), hq.size()));
  }
  { size_t c1 = 0, c2 = 0, c3 = 0; CHK(seq_grow(ctx, ctx->Of, c1, (size_t)B * K)); CHK(seq_grow(ctx, ctx->Ef, c2, (size_t)B * K)); CHK(seq_grow(ctx, ctx->Mtab, c3, (size_t)B * K));
    HIPCHK(hipMemsetAsync(ctx->Of, 0, sizeof(float) * (size_t)B * K, ctx->L.stream)); HIPCHK(hipMemsetAsync(ctx->Ef, 0, sizeof(float) * (size_t)B * K, ctx->L.stream)); }
  if (ctx->oe_arith) {
    // head: E = sum(R, 1) Pr_b^T, O = R Phi^T (:149-150): one chain set over all cells in original order
    CHK(seq_plan_build(ctx, ctx->plan_head, {{0, n}}, 256));
    // rounds: one chain set per block over the round's shuffled order; block j = positions [j cpb, (j + 1) cpb), the last takes the rest (:296-300)
    std::vector<std::pair<int, int>> ch;
    for (int j = 0; j < ctx->nb; j++) {
      const int lo = (int)std::min<uint64_t>((uint64_t)n, (uint64_t)j * ctx->cells_per_block);
      const int hi = (j == ctx->nb - 1) ? n : (int)std::min<uint64_t>((uint64_t)n, (uint64_t)(j + 1) * ctx->cells_per_block);
      ch.push_back({lo, hi - lo});
    }
    CHK(seq_plan_build(ctx, ctx->plan_round, ch, 128));
    size_t cap = 0; CHK(seq_grow(ctx, ctx->roundlist, cap, (size_t)n));
    { const int Cl = std::min(C, 4); size_t c1 = 0, c2 = 0; CHK(seq_grow(ctx, ctx->roundlev, c1, (size_t)Cl * n)); CHK(seq_grow(ctx, ctx->headlev, c2, (size_t)Cl * n));
      std::vector<int> hv((size_t)Cl * n);
      for (int i = 0; i < n; i++) for (int c = 0; c < Cl; c++) hv[(size_t)c * n + i] = ctx->qlev[(size_t)ctx->combo_h[ctx->invperm_h[i]] * C + c];
      CHK(h2d(ctx, ctx->headlev, hv.data(), hv.size())); }
  }
  if (ctx->ridge_arith) {
    if (ctx->d > 62) return fail(ctx, HMX_ERR_LIMIT, "ridge_arith = 1 supports d <= 62");
    if (!ctx->solve_on_device) return fail(ctx, HMX_ERR_ARG, "ridge_arith = 1 needs the device-side ridge solve");
    std::vector<std::pair<int, int>> ch; ch.push_back({0, n});      // the intercept row's chain: all (kept) cells in original order
    if (C == 1) { for (int q = 0; q < Q; q++) { const int b = ctx->qlev[q]; ch.push_back({ctx->lev_off[b], ctx->lev_cnt[b]}); } }    // a level's cells, ascending
    else for (int b = 0; b < B; b++) ch.push_back({ctx->lev_off[b], ctx->lev_cnt[b]});      // several covariates: one chain per LEVEL (chain 1 + b)
    CHK(seq_plan_build(ctx, ctx->plan_ridge, ch, 1024));
    size_t cap = 0; CHK(seq_grow(ctx, ctx->inset, cap, (size_t)Q * ((K + 7) / 8 * 8) + 8));
    if (C > 1) {
      // level pairs across covariates: cells stably sorted by (level of c, level of c2), original order inside a pair
      std::vector<int> pl; std::vector<int> pidx((size_t)B * B, -1); std::vector<std::pair<int, int>> pch;
      for (int c = 0; c < C; c++) for (int c2 = c + 1; c2 < C; c2++) {
        std::map<std::pair<int, int>, std::vector<int>> by;
        for (int i = 0; i < n; i++) { const int cell = ctx->invperm_h[i]; const int* lv = &ctx->qlev[(size_t)ctx->combo_h[cell] * C]; by[{lv[c], lv[c2]}].push_back(cell); }
        for (auto& kv : by) {
          pidx[(size_t)kv.first.first * B + kv.first.second] = (int)pch.size();
          pch.push_back({(int)pl.size(), (int)kv.second.size()});
          pl.insert(pl.end(), kv.second.begin(), kv.second.end());
        }
      }
      ctx->npairs = (int)pch.size();
      { size_t c1 = 0, c2 = 0; CHK(seq_grow(ctx, ctx->pairlist, c1, pl.size())); CHK(seq_grow(ctx, ctx->pair_idx, c2, pidx.size()));
        CHK(h2d(ctx, ctx->pairlist, pl.data(), pl.size())); CHK(h2d(ctx, ctx->pair_idx, pidx.data(), pidx.size())); }
      CHK(seq_plan_build(ctx, ctx->plan_pair, pch, 256));
      { size_t c1 = 0, c2 = 0, c3 = 0; CHK(seq_grow(ctx, ctx->rg_tot, c1, (size_t)(1 + B) * K * 64)); CHK(seq_grow(ctx, ctx->rp_tot, c2, (size_t)std::max(ctx->npairs, 1) * K));
        CHK(seq_grow(ctx, ctx->rp_start, c3, (size_t)std::max(ctx->plan_pair.nsegs, 1) * K)); }
    }
  }
  if ((ctx->solve_arith || ctx->oe_arith) && !ctx->solve_on_device) return fail(ctx, HMX_ERR_ARG, "solve_arith / oe_arith need the device-side ridge solve");
  return 0;
}
// E, O of the head in the reference's arithmetic: R has just been rewritten
int oe_head(hmx_ctx* ctx) {
  CHK(seq_run_oe(ctx, ctx->plan_head, ctx->headlist, ctx->headlev, 0, 1));
  l_oe_fold(ctx->L, ctx->D, ctx->Of, ctx->Ef, nullptr, ctx->sq_total, nullptr, 1); KCHK();
  return 0;
}
// compute_objective's three my_accu sums (src/harmony.cpp:160-162) as sequential fp32 chains over K*N terms each -> obj[2..4]
// Dterms: the state the k-means term's distances are taken from (default: the current Z_corr / Y; stale_dist: the snapshot of the last head,
// i.e. the reference's stored dist_mat, src/harmony.cpp:160)
int seq_objective(hmx_ctx* ctx, const Dev& D) {
  const long long nt = (long long)ctx->N * ctx->K;
  constexpr int LSEG = 512;      // (round 5: 2048 -> 512 terms per segment: four times the threads for the thread-per-segment passes, which are latency-bound; measured 256 / 512 / 1024: objective 29.6 / 25.9 / 30.4 ms per run)
  const int nsegs = (int)((nt + LSEG - 1) / LSEG);
  CHK(seq_grow(ctx, ctx->objT, ctx->objT_cap, (size_t)3 * (size_t)nt));
  CHK(seq_workspace(ctx, (size_t)3 * nsegs, 3));
  if ((size_t)3 * nsegs > ctx->obj_start_cap) { CHK(seq_grow(ctx, ctx->obj_start, ctx->obj_start_cap, (size_t)3 * nsegs)); ctx->obj_warm = false; }
  CHK(seq_grow(ctx, ctx->obj_partial, ctx->obj_partial_cap, (size_t)3 * ((nsegs + 255) / 256)));
  const int mat = l_obj_terms(ctx->L, D, ctx->oe_arith ? ctx->Of : nullptr, ctx->oe_arith ? ctx->Ef : nullptr, ctx->Mtab, ctx->objT, nt); KCHK();
  // mat == 1: only R % dist is materialised; the entropy / cross-entropy chains are summed straight from R (k_seq_objr_pass)
  CHK(seq_iterate(ctx, 1, ctx->obj_warm, nt >= ctx->seq_adaptive_cells,
                  [&](bool zero, unsigned* cz) -> int {
                    l_seq_arr_pass(ctx->L, ctx->objT, nt, nt, mat, LSEG, nsegs, ctx->obj_start, ctx->sq_end, zero ? 1 : 0, ctx->obj_partial, cz); KCHK();
                    if (mat == 1) { l_seq_objr_pass(ctx->L, D, ctx->Mtab, nt, LSEG, nsegs, ctx->obj_start, ctx->sq_end, zero ? 1 : 0, ctx->obj_partial); KCHK(); }
                    return 0; },
                  [&](bool zero, unsigned* conv) -> int { l_seq_scan1(ctx->L, 3, nsegs, ctx->obj_start, ctx->sq_end, ctx->obj_start, ctx->sq_total, conv, zero ? 1 : 0, ctx->obj_partial); KCHK(); return 0; }));
  ctx->obj_warm = true;
  l_obj_store(ctx->L, ctx->sq_total, D.obj); KCHK();
  ctx->seq_runs++;
  return 0;
}
// the objective snapshot obj[2..4] from what the last pass over the cells left behind (obj[0..1]: exact per-cell sums)
int objective_snapshot(hmx_ctx* ctx, const Dev* Dterms) {
  l_objective_tables(ctx->L, ctx->D); KCHK();      // (also resets the block chain's control words)
  if (ctx->obj_arith) return seq_objective(ctx, Dterms ? *Dterms : ctx->D);
  if (ctx->oe_arith) { l_obj_cross_f32(ctx->L, ctx->D, ctx->Of, ctx->Ef, ctx->Mtab); KCHK(); }
  return 0;
}
// ridge statistics in the reference's arithmetic (ridge_arith = 1, one covariate): per cluster the intercept row's chain over all kept
// cells in original order (sum(Z_tmp, 1), :599) and a chain per level over its cells (sum(Z_tmp.cols(index[b]), 1), :605-608;
// Phi_Rk * Phi_moe_t, :567) -> S0 / n0, Sq / nq
int seq_ridge_stats(hmx_ctx* ctx) {
  const Dev& D = ctx->D;
  const hmx_ctx::SeqPlan& P = ctx->plan_ridge;
  const int W = ctx->K * 64;
  const bool multi = ctx->C > 1;
  CHK(seq_workspace(ctx, (size_t)P.nsegs * W, (size_t)P.nchains * W));
  float* const tot = multi ? ctx->rg_tot : ctx->sq_total;          // (after the workspace call: it may have re-allocated sq_total)
  if ((size_t)P.nsegs * W > ctx->rg_start_cap) { CHK(seq_grow(ctx, ctx->rg_start, ctx->rg_start_cap, (size_t)P.nsegs * W)); ctx->rg_warm = false; }
  l_seq_inset(ctx->L, D, ctx->oe_arith ? ctx->Of : nullptr, ctx->sv_cov_bounds, ctx->cutoff, ctx->inset); KCHK();
  CHK(seq_iterate(ctx, 2, ctx->rg_warm, ctx->N >= ctx->seq_adaptive_cells,
                  [&](bool zero, unsigned* cz) -> int { l_seq_ridge_pass(ctx->L, D, ctx->headlist, ctx->headq, P.d_segs, 0, P.nsegs, ctx->inset, ctx->rg_start, ctx->sq_end, zero ? 1 : 0, cz); KCHK(); return 0; },
                  [&](bool zero, unsigned* conv) -> int { l_seq_scan(ctx->L, P.d_chains, 0, P.nchains, W, ctx->rg_start, ctx->sq_end, ctx->rg_start, tot, conv, zero ? 1 : 0); KCHK(); return 0; }));
  ctx->rg_warm = true;
  ctx->seq_runs++;
  if (!multi) { l_seq_ridge_store(ctx->L, D, ctx->sq_total); KCHK(); return 0; }
  // several covariates: the level-pair entries of Phi_Rk * Phi_moe_t (:561-568): plain sequential sums of R_k over each pair's cells
  const hmx_ctx::SeqPlan& PP = ctx->plan_pair;
  if (PP.nchains > 0) {
    CHK(seq_workspace(ctx, (size_t)PP.nsegs * ctx->K, 1));
    int longest = 0;
    for (int c = 0; c < PP.nchains; c++) longest = std::max(longest, PP.seg0[c + 1] - PP.seg0[c]);
    CHK(seq_iterate(ctx, 3, ctx->rp_warm, (int64_t)longest * PP.seg_cells >= ctx->seq_adaptive_cells,
                    [&](bool zero, unsigned* cz) -> int { l_seq_sum_pass(ctx->L, D, ctx->pairlist, PP.d_segs, 0, PP.nsegs, ctx->rp_start, ctx->sq_end, zero ? 1 : 0, cz); KCHK(); return 0; },
                    [&](bool zero, unsigned* conv) -> int { l_seq_scan(ctx->L, PP.d_chains, 0, PP.nchains, ctx->K, ctx->rp_start, ctx->sq_end, ctx->rp_start, ctx->rp_tot, conv, zero ? 1 : 0); KCHK(); return 0; }));
    ctx->rp_warm = true;
    ctx->seq_runs++;
  }
  return 0;
}

// update_R with the reference's O / E arithmetic (oe_arith): the tables are fp32 and every block removes / puts back its cells'
// sums exactly as src/harmony.cpp:312-313,329-330 -- sequential fp32 sums in the round's shuffled order, formed first, then one
// subtraction / addition per table entry.  One launch of the tile kernel per block (penalty table from memory), the block chain's
// persistent launch does not apply.  Host-visible shuffle: the order is materialised on the host whatever its source.
int update_R_ref(hmx_ctx* ctx) {
  Dev& D = ctx->D;
  const double t0 = now_ms();
  const int n = (int)ctx->N, B = ctx->B, K = ctx->K, nb = ctx->nb;
  const hmx_ctx::SeqPlan& P = ctx->plan_round;
  ctx->R_valid = false; D.r_store = 1;
  { PhaseScope ph(ctx, "randomize");
    if (!ctx->injected.empty() || ctx->rng_mode == 1) {       // the host owns the shuffle: its order goes to the device as it is
      if (ctx->injected.empty()) { ensure_rrng(ctx); std::vector<int64_t> o; ctx->rrng.arma_shuffle(ctx->N_global, o); ctx->injected.push_back(std::move(o)); }
      const std::vector<int64_t>& order = ctx->injected.front();
      std::vector<int> po((size_t)n);
      for (int p = 0; p < n; p++) po[p] = ctx->invperm_h[(size_t)order[p]];
      CHK(h2d(ctx, ctx->roundlist, po.data(), po.size()));
      { const int Cl = std::min(ctx->C, 4); std::vector<int> lv((size_t)Cl * n);
        for (int p = 0; p < n; p++) for (int c = 0; c < Cl; c++) lv[(size_t)c * n + p] = ctx->qlev[(size_t)ctx->combo_h[po[p]] * ctx->C + c];
        CHK(h2d(ctx, ctx->roundlev, lv.data(), lv.size())); }
    } else { l_ref_posord(ctx->L, D, ctx->seed, ctx->round_counter, (uint64_t)ctx->N_global, ctx->roundlist, ctx->roundlev); KCHK(); }
    CHK(prepare_round(ctx, ctx->round_counter)); }             // the tile kernels' padded block order, from the same shuffle
  ctx->round_counter++;
  HIPCHK(hipMemsetAsync(D.Snew_fx, 0, sizeof(long long) * (size_t)D.nrep * B * K, ctx->L.stream));   // (the kernel's fixed-point sums are not used here)
  const int W = (1 + B) * K;
  { PhaseScope ph(ctx, "EO_update");     // every block's cells are still untouched at this point: the sums each block will remove (:312-313), all at once
    CHK(seq_run_oe(ctx, P, ctx->roundlist, ctx->roundlev, 0, nb)); }
  D.fused_fold = 0; D.Sold_next = nullptr;
  const float* put_back = nullptr;                                   // the sums of the block updated last, still to be added back (:329-330)
  for (int j = 0; j < nb; j++) {
    if (P.seg0[j + 1] == P.seg0[j]) continue;                      // N * block_size rounding can leave trailing empty blocks
    float* tot = ctx->sq_total + (size_t)j * W;
    // one launch: the previous block goes back in, this block comes out, this block's penalty table (:329-330, :312-313, :322)
    { PhaseScope ph(ctx, "EO_update"); l_oe_fold(ctx->L, D, ctx->Of, ctx->Ef, put_back, tot, D.pen, 0); KCHK(); }
    { Launch Le; CHK(launch_with_events(ctx, Le)); l_update(Le, D, j); KCHK(); if (ctx->profile) ctx->prof_update_steps++; }
    { PhaseScope ph(ctx, "EO_update");
      // the same cells in the same order as the sums removed above, their R rows updated: that run's segment starts are this run's first guess
      CHK(seq_run_oe(ctx, P, ctx->roundlist, ctx->roundlev, j, 1, true)); }
    put_back = tot;
  }
  if (put_back) { PhaseScope ph(ctx, "EO_update"); l_oe_fold(ctx->L, D, ctx->Of, ctx->Ef, put_back, nullptr, nullptr, 0); KCHK(); }
  { PhaseScope ph(ctx, "objective");
    l_obj_reduce(ctx->L, D); KCHK();
    CHK(objective_snapshot(ctx)); }
  CHK(push_objective(ctx));
  ctx->sets_clean = false;
  for (int i = 0; i < 2; i++) if (ctx->sold_state[i] == 2) ctx->sold_state[i] = 1;
  if (ctx->profile) ctx->prof_update_cells += ctx->N;
  ctx->R_valid = true;
  ctx->timers["update_R"] += now_ms() - t0;
  return 0;
}

// ---- update_R (src/harmony.cpp:269-342) ---------------------------------------------------------
int update_R(hmx_ctx* ctx) {
  if (ctx->oe_arith) return update_R_ref(ctx);
  Dev& D = ctx->D;
  const bool sharded = ctx->world > 1 || ctx->comm_force;
  const char* fold_env = getenv("HMX_FOLD_IMPL");   // "split": force the two-kernel fold + penalty fallback (tests)
  // k_foldpen (one launch, K/16 workgroups, every thread walks B/16 levels x the replicas) suits small tables; with thousands of
  // entries (configs[4]: 200 levels x 200 clusters) one thread per entry in two launches is faster, unless the fused / chain paths apply
  const bool merged = (size_t)D.B * 128 <= 64 * 1024 && !(fold_env && std::string(fold_env) == "split") &&
                      (ctx->fused_ok || (size_t)D.B * D.K <= 8192 || (fold_env && std::string(fold_env) == "merged"));   // LDS budget of k_foldpen
  const double t0 = now_ms();
  ctx->R_valid = false;
  { PhaseScope ph(ctx, "randomize");      // the round's shuffle (:272-291, timers "randomize")
    CHK(prepare_round(ctx, ctx->round_counter)); }
  ctx->round_counter++;
  // sharded: the chain needs the in-launch exchange over the peers' inboxes (hmx_p2p_*); without it, one launch + one collective per block
  const bool p2p = sharded && ctx->p2p_on && ctx->p2p_world == ctx->world && !ctx->comm_force && (size_t)D.B * D.K <= (size_t)P2P_CAP;
  const bool chain_path = merged && ctx->fused_ok && ctx->chain_ok && (!sharded || p2p);
  D.p2p_world = p2p ? ctx->p2p_world : 0; D.p2p_rank = ctx->p2p_rank;
  for (int g = 0; g < 8; g++) D.p2p_inbox[g] = ctx->p2p_peer[g];
  { PhaseScope ph(ctx, "EO_update");      // removal of every block's old contribution (:312-313)
    D.r_store = 1;
    {
      const size_t nBKs = (size_t)D.B * D.K, nSold = (size_t)D.nb * nBKs, nSets = 3 * (size_t)D.nrep * nBKs;
      const int cur = ctx->sold_cur, oth = cur ^ 1;
      const int64_t rnd = (int64_t)ctx->round_counter - 1;          // this round
      D.Sold_fx = ctx->sold_buf[cur];
      if (!ctx->sets_clean) { HIPCHK(hipMemsetAsync(D.Snew_set[0], 0, sizeof(long long) * nSets, ctx->L.stream)); ctx->sets_clean = true; }
      const bool carried = ctx->sold_state[cur] == 2 && ctx->sold_round[cur] == rnd && ctx->sold_seed[cur] == ctx->seed &&
                           ctx->sorted_round[rnd & ctx->oset_mask] == rnd && ctx->sorted_seed[rnd & ctx->oset_mask] == ctx->seed;   // (same Feistel permutation as the sort's)
      if (carried) ctx->carried_rounds++;     // filled by the previous round's tile kernels: no pass over R
      else {             // all blocks in one pass over R
        if (ctx->sold_state[cur] != 0) HIPCHK(hipMemsetAsync(D.Sold_fx, 0, sizeof(long long) * nSold, ctx->L.stream));
        if (ctx->shuf_inv && ctx->injected_round != rnd) {      // (the sort-free shuffle leaves D.blk alone: block ids of this round's cells, on demand)
          l_shuffle_blocks(ctx->L, D, ctx->seed, (uint64_t)rnd, (uint64_t)ctx->N_global, (uint64_t)ctx->goff, ctx->cells_per_block); KCHK(); }
        l_oldsum(ctx->L, D); KCHK();
      }
      ctx->sold_state[cur] = 1;
      if (!(chain_path && p2p)) CHK(allreduce(ctx, D.Sold_fx, (int64_t)nSold, 0));      // (p2p chain: the folder exchanges new(j - 1) - old_local(j), the ranks' old sums meet there)
      // this round's tile kernels collect the next round's old contributions if this round's tiles are keyed by the next block
      const bool write_next = ctx->carry_ok && ctx->sorted_nxt[rnd & ctx->oset_mask] && !ctx->last_round_hint && D.upd_impl == 0;
      D.Sold_next = nullptr;
      if (write_next) {
        if (ctx->sold_state[oth] != 0) HIPCHK(hipMemsetAsync(ctx->sold_buf[oth], 0, sizeof(long long) * nSold, ctx->L.stream));
        D.Sold_next = ctx->sold_buf[oth];
        ctx->sold_state[oth] = 2; ctx->sold_round[oth] = rnd + 1; ctx->sold_seed[oth] = ctx->seed;
      }
      ctx->sets_clean = false;
      // R rows nobody reads are not written: this round's rows are dead if the NEXT round takes its old contributions from the carried sums
      // (write_next) and this round cannot be the call's last (round_may_be_last, set by hmx_cluster) -- moe_correct_ridge_cpp, the getters
      // and a stand-alone compute_objective only ever see the last round's R.  (A host with an abort poll may leave the call early: it
      // always gets its rows.  HMX_R_STORE=1: always store.)
      D.r_store = (write_next && !ctx->round_may_be_last && !ctx->poll && !ctx->r_store_always) ? 0 : 1;
      if (!D.r_store) ctx->rounds_without_R++;
    } }
  // (objpart needs no memset here: k_obj_reduce zeroes every slot it reads, setup / head_pass zero it initially)
  bool round_done = false;   // set by the fused path: all block steps done, skip the step loop below
  bool chain_tail = false;   // the persistent chain closed the round by itself
  const bool fused = merged && ctx->fused_ok;
  if (chain_path) {
    // default on one GPU: the whole block chain in ONE persistent launch (k_tile MODE 4)
    // (chain_ctl was reset by the launch that closed the previous round: k_round_tail / k_objective_tables.  The shuffle kernels must
    //  not touch chain_ctl, pen_g or the Sold buffers: prefetch_next() runs them on the side stream while a chain may be in flight)
    D.chain_tag = (unsigned)(1 + (ctx->chain_rounds++ % (1u << 24)) * 64);
    D.chain_xseq = ctx->p2p_xseq;
    long long* const keep_snew = D.Snew_fx;
    D.Snew_fx = D.Snew_set[0];     // one replica set: the folder resets it by exchange (zeroed by the round's memset)
    // one GPU: the chain's folder also closes the round (objective snapshot, table clears, control reset): no k_round_tail launch
    // (sharded runs with the in-launch exchange too: the ranks' objective sums travel through the inboxes, entries nBK and nBK + 1)
    chain_tail = (!sharded || (p2p && (size_t)D.B * D.K + 2 <= (size_t)P2P_CAP && D.nb <= 62)) && !ctx->obj_arith;
    D.chain_tail = chain_tail ? 1 : 0;
    if (p2p) ctx->p2p_xseq += (unsigned)D.nb + 1u + (chain_tail ? 1u : 0u);      // exchanges of this round: nb + 1 block steps (+ the objective's)
    if (chain_tail) {
      double* slot = nullptr;
      CHK(objective_slot(ctx, &slot));
      const size_t nBKs = (size_t)D.B * D.K;
      D.tail_host_slot = slot; D.tail_z0 = D.Sold_fx; D.tail_n0 = (unsigned long long)D.nb * nBKs;
      D.tail_z1 = D.Snew_set[0]; D.tail_n1 = 3ull * (unsigned long long)D.nrep * nBKs;
    }
    {
      ChainGate& gate = chain_gate();
      std::lock_guard<std::mutex> lk(gate.mu);
      hipEvent_t& ev = gate.last[ctx->device];
      const void*& owner = gate.owner[ctx->device];
      if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      else if (owner != (const void*)ctx->L.stream) HIPCHK(hipStreamWaitEvent(ctx->L.stream, ev, 0));   // (the same stream orders its own launches: no event, ~20 us of barrier packet less per round)
      { Launch Le; CHK(launch_with_events(ctx, Le)); l_chain(Le, D, ctx->chain_wgs); KCHK(); }
      HIPCHK(hipEventRecord(ev, ctx->L.stream));
      owner = (const void*)ctx->L.stream;
    }
    D.chain_tail = 0;
    if (ctx->profile) ctx->prof_update_steps += D.nb;
    D.Snew_fx = keep_snew;
    ctx->chain_check = true;
    round_done = true;
  } else if (fused) {
    // default: the fold + penalty of step j happens in the prologue of its own update launch.  Sharded: the replica set a
    // launch has filled is all-reduced IN PLACE (nrep*B*K int64, 64 KB at C4: latency-bound like the 8 KB of one table), so
    // the next launch's prologue sums global replicas exactly as it sums local ones -- one kernel + one collective per block
    // step instead of three kernels + one collective.
    long long* const keep_snew = D.Snew_fx;
    D.fused_fold = 1;
    for (int j = 0; j < D.nb; j++) {
      D.fold_prev = D.Snew_set[(j + 2) % 3]; D.Snew_fx = D.Snew_set[j % 3]; D.fold_zero = D.Snew_set[(j + 1) % 3];
      { Launch Le; CHK(launch_with_events(ctx, Le)); l_update(Le, D, j); KCHK(); }
      if (ctx->profile) ctx->prof_update_steps++;
      if (sharded) CHK(allreduce(ctx, D.Snew_fx, (int64_t)D.nrep * D.B * D.K, 0));   // this block's new contribution, all ranks
      std::swap(D.O_fx, D.O_alt);   // workgroup 0 published O' into O_alt
    }
    D.fused_fold = 0;
    // O += new(last block): fold-only launch; the set it zeroes is one of the three (re-zeroed next round anyway)
    l_foldpen(ctx->L, D, -1, D.O_fx, D.O_alt, D.Snew_set[(D.nb - 1) % 3], D.Snew_set[D.nb % 3]); KCHK();
    std::swap(D.O_fx, D.O_alt);
    D.Snew_fx = keep_snew;
    round_done = true;
  }
  for (int j = 0; j <= D.nb && !round_done; j++) {
    // fold the previous block's new contribution into O, remove block j's old one (src/harmony.cpp:312-313,329-330)
    if (sharded) {  // shard-local replicas -> one table, summed over the ranks (the only collective of a block step)
      l_fold(ctx->L, D, j, 1); KCHK();
      CHK(allreduce(ctx, D.Snew_fx, (int64_t)D.B * D.K, 0));
    }
    if (merged) {
      // one launch: O' = O + new(prev) - old(j) and the penalty table; ping-pong so nothing is read while written
      l_foldpen(ctx->L, D, j < D.nb ? j : -1, D.O_fx, D.O_alt, D.Snew_fx, D.Snew_alt); KCHK();
      std::swap(D.O_fx, D.O_alt); std::swap(D.Snew_fx, D.Snew_alt);
    } else {
      l_fold(ctx->L, D, j < D.nb ? j : -1, sharded ? 2 : 0); KCHK();
      if (j < D.nb) { l_penalty(ctx->L, D); KCHK(); }
    }
    if (j == D.nb) break;
    { Launch Le; CHK(launch_with_events(ctx, Le)); l_update(Le, D, j); KCHK(); if (ctx->profile) ctx->prof_update_steps++; }
  }
  if (chain_tail) {
    ctx->sold_state[ctx->sold_cur] = 0;
    ctx->sets_clean = true;
    HIPCHK(hipEventRecord(ctx->obj_event, ctx->L.stream));
    ctx->obj_pending++;
  } else if (!sharded && !ctx->obj_arith) {
    // one launch: slot rows -> objective terms -> snapshot written STRAIGHT into the pinned host slot (no copy engine, no
    // second launch), chain control reset.  Resolved by flush_objectives (event) when a value is needed.
    double* slot = nullptr;
    CHK(objective_slot(ctx, &slot));
    {   // the table this round consumed and the replica sets are cleared by the same launch
      const size_t nBKs = (size_t)D.B * D.K;
      l_round_tail(ctx->L, D, slot, D.Sold_fx, (size_t)D.nb * nBKs, D.Snew_set[0], 3 * (size_t)D.nrep * nBKs); KCHK();
      ctx->sold_state[ctx->sold_cur] = 0;
      ctx->sets_clean = true;
    }
    HIPCHK(hipEventRecord(ctx->obj_event, ctx->L.stream));
    ctx->obj_pending++;
  } else {
    l_obj_reduce(ctx->L, D); KCHK();
    CHK(allreduce(ctx, D.obj, 2, 1));
    CHK(objective_snapshot(ctx));
    CHK(push_objective(ctx));  // asynchronous: resolved by flush_objectives when a value is needed
  }
  ctx->sold_cur ^= 1;      // next round subtracts what this round's tile kernels collected (or a fresh k_oldsum pass)
  D.Sold_next = nullptr;
  ctx->R_valid = D.r_store != 0;
  if (ctx->profile) { ctx->prof_update_cells += ctx->N; }   // the event pairs are resolved when a "prof:*" field is read
  ctx->timers["update_R"] += now_ms() - t0;
  return 0;
}

// ---- ridge solves (src/harmony.cpp:358-611), fp64, one cluster at a time --------------------------
bool chol_solve(std::vector<double>& A, int n, std::vector<double>& Bm, int m) {  // column-major, in place
  for (int c = 0; c < n; c++) {
    double s = A[(size_t)c * n + c];
    for (int k = 0; k < c; k++) s -= A[(size_t)k * n + c] * A[(size_t)k * n + c];
    if (!(s > 0)) return false;
    const double l = std::sqrt(s);
    A[(size_t)c * n + c] = l;
    for (int r = c + 1; r < n; r++) {
      double t = A[(size_t)c * n + r];
      for (int k = 0; k < c; k++) t -= A[(size_t)k * n + r] * A[(size_t)k * n + c];
      A[(size_t)c * n + r] = t / l;
    }
  }
  for (int j = 0; j < m; j++) {
    double* b = &Bm[(size_t)j * n];
    for (int r = 0; r < n; r++) { double t = b[r]; for (int k = 0; k < r; k++) t -= A[(size_t)k * n + r] * b[k]; b[r] = t / A[(size_t)r * n + r]; }
    for (int r = n - 1; r >= 0; r--) { double t = b[r]; for (int k = r + 1; k < n; k++) t -= A[(size_t)r * n + k] * b[k]; b[r] = t / A[(size_t)r * n + r]; }
  }
  return true;
}
bool lu_solve(std::vector<double>& A, int n, std::vector<double>& Bm, int m) {
  for (int c = 0; c < n; c++) {
    int p = c; double best = std::fabs(A[(size_t)c * n + c]);
    for (int r = c + 1; r < n; r++) if (std::fabs(A[(size_t)c * n + r]) > best) { best = std::fabs(A[(size_t)c * n + r]); p = r; }
    if (best == 0) return false;
    if (p != c) {
      for (int j = 0; j < n; j++) std::swap(A[(size_t)j * n + c], A[(size_t)j * n + p]);
      for (int j = 0; j < m; j++) std::swap(Bm[(size_t)j * n + c], Bm[(size_t)j * n + p]);
    }
    const double inv = 1 / A[(size_t)c * n + c];
    for (int r = c + 1; r < n; r++) {
      const double f = A[(size_t)c * n + r] * inv; if (f == 0) continue;
      for (int j = c + 1; j < n; j++) A[(size_t)j * n + r] -= f * A[(size_t)j * n + c];
      for (int j = 0; j < m; j++) Bm[(size_t)j * n + r] -= f * Bm[(size_t)j * n + c];
    }
  }
  for (int j = 0; j < m; j++)
    for (int r = n - 1; r >= 0; r--) {
      double s = Bm[(size_t)j * n + r];
      for (int c = r + 1; c < n; c++) s -= A[(size_t)c * n + r] * Bm[(size_t)j * n + c];
      Bm[(size_t)j * n + r] = s / A[(size_t)r * n + r];
    }
  return true;
}

struct SolveOut { int status = 0; bool skipped = false, subset = false; std::vector<float> W; int m = 0; };

// O, E: K x B column-major floats; Sq [Q][K][d], nq [Q][K] doubles; Wq [Q][K][d] floats (output)
void solve_cluster(const hmx_ctx* ctx, int k, const std::vector<float>& O, const std::vector<float>& E,
                   const std::vector<double>& Sq, const std::vector<double>& nq, std::vector<float>& Wq,
                   std::vector<float>& Ynew, SolveOut& out, const double* S0 = nullptr, const double* n0 = nullptr) {
  const int K = ctx->K, B = ctx->B, C = ctx->C, d = ctx->d, Q = ctx->Q;
  std::vector<int> cov_levels(C, 0);
  for (int b = 0, cov = 0; b < B; b++) {  // :368-380
    if (!(b < ctx->cov_bounds[cov])) cov++;
    const float rep = O[(size_t)b * K + k] / ctx->sizes[b];
    if (rep > ctx->cutoff) cov_levels[cov]++;
  }
  std::vector<int> keep;
  for (int b = 0, cov = 0; b < B; b++) {  // :389-402
    if (cov < C && !(b < ctx->cov_bounds[cov])) cov++;
    const float rep = O[(size_t)b * K + k] / ctx->sizes[b];
    if (rep > ctx->cutoff && cov_levels[cov] > 1) keep.push_back(b);
  }
  int active = 0; for (int l : cov_levels) if (l > 1) active++;
  const bool full = ((int)keep.size() == B);
  out.subset = !full;
  for (int q = 0; q < Q; q++) std::fill_n(&Wq[((size_t)q * K + k) * d], d, 0.f);
  if (!full && active == 0) { out.skipped = true; return; }  // :449-452
  const int m = (int)keep.size() + 1;
  if (C == 1) {
    // One covariate: Phi* diag(R_k) Phi*^T + Lambda is an arrowhead matrix; solve it in closed form in fp64
    // (the reference inverts it in closed form too, src/harmony.cpp:575-586).  Combinations == levels here.
    thread_local std::vector<double> w0, coef;
    thread_local std::vector<int> qof;
    w0.assign(d, 0.0); coef.assign(B, 0.0); qof.assign(B, -1);
    for (int q = 0; q < Q; q++) qof[ctx->qlev[q]] = q;
    double N = 0.0, u = 0.0;
    bool ok = true;
    for (int b : keep) {
      const int q = qof[b];
      const double n = (q >= 0) ? nq[(size_t)q * K + k] : 0.0;
      const float lam = ctx->lambda_estimation ? E[(size_t)b * K + k] * ctx->alpha : ctx->lambda[b + 1];
      const double den = n + (double)lam;
      if (!(den > 0.0)) { ok = false; break; }
      coef[b] = n / den;                       // n_b / (n_b + lambda_b)
      N += n; u += n * coef[b];
      if (q >= 0) { const double* sq = &Sq[((size_t)q * K + k) * d]; for (int j = 0; j < d; j++) w0[j] += (S0 ? -coef[b] : (1.0 - coef[b])) * sq[j]; }
    }
    // the arrowhead system's first row: (sum_i R_ki) w0 + sum_b n_b w_b = sum_i R_ki z_i.  Exact statistics: both totals are
    // the sums of the level rows.  ridge_arith = 1: the reference's OWN sequential fp32 totals (separate chains, :567,:599).
    if (S0) { N = *n0; for (int j = 0; j < d; j++) w0[j] += S0[j]; }
    u = N - u;
    if (ok && u > 0.0 && std::isfinite(u)) {
      for (int j = 0; j < d; j++) { w0[j] /= u; Ynew[(size_t)k * d + j] = (float)w0[j]; }          // intercept row :610
      out.m = m; out.W.assign((size_t)m * d, 0.f);
      int a = 1;
      for (int b : keep) {
        const int q = qof[b];
        const double n = (q >= 0) ? nq[(size_t)q * K + k] : 0.0;
        const float lam = ctx->lambda_estimation ? E[(size_t)b * K + k] * ctx->alpha : ctx->lambda[b + 1];
        const double inv = 1.0 / (n + (double)lam);
        float* wq = (q >= 0) ? &Wq[((size_t)q * K + k) * d] : nullptr;
        for (int j = 0; j < d; j++) {
          const double s = (q >= 0) ? Sq[((size_t)q * K + k) * d + j] : 0.0;
          const float w = (float)((s - n * w0[j]) * inv);
          out.W[(size_t)j * m + a] = w;
          if (wq) wq[j] = w;
        }
        a++;
      }
      return;
    }
  }
  std::vector<int> row_of(B, -1);
  for (int a = 0; a < (int)keep.size(); a++) row_of[keep[a]] = a + 1;
  std::vector<double> cov((size_t)m * m, 0.0), rhs((size_t)m * d, 0.0);
  std::vector<int> rows(C + 1);
  for (int q = 0; q < Q; q++) {
    int nr = 0; rows[nr++] = 0;
    for (int c = 0; c < C; c++) { const int ro = row_of[ctx->qlev[(size_t)q * C + c]]; if (ro >= 0) rows[nr++] = ro; }
    if (nr == 1) continue;  // none of this combination's levels is kept: its cells do not enter (:400,456-460)
    const double n = nq[(size_t)q * K + k];
    for (int a = 0; a < nr; a++) for (int b2 = 0; b2 < nr; b2++) cov[(size_t)rows[b2] * m + rows[a]] += n;
    const double* sq = &Sq[((size_t)q * K + k) * d];
    for (int j = 0; j < d; j++) { const double s = sq[j]; for (int a = 0; a < nr; a++) rhs[(size_t)j * m + rows[a]] += s; }
  }
  for (int a = 1; a < m; a++) {  // :434-439, :533-544
    const float lam = ctx->lambda_estimation ? E[(size_t)keep[a - 1] * K + k] * ctx->alpha : ctx->lambda[keep[a - 1] + 1];
    cov[(size_t)a * m + a] += (double)lam;
  }
  std::vector<double> A = cov, X = rhs;
  if (!chol_solve(A, m, X, d)) { A = cov; X = rhs; if (!lu_solve(A, m, X, d)) { out.status = HMX_ERR_SOLVE; return; } }
  for (int j = 0; j < d; j++) { Ynew[(size_t)k * d + j] = (float)X[(size_t)j * m]; X[(size_t)j * m] = 0.0; }  // :610-611
  out.m = m; out.W.resize((size_t)m * d);
  for (size_t i = 0; i < out.W.size(); i++) out.W[i] = (float)X[i];
  for (int q = 0; q < Q; q++) {  // correction of a cell of combination q from cluster k: sum of its kept levels' rows
    float* w = &Wq[((size_t)q * K + k) * d];
    for (int c = 0; c < C; c++) { const int ro = row_of[ctx->qlev[(size_t)q * C + c]]; if (ro < 0) continue;
      for (int j = 0; j < d; j++) w[j] += out.W[(size_t)j * m + ro]; }
  }
}

std::vector<float> table_O(const hmx_ctx* ctx, const std::vector<long long>& ofx) {
  std::vector<float> O(ofx.size());
  for (size_t i = 0; i < ofx.size(); i++) O[i] = (float)((double)ofx[i] * FX_INV);
  return O;
}
std::vector<float> table_E(const hmx_ctx* ctx, const std::vector<long long>& ofx) {
  const int K = ctx->K, B = ctx->B;
  std::vector<float> E((size_t)K * B);
  for (int k = 0; k < K; k++) {
    long long rs = 0; for (int b = 0; b < ctx->B_vec[0]; b++) rs += ofx[(size_t)b * K + k];
    const double rsd = (double)rs * FX_INV;
    for (int b = 0; b < B; b++) E[(size_t)b * K + k] = (float)(rsd * (double)ctx->Pr_b[b]);
  }
  return E;
}

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
// ---- peer-to-peer block chain ------------------------------------------------------------------------------------------
int hmx_p2p_export(hmx_ctx* ctx, uint8_t* handle_out) {
  if (!ctx || !handle_out) return ctx ? fail(ctx, HMX_ERR_ARG, "null handle buffer") : HMX_ERR_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == HMX_P2P_HANDLE_BYTES, "HIP IPC handle size");
  if (ctx->device < 0) { int cur = 0; (void)hipGetDevice(&cur); ctx->device = cur; }
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->p2p_self) {
    // fine-grained device memory: peers write it over xGMI while this GPU's folder polls it (coarse-grained memory is only
    // coherent at kernel boundaries).  HMX_P2P_MEM=uncached selects hipDeviceMallocUncached instead.
    const char* m = getenv("HMX_P2P_MEM");
    const unsigned flags = (m && std::string(m) == "uncached") ? hipDeviceMallocUncached : hipDeviceMallocFinegrained;
    void* p = nullptr;
    HIPCHK(hipExtMallocWithFlags(&p, P2P_INBOX_GRANULES * sizeof(unsigned long long), flags));
    HIPCHK(hipMemset(p, 0, P2P_INBOX_GRANULES * sizeof(unsigned long long)));
    HIPCHK(hipDeviceSynchronize());
    ctx->p2p_self = (unsigned long long*)p;
  }
  hipIpcMemHandle_t h;
  HIPCHK(hipIpcGetMemHandle(&h, ctx->p2p_self));
  std::memcpy(handle_out, &h, sizeof(h));
  return 0;
}
int hmx_p2p_connect(hmx_ctx* ctx, int32_t rank, int32_t world, const uint8_t* handles) {
  if (!ctx || !handles) return ctx ? fail(ctx, HMX_ERR_ARG, "null handle table") : HMX_ERR_ARG;
  if (world < 2 || world > 8 || rank < 0 || rank >= world) return fail(ctx, HMX_ERR_ARG, "the peer-to-peer chain takes 2..8 ranks");
  if (!ctx->p2p_self) return fail(ctx, HMX_ERR_STATE, "hmx_p2p_export first");
  HIPCHK(hipSetDevice(ctx->device));
  for (int g = 0; g < world; g++) {
    if (g == rank) { ctx->p2p_peer[g] = ctx->p2p_self; continue; }
    if (ctx->p2p_peer[g]) continue;
    hipIpcMemHandle_t h; std::memcpy(&h, handles + (size_t)g * HMX_P2P_HANDLE_BYTES, sizeof(h));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); ctx->p2p_note = std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e); return fail(ctx, HMX_ERR_COMM, ctx->p2p_note); }
    ctx->p2p_peer[g] = (unsigned long long*)p;
  }
  ctx->p2p_rank = rank; ctx->p2p_world = world; ctx->p2p_note = "connected, not tested";
  return 0;
}
int hmx_p2p_selftest(hmx_ctx* ctx) {
  if (!ctx || ctx->p2p_world < 2) return ctx ? fail(ctx, HMX_ERR_STATE, "hmx_p2p_connect first") : HMX_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->L.stream) { HIPCHK(hipStreamCreateWithFlags(&ctx->L.stream, hipStreamNonBlocking)); ctx->own_stream = true; }
  if (!ctx->p2p_result) HIPCHK(hipMalloc((void**)&ctx->p2p_result, 2 * sizeof(int)));
  Dev T{};
  T.p2p_world = ctx->p2p_world; T.p2p_rank = ctx->p2p_rank;
  for (int g = 0; g < 8; g++) T.p2p_inbox[g] = ctx->p2p_peer[g];
  // tags of the test live above 2^31 (the chain's stay below 2^30 + 64), 128 per test
  l_p2p_selftest(ctx->L, T, 0x80000000u + ((++ctx->p2p_tests) & 0xffffffu) * 128u, ctx->p2p_result); KCHK();
  int res[2] = {-1, 0};
  HIPCHK(hipMemcpyAsync(res, ctx->p2p_result, sizeof(res), hipMemcpyDeviceToHost, ctx->L.stream));
  HIPCHK(hipStreamSynchronize(ctx->L.stream));
  if (res[0] != 0) {
    ctx->p2p_note = "self-test: " + std::to_string(res[0]) + " wrong or missing values";
    return fail(ctx, HMX_ERR_COMM, "peer-to-peer " + ctx->p2p_note);
  }
  ctx->p2p_exchange_us = (double)res[1] / 100.0 / 63.0;    // 100 MHz ticks over the 63 steps after the first
  char buf[96]; snprintf(buf, sizeof(buf), "self-test passed (%.2f us per exchange step)", ctx->p2p_exchange_us);
  ctx->p2p_note = buf;
  return 0;
}
int hmx_p2p_enable(hmx_ctx* ctx, int32_t on) {
  if (!ctx) return HMX_ERR_ARG;
  if (on && ctx->p2p_world < 2) return fail(ctx, HMX_ERR_STATE, "hmx_p2p_connect first");
  ctx->p2p_on = on != 0;
  if (on && ctx->p2p_note.find("passed") == std::string::npos) ctx->p2p_note = "on (self-test not run)";
  return 0;
}
// with the built-in communicator the whole bootstrap is automatic; every failure just leaves the per-block collectives in place
static void p2p_auto(hmx_ctx* ctx, RcclApi* api, int rank, int world) {
  const char* e = getenv("HMX_P2P");
  if ((e && std::string(e) == "0") || world < 2 || world > 8 || !api->AllGather) { ctx->p2p_note = "off"; return; }
  auto give_up = [&](const std::string& why) { ctx->p2p_note = why; ctx->p2p_on = false; ctx->err.clear(); };
  if (!ctx->L.stream) { if (hipStreamCreateWithFlags(&ctx->L.stream, hipStreamNonBlocking) != hipSuccess) return give_up("no stream"); ctx->own_stream = true; }
  uint8_t mine[HMX_P2P_HANDLE_BYTES] = {};
  long long ok = hmx_p2p_export(ctx, mine) == 0 ? 1 : 0;    // (a rank that cannot export still takes part in the collectives)
  uint8_t* dev = nullptr; long long* flag = nullptr;
  std::vector<uint8_t> all((size_t)world * HMX_P2P_HANDLE_BYTES);
  if (hipMalloc((void**)&dev, all.size()) != hipSuccess || hipMalloc((void**)&flag, 8) != hipSuccess) return give_up("hipMalloc");
  bool comm_ok = hipMemcpyAsync(dev + (size_t)rank * HMX_P2P_HANDLE_BYTES, mine, sizeof(mine), hipMemcpyHostToDevice, ctx->L.stream) == hipSuccess &&
                 api->AllGather(dev + (size_t)rank * HMX_P2P_HANDLE_BYTES, dev, HMX_P2P_HANDLE_BYTES, ncclInt8, ctx->comm, ctx->L.stream) == ncclSuccess &&
                 hipMemcpyAsync(all.data(), dev, all.size(), hipMemcpyDeviceToHost, ctx->L.stream) == hipSuccess &&
                 hipStreamSynchronize(ctx->L.stream) == hipSuccess;
  auto agree = [&]() {     // min over the ranks of `ok`
    if (!comm_ok) return;
    comm_ok = hipMemcpyAsync(flag, &ok, 8, hipMemcpyHostToDevice, ctx->L.stream) == hipSuccess &&
              api->AllReduce(flag, flag, 1, ncclInt64, ncclMin, ctx->comm, ctx->L.stream) == ncclSuccess &&
              hipMemcpyAsync(&ok, flag, 8, hipMemcpyDeviceToHost, ctx->L.stream) == hipSuccess &&
              hipStreamSynchronize(ctx->L.stream) == hipSuccess;
  };
  agree();                                                                       // everyone exported (also a barrier)
  if (comm_ok && ok) { ok = hmx_p2p_connect(ctx, rank, world, all.data()) == 0 ? 1 : 0; agree(); }     // everyone connected
  if (comm_ok && ok) { ok = hmx_p2p_selftest(ctx) == 0 ? 1 : 0; agree(); }                             // everyone heard everyone
  (void)hipFree(dev); (void)hipFree(flag);
  if (!comm_ok) return give_up("bootstrap collectives failed");
  if (!ok) return give_up(ctx->p2p_note.empty() ? "a rank failed" : ctx->p2p_note + " (on some rank)");
  (void)hmx_p2p_enable(ctx, 1);
}
int hmx_set_stream(hmx_ctx* ctx, void* s) {
  if (!ctx) return HMX_ERR_ARG;
  if (ctx->own_stream && ctx->L.stream) (void)hipStreamDestroy(ctx->L.stream);
  ctx->L.stream = (hipStream_t)s; ctx->own_stream = false;
  return 0;
}
int hmx_set_abort_poll(hmx_ctx* ctx, int (*poll)(void*), void* user) {
  if (!ctx) return HMX_ERR_ARG;
  ctx->poll = poll; ctx->poll_user = user;
  return 0;
}
// probes of the R-compatible stream (no device needed): n uniforms after set.seed(seed); arma::shuffle(0..N-1) after set.seed(seed)
void hmx_r_runif(uint32_t seed, int32_t n, double* out) { hmx::RRng r; r.set_seed(seed); for (int i = 0; i < n; i++) out[i] = r.unif_rand(); }
void hmx_r_shuffle(uint32_t seed, int64_t N, int64_t* out) {
  hmx::RRng r; r.set_seed(seed);
  std::vector<int64_t> o; r.arma_shuffle(N, o);
  std::copy(o.begin(), o.end(), out);
}
void hmx_mt19937_by_array(const uint32_t* key, int32_t len, int32_t n, uint32_t* out) {   // MT19937's published known-answer vector
  hmx::RRng r; r.mt_init_by_array(key, len);
  for (int i = 0; i < n; i++) out[i] = r.genrand_int32();
}
int hmx_set_uniform_source(hmx_ctx* ctx, double (*unif_rand)(void*), void* user) {
  if (!ctx) return HMX_ERR_ARG;
  ctx->rrng.set_source(unif_rand, user);
  return 0;
}
int hmx_push_update_order(hmx_ctx* ctx, const int64_t* order) {
  if (!ctx || !ctx->ran_setup) return ctx ? fail(ctx, HMX_ERR_STATE, "setup first") : HMX_ERR_ARG;
  if (!order) return fail(ctx, HMX_ERR_ARG, "null update_order");
  {  // must be a permutation of 0..N_global-1: prepare_round indexes a position table with these values
    std::vector<bool> seen((size_t)ctx->N_global, false);
    for (int64_t p = 0; p < ctx->N_global; p++) {
      const int64_t v = order[p];
      if (v < 0 || v >= ctx->N_global || seen[(size_t)v]) return fail(ctx, HMX_ERR_ARG, "update_order is not a permutation of 0..N-1");
      seen[(size_t)v] = true;
    }
  }
  ctx->injected.emplace_back(order, order + ctx->N_global);
  return 0;
}

int hmx_set_int(hmx_ctx* ctx, const char* field, int64_t v) {
  if (!ctx || !field) return HMX_ERR_ARG;
  const std::string f(field);
  if (f == "max_iter_kmeans") ctx->max_iter_kmeans = (int)v;
  else if (f == "seed") { ctx->seed = (uint64_t)v; ctx->rrng_seeded = false; }
  else if (f == "rng") { if (v != 0 && v != 1) return fail(ctx, HMX_ERR_ARG, "rng: 0 (counter-based) or 1 (R-compatible)"); ctx->rng_mode = (int)v; ctx->rrng_seeded = false; }
  else if (f == "ridge_arith" || f == "oe_arith" || f == "obj_arith" || f == "solve_arith" || f == "ref_arith") {
    if (v != 0 && v != 1) return fail(ctx, HMX_ERR_ARG, f + ": 0 (exact accumulators) or 1 (the reference's fp32 operation order)");
    if (ctx->ran_setup) return fail(ctx, HMX_ERR_STATE, f + " must be set before setup");
    if (f == "ridge_arith" || f == "ref_arith") ctx->ridge_arith = (int)v;
    if (f == "oe_arith" || f == "ref_arith") ctx->oe_arith = (int)v;
    if (f == "obj_arith" || f == "ref_arith") ctx->obj_arith = (int)v;
    if (f == "solve_arith" || f == "ref_arith") ctx->solve_arith = (int)v;
  }
  else if (f == "stale_dist") { if (ctx->ran_setup) return fail(ctx, HMX_ERR_STATE, "stale_dist must be set before setup"); ctx->stale_dist = v != 0; }
  else if (f == "seq_passes") { if (v < 2 || v > 64) return fail(ctx, HMX_ERR_ARG, "seq_passes: 2..64"); ctx->seq_passes = (int)v; if (ctx->seq_max_passes < (int)v) ctx->seq_max_passes = (int)v; }
  else if (f == "seq_warm_passes") { if (v < 1 || v > 64) return fail(ctx, HMX_ERR_ARG, "seq_warm_passes: 1..64"); ctx->seq_warm_passes = (int)v; }
  else if (f == "seq_stats") ctx->seq_stats = v != 0;
  else if (f == "seq_tol_ppb") { if (v < 0 || v > 100000000) return fail(ctx, HMX_ERR_ARG, "seq_tol_ppb: 0 .. 1e8 (parts per billion)"); ctx->seq_tol = 1e-9 * (double)v; }
  else if (f == "seq_strict") { ctx->seq_strict = v != 0; if (v && ctx->seq_max_passes < 64) ctx->seq_max_passes = 64; }
  else if (f == "seq_max_passes") { if (v < 2 || v > 256) return fail(ctx, HMX_ERR_ARG, "seq_max_passes: 2..256"); ctx->seq_max_passes = (int)v; }
  else if (f == "device") ctx->device = (int)v;
  else if (f == "profile") { ctx->profile = (int)(v < 0 ? 0 : v > 2 ? 2 : v); ctx->prof_update_ms = 0; ctx->prof_update_launches = 0; ctx->prof_update_cells = 0; ctx->prof_update_steps = 0; ctx->ev_used = 0;
                             ctx->ph_used = 0; ctx->gpu_timers.clear(); }
  else if (f == "grid") { if (ctx->ran_setup) return fail(ctx, HMX_ERR_STATE, "grid must be set before setup"); ctx->L.grid = (int)v; }
  else if (f == "upd_cpw") { ctx->tun_cpw = (int)v; if (ctx->ran_setup) ctx->D.upd_cpw = (int)(v < 4 ? 4 : v); }
  else if (f == "comm_force") ctx->comm_force = v != 0;
  else if (f == "upd_impl") { ctx->tun_impl = (int)v; if (ctx->ran_setup) { ctx->D.upd_impl = (int)v;
                                if (v == 1 && !ctx->D.need_lorder) { ctx->D.need_lorder = 1; for (int i = 0; i < 4; i++) ctx->sorted_round[i] = -1; } } }   // (the v1 kernel reads lorder: re-sort with it; the histogram slots stay valid)
  else if (f == "upd_wps") { if (ctx->ran_setup) return fail(ctx, HMX_ERR_STATE, "upd_wps must be set before setup"); ctx->tun_wps = (int)v; }
  else if (f == "upd_debug") { if (ctx->ran_setup) ctx->D.upd_debug = (int)v; }
  else if (f == "upd_tpw") { ctx->tun_tpw = (int)v; if (ctx->ran_setup) ctx->D.upd_tpw = (int)(v < 1 ? 1 : v); }
  else return fail(ctx, HMX_ERR_ARG, "unknown or read-only field: " + f);
  return 0;
}

// ---- setup (src/harmony.cpp:29-128) -------------------------------------------------------------------
int hmx_setup(hmx_ctx* ctx, const double* Z, int64_t N, int32_t d, const int32_t* phi_i, const int32_t* phi_p,
              const double* phi_x, int32_t B, const double* sigma, const double* theta, const double* lambda,
              int32_t n_lambda, double alpha, int32_t max_iter_kmeans, double epsilon_kmeans, double epsilon_harmony,
              int32_t K, double block_size, const int32_t* B_vec, int32_t C, double cutoff, int32_t verbose) {
  return hmx_setup_ex(ctx, Z, HMX_F64, HMX_HOST, N, d, phi_i, phi_p, phi_x, B, sigma, theta, lambda, n_lambda, alpha, max_iter_kmeans,
                      epsilon_kmeans, epsilon_harmony, K, block_size, B_vec, C, cutoff, verbose);
}

int hmx_setup_ex(hmx_ctx* ctx, const void* Z, int32_t z_dtype, int32_t z_location, int64_t N, int32_t d, const int32_t* phi_i,
                 const int32_t* phi_p, const double* phi_x, int32_t B, const double* sigma, const double* theta,
                 const double* lambda, int32_t n_lambda, double alpha, int32_t max_iter_kmeans, double epsilon_kmeans,
                 double epsilon_harmony, int32_t K, double block_size, const int32_t* B_vec, int32_t C, double cutoff,
                 int32_t verbose) {
  if (!ctx) return HMX_ERR_ARG;
  if ((z_dtype != HMX_F64 && z_dtype != HMX_F32) || (z_location != HMX_HOST && z_location != HMX_DEVICE))
    return fail(ctx, HMX_ERR_ARG, "bad dtype / location of Z");
  ctx->err.clear(); ctx->warn.clear();
  if (!Z || !phi_i || !phi_p || !sigma || !theta || !lambda || !B_vec) return fail(ctx, HMX_ERR_ARG, "null argument");
  if (N <= 0 || d <= 0 || K <= 0 || B <= 0 || C <= 0) return fail(ctx, HMX_ERR_ARG, "non-positive dimension");
  if (d > 128 || K > 256 || C > 15) return fail(ctx, HMX_ERR_LIMIT, "supported envelope: d <= 128, K <= 256, covariates <= 15");
  if (N > 2000000000ll) return fail(ctx, HMX_ERR_LIMIT, "at most 2e9 cells per GPU shard");
  if ((ctx->ridge_arith || ctx->oe_arith || ctx->obj_arith || ctx->solve_arith) && (ctx->world > 1 || ctx->comm_force))
    return fail(ctx, HMX_ERR_ARG, "the reference-arithmetic modes (ridge_arith / oe_arith / obj_arith / solve_arith) run on one GPU");
  if (ctx->world <= 1) { ctx->N_global = N; ctx->goff = 0; }
  if (ctx->N_global > 4000000000ll) return fail(ctx, HMX_ERR_LIMIT, "at most 4e9 cells in total");
  if (ctx->N_global < 6) return fail(ctx, HMX_ERR_TOO_FEW, "Refusing to run with less than 6 cells");
  if (n_lambda != 1 && n_lambda != B + 1) return fail(ctx, HMX_ERR_ARG, "lambda must have length B+1 (or be the single value -1)");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(ctx, HMX_ERR_DEVICE, "no HIP device available: libharmony_mi355x has no CPU fallback");
  if (ctx->device < 0) { int cur = 0; (void)hipGetDevice(&cur); ctx->device = cur; }
  HIPCHK(hipSetDevice(ctx->device));
  free_all(ctx);
  if (!ctx->L.stream) { HIPCHK(hipStreamCreateWithFlags(&ctx->L.stream, hipStreamNonBlocking)); ctx->own_stream = true; }
  if (ctx->L.grid <= 0) {
    const char* e = getenv("HMX_GRID");
    ctx->L.grid = e ? atoi(e) : 2048;
  }

  ctx->N = N; ctx->d = d; ctx->K = K; ctx->B = B; ctx->C = C; ctx->verbose = verbose;
  ctx->B_vec.assign(B_vec, B_vec + C);
  ctx->cov_bounds.resize(C);
  std::partial_sum(ctx->B_vec.begin(), ctx->B_vec.end(), ctx->cov_bounds.begin());
  if (ctx->cov_bounds.back() != B) return fail(ctx, HMX_ERR_ARG, "sum(B_vec) != nrow(Phi)");
  ctx->sigma.resize(K); for (int k = 0; k < K; k++) ctx->sigma[k] = (float)sigma[k];
  ctx->theta.resize(B); for (int b = 0; b < B; b++) ctx->theta[b] = (float)theta[b];
  if (lambda[0] == -1) { ctx->lambda_estimation = true; ctx->lambda.clear(); }
  else {
    if (n_lambda != B + 1) return fail(ctx, HMX_ERR_ARG, "fixed lambda must have length B+1");
    ctx->lambda_estimation = false; ctx->lambda.resize(B + 1); for (int i = 0; i <= B; i++) ctx->lambda[i] = (float)lambda[i];
  }
  ctx->alpha = (float)alpha; ctx->max_iter_kmeans = max_iter_kmeans; ctx->eps_k = (float)epsilon_kmeans;
  ctx->eps_h = (float)epsilon_harmony; ctx->cutoff = (float)cutoff;
  if (ctx->N_global < 40) { ctx->warn = "Too few cells. Setting block_size to 0.2"; ctx->block_size = 0.2f; }  // :86-88
  else ctx->block_size = (float)block_size;
  ctx->nb = my_ceil(1.0 / ctx->block_size);                                         // :280
  ctx->cells_per_block = (uint64_t)(unsigned)((float)ctx->N_global * ctx->block_size);  // :281 (fp32 product, truncated)
  if (ctx->cells_per_block < 1) ctx->cells_per_block = 1;
  if (ctx->nb < 1) ctx->nb = 1;

  // ---- per-covariate level codes from the C-hot CSC design (src/harmony.cpp:49-65, R/ui.R:210-213)
  std::vector<int> codes((size_t)C * N);
  for (int64_t i = 0; i < N; i++) {
    if (phi_p[i + 1] - phi_p[i] != C) return fail(ctx, HMX_ERR_PHI, "Phi column does not hold exactly one level per covariate");
    for (int c = 0; c < C; c++) {
      const int b = phi_i[phi_p[i] + c];
      if (b < 0 || b >= B || b >= ctx->cov_bounds[c] || (c > 0 && b < ctx->cov_bounds[c - 1]))
        return fail(ctx, HMX_ERR_PHI, "Phi rows are not grouped by covariate");
      if (phi_x && phi_x[phi_p[i] + c] != 1.0) return fail(ctx, HMX_ERR_PHI, "Phi must be a 0/1 design");
      codes[(size_t)c * N + i] = b;
    }
  }
  // ---- level combinations: dense mixed-radix key -> compact id (identical on every rank)
  double dense = 1; for (int c = 0; c < C; c++) dense *= ctx->B_vec[c];
  if (dense > 16777216.0) return fail(ctx, HMX_ERR_LIMIT, "product of covariate level counts exceeds 2^24");
  const int64_t P = (int64_t)dense;
  std::vector<long long> present((size_t)P, 0);
  std::vector<int> key((size_t)N);
  for (int64_t i = 0; i < N; i++) {
    int64_t kk = 0, mul = 1;
    for (int c = 0; c < C; c++) { kk += mul * (codes[(size_t)c * N + i] - (c ? ctx->cov_bounds[c - 1] : 0)); mul *= ctx->B_vec[c]; }
    key[i] = (int)kk; present[(size_t)kk]++;
  }
  // global level sizes N_b and global presence (one all-reduce each when sharded)
  std::vector<long long> nbcount((size_t)B, 0);
  for (int c = 0; c < C; c++) for (int64_t i = 0; i < N; i++) nbcount[codes[(size_t)c * N + i]]++;
  if (ctx->world > 1 || ctx->comm_force) {
    long long* dtmp; const size_t cnt = (size_t)P + B;
    HIPCHK(hipMalloc((void**)&dtmp, cnt * sizeof(long long)));
    std::vector<long long> tmp(present); tmp.insert(tmp.end(), nbcount.begin(), nbcount.end());
    int st = h2d(ctx, dtmp, tmp.data(), cnt);
    if (!st) st = allreduce(ctx, dtmp, (int64_t)cnt, 0);
    if (!st) st = d2h(ctx, tmp.data(), dtmp, cnt);
    (void)hipFree(dtmp);
    if (st) return st;
    std::copy(tmp.begin(), tmp.begin() + P, present.begin());
    std::copy(tmp.begin() + P, tmp.end(), nbcount.begin());
  }
  std::vector<int> qid((size_t)P, -1);
  ctx->Q = 0; ctx->qlev.clear();
  for (int64_t kk = 0; kk < P; kk++) if (present[(size_t)kk] > 0) {
    qid[(size_t)kk] = ctx->Q++;
    int64_t rem = kk;
    for (int c = 0; c < C; c++) { ctx->qlev.push_back((int)(rem % ctx->B_vec[c]) + (c ? ctx->cov_bounds[c - 1] : 0)); rem /= ctx->B_vec[c]; }
  }
  const int Q = ctx->Q;
  ctx->sizes.resize(B); ctx->Pr_b.resize(B);
  for (int b = 0; b < B; b++) { ctx->sizes[b] = (float)nbcount[b]; ctx->Pr_b[b] = ctx->sizes[b] / (float)ctx->N_global; }  // :67
  // ---- internal order: cells sorted (stably) by combination
  std::vector<int> combo_of((size_t)N), start((size_t)Q + 1, 0), invperm((size_t)N), combo_sorted((size_t)N);
  for (int64_t i = 0; i < N; i++) { combo_of[i] = qid[(size_t)key[i]]; start[(size_t)combo_of[i] + 1]++; }
  for (int q = 0; q < Q; q++) start[q + 1] += start[q];
  ctx->perm.assign((size_t)N, 0);
  { std::vector<int> cur(start.begin(), start.end() - 1);
    for (int64_t i = 0; i < N; i++) { const int p = cur[combo_of[i]]++; ctx->perm[p] = (int)i; invperm[i] = p; combo_sorted[p] = combo_of[i]; } }
  std::vector<Item> items, aitems, titems;
  for (int q = 0; q < Q; q++) {
    for (int s = start[q]; s < start[q + 1]; s += ITEM_CELLS) items.push_back({q, s, std::min(ITEM_CELLS, start[q + 1] - s)});
    for (int s = start[q]; s < start[q + 1]; s += APPLY_CELLS) aitems.push_back({q, s, std::min(APPLY_CELLS, start[q + 1] - s)});
    for (int s = start[q]; s < start[q + 1]; s += 16) titems.push_back({q, s, std::min(16, start[q + 1] - s)});
  }

  // ---- device state
  Dev& D = ctx->D;
  D = Dev{};
  D.n = (int)N; D.d = d; D.K = K; D.B = B; D.C = C; D.Q = Q; D.B0 = ctx->B_vec[0];
  D.KP = (K + 63) / 64 * 64; D.nb = ctx->nb;
  D.zs = (d + 3) / 4 * 4;
  // (rows padded to whole 128-byte lines -- 208 -> 256 B at d = 50 -- were measured in round 4: the block step stayed where it was for 23 % more memory; the switch is gone)
  { const char* e = getenv("HMX_NREP"); int want = e ? atoi(e) : 8; if (want > 8) want = 8; D.nrep = 1; while (D.nrep * 2 <= want && (size_t)D.nrep * 2 * B * K <= (1u << 20)) D.nrep *= 2; }
  { static const int sup[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 14, 16};     // (13: K = 200, BASELINE configs[4])
    const int need = (K + 15) / 16; D.NCT = 16; for (int v : sup) if (v >= need) { D.NCT = v; break; } }
  D.lloyd_lds = ((size_t)d * D.KP * 4 + ((size_t)K * d + K) * 8 <= 98304) ? 1 : 0;
  { const char* e = getenv("HMX_MOE_IMPL");
    D.moe_mfma = (K % 4 == 0 && d <= 64 && K <= 256 && !(e && std::string(e) == "v1")) ? 1 : 0;   // K > 128: split statistics kernel
    D.wNT4 = K / 16; D.wtail = (K - 16 * D.wNT4) / 4; D.wNS = 4 * D.wNT4 + D.wtail; D.wNQ = ((d + 15) / 16 + 3) / 4; }
  D.r_store = 1;
  { const char* rs = getenv("HMX_R_STORE"); ctx->r_store_always = rs && atoi(rs) == 1; }
  D.nwmax = 4 * ctx->L.grid; D.objslots = std::min(D.nb, 64);
  D.pen_lds = ((size_t)D.NQ * 0 + (size_t)B * K * 4 + (size_t)Q * C * 4 <= 24576) ? 1 : 0;
  D.rvec = (K % 4 == 0) ? 1 : 0;
  D.NQ = (D.NCT + 3) / 4; D.NT4 = D.zs / 16; D.tail = (D.zs - 16 * D.NT4) / 4; D.NS = 4 * D.NT4 + D.tail;
  // split-bf16 form of the tile kernels' distance GEMM (hmx_tile_bf.hip): offered where its register form covers the shapes the fp32
  // register form covers (rows of <= 64 PCs in four 16-byte groups); each launch takes it when its LDS image fits (l_update & co)
  D.NS2 = (D.zs + 31) / 32;
  { const char* e = getenv("HMX_DOT"); D.dot_bf = !(e && std::string(e) == "f32") && (D.NT4 > 4 || D.NS2 <= 2) && D.NS2 <= 4; }
  { const char* e = getenv("HMX_UPDATE_IMPL"); D.upd_impl = ctx->tun_impl >= 0 ? ctx->tun_impl : ((e && std::string(e) == "v1") ? 1 : 0); }
  { const char* e = getenv("HMX_UPD_THREADS"); D.upd_threads = (e && atoi(e) == 256) ? 256 : 512; }
  { // uniform sigma (the reference's default): scalar-constant kernel variants; with K <= 64 they also fit the register
    // budget of 4 waves per SIMD (1024-thread workgroups) -- measured 16 % faster per launch than 2 waves at K = 64
    bool usig = true; for (int k = 1; k < K; k++) usig = usig && ctx->sigma[k] == ctx->sigma[0];
    { const char* e = getenv("HMX_USIG"); if (e && atoi(e) == 0) usig = false; }
    D.usig = usig ? 1 : 0;
    const char* e = getenv("HMX_UPD_WPS");
    int w = ctx->tun_wps > 0 ? ctx->tun_wps : (e ? atoi(e) : 4);
    if (w != 4 || !usig || D.NCT > 4 || D.upd_impl != 0) w = 2;
    D.upd_wps = w;
    if (w == 4) D.upd_threads = 1024; }
  { const char* e = getenv("HMX_UPD_MAXBLOCKS"); D.upd_maxblocks = e ? atoi(e) : (D.upd_threads >= 512 ? 256 : 512); if (D.upd_maxblocks < 1) D.upd_maxblocks = 1; }
  D.upd_debug = 0;
  // static-tile launches (head / Lloyd / seeding): one resident generation of 256-thread workgroups (2 per CU at the 2 waves
  // per SIMD the K > 64 kernels get) re-stages the centroid image once instead of four times: head 213 -> 200 us at 1M
  { const char* e = getenv("HMX_STATIC_MAXBLOCKS"); D.static_maxblocks = e ? atoi(e) : (D.NCT >= 5 ? 512 : D.NCT >= 3 ? 768 : 1024); }
  { const char* e = getenv("HMX_OLDSUM_IMPL"); D.oldsum_stream = (e && std::string(e) == "gather") ? 0 : (e && std::string(e) == "stream1") ? 2 : 1; }   // 1: 16-byte stream, 2: dword stream
  D.need_lorder = (D.upd_impl == 1 || D.oldsum_stream == 0 || (size_t)D.nb * K * 8 > 64 * 1024) ? 1 : 0;
  { const char* e = getenv("HMX_UPD_TPW"); D.upd_tpw = ctx->tun_tpw > 0 ? ctx->tun_tpw : (e ? atoi(e) : 1); if (D.upd_tpw < 1) D.upd_tpw = 1; }
  { const char* e = getenv("HMX_UPD_CPW"); D.upd_cpw = ctx->tun_cpw > 0 ? ctx->tun_cpw : (e ? atoi(e) : 128); if (D.upd_cpw < 4) D.upd_cpw = 4; }
  std::vector<Item> schunks; std::vector<int> qchunk((size_t)Q + 1, 0);
  for (int q = 0; q < Q; q++) {
    qchunk[q] = (int)schunks.size();
    for (int s = start[q]; s < start[q + 1]; s += SORT_CHUNK) schunks.push_back({q, s, std::min(SORT_CHUNK, start[q + 1] - s)});
  }
  qchunk[Q] = (int)schunks.size();
  D.nchunks = (int)schunks.size();
  { // Old contributions carried from round to round (update_R): tiles keyed by (block, combination, NEXT block) cost up to 16
    // padding slots per key -- worth it while the expected padding (8 per key) stays below 4 % (12 % with the chain) of the cells.  HMX_SOLD_CARRY=0|1.
    const char* e = getenv("HMX_SOLD_CARRY");
    const bool fits = D.nb <= 63 && Q < (1 << 19) && D.upd_impl == 0 &&
                      (int64_t)N + (int64_t)D.nb * D.nb * Q * 16 <= 2147483000ll;
    // (round 4: with the R stores of carried rounds gone as well -- Dev::r_store -- the carry saves ~200 us per round at 1M cells where the
    //  persistent chain runs (K <= 112): worth up to ~12 % of padding there; measured at 1.25M cells / 20 batches, 5.1 %: 15.5 -> 12.7 ms per run.
    //  On the launch-per-step path (configs[4] shape, 41 %: 63 -> 70 ms) the old bound stays.)
    const bool pays = (int64_t)D.nb * D.nb * Q * 8 * (K <= 112 ? 8 : 25) <= (int64_t)N;
    ctx->carry_ok = fits && (e ? atoi(e) == 1 : pays) && !ctx->oe_arith;      // (oe_arith: the tables follow the reference, nothing is carried)
    D.nxt = 0; D.Sold_next = nullptr; D.Sold_head = nullptr; D.head_gather = 0; ctx->carried_rounds = 0;
    D.qmask = ctx->carry_ok ? 0x7FFFF : 0x7FFFFFFF; }
  const int nV = ctx->carry_ok ? D.nb * D.nb : D.nb;      // sort keys of a round
  if ((int64_t)N + (int64_t)nV * Q * 16 > 2147483000ll)
    return fail(ctx, HMX_ERR_LIMIT, "padded block order (N + n_blocks * combinations * 16) exceeds the int32 index range of one shard");
  D.npad = (int)((int64_t)N + (int64_t)nV * Q * 16);
  D.nitems = (int)items.size(); D.naitems = (int)aitems.size(); D.ntitems = (int)titems.size();
  { const char* e = getenv("HMX_TILE_IMPL"); D.tile_impl = (e && std::string(e) == "v1") ? 0 : 1; }
  CHK(dalloc(ctx, &D.Zo, (size_t)N * D.zs)); CHK(dalloc(ctx, &D.Zc, (size_t)N * D.zs)); CHK(dalloc(ctx, &D.R, ((size_t)N + 1) * K));   // + one dummy row (target of masked stores)
  CHK(dalloc(ctx, &D.perm, (size_t)N)); CHK(dalloc(ctx, &D.invperm, (size_t)N)); CHK(dalloc(ctx, &D.combo, (size_t)N));
  CHK(dalloc(ctx, &D.qlev, (size_t)Q * C));
  CHK(dalloc(ctx, &D.Yt, (size_t)d * K)); CHK(dalloc(ctx, &D.Ycur, (size_t)d * K)); CHK(dalloc(ctx, &D.Yimg, (size_t)D.NQ * D.NS * 256)); HIPCHK(hipMemsetAsync(D.Yimg, 0, (size_t)D.NQ * D.NS * 1024, ctx->L.stream)); CHK(dalloc(ctx, &D.Yimg3, (size_t)D.NCT * D.NS2 * 3 * 512)); HIPCHK(hipMemsetAsync(D.Yimg3, 0, (size_t)D.NCT * D.NS2 * 3 * 1024, ctx->L.stream)); CHK(dalloc(ctx, &D.sigma, (size_t)K)); CHK(dalloc(ctx, &D.theta, (size_t)B)); CHK(dalloc(ctx, &D.Pr_b, (size_t)B));
  CHK(dalloc(ctx, &D.O_fx, (size_t)B * K)); CHK(dalloc(ctx, &D.Snew_fx, (size_t)D.nrep * B * K));
  // Sold_fx [nb][B][K] and the three rotating replica sets of the fused path share one buffer: one memset per round
  { long long* s3; CHK(dalloc(ctx, &s3, (size_t)2 * D.nb * B * K + (size_t)3 * D.nrep * B * K)); D.Sold_fx = s3;
    ctx->sold_buf[0] = s3; ctx->sold_buf[1] = s3 + (size_t)D.nb * B * K; ctx->sold_cur = 0; ctx->sold_state[0] = ctx->sold_state[1] = 1; ctx->sets_clean = false;
    for (int i = 0; i < 3; i++) D.Snew_set[i] = s3 + (size_t)2 * D.nb * B * K + (size_t)i * D.nrep * B * K; }
  CHK(dalloc(ctx, &D.O_alt, (size_t)B * K)); CHK(dalloc(ctx, &D.Snew_alt, (size_t)D.nrep * B * K)); CHK(dalloc(ctx, &D.objpart, (size_t)2 * D.objslots * D.nwmax)); CHK(dalloc(ctx, &D.objrow, (size_t)2 * D.objslots));
  D.trace = nullptr;
  if (const char* e = getenv("HMX_TRACE")) if (atoi(e)) { CHK(dalloc(ctx, &D.trace, (size_t)16 * D.nwmax)); HIPCHK(hipMemsetAsync(D.trace, 0, sizeof(unsigned long long) * 16 * (size_t)D.nwmax, ctx->L.stream)); }
  CHK(dalloc(ctx, &D.pen, (size_t)B * K)); CHK(dalloc(ctx, &D.obj, (size_t)8));
  CHK(dalloc(ctx, &D.blk, (size_t)N)); CHK(dalloc(ctx, &D.lorder, (size_t)3 * D.npad + 2)); D.lpair = reinterpret_cast<int2*>(D.lorder + (((size_t)D.npad + 1) & ~(size_t)1)); /* lorder + lpair: one 0xFF memset per round */ CHK(dalloc(ctx, &D.lcombo, (size_t)D.npad));
  CHK(dalloc(ctx, &D.binoff, (size_t)nV * Q + 1)); CHK(dalloc(ctx, &D.schunks, schunks.size())); CHK(dalloc(ctx, &D.qchunk, (size_t)Q + 1));
  CHK(dalloc(ctx, &D.blkv, (size_t)N)); CHK(dalloc(ctx, &D.bincnt, (size_t)nV * Q));
  CHK(dalloc(ctx, &D.ce, (size_t)K)); CHK(dalloc(ctx, &D.cl, (size_t)K)); CHK(dalloc(ctx, &D.boff, (size_t)D.nb + 1));
  CHK(dalloc(ctx, &D.counts, (size_t)nV * D.nchunks)); CHK(dalloc(ctx, &D.offs, (size_t)nV * D.nchunks));
  { // second buffer set + side stream for the overlapped shuffle of the next round 
    ctx->sets[0] = {D.blk, D.lorder, D.lpair, D.lcombo, D.boff, D.binoff, D.counts, D.offs, D.blkv, D.bincnt};
    hmx_ctx::SortSet& t = ctx->sets[1];
    CHK(dalloc(ctx, &t.blk, (size_t)N)); CHK(dalloc(ctx, &t.lorder, (size_t)3 * D.npad + 2)); t.lpair = reinterpret_cast<int2*>(t.lorder + (((size_t)D.npad + 1) & ~(size_t)1));
    CHK(dalloc(ctx, &t.lcombo, (size_t)D.npad)); CHK(dalloc(ctx, &t.binoff, (size_t)nV * Q + 1)); CHK(dalloc(ctx, &t.boff, (size_t)D.nb + 1));
    CHK(dalloc(ctx, &t.counts, (size_t)nV * D.nchunks)); CHK(dalloc(ctx, &t.offs, (size_t)nV * D.nchunks)); CHK(dalloc(ctx, &t.blkv, (size_t)N)); CHK(dalloc(ctx, &t.bincnt, (size_t)nV * Q));
    ctx->sort_overlap = true;
    { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // lowest priority: the shuffle only fills gaps
      HIPCHK(hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, lo)); }
    for (int i = 0; i < 2; i++) { HIPCHK(hipEventCreateWithFlags(&ctx->ev_sorted[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ctx->ev_free[i], hipEventDisableTiming)); }
    ctx->sort_sched = 3;      // (the per-round schedule, sort_sched = 1, lost round 4 to the batched shuffle; its switches are gone)
    ctx->oset_mask = ctx->sort_sched == 3 ? 3 : 1;
    { const char* si = getenv("HMX_SHUFFLE_INV"); const int v = si ? atoi(si) : 1;      // 0: counting sort always; 2: sort-free form on sharded runs too
      ctx->shuf_inv = ctx->sort_sched == 3 && v != 0 && (ctx->world == 1 || v == 2) && ctx->carry_ok &&      /* (without the carry every round needs D.blk: the counting sort has it for free) */
                      D.nb < 64 && Q < 2048 && ctx->N_global < ((int64_t)1 << 31) &&
                      ((size_t)D.nb * Q + (size_t)Q + 1) * sizeof(int) + 5 * 4096 <= 64 * 1024; }     // (lpair packs the combination in 19 bits and the blocks in 6; posr the combination in 11)
    if (ctx->shuf_inv) {
      const int P = shuffle_parts((uint64_t)ctx->N_global, D.nb, ctx->cells_per_block);
      for (int i = 0; i < 4; i++) { CHK(dalloc(ctx, &ctx->posr[i], (size_t)ctx->N_global)); CHK(dalloc(ctx, &ctx->shuf_partcnt[i], (size_t)nV * Q * P));
        CHK(dalloc(ctx, &ctx->shuf_binacc[i], (size_t)nV * Q)); HIPCHK(hipMemsetAsync(ctx->shuf_binacc[i], 0, sizeof(int) * (size_t)nV * Q, ctx->L.stream)); }
    }
    if (ctx->sort_sched == 3) for (int i = 2; i < 4; i++) {
      hmx_ctx::SortSet& u = ctx->sets[i];
      CHK(dalloc(ctx, &u.blk, (size_t)N)); CHK(dalloc(ctx, &u.lorder, (size_t)3 * D.npad + 2)); u.lpair = reinterpret_cast<int2*>(u.lorder + (((size_t)D.npad + 1) & ~(size_t)1));
      CHK(dalloc(ctx, &u.lcombo, (size_t)D.npad)); CHK(dalloc(ctx, &u.binoff, (size_t)nV * Q + 1)); CHK(dalloc(ctx, &u.boff, (size_t)D.nb + 1));
      CHK(dalloc(ctx, &u.counts, (size_t)nV * D.nchunks)); CHK(dalloc(ctx, &u.offs, (size_t)nV * D.nchunks)); CHK(dalloc(ctx, &u.blkv, (size_t)N)); CHK(dalloc(ctx, &u.bincnt, (size_t)nV * Q));
    }
  }
  CHK(dalloc(ctx, &D.items, items.size())); CHK(dalloc(ctx, &D.aitems, aitems.size())); CHK(dalloc(ctx, &D.titems, titems.size()));
  CHK(dalloc(ctx, &D.Sq, (size_t)Q * d * K)); CHK(dalloc(ctx, &D.nq, (size_t)Q * K));
  { const char* e = getenv("HMX_MOE_SOLVE"); ctx->solve_on_device = !(e && std::string(e) == "host") && (size_t)(B + 1) * 16 * 8 + (size_t)(4 * B + 8 + C) * 4 <= 158 * 1024; }   // (LDS panel of the device Cholesky)
  ctx->y_on_device = false; ctx->solve_pending = false;
  if (ctx->solve_on_device) {
    const size_t M = (size_t)B + 1;
    CHK(dalloc(ctx, &ctx->sv_cov, (size_t)K * M * M)); CHK(dalloc(ctx, &ctx->sv_rhs, (size_t)K * d * M)); CHK(dalloc(ctx, &ctx->sv_Wall, (size_t)K * d * M));
    CHK(dalloc(ctx, &ctx->sv_mrows, (size_t)K)); CHK(dalloc(ctx, &ctx->sv_flags, (size_t)K)); CHK(dalloc(ctx, &ctx->sv_lambda, M)); CHK(dalloc(ctx, &ctx->sv_cov_bounds, (size_t)C));
    if (!ctx->lambda_estimation) CHK(h2d(ctx, ctx->sv_lambda, ctx->lambda.data(), M));
    CHK(h2d(ctx, ctx->sv_cov_bounds, ctx->cov_bounds.data(), (size_t)C));
    HIPCHK(hipMemsetAsync(ctx->sv_flags, 0, sizeof(int) * (size_t)K, ctx->L.stream));
    HIPCHK(hipMemsetAsync(ctx->sv_mrows, 0, sizeof(int) * (size_t)K, ctx->L.stream));
  }
  CHK(dalloc(ctx, &D.solve_err, (size_t)1)); HIPCHK(hipMemsetAsync(D.solve_err, 0, sizeof(int), ctx->L.stream));
  CHK(dalloc(ctx, &D.S0, (size_t)K * d)); CHK(dalloc(ctx, &D.n0, (size_t)K)); CHK(dalloc(ctx, &D.qstart, (size_t)Q + 1)); CHK(dalloc(ctx, &D.sizes, (size_t)B));
  CHK(h2d(ctx, D.qstart, start.data(), (size_t)Q + 1)); CHK(h2d(ctx, D.sizes, ctx->sizes.data(), (size_t)B)); CHK(dalloc(ctx, &D.Wq, (size_t)Q * K * d)); CHK(dalloc(ctx, &D.Wimg, D.moe_mfma ? (size_t)Q * D.wNQ * D.wNS * 256 : 1));
  { // deterministic statistics pass (k_moe_stats_q): static split of the 16-cell tiles over ~2 workgroups per CU; one partial
    // slot per (workgroup, combination met) -- known here because the tiles are listed by combination
    const char* e = getenv("HMX_MOE_STATS");
    D.st_dma = (D.moe_mfma && D.NCT <= 8 && !(e && std::string(e) == "atomic")) ? 1 : 0;
    if (D.st_dma) {
      const int nt = (int)titems.size();
      int tpw = (nt + 2 * 256 - 1) / (2 * 256); if (tpw < 16) tpw = 16;
      D.st_cpw = tpw; D.st_nwg = (nt + tpw - 1) / tpw;
      std::vector<int> slot0((size_t)D.st_nwg), qptr((size_t)Q + 1, 0);
      std::vector<std::vector<int>> byq((size_t)Q);
      int nslots = 0;
      for (int w = 0; w < D.st_nwg; w++) {
        slot0[w] = nslots;
        int last = -1;
        for (int t = w * tpw; t < std::min(nt, (w + 1) * tpw); t++) if (titems[t].q != last) { last = titems[t].q; byq[(size_t)last].push_back(nslots++); }
      }
      std::vector<int> qslots; qslots.reserve((size_t)nslots);
      for (int q = 0; q < Q; q++) { qptr[q] = (int)qslots.size(); qslots.insert(qslots.end(), byq[q].begin(), byq[q].end()); }
      qptr[Q] = (int)qslots.size();
      CHK(dalloc(ctx, &D.st_part, (size_t)std::max(nslots, 1) * ((size_t)K * d + K))); CHK(dalloc(ctx, &D.st_slot0, slot0.size()));
      CHK(dalloc(ctx, &D.st_qptr, qptr.size())); CHK(dalloc(ctx, &D.st_qslots, std::max<size_t>(qslots.size(), 1)));
      CHK(h2d(ctx, D.st_slot0, slot0.data(), slot0.size())); CHK(h2d(ctx, D.st_qptr, qptr.data(), qptr.size()));
      if (!qslots.empty()) CHK(h2d(ctx, D.st_qslots, qslots.data(), qslots.size()));
    }
  }
  CHK(dalloc(ctx, &D.km_gcells, (size_t)K)); CHK(dalloc(ctx, &D.km_rows, (size_t)K * d)); CHK(dalloc(ctx, &D.km_excl, (size_t)K));
  CHK(dalloc(ctx, &D.seedmin, (size_t)K)); CHK(dalloc(ctx, &D.lsum, (size_t)K * d + K)); D.lcnt = reinterpret_cast<unsigned long long*>(D.lsum + (size_t)K * d); CHK(dalloc(ctx, &D.ynorm, (size_t)K));
  CHK(h2d(ctx, D.perm, ctx->perm.data(), (size_t)N)); CHK(h2d(ctx, D.invperm, invperm.data(), (size_t)N));
  CHK(h2d(ctx, D.combo, combo_sorted.data(), (size_t)N)); CHK(h2d(ctx, D.qlev, ctx->qlev.data(), ctx->qlev.size()));
  CHK(h2d(ctx, D.sigma, ctx->sigma.data(), (size_t)K)); CHK(h2d(ctx, D.theta, ctx->theta.data(), (size_t)B)); CHK(h2d(ctx, D.Pr_b, ctx->Pr_b.data(), (size_t)B));
  CHK(h2d(ctx, D.schunks, schunks.data(), schunks.size())); CHK(h2d(ctx, D.qchunk, qchunk.data(), qchunk.size()));
  { std::vector<float> ce(K), cl(K);
    for (int k = 0; k < K; k++) { ce[k] = -1.44269504088896341f / ctx->sigma[k]; cl[k] = ctx->sigma[k] * 0.693147180559945309f; }
    CHK(h2d(ctx, D.ce, ce.data(), (size_t)K)); CHK(h2d(ctx, D.cl, cl.data(), (size_t)K)); }
  CHK(h2d(ctx, D.items, items.data(), items.size())); CHK(h2d(ctx, D.aitems, aitems.data(), aitems.size())); CHK(h2d(ctx, D.titems, titems.data(), titems.size()));
  HIPCHK(hipMemsetAsync(D.O_fx, 0, sizeof(long long) * B * K, ctx->L.stream));
  HIPCHK(hipMemsetAsync(D.Snew_fx, 0, sizeof(long long) * (size_t)D.nrep * B * K, ctx->L.stream));
  HIPCHK(hipMemsetAsync(D.Snew_alt, 0, sizeof(long long) * (size_t)D.nrep * B * K, ctx->L.stream));
  HIPCHK(hipMemsetAsync(D.O_alt, 0, sizeof(long long) * B * K, ctx->L.stream));
  HIPCHK(hipMemsetAsync(D.objpart, 0, sizeof(double) * 2 * (size_t)D.objslots * D.nwmax, ctx->L.stream));
  HIPCHK(hipMemsetAsync(D.obj, 0, sizeof(double) * 8, ctx->L.stream));
  HIPCHK(hipMemsetAsync(D.R, 0, sizeof(float) * (size_t)N * K, ctx->L.stream));
  HIPCHK(hipMemsetAsync(D.Zo, 0, sizeof(float) * (size_t)N * D.zs, ctx->L.stream));
  HIPCHK(hipMemsetAsync(D.Zc, 0, sizeof(float) * (size_t)N * D.zs, ctx->L.stream));
  HIPCHK(hipMemsetAsync(D.Wq, 0, sizeof(float) * (size_t)Q * K * d, ctx->L.stream));
  // Z: d x N (cell-major), double (the R seam, conv_to :41) or float, on the host or already in HBM -> fp32 rows in internal
  // order.  Host input goes through two HBM staging slabs: the copy of slab s+1 (copy stream) overlaps the conversion of slab s.
  if (z_location != HMX_DEVICE && xfer_mode() == 2) (void)xfer_ring(ctx->device).ensure();    // (once per process: not part of a matrix's transfer time)
  // (the buffers' first touch by the clears above is allocate_buffers' time, and the first launch of a library kernel in a process loads the
  //  code object -- tens of ms once per process --: neither is the ingest's)
  l_copy(ctx->L, D.Zo, D.Zo, 0); KCHK();
  HIPCHK(hipStreamSynchronize(ctx->L.stream));
  {
    const double t_in = now_ms();
    const int f32 = z_dtype == HMX_F32;
    const size_t esz = f32 ? 4 : 8;
    if (z_location == HMX_DEVICE) {
      l_convert_in(ctx->L, Z, f32, D.Zo, D.invperm, (int)N, d, D.zs); KCHK();
      HIPCHK(hipStreamSynchronize(ctx->L.stream));
    } else if (xfer_mode() == 2 && xfer_ring(ctx->device).ensure()) {
      // ring of page-locked slots: host threads fill slot b while the DMA engine drains the earlier ones and the conversion kernel
      // consumes what has landed (two HBM staging slabs)
      XferRing& ring = xfer_ring(ctx->device);
      std::lock_guard<std::mutex> ring_lock(ring.mu);
      XferPool pool; pool.start(xfer_threads());
      const int64_t slab = std::max<int64_t>(1, (int64_t)XferRing::SLOT / ((int64_t)esz * d));
      hipStream_t cs = ring.cs;
      hipEvent_t* copied = ring.ev_a; hipEvent_t* used = ring.ev_b; hipEvent_t* left = ring.ev_slot;   // left[b]: slot b's bytes have left for the device
      ctx->timers["ingest_pinned"] = 2.0;
      hipError_t e = hipSuccess;
      int it = 0;
      for (int64_t s0 = 0; s0 < N && e == hipSuccess; s0 += slab, it++) {
        const int64_t cnt = std::min<int64_t>(slab, N - s0);
        const size_t nbytes = (size_t)cnt * d * esz;
        const int b = it & 1, rb = it % XferRing::NB;
        if (it >= XferRing::NB) e = hipEventSynchronize(left[rb]);
        if (e != hipSuccess) break;
        pool.copy(ring.slot[rb], (const char*)Z + (size_t)s0 * d * esz, nbytes);
        if (it >= 2) e = hipStreamWaitEvent(cs, used[b], 0);           // the slab's previous conversion has read it
        if (e == hipSuccess) e = hipMemcpyAsync(ring.stage[b], ring.slot[rb], nbytes, hipMemcpyHostToDevice, cs);
        if (e == hipSuccess) e = hipEventRecord(left[rb], cs);
        if (e == hipSuccess) e = hipEventRecord(copied[b], cs);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->L.stream, copied[b], 0);
        if (e == hipSuccess) { l_convert_in(ctx->L, ring.stage[b], f32, D.Zo, D.invperm + s0, (int)cnt, d, D.zs); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipEventRecord(used[b], ctx->L.stream);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->L.stream);
      (void)hipStreamSynchronize(cs);
      if (e != hipSuccess) return fail(ctx, HMX_ERR_DEVICE, hipGetErrorString(e));
    } else {
      const int64_t slab = std::max<int64_t>(1, (int64_t)(128ll << 20) / ((int64_t)esz * d));
      const int64_t scnt = std::min<int64_t>(slab, N);
      void* stage[2] = {nullptr, nullptr}; hipStream_t cs = nullptr; hipEvent_t copied[2] = {nullptr, nullptr}, used[2] = {nullptr, nullptr};
      // The caller's matrix is pageable (R's heap): page-lock it for the duration of the ingest, so that the slab copies are real DMA
      // at PCIe speed instead of the runtime's staged pageable path (HMX_PIN=0 leaves it pageable; a failed registration is not an error).
      const bool pinned = xfer_mode() >= 1 && hipHostRegister(const_cast<void*>(Z), (size_t)N * d * esz, hipHostRegisterDefault) == hipSuccess;
      if (!pinned) (void)hipGetLastError();
      ctx->timers["ingest_pinned"] = pinned ? 1.0 : 0.0;
      hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
      for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipMalloc(&stage[i], (size_t)scnt * d * esz);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&copied[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&used[i], hipEventDisableTiming);
      }
      int it = 0;
      for (int64_t s0 = 0; s0 < N && e == hipSuccess; s0 += slab, it++) {
        const int64_t cnt = std::min<int64_t>(slab, N - s0);
        const int b = it & 1;
        if (it >= 2) e = hipStreamWaitEvent(cs, used[b], 0);           // the slab's previous conversion has read it
        if (e == hipSuccess) e = hipMemcpyAsync(stage[b], (const char*)Z + (size_t)s0 * d * esz, (size_t)cnt * d * esz, hipMemcpyHostToDevice, cs);
        if (e == hipSuccess) e = hipEventRecord(copied[b], cs);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->L.stream, copied[b], 0);
        if (e == hipSuccess) { l_convert_in(ctx->L, stage[b], f32, D.Zo, D.invperm + s0, (int)cnt, d, D.zs); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipEventRecord(used[b], ctx->L.stream);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->L.stream);
      for (int i = 0; i < 2; i++) { if (stage[i]) (void)hipFree(stage[i]); if (copied[i]) (void)hipEventDestroy(copied[i]); if (used[i]) (void)hipEventDestroy(used[i]); }
      if (cs) (void)hipStreamDestroy(cs);
      if (pinned) (void)hipHostUnregister(const_cast<void*>(Z));
      if (e != hipSuccess) return fail(ctx, HMX_ERR_DEVICE, hipGetErrorString(e));
    }
    ctx->timers["ingest_Z"] = now_ms() - t_in;
  }
  ctx->W.assign((size_t)(B + 1) * d, 0.f); ctx->W_rows = B + 1;  // allocate_buffers :127
  ctx->Y.assign((size_t)d * K, 0.f);
  { const char* e = getenv("HMX_FUSED_FOLD");
    ctx->fused_ok = !(e && std::string(e) == "0") && D.upd_impl == 0 &&
                    (size_t)D.NQ * D.NS * 1024 + (size_t)B * K * 12 + (size_t)Q * C * 4 + 64 <= 150 * 1024; }
  { // persistent block chain: one workgroup per CU must be resident at once (they synchronise inside the launch)
    const char* e = getenv("HMX_CHAIN");
    int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    ctx->chain_wgs = cus;
    if (const char* w = getenv("HMX_CHAIN_WGS")) ctx->chain_wgs = std::max(8, std::min(cus, atoi(w)));   // (tests: two ranks sharing one GPU)
    // The chain pays off while a block step is latency-bound: a few 16-cell tiles per resident wave (1.5 at 1M cells, where a
    // step takes 21 us against 27.5 + 3 us of launch gap).  At 10M cells per GPU (15 tiles per wave) the per-step launches
    // stream just as well and were measured 6 % faster (142 vs 151 us per step): HMX_CHAIN=1 forces the chain there.
    const double tiles_per_wave = (double)N / std::max(D.nb, 1) / 16.0 / (8.0 * std::max(cus - 1, 1));
    double max_tpw = 6.0; if (const char* m = getenv("HMX_CHAIN_MAX_TPW")) max_tpw = atof(m);       // (tests: move the threshold between two shards)
    const bool chain_fits = (e && std::string(e) == "1") || tiles_per_wave <= max_tpw;
    ctx->chain_ok = !(e && std::string(e) == "0") && chain_fits && ctx->fused_ok && cus >= 8 && D.NCT <= 7 && D.NT4 <= 4 && D.nb <= 64 &&
                    (size_t)D.NQ * D.NS * 1024 + (size_t)B * K * 12 + (size_t)Q * C * 4 + 64 <= 150 * 1024;
    if (ctx->world > 1 || ctx->comm_force) {
      // The flags pick the inter-rank PROTOCOL of update_R (in-launch exchange of the persistent chain / one all-reduce per block
      // step): every rank must take the same path, but chain_ok depends on the LOCAL cell count and CU count.  Agree on the minimum
      // -- before anything is derived from the flags (the replica count below sizes a per-block all-reduce).
      long long* dflag; long long hf[2] = {ctx->chain_ok ? 1 : 0, ctx->fused_ok ? 1 : 0};
      CHK(dalloc(ctx, &dflag, (size_t)2));
      CHK(h2d(ctx, dflag, hf, 2)); CHK(allreduce(ctx, dflag, 2, 2)); CHK(d2h(ctx, hf, dflag, 2));
      ctx->chain_ok = hf[0] != 0; ctx->fused_ok = hf[1] != 0;
    }
    CHK(dalloc(ctx, &D.tail_ticket, (size_t)1)); HIPCHK(hipMemsetAsync(D.tail_ticket, 0, sizeof(int), ctx->L.stream));
    CHK(dalloc(ctx, &D.pen_g, (size_t)B * K)); CHK(dalloc(ctx, &D.chain_ctl, (size_t)8 * D.nb + 24)); CHK(dalloc(ctx, &D.chain_dbg, (size_t)64));
    HIPCHK(hipMemsetAsync(D.chain_dbg, 0, sizeof(unsigned long long) * 64, ctx->L.stream));
    HIPCHK(hipMemsetAsync(D.pen_g, 0, sizeof(unsigned long long) * (size_t)B * K, ctx->L.stream));
    HIPCHK(hipMemsetAsync(D.chain_ctl, 0, sizeof(int) * ((size_t)8 * D.nb + 24), ctx->L.stream));
    D.chain_wps = 2;      // (the 3- / 4-waves-per-SIMD chain variants and the in-chain gathering of the old contributions lost rounds 2 and 3: removed in round 5)
    // the folder reads AND resets every replica of the contribution table inside a block step (atomic exchanges on its critical
    // path): 4 replicas measured 0.4 us per step faster than 8 there (2: the workers' atomics start to queue, +2 us)
    if (ctx->chain_ok && !getenv("HMX_NREP") && D.nrep > 4) {
      D.nrep = 4;
      for (int i = 0; i < 3; i++) D.Snew_set[i] = ctx->sold_buf[0] + (size_t)2 * D.nb * B * K + (size_t)i * D.nrep * B * K;   // keep the three sets contiguous
    }
    D.upd_contig = (!ctx->chain_ok && tiles_per_wave >= 4.0) ? 1 : 0;     // launch-per-step path: contiguous tile ranges once a wave has several tiles per block
    ctx->chain_rounds = 0;
    D.p2p_world = 0; D.p2p_rank = ctx->p2p_rank;
    for (int g = 0; g < 8; g++) D.p2p_inbox[g] = ctx->p2p_peer[g]; }
  ctx->invperm_h = invperm; ctx->combo_h = combo_sorted;
  ctx->Zc_head = nullptr; ctx->Yt_head = nullptr; ctx->head_is_stale = false;
  if (ctx->stale_dist) { CHK(dalloc(ctx, &ctx->Zc_head, (size_t)N * D.zs)); CHK(dalloc(ctx, &ctx->Yt_head, (size_t)d * K)); }
  CHK(seq_setup_static(ctx));
  ctx->ran_setup = true;
  return hmx_restart(ctx);
}

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
    HIPCHK(hipMemsetAsync(D.Sq, 0, sizeof(double) * (size_t)Q * d * K, ctx->L.stream));
    HIPCHK(hipMemsetAsync(D.nq, 0, sizeof(double) * (size_t)Q * K, ctx->L.stream));
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

// ---- diagnostics of the restarted sequential sums (tests/test_gpu_seq.py): the machinery alone, on caller-provided data -----------
// totals[c][0][k] = the fp32 value of   s = 0; for i in chain c: s += R[list[off_c + i]][k]   (one add after the other);
// totals[c][1 + b][k] = the same loop over the chain's cells of level b only
int hmx_debug_seq_oe(const float* R, int64_t n, int32_t K, const int32_t* level, int32_t B, const int32_t* list, int64_t nlist,
                     const int32_t* chain_off, const int32_t* chain_cnt, int32_t nchains, int32_t seg_cells, int32_t passes, float* totals,
                     int64_t* mismatch, double* residual) {
  if (!R || !list || !level || !chain_off || !chain_cnt || !totals || n <= 0 || K <= 0 || B <= 0 || nchains <= 0 || seg_cells <= 0 || passes < 2) return HMX_ERR_ARG;
  // (a probe, but an exported one: every index the kernels will use is checked here -- cells inside R, levels inside the LDS rows, chains
  //  inside the list, the level rows inside the LDS budget of a workgroup)
  if (nlist <= 0 || n > 2000000000ll || nlist > 2000000000ll || (size_t)B * 256 > 60 * 1024) return HMX_ERR_ARG;
  for (int64_t i = 0; i < nlist; i++) if (list[i] < 0 || list[i] >= n) return HMX_ERR_ARG;
  for (int64_t i = 0; i < n; i++) if (level[i] < 0 || level[i] >= B) return HMX_ERR_ARG;
  for (int c = 0; c < nchains; c++) if (chain_off[c] < 0 || chain_cnt[c] < 0 || (int64_t)chain_off[c] + chain_cnt[c] > nlist) return HMX_ERR_ARG;
  hmx_ctx* ctx = hmx_create();
  float* dR = nullptr; int* dl = nullptr; int* dlev = nullptr; int* dq = nullptr;
  auto run = [&]() -> int {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ctx, HMX_ERR_DEVICE, "no HIP device");
    HIPCHK(hipStreamCreateWithFlags(&ctx->L.stream, hipStreamNonBlocking)); ctx->own_stream = true;
    HIPCHK(hipMalloc((void**)&dR, sizeof(float) * (size_t)n * K)); HIPCHK(hipMalloc((void**)&dl, sizeof(int) * (size_t)nlist));
    HIPCHK(hipMalloc((void**)&dlev, sizeof(int) * (size_t)n)); HIPCHK(hipMalloc((void**)&dq, sizeof(int) * (size_t)B));
    std::vector<int> ident(B); std::iota(ident.begin(), ident.end(), 0);
    CHK(h2d(ctx, dR, R, (size_t)n * K)); CHK(h2d(ctx, dl, list, (size_t)nlist)); CHK(h2d(ctx, dlev, level, (size_t)n)); CHK(h2d(ctx, dq, ident.data(), (size_t)B));
    ctx->K = K; ctx->B = B; ctx->seq_passes = passes; ctx->seq_stats = true;
    ctx->D.R = dR; ctx->D.K = K; ctx->D.B = B; ctx->D.C = 1; ctx->D.combo = dlev; ctx->D.qlev = dq;      // one covariate: combination == level
    std::vector<std::pair<int, int>> ch;
    for (int c = 0; c < nchains; c++) ch.push_back({chain_off[c], chain_cnt[c]});
    CHK(seq_plan_build(ctx, ctx->plan_round, ch, seg_cells));
    CHK(seq_run_oe(ctx, ctx->plan_round, dl, nullptr, 0, nchains));
    CHK(d2h(ctx, totals, ctx->sq_total, (size_t)nchains * (1 + B) * K));
    unsigned mm[2] = {0, 0}; CHK(d2h(ctx, mm, ctx->sq_conv, 2));
    if (mismatch) *mismatch = (int64_t)mm[0];
    if (residual) { float r; std::memcpy(&r, &mm[1], 4); *residual = (double)r; }
    return 0;
  };
  const int st = run();
  for (void* q : {(void*)dR, (void*)dl, (void*)dlev, (void*)dq}) if (q) (void)hipFree(q);
  hmx_destroy(ctx);
  return st;
}
// total[a] = the fp32 value of   s = 0; for i < n: s += T[a * n + i]
int hmx_debug_seq_arr(const float* T, int64_t n, int32_t narr, int32_t seg_terms, int32_t passes, float* total, int64_t* mismatch, double* residual) {
  if (!T || !total || n <= 0 || narr <= 0 || narr > 64 || seg_terms <= 0 || passes < 2) return HMX_ERR_ARG;
  hmx_ctx* ctx = hmx_create();
  float* dT = nullptr;
  auto run = [&]() -> int {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ctx, HMX_ERR_DEVICE, "no HIP device");
    HIPCHK(hipStreamCreateWithFlags(&ctx->L.stream, hipStreamNonBlocking)); ctx->own_stream = true;
    HIPCHK(hipMalloc((void**)&dT, sizeof(float) * (size_t)n * narr));
    CHK(h2d(ctx, dT, T, (size_t)n * narr));
    const int nsegs = (int)((n + seg_terms - 1) / seg_terms);
    CHK(seq_workspace(ctx, (size_t)narr * nsegs, (size_t)narr));
    CHK(seq_grow(ctx, ctx->obj_partial, ctx->obj_partial_cap, (size_t)narr * ((nsegs + 255) / 256)));
    for (int p = 0; p < passes; p++) {
      l_seq_arr_pass(ctx->L, dT, n, n, narr, seg_terms, nsegs, ctx->sq_start, ctx->sq_end, p == 0, ctx->obj_partial, p == passes - 1 ? ctx->sq_mismatch : nullptr); KCHK();
      l_seq_scan1(ctx->L, narr, nsegs, ctx->sq_start, ctx->sq_end, ctx->sq_start, ctx->sq_total, p == passes - 1 ? ctx->sq_mismatch : nullptr, p == 0, ctx->obj_partial); KCHK();
    }
    CHK(d2h(ctx, total, ctx->sq_total, (size_t)narr));
    unsigned mm[2] = {0, 0}; CHK(d2h(ctx, mm, ctx->sq_mismatch, 2));
    if (mismatch) *mismatch = (int64_t)mm[0];
    if (residual) { float r; std::memcpy(&r, &mm[1], 4); *residual = (double)r; }
    return 0;
  };
  const int st = run();
  if (dT) (void)hipFree(dT);
  hmx_destroy(ctx);
  return st;
}

// host-only micro-benchmark of the ridge solves (no device needed): returns microseconds per call of the full K-cluster loop
double hmx_debug_solve_bench(int K, int B, int d, int reps) {
  hmx_ctx c; c.K = K; c.B = B; c.C = 1; c.d = d; c.Q = B; c.B_vec = {B}; c.cov_bounds = {B};
  c.sizes.assign(B, 1000.f); c.lambda_estimation = true; c.alpha = 0.2f; c.cutoff = 1e-5f;
  c.qlev.resize(B); for (int b = 0; b < B; b++) c.qlev[b] = b;
  std::vector<float> O((size_t)K * B, 50.f), E((size_t)K * B, 40.f);
  std::vector<double> Sq((size_t)B * K * d, 0.3), nq((size_t)B * K, 50.0);
  for (size_t i = 0; i < Sq.size(); i++) Sq[i] = 0.1 + 1e-3 * (double)(i % 97);
  std::vector<float> Wq((size_t)B * K * d), Ynew((size_t)K * d);
  std::vector<SolveOut> outs(K);
  const double t0 = now_ms();
  for (int r = 0; r < reps; r++) for (int k = 0; k < K; k++) solve_cluster(&c, k, O, E, Sq, nq, Wq, Ynew, outs[k]);
  return 1e3 * (now_ms() - t0) / reps;
}

int64_t hmx_get(hmx_ctx* ctx, const char* field, double* out, int64_t cap) {
  if (!ctx || !field) return -1;
  const std::string f(field);
  if (f.rfind("objective_", 0) == 0 || f == "kmeans_rounds") { if (flush_objectives(ctx)) return -1; }
  // (-1 with hmx_last_error = "singular ridge system": the deferred verdict of the last device-side correction, not an unknown field)
  if (f == "Y" || f == "W" || f == "W_rows" || f == "subset_clusters" || f == "skipped_clusters") { if (sync_solve_results(ctx)) return -1; }
  auto scalar = [&](double v) -> int64_t { if (out && cap >= 1) out[0] = v; return 1; };
  auto vec = [&](const auto& v) -> int64_t {
    if (out) for (size_t i = 0; i < v.size() && (int64_t)i < cap; i++) out[i] = (double)v[i];
    return (int64_t)v.size();
  };
  if (f == "N") return scalar((double)ctx->N_global);
  if (f == "N_local") return scalar((double)ctx->N);
  if (f == "B") return scalar(ctx->B);
  if (f == "K") return scalar(ctx->K);
  if (f == "d") return scalar(ctx->d);
  if (f == "alpha") return scalar(ctx->alpha);
  if (f == "max_iter_kmeans") return scalar(ctx->max_iter_kmeans);
  if (f == "block_size") return scalar(ctx->block_size);
  if (f == "n_blocks") return scalar(ctx->nb);
  if (f == "cells_per_block") return scalar((double)ctx->cells_per_block);
  if (f == "W_rows") return scalar(ctx->W_rows);
  if (f == "n_combos") return scalar(ctx->Q);
  if (f == "subset_clusters") return scalar((double)ctx->subset_clusters);
  if (f == "skipped_clusters") return scalar((double)ctx->skipped_clusters);
  if (f == "comm:calls") return scalar((double)ctx->comm_calls);
  if (f == "comm:bytes") return scalar((double)ctx->comm_bytes);
  if (f == "trace") {   // HMX_TRACE=1: raw per-wave stamps of the last block-update launch
    if (!ctx->D.trace) return -1;
    const int64_t n = (int64_t)ctx->D.nwmax * 16;
    if (out) {
      std::vector<unsigned long long> h((size_t)n);
      (void)hipStreamSynchronize(ctx->L.stream);
      (void)hipMemcpy(h.data(), ctx->D.trace, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost);
      for (int64_t i = 0; i < std::min(n, cap); i++) out[i] = (double)(h[(size_t)i] & ((1ull << 52) - 1));
    }
    return n;
  }
  if (f == "usig") return scalar((double)ctx->D.usig);
  if (f == "upd_wps") return scalar((double)ctx->D.upd_wps);
  if (f.rfind("prof:", 0) == 0 && ctx->ev_used) {   // resolve the pending event pairs (one sync, outside any timed region)
    (void)hipStreamSynchronize(ctx->L.stream);
    for (size_t i = 0; i < ctx->ev_used; i++) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ctx->ev_pool[i].first, ctx->ev_pool[i].second) == hipSuccess) ctx->prof_update_ms += ms;
    }
    ctx->prof_update_launches += (int64_t)ctx->ev_used; ctx->ev_used = 0;
  }
  if (f == "prof:update_ms") return scalar(ctx->prof_update_ms);
  if (f == "prof:update_launches") return scalar((double)ctx->prof_update_launches);
  if (f == "prof:update_cells") return scalar((double)ctx->prof_update_cells);
  if (f == "prof:update_steps") return scalar((double)ctx->prof_update_steps);
  if (f == "sync") {        // everything queued on the handle's streams has completed (hosts without a HIP runtime of their own: bench.py --bootstrap file)
    if (ctx->L.stream && hipStreamSynchronize(ctx->L.stream) != hipSuccess) return -1;
    if (ctx->side && hipStreamSynchronize(ctx->side) != hipSuccess) return -1;
    return scalar(1.0);
  }
  if (f == "chain") return scalar(ctx->chain_ok ? 1.0 : 0.0);
  if (f == "dot_bf") return scalar(ctx->D.dot_bf ? 1.0 : 0.0);     // split-bf16 tile kernels offered (each launch still checks its LDS budget)
  if (f == "sold_carry") return scalar(ctx->carry_ok ? 1.0 : 0.0);
  if (f == "carried_rounds") return scalar((double)ctx->carried_rounds);
  if (f == "rounds_without_R") return scalar((double)ctx->rounds_without_R);
  if (f == "chain_rounds") return scalar((double)ctx->chain_rounds);
  if (f == "shuffle_inv") return scalar(ctx->shuf_inv ? 1.0 : 0.0);
  if (f == "p2p:exchange_us") return scalar(ctx->p2p_exchange_us);
  if (f == "p2p:allreduce_calls") return scalar((double)ctx->p2p_ar_calls);
  if (f == "p2p:allreduce_big_windows") return scalar((double)ctx->p2p_ar_big_windows);
  if (f == "p2p") return scalar(ctx->p2p_on && ctx->p2p_world == ctx->world ? 1.0 : 0.0);
  if (f == "chain_dbg") {   // accumulated 100 MHz ticks of the persistent chain's phases (see hmx_internal.h); reading resets them
    if (!ctx->ran_setup) return -1;
    if (!out) return 64;      // [0..12] folder / workgroup 0 wave 0 phases, [16..31] workgroup 0 and [32..47] workgroup 100: per wave busy ticks, tiles; [48..55] / [56..63]: wave 4 / 5 of workgroup 0, the phases of [4..11]
    std::vector<unsigned long long> h(64);
    if (d2h(ctx, h.data(), ctx->D.chain_dbg, 64)) return -1;
    (void)hipMemsetAsync(ctx->D.chain_dbg, 0, sizeof(unsigned long long) * 64, ctx->L.stream);
    return vec(h);
  }
  if (f.rfind("gputimer:", 0) == 0) {   // GPU time of a phase (profile mode), ms; "gputimer:Rcells_update" == "prof:update_ms"
    resolve_phases(ctx);
    const std::string nm = f.substr(9);
    if (nm == "Rcells_update") return hmx_get(ctx, "prof:update_ms", out, cap);
    auto it = ctx->gpu_timers.find(nm); return scalar(it == ctx->gpu_timers.end() ? 0.0 : it->second);
  }
  if (f.rfind("timer:", 0) == 0) { auto it = ctx->timers.find(f.substr(6)); return scalar(it == ctx->timers.end() ? 0.0 : it->second); }
  if (f == "seed_cells") return vec(ctx->seed_cells);
  if (f == "Y") return vec(ctx->Y);
  if (f == "W") return vec(ctx->W);
  if (f == "Pr_b") return vec(ctx->Pr_b);
  if (f == "theta") return vec(ctx->theta);
  if (f == "sigma") return vec(ctx->sigma);
  if (f == "lambda") return vec(ctx->lambda);
  if (f == "B_vec") return vec(ctx->B_vec);
  if (f == "objective_kmeans") return vec(ctx->obj_kmeans);
  if (f == "objective_kmeans_dist") return vec(ctx->obj_dist);
  if (f == "objective_kmeans_entropy") return vec(ctx->obj_entropy);
  if (f == "objective_kmeans_cross") return vec(ctx->obj_cross);
  if (f == "objective_harmony") return vec(ctx->obj_harmony);
  if (f == "kmeans_rounds") return vec(ctx->kmeans_rounds);
  if (!ctx->ran_setup) return -1;
  if (hipSetDevice(ctx->device) != hipSuccess) return -1;
  if (f == "seq:mismatch" || f == "seq:residual") {      // last scans of the restarted sequential sums: segment starts that still moved / the largest move relative to its chain's largest start
    if (!ctx->sq_mismatch) return scalar(0.0);
    bool dummy = false;
    if (seq_settled(ctx, &dummy)) return -1;          // (folds the last unchecked scan's words in)
    (void)hipMemsetAsync(ctx->sq_conv, 0, 2 * sizeof(unsigned), ctx->L.stream);
    return scalar(f == "seq:mismatch" ? (double)ctx->seq_mismatch_sum : ctx->seq_resid_max);
  }
  if (f == "seq:extra_passes") return scalar((double)ctx->seq_extra_passes);
  if (f == "seq:unsettled") return scalar((double)ctx->seq_unsettled);
  if (f == "seq:group_passes" || f == "seq:group_runs") {
    if (!out) return 4;
    for (int g = 0; g < 4; g++) out[g] = (double)(f == "seq:group_passes" ? ctx->seq_group_passes[g] : ctx->seq_group_runs[g]);
    return 4;
  }
  if (f == "seq:runs") return scalar((double)ctx->seq_runs);
  if (f == "O" || f == "E" || f == "Lambda") {
    const int K = ctx->K, B = ctx->B;
    const int64_t cnt = (f == "Lambda") ? (int64_t)K * (B + 1) : (int64_t)K * B;
    if (!out) return cnt;
    std::vector<long long> ofx((size_t)B * K);
    if (d2h(ctx, ofx.data(), ctx->D.O_fx, ofx.size())) return -1;
    std::vector<float> Of_h, Ef_h;
    if (ctx->oe_arith) {       // the tables ARE fp32 in this mode
      Of_h.resize((size_t)B * K); Ef_h.resize((size_t)B * K);
      if (d2h(ctx, Of_h.data(), ctx->Of, Of_h.size()) || d2h(ctx, Ef_h.data(), ctx->Ef, Ef_h.size())) return -1;
    }
    if (f == "O") return vec(ctx->oe_arith ? Of_h : table_O(ctx, ofx));
    const std::vector<float> E = ctx->oe_arith ? Ef_h : table_E(ctx, ofx);
    if (f == "E") return vec(E);
    std::vector<double> L((size_t)K * (B + 1), 0.0);  // getLambda :657-669
    // (estimated: find_lambda_cpp's [0, alpha * E[k,:]], src/utils.cpp:159-163; fixed: the caller's whole vector, its entry 0 included)
    for (int k = 0; k < K; k++) {
      if (!ctx->lambda_estimation) L[k] = (double)ctx->lambda[0];
      for (int b = 0; b < B; b++)
        L[(size_t)(b + 1) * K + k] = ctx->lambda_estimation ? (double)(E[(size_t)b * K + k] * ctx->alpha) : (double)ctx->lambda[b + 1];
    }
    return vec(L);
  }
  if (f == "Z_corr" || f == "Z_orig" || f == "R") return hmx_get_matrix(ctx, field, out, HMX_F64, HMX_HOST, cap);
  ctx->err = "hmx_get: unknown field '" + f + "'";
  return -1;
}

// getZcorr / getZorig / getR (src/harmony.cpp:640-655) with a choice of element type and destination: double on the host is the
// R seam; float and/or a device pointer avoid the fp64 blow-up and the PCIe trip for hosts that keep working on the GPU.
// Host destinations are filled slab by slab through two HBM staging buffers (conversion of slab s+1 overlaps the copy of slab s).
int64_t hmx_get_matrix(hmx_ctx* ctx, const char* field, void* out, int32_t dtype, int32_t location, int64_t cap) {
  if (!ctx || !field || !ctx->ran_setup) return -1;
  const std::string f(field);
  if (f != "Z_corr" && f != "Z_orig" && f != "R") { ctx->err = "hmx_get_matrix: unknown field " + f; return -1; }
  if ((dtype != HMX_F64 && dtype != HMX_F32) || (location != HMX_HOST && location != HMX_DEVICE)) { ctx->err = "bad dtype / location"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) return -1;
  if (out && f == "Z_corr" && sync_solve_results(ctx)) return -1;      // a singular system of the last correction surfaces here (hmx_last_error says so)
  if (f == "R" && ctx->ran_init && !ctx->R_valid) { ctx->err = "R is not available: the clustering call that would have stored it did not complete"; return -1; }
  const int w = (f == "R") ? ctx->K : ctx->d;
  const int64_t cnt = ctx->N * w;
  if (!out || cap < cnt) return cnt;
  const float* src = (f == "R") ? ctx->D.R : (f == "Z_corr" ? ctx->D.Zc : ctx->D.Zo);
  const int ws = (f == "R") ? w : ctx->D.zs;
  const int f32 = dtype == HMX_F32;
  const size_t esz = f32 ? 4 : 8;
  const double t0 = now_ms();
  hipError_t e = hipSuccess;
  if (location == HMX_DEVICE) {
    l_convert_out(ctx->L, src, out, f32, ctx->D.invperm, ctx->D.n, w, ws);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->L.stream);
  } else if (xfer_mode() == 2 && xfer_ring(ctx->device).ensure()) {
    // conversion -> HBM staging slab -> DMA into a page-locked ring slot -> host threads copy the slot into the caller's (fresh, pageable)
    // matrix, touching its pages in parallel, while the next slots are on their way
    XferRing& ring = xfer_ring(ctx->device);
    std::lock_guard<std::mutex> ring_lock(ring.mu);
    XferPool pool; pool.start(xfer_threads());
    const int64_t slab = std::max<int64_t>(1, (int64_t)XferRing::SLOT / ((int64_t)esz * w));
    constexpr int NB = XferRing::NB, LAG = NB - 1;
    hipStream_t cs = ring.cs;
    hipEvent_t* conv = ring.ev_a; hipEvent_t* copied = ring.ev_b; hipEvent_t* arrived = ring.ev_slot;
    const int nslabs = (int)((ctx->N + slab - 1) / slab);
    auto drain = [&](int j) {          // slab j has arrived in its ring slot: out of the ring, into the caller's matrix
      const int64_t s0 = (int64_t)j * slab, c = std::min<int64_t>(slab, ctx->N - s0);
      hipError_t ee = hipEventSynchronize(arrived[j % NB]);
      if (ee == hipSuccess) pool.copy((char*)out + (size_t)s0 * w * esz, ring.slot[j % NB], (size_t)c * w * esz);
      return ee;
    };
    for (int it = 0; it < nslabs && e == hipSuccess; it++) {
      const int64_t s0 = (int64_t)it * slab, c = std::min<int64_t>(slab, ctx->N - s0);
      const int b = it & 1;
      if (it >= 2) e = hipStreamWaitEvent(ctx->L.stream, copied[b], 0);   // the slab's previous contents have left for the host
      if (e == hipSuccess) { l_convert_out(ctx->L, src, ring.stage[b], f32, ctx->D.invperm + s0, (int)c, w, ws); e = hipGetLastError(); }
      if (e == hipSuccess) e = hipEventRecord(conv[b], ctx->L.stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(cs, conv[b], 0);
      // (ring slot it % NB was drained by the host in iteration it - NB + LAG, i.e. before this point)
      if (e == hipSuccess) e = hipMemcpyAsync(ring.slot[it % NB], ring.stage[b], (size_t)c * w * esz, hipMemcpyDeviceToHost, cs);
      if (e == hipSuccess) e = hipEventRecord(copied[b], cs);
      if (e == hipSuccess) e = hipEventRecord(arrived[it % NB], cs);
      if (e == hipSuccess && it >= LAG) e = drain(it - LAG);
    }
    for (int j = std::max(0, nslabs - LAG); j < nslabs && e == hipSuccess; j++) e = drain(j);
    (void)hipStreamSynchronize(cs);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->L.stream);
  } else {
    const int64_t slab = std::max<int64_t>(1, (int64_t)(128ll << 20) / ((int64_t)esz * w));
    const int64_t scnt = std::min<int64_t>(slab, ctx->N);
    void* stage[2] = {nullptr, nullptr}; hipStream_t cs = nullptr; hipEvent_t conv[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
    const bool pinned = xfer_mode() >= 1 && hipHostRegister(out, (size_t)cnt * esz, hipHostRegisterDefault) == hipSuccess;     // (see hmx_setup_ex)
    if (!pinned) (void)hipGetLastError();
    e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
      e = hipMalloc(&stage[i], (size_t)scnt * w * esz);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&conv[i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&copied[i], hipEventDisableTiming);
    }
    int it = 0;
    for (int64_t s0 = 0; s0 < ctx->N && e == hipSuccess; s0 += slab, it++) {
      const int64_t c = std::min<int64_t>(slab, ctx->N - s0);
      const int b = it & 1;
      if (it >= 2) e = hipStreamWaitEvent(ctx->L.stream, copied[b], 0);   // the slab's previous contents have left for the host
      if (e == hipSuccess) { l_convert_out(ctx->L, src, stage[b], f32, ctx->D.invperm + s0, (int)c, w, ws); e = hipGetLastError(); }
      if (e == hipSuccess) e = hipEventRecord(conv[b], ctx->L.stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(cs, conv[b], 0);
      if (e == hipSuccess) e = hipMemcpyAsync((char*)out + (size_t)s0 * w * esz, stage[b], (size_t)c * w * esz, hipMemcpyDeviceToHost, cs);
      if (e == hipSuccess) e = hipEventRecord(copied[b], cs);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(cs);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->L.stream);
    for (int i = 0; i < 2; i++) { if (stage[i]) (void)hipFree(stage[i]); if (conv[i]) (void)hipEventDestroy(conv[i]); if (copied[i]) (void)hipEventDestroy(copied[i]); }
    if (cs) (void)hipStreamDestroy(cs);
    if (pinned) (void)hipHostUnregister(out);
  }
  if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return -1; }
  ctx->timers["egress_" + f] = now_ms() - t0;
  return cnt;
}

}  // extern "C"
