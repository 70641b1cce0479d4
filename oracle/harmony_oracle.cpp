// =============================================================================
// TEST INFRASTRUCTURE ONLY -- CPU oracle for the Harmony clustering+correction
// loop.  Nothing under oracle/ is part of the product: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// What it is: a from-scratch C++17 restatement (no Armadillo/Rcpp -- neither is
// installable here) of the reference engine, following it function by function:
//   setup / allocate_buffers     /root/reference/src/harmony.cpp:29-128
//   init_cluster_cpp             src/harmony.cpp:131-156
//   compute_objective            src/harmony.cpp:158-170, src/utils.cpp:67-81
//   check_convergence            src/harmony.cpp:173-205
//   cluster_cpp                  src/harmony.cpp:208-262
//   update_R                     src/harmony.cpp:269-342
//   moe_correct_ridge_cpp        src/harmony.cpp:345-638
//   kmeans_centers et al.        src/utils.cpp:10-108,159-163
//
// HOW FAR PARITY IS PINNED.  The reference's tests hold no golden numeric vectors (tests/testthat/*.R assert shapes, probability-simplex,
// finiteness and a chi2 ordering only) and the package as R builds it (R, Rcpp, RcppArmadillo, RcppProgress, BLAS/LAPACK) cannot be
// built here.  Two pins instead:
//   (1) the reference's OWN engine sources -- src/harmony.cpp, utils.cpp, timer.cpp, compiled where they lie, unmodified -- over
//       oracle/shim/ (a minimal stand-in for the Armadillo / Rcpp headers) give oracle/_ref/libharmony_ref.so, and this oracle's faithful
//       mode must equal it BIT FOR BIT after every call on R's random stream (tests/test_oracle_ref.py: the bundled fixtures, one / two /
//       three covariates, the subset and skip branches, fixed lambda, vector sigma, tau, tiny inputs).  That pins every line of control
//       flow and every expression order restated below to the reference's source.
//   (2) the reference's test invariants on its bundled fixtures (tests/test_oracle.py).
// STILL UNPINNED: the arithmetic INSIDE Armadillo's own loops (op_sum, accumulate, the short-vector norms, dense x sparse, sparse x sparse,
// kmeans) -- the stand-in restates them exactly as this file does ("Third-party arithmetic" and "LIBERTIES" below), so pin (1) cannot see
// a misreading of Armadillo shared by both.  What Armadillo takes from BLAS / LAPACK is no longer in that class: the one real BLAS on this
// machine (OpenBLAS 0.3.28 inside scipy, the release the reference's docs were built on) can be injected into both libraries for sgemm,
// sasum, snrm2, sgetrf + sgetri and spotrf + spotri (liberty bits 5 - 7, lapack_inv.hpp), the two still agree bit for bit, and the run
// moves by 2e-6 .. 8e-6 of Z_corr -- the width of every other liberty (profiles/r5_oracle_liberties.json).
//
// Third-party arithmetic whose source is NOT under /root/reference (RcppArmadillo,
// unpinned version; DESCRIPTION:54) is restated by its published semantics:
//   arma::normalise(X,p,0)  column / ||column||_p, zero norm -> divide by 1
//   arma::trunc_log         log(x) with x<=0 -> log(FLT_MIN), +inf -> log(FLT_MAX)
//   arma::shuffle / randu   replaced by an injectable permutation / a documented
//                           counter-based generator (R's RNG is not reproducible here)
//   arma::kmeans(...,keep_existing,1)  ONE Lloyd iteration: assign to the nearest
//                           mean (Euclidean), new mean = average of members, an
//                           empty cluster keeps its previous mean  [our definition]
//   arma::inv               fp32 LU with partial pivoting (faithful mode): UNBLOCKED, row by row.  Armadillo calls
//                           LAPACK sgetrf + sgetri, whose blocked update order depends on the BLAS the reference
//                           package was linked with: the same factorisation, not the same rounding sequence --
//                           this branch (several covariates, src/harmony.cpp:573) cannot be pinned bit for bit
//                           without that BLAS.  The GPU's reference-arithmetic mode matches THIS restatement.
//                           (Liberty bits 5 / 6 below run it through the one real LAPACK on this machine instead.)
//
// LIBERTIES of the faithful mode -- places where "the reference's operation order" is not determined by /root/reference itself but by the
// (unpinned) Armadillo / BLAS underneath it, what this restatement does there, and the switch that flips it (set_int("liberty", mask);
// tools/oracle_liberties.py measures how far each one moves a faithful run: "faithful" is an INTERVAL of that width, not a point, and
// the GPU's reference-arithmetic mode is tuned to the default point of it):
//   bit 0 (1)  L1 normalisations (src/harmony.cpp:145,150,323,326).  normalise(X, 1, 0) takes norm(col, 1); Armadillo's op_norm sums a
//              column of fewer than 32 elements with TWO accumulators (even / odd elements, added at the end) and hands longer ones to BLAS
//              sasum, whose order is the BLAS build's (SIMD partial sums).  Default here: one sequential accumulator.  Bit 0: the
//              two-accumulator form; bit 1 (2): eight strided partial sums combined pairwise (a SIMD sasum's shape).
//   bit 2 (4)  Z_corr -= W.t() * Phi_Rk for SEVERAL covariates (:615, dense x sparse): Armadillo's dense-times-sparse walks the sparse
//              operand's non-zeros and adds A.col(row) * value per non-zero, i.e. fl(W[row_1] r) + fl(W[row_2] r) + ...; default here:
//              (sum_c W[row_c]) * r -- one product.  Bit 2: one rounded product per non-zero, added in row order.  (One covariate: the same.)
//   bit 3 (8)  Lloyd iterations of kmeans_centers (src/utils.cpp:56-61): arma::kmeans' accumulation is internal to Armadillo (running
//              means, its own threading); default here: exact fp64 sums, an empty cluster keeps its mean.  Bit 3: fp32 sums.  (The
//              parity tests share the centres between the backends, so this liberty is outside every GPU comparison.)
//   bit 4 (16) L2 normalisations (:42,136,220,633): norm(col, 2) of >= 32 elements is BLAS snrm2 (OpenBLAS accumulates it in double on
//              x86-64; Armadillo's own loop for short columns uses two fp32 accumulators).  Default: one fp32 accumulator; bit 4: fp64 sum.
//   bit 5 (32) arma::inv of the several-covariate ridge system (:573) through a REAL LAPACK -- OpenBLAS 0.3.28's sgetrf + sgetri (the
//              release the reference's docs were built on; it sits inside scipy here and is injected from Python, oracle.use_lapack()) --
//              in Armadillo's auxlib::inv call sequence; bit 6 (64): spotrf + spotri + mirror (auxlib::inv_sympd, which Armadillo's inv()
//              tries first for a matrix that looks symmetric positive definite, as this one is).  Default: the unblocked LU below.
//              lapack_inv.hpp; the reference's own sources take the same two routes through the stand-in header (ref_set_inv_mode), and
//              tests/test_oracle_ref.py requires the two to agree bit for bit there as well.
//   bit 7 (128) every norm of a normalise() call exactly as Armadillo's op_norm forms it on a BLAS: two accumulators below 32 elements, the REAL
//              sasum / snrm2 of OpenBLAS 0.3.28 from 32 on (lapack_inv.hpp, namespace blas1; injected by oracle.use_lapack()); and the head's
//              column sums sum(R, 0) (:146,225 -- op_sum, not normalise) as arrayops::accumulate: two accumulators at every length.  Supersedes
//              the modelled shapes of bits 0, 1 and 4.  Bits 2 + 6 + 7 with the distance GEMM through the same library's sgemm
//              (oracle.use_openblas) is "what a current RcppArmadillo linked against OpenBLAS 0.3.28 runs": tools/oracle_liberties.py's last row.
//   not switchable, stated: abs() in check_convergence (:185,194) -- with <cmath> in scope and a float argument, overload resolution
//              takes std::abs(float) (exact match; ::abs(int) would need a conversion), so the quotient is a float expression, as here.
//
// Two arithmetic modes (hard part 3):
//   faithful  fp32 state and fp32 accumulators in the reference's operation order
//             (sequential fp32 my_accu, fp32 O/E += / -= drift, fp32 inverse)
//   accurate  same algorithm, but O/E, objective terms and the ridge sufficient
//             statistics / solve are carried in fp64.  This is the parity target
//             for the GPU path; the faithful-vs-accurate gap is the noise floor.
// =============================================================================
#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <numeric>
#include <set>
#include <string>
#include <type_traits>
#include <vector>

#include "lapack_inv.hpp"

namespace {

typedef void (*sgemm_fn)(int order, int transa, int transb, int M, int N, int K, float alpha,
                         const float* A, int lda, const float* B, int ldb, float beta, float* C,
                         int ldc);
sgemm_fn g_sgemm = nullptr;  // optional cblas_sgemm (OpenBLAS) injected from Python

// ---- documented counter-based generators (shared SPEC with the product; the
// ---- implementations are independent).  See include/harmony_mi355x.h. --------
inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
inline float u01(uint64_t seed, uint64_t stream, uint64_t idx) {
  uint64_t h = splitmix64(splitmix64(seed ^ (stream * 0xD1342543DE82EF95ull)) + idx);
  return ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
}
inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
// position of cell g in round `round`'s pseudo-random permutation of [0,N)
inline uint64_t feistel_pos(uint64_t seed, uint64_t round, uint64_t N, uint64_t g) {
  int bits = 2;
  while (((uint64_t)1 << bits) < N) bits += 2;
  const int half = bits / 2;
  const uint32_t mask = (uint32_t)(((uint64_t)1 << half) - 1);
  uint32_t keys[6];
  for (int r = 0; r < 6; r++)
    keys[r] = (uint32_t)(splitmix64(seed ^ (round * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(r + 1) << 56)) >> 32);
  uint64_t x = g;
  do {
    uint32_t L = (uint32_t)(x >> half), R = (uint32_t)(x & mask);
    for (int r = 0; r < 6; r++) {
      uint32_t t = L ^ (fmix32(R * 0x9E3779B1u + keys[r]) & mask);
      L = R; R = t;
    }
    x = ((uint64_t)L << half) | R;
  } while (x >= N);
  return x;
}

// ---- R-compatible stream (optional, set_int("rng", 1)): what the reference really draws from.  RcppArmadillo maps
// arma's generator to R's (randu -> Rf_runif(0,1), randi -> int(Rf_runif(0, RAND_MAX))); R's default generator is
// MT19937 seeded by set.seed (initial scrambling with the LCG 69069 x + 1; R: src/main/RNG.c).  Written independently of
// the product's harmony_amd/csrc/hmx_rrng.h; both are checked against R's documented set.seed(1); runif(3) values.
struct RStream {
  uint32_t mt[624]; int mti = 625;
  void set_seed(uint32_t seed) {
    for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;
    seed = 69069u * seed + 1u;                      // dummy[0] (overwritten with mti = 624 by FixupSeeds)
    for (int j = 0; j < 624; j++) { seed = 69069u * seed + 1u; mt[j] = seed; }
    mti = 624;
  }
  uint32_t next() {
    if (mti >= 624) {
      for (int kk = 0; kk < 624; kk++) {
        const uint32_t y = (mt[kk] & 0x80000000u) | (mt[(kk + 1) % 624] & 0x7fffffffu);
        mt[kk] = mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      mti = 0;
    }
    uint32_t y = mt[mti++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
  double unif_rand() {
    const double x = (double)next() * 2.3283064365386963e-10, h = 0.5 * 2.328306437080797e-10;
    return x <= 0.0 ? h : ((1.0 - x) <= 0.0 ? 1.0 - h : x);
  }
  double runif(double a, double b) { if (a == b) return a; double u; do { u = unif_rand(); } while (u <= 0 || u >= 1); return a + (b - a) * u; }
  float randu() { return (float)runif(0.0, 1.0); }
  int randi() { return (int)runif(0.0, (double)RAND_MAX); }
};

inline float trunc_logf(float x) {  // arma::trunc_log
  if (!(x > 0.0f)) return std::log(FLT_MIN);
  if (std::isinf(x)) return std::log(FLT_MAX);
  return std::log(x);
}
int my_ceil(float num) {  // src/utils.cpp:102-108
  int inum = (int)num;
  if (num == (float)inum) return inum;
  return inum + 1;
}

struct Timers {
  std::map<std::string, double> ms;
  struct Scope {
    double& acc; std::chrono::steady_clock::time_point t0;
    Scope(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~Scope() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
  };
};

// C[M x N] = A^T[M x Kd] * B[Kd x N], all column-major (A is Kd x M).
void gemm_tn(int M, int N, int Kd, const float* A, const float* B, float* C) {
  if (g_sgemm) {
    g_sgemm(102 /*ColMajor*/, 112 /*Trans*/, 111 /*NoTrans*/, M, N, Kd, 1.0f, A, Kd, B, Kd, 0.0f, C, M);
    return;
  }
  for (int64_t n = 0; n < N; n++)
    for (int m = 0; m < M; m++) {
      float s = 0.0f;
      const float* a = A + (int64_t)m * Kd; const float* b = B + n * Kd;
      for (int k = 0; k < Kd; k++) s += a[k] * b[k];
      C[n * M + m] = s;
    }
}

template <class T> void normalise_cols_l2(T* X, int rows, int64_t cols, int dacc = 0) {
  for (int64_t c = 0; c < cols; c++) {
    T* x = X + c * rows; T nrm;
    if (dacc == 2 && std::is_same<T, float>::value && blas1::ready()) nrm = (T)blas1::norm2((const float*)x, rows);   // (liberty bit 7: Armadillo's op_norm on a real BLAS)
    else if (dacc) { double sd = 0; for (int r = 0; r < rows; r++) sd += (double)x[r] * (double)x[r]; nrm = (T)std::sqrt(sd); }   // (liberty bit 4: snrm2 in double)
    else { T s = 0; for (int r = 0; r < rows; r++) s += x[r] * x[r]; nrm = std::sqrt(s); }
    if (nrm == 0) nrm = 1;
    for (int r = 0; r < rows; r++) x[r] /= nrm;
  }
}

// dense solve A X = Bm (n x n, n x m) in place, LU with partial pivoting; returns false if singular
template <class T> bool lu_solve(std::vector<T>& A, int n, std::vector<T>& Bm, int m) {
  std::vector<int> piv(n);
  for (int c = 0; c < n; c++) {
    int p = c; T best = std::fabs(A[c * n + c]);
    for (int r = c + 1; r < n; r++) if (std::fabs(A[c * n + r]) > best) { best = std::fabs(A[c * n + r]); p = r; }
    if (best == 0) return false;
    if (p != c) {
      for (int j = 0; j < n; j++) std::swap(A[j * n + c], A[j * n + p]);
      for (int j = 0; j < m; j++) std::swap(Bm[j * n + c], Bm[j * n + p]);
    }
    T inv = 1 / A[c * n + c];
    for (int r = c + 1; r < n; r++) {
      T f = A[c * n + r] * inv; if (f == 0) continue;
      A[c * n + r] = f;
      for (int j = c + 1; j < n; j++) A[j * n + r] -= f * A[j * n + c];
      for (int j = 0; j < m; j++) Bm[j * n + r] -= f * Bm[j * n + c];
    }
  }
  for (int j = 0; j < m; j++)
    for (int r = n - 1; r >= 0; r--) {
      T s = Bm[j * n + r];
      for (int c = r + 1; c < n; c++) s -= A[c * n + r] * Bm[j * n + c];
      Bm[j * n + r] = s / A[r * n + r];
    }
  return true;
}

// Arithmetic mask (one bit per group of accumulators; 0 = the reference's fp32, 1 = fp64):
//   bit 0 (1)  O / E tables and the row sums that feed them        src/harmony.cpp:149-150,312-313,329-330
//   bit 1 (2)  objective accumulators (my_accu)                     src/utils.cpp:67-75
//   bit 2 (4)  ridge statistics  Phi* diag(R_k) Phi*^T,  Phi* diag(R_k) Z^T   src/harmony.cpp:561-609
//   bit 3 (8)  ridge solve (inverse and the W products)             src/harmony.cpp:571-609
// MASK = 0 -> faithful mode;  MASK = 15 -> accurate mode;  the single-bit flips attribute the faithful-vs-accurate gap
// to one group at a time (tools/arith_gap.py).
struct OracleBase {
  virtual ~OracleBase() {}
  virtual int setup(const double* Z, int64_t N, int d, const int32_t* phi_i, const int32_t* phi_p, int B,
                    const double* sigma, const double* theta, const double* lambda, int n_lambda,
                    double alpha, int max_iter_kmeans, double eps_k, double eps_h, int K, double block_size,
                    const int32_t* B_vec, int C, double cutoff) = 0;
  virtual int init_cluster(const double* Y0, uint64_t seed) = 0;
  virtual int cluster() = 0;
  virtual int moe_correct_ridge() = 0;
  virtual int check_convergence(int type) = 0;
  virtual void compute_objective() = 0;
  virtual int64_t get(const char* what, double* out) = 0;
  virtual void push_update_order(const int64_t* order) = 0;
  virtual void set_int(const char* what, int64_t v) = 0;
  std::string err;
  Timers timers;
};

template <unsigned MASK> struct Oracle : OracleBase {
  using ACC = typename std::conditional<(MASK & 1u) != 0, double, float>::type;    // O / E tables
  using AOBJ = typename std::conditional<(MASK & 2u) != 0, double, float>::type;   // objective sums
  using AST = typename std::conditional<(MASK & 4u) != 0, double, float>::type;    // ridge statistics
  using ASV = typename std::conditional<(MASK & 8u) != 0, double, float>::type;    // ridge solve
  int64_t N = 0; int d = 0, K = 0, B = 0, C = 0;
  std::vector<float> Z_orig, Z_corr, R, dist, Y, W;  // column-major like the reference
  std::vector<ACC> O, E;                              // K x B, column-major: [b*K + k]
  std::vector<float> Pr_b, theta, sigma, lambda, sizes;
  std::vector<int> B_vec, covariate_bounds;
  std::vector<std::vector<int64_t>> index;            // cells of each level, ascending
  std::vector<int32_t> codes;                         // [c*N + i] global level of cell i in covariate c
  std::vector<float> objective_kmeans, objective_kmeans_dist, objective_kmeans_entropy,
      objective_kmeans_cross, objective_harmony;
  std::vector<int> kmeans_rounds;
  std::vector<int64_t> seed_cells;   // diagnostics: the cells initialize_centroids chose
  float block_size = 0.05f, epsilon_kmeans = 1e-3f, epsilon_harmony = 1e-2f, alpha = 0.2f,
        batch_proportion_cutoff = 1e-5f;
  int max_iter_kmeans = 4, window_size = 3;
  bool lambda_estimation = false;
  uint64_t seed = 0, round_counter = 0;
  int rng_mode = 0; RStream rs_; bool rs_seeded = false;   // 1: R-compatible stream (set.seed(seed) at the first draw)
  void ensure_rs() { if (!rs_seeded) { rs_.set_seed((uint32_t)seed); rs_seeded = true; } }
  std::deque<std::vector<int64_t>> injected;
  int W_rows = 0;
  int64_t subset_clusters = 0, skipped_clusters = 0;
  int liberty = 0;      // header, "LIBERTIES of the faithful mode"
  // the L1 norm of a column of R as normalise(X, 1, 0) forms it (all entries are >= 0: |r| = r)
  int l2_mode() const { return (liberty & 128) ? 2 : ((liberty & 16) ? 1 : 0); }
  // sum(R, 0) of the head (:146,225) is op_sum, not normalise: Armadillo's arrayops::accumulate = two accumulators at every length, no BLAS
  float head_sum(const float* r, int n) const {
    if (liberty & 128) { float a1 = 0.f, a2 = 0.f; int i = 0; for (; i + 1 < n; i += 2) { a1 += r[i]; a2 += r[i + 1]; } if (i < n) a1 += r[i]; return a1 + a2; }
    return l1_sum(r, n);
  }
  float l1_sum(const float* r, int n) const {
    if ((liberty & 128) && blas1::ready()) return blas1::norm1(r, n);
    if (liberty & 1) { float a1 = 0.f, a2 = 0.f; int i = 0; for (; i + 1 < n; i += 2) { a1 += std::fabs(r[i]); a2 += std::fabs(r[i + 1]); } if (i < n) a1 += std::fabs(r[i]); return a1 + a2; }
    if (liberty & 2) { float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}; for (int i = 0; i < n; i++) a[i & 7] += std::fabs(r[i]); return ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7])); }
    float s = 0.f; for (int i = 0; i < n; i++) s += std::fabs(r[i]); return s;
  }

  // ---- setup: src/harmony.cpp:29-128 ----------------------------------------
  int setup(const double* Z, int64_t N_, int d_, const int32_t* phi_i, const int32_t* phi_p, int B_,
            const double* sigma_, const double* theta_, const double* lambda_, int n_lambda, double alpha_,
            int max_iter_kmeans_, double eps_k, double eps_h, int K_, double block_size_, const int32_t* B_vec_,
            int C_, double cutoff) override {
    N = N_; d = d_; B = B_; K = K_; C = C_;
    if (N < 6) { err = "Refusing to run with less than 6 cells"; return 2; }  // :83-85
    Z_orig.resize((size_t)d * N);
    for (size_t i = 0; i < Z_orig.size(); i++) Z_orig[i] = (float)Z[i];          // conv_to :41
    Z_corr = Z_orig; normalise_cols_l2(Z_corr.data(), d, N, l2_mode());   // :42
    B_vec.assign(B_vec_, B_vec_ + C);
    covariate_bounds.resize(C); std::partial_sum(B_vec.begin(), B_vec.end(), covariate_bounds.begin());
    if (covariate_bounds.back() != B) { err = "sum(B_vec) != nrow(Phi)"; return 3; }
    // Phi is C-hot with unit values (R/ui.R:210-213); keep per-covariate level codes + index lists (:49-65)
    index.assign(B, {}); codes.assign((size_t)C * N, -1);
    for (int64_t i = 0; i < N; i++) {
      if (phi_p[i + 1] - phi_p[i] != C) { err = "Phi column is not C-hot"; return 3; }
      for (int c = 0; c < C; c++) {
        int b = phi_i[phi_p[i] + c];
        int cov = 0; while (b >= covariate_bounds[cov]) cov++;
        if (cov != c) { err = "Phi rows not grouped by covariate"; return 3; }
        codes[(size_t)c * N + i] = b; index[b].push_back(i);
      }
    }
    sizes.resize(B); Pr_b.resize(B);
    for (int b = 0; b < B; b++) { sizes[b] = (float)index[b].size(); Pr_b[b] = sizes[b] / (float)N; }  // :67
    epsilon_kmeans = (float)eps_k; epsilon_harmony = (float)eps_h;
    if (lambda_[0] == -1) lambda_estimation = true;                               // :75-79
    else { lambda_estimation = false; lambda.assign(n_lambda, 0.f); for (int i = 0; i < n_lambda; i++) lambda[i] = (float)lambda_[i]; }
    sigma.resize(K); for (int k = 0; k < K; k++) sigma[k] = (float)sigma_[k];
    block_size = (N < 40) ? 0.2f : (float)block_size_;                            // :86-91
    theta.resize(B); for (int b = 0; b < B; b++) theta[b] = (float)theta_[b];
    max_iter_kmeans = max_iter_kmeans_; alpha = (float)alpha_; batch_proportion_cutoff = (float)cutoff;
    // allocate_buffers :114-128
    dist.assign((size_t)K * N, 0.f); R.assign((size_t)K * N, 0.f);
    O.assign((size_t)K * B, 0); E.assign((size_t)K * B, 0);
    W.assign((size_t)(B + 1) * d, 0.f); W_rows = B + 1;
    objective_kmeans.clear(); objective_kmeans_dist.clear(); objective_kmeans_entropy.clear();
    objective_kmeans_cross.clear(); objective_harmony.clear(); kmeans_rounds.clear();
    round_counter = 0;
    return 0;
  }

  // ---- kmeans_centers: src/utils.cpp:10-64 -----------------------------------
  void kmeans_centers(uint64_t seed_) {
    const float* X = Z_corr.data();
    Y.assign((size_t)d * K, 0.f);
    // initialize_centroids :10-49
    const uint64_t Nm1 = (uint64_t)N - 1;
    if (rng_mode == 1) ensure_rs();
    for (int i = 0; i < K; i++) {
      int64_t idx = (int64_t)std::floor((rng_mode == 1 ? rs_.randu() : u01(seed_, 0, (uint64_t)i)) * (float)Nm1);
      std::memcpy(&Y[(size_t)i * d], X + idx * d, sizeof(float) * d);
    }
    std::set<int64_t> sup;
    seed_cells.clear();
    std::vector<float> prob(N);
    for (int i = 0; i < K; i++) {
      const float* y = &Y[(size_t)i * d];
      for (int64_t n = 0; n < N; n++) {
        float dot = 0.f; const float* x = X + n * d;
        for (int j = 0; j < d; j++) dot += y[j] * x[j];
        float dis = std::fabs(2.f * (1.f - dot));
        prob[n] = -std::log(rng_mode == 1 ? rs_.randu() : u01(seed_, 1 + (uint64_t)i, (uint64_t)n)) / dis;
      }
      int64_t best = std::min_element(prob.begin(), prob.end()) - prob.begin();
      while (sup.count(best)) {                                                    // :38-43
        prob[best] = *std::max_element(prob.begin(), prob.end());
        best = std::min_element(prob.begin(), prob.end()) - prob.begin();
      }
      sup.insert(best);
      seed_cells.push_back(best);
      std::memcpy(&Y[(size_t)i * d], X + best * d, sizeof(float) * d);
    }
    // 10 x one Lloyd iteration :53-64 (arma::kmeans semantics: see header)
    std::vector<double> sums((size_t)d * K); std::vector<int64_t> cnt(K); std::vector<float> ynorm(K);
    for (int it = 0; it < 10; it++) {
      std::fill(sums.begin(), sums.end(), 0.0); std::fill(cnt.begin(), cnt.end(), 0);
      for (int k = 0; k < K; k++) { float s = 0.f; for (int j = 0; j < d; j++) s += Y[(size_t)k * d + j] * Y[(size_t)k * d + j]; ynorm[k] = s; }
      for (int64_t n = 0; n < N; n++) {
        const float* x = X + n * d; int bk = 0; float bs = FLT_MAX;
        for (int k = 0; k < K; k++) {
          float dot = 0.f; const float* y = &Y[(size_t)k * d];
          for (int j = 0; j < d; j++) dot += y[j] * x[j];
          float sc = ynorm[k] - 2.f * dot;  // ||x-y||^2 - ||x||^2
          if (sc < bs) { bs = sc; bk = k; }
        }
        cnt[bk]++;
        if (liberty & 8) for (int j = 0; j < d; j++) sums[(size_t)bk * d + j] = (double)((float)sums[(size_t)bk * d + j] + x[j]);     // fp32 running sums
        else for (int j = 0; j < d; j++) sums[(size_t)bk * d + j] += x[j];
      }
      for (int k = 0; k < K; k++) if (cnt[k] > 0)
        for (int j = 0; j < d; j++) Y[(size_t)k * d + j] = (float)(sums[(size_t)k * d + j] / (double)cnt[k]);
    }
  }

  // R = softmax_k(-dist/sigma), E, O from scratch: src/harmony.cpp:141-150 and :221-227
  void dist_R_EO() {
    gemm_tn(K, (int)N, d, Y.data(), Z_corr.data(), dist.data());
    for (size_t i = 0; i < dist.size(); i++) dist[i] = 2.f * (1.f - dist[i]);
    for (int64_t n = 0; n < N; n++) {
      float* r = &R[n * K]; const float* dm = &dist[n * K];
      for (int k = 0; k < K; k++) r[k] = std::exp(-dm[k] / sigma[k]);
      float s = head_sum(r, K);      // R.each_row() /= sum(R, 0)  (:146,225)
      if (s == 0.f) s = 1.f;
      for (int k = 0; k < K; k++) r[k] /= s;
    }
    std::vector<ACC> rs(K, 0);
    for (int64_t n = 0; n < N; n++) for (int k = 0; k < K; k++) rs[k] += R[n * K + k];
    for (int b = 0; b < B; b++) for (int k = 0; k < K; k++) E[(size_t)b * K + k] = rs[k] * (ACC)Pr_b[b];
    std::fill(O.begin(), O.end(), (ACC)0);
    for (int b = 0; b < B; b++) for (int64_t i : index[b]) for (int k = 0; k < K; k++) O[(size_t)b * K + k] += R[i * K + k];
  }

  int init_cluster(const double* Y0, uint64_t seed_) override {  // :131-156
    seed = seed_; rs_seeded = false;
    if (Y0) { Y.resize((size_t)d * K); for (size_t i = 0; i < Y.size(); i++) Y[i] = (float)Y0[i]; }
    else kmeans_centers(seed_);
    normalise_cols_l2(Y.data(), d, K, l2_mode());
    dist_R_EO();
    compute_objective();
    objective_harmony.push_back(objective_kmeans.back());
    return 0;
  }

  void compute_objective() override {  // :158-170
    const float norm_const = 2000 / ((float)N);
    AOBJ kmeans_error = 0, entropy = 0, cross = 0;
    std::vector<float> M((size_t)K * B);  // theta_b * log((O+E+1)/(2E+1))
    for (int b = 0; b < B; b++) for (int k = 0; k < K; k++) {
      float o = (float)O[(size_t)b * K + k], e = (float)E[(size_t)b * K + k];
      M[(size_t)b * K + k] = theta[b] * std::log((o + e + 1.f) / ((2.f * e) + 1.f));
    }
    for (int64_t n = 0; n < N; n++) for (int k = 0; k < K; k++) kmeans_error += R[n * K + k] * dist[n * K + k];
    for (int64_t n = 0; n < N; n++) for (int k = 0; k < K; k++) { float r = R[n * K + k]; entropy += (r * trunc_logf(r)) * sigma[k]; }
    for (int64_t n = 0; n < N; n++) for (int k = 0; k < K; k++) {
      float m = 0.f; for (int c = 0; c < C; c++) m += M[(size_t)codes[(size_t)c * N + n] * K + k];
      cross += (R[n * K + k] * sigma[k]) * m;
    }
    objective_kmeans.push_back((float)((kmeans_error + entropy + cross) * norm_const));
    objective_kmeans_dist.push_back((float)(kmeans_error * norm_const));
    objective_kmeans_entropy.push_back((float)(entropy * norm_const));
    objective_kmeans_cross.push_back((float)(cross * norm_const));
  }

  int check_convergence(int type) override {  // :173-205
    float obj_new, obj_old;
    if (type == 0) {
      obj_old = 0; obj_new = 0;
      for (int i = 0; i < window_size; i++) {
        obj_old += objective_kmeans[objective_kmeans.size() - 2 - i];
        obj_new += objective_kmeans[objective_kmeans.size() - 1 - i];
      }
      return (std::fabs(obj_old - obj_new) / std::fabs(obj_old) < epsilon_kmeans) ? 1 : 0;
    } else if (type == 1) {
      obj_old = objective_harmony[objective_harmony.size() - 2];
      obj_new = objective_harmony[objective_harmony.size() - 1];
      return ((obj_old - obj_new) / std::fabs(obj_old) < epsilon_harmony) ? 1 : 0;
    }
    return 1;
  }

  int cluster() override {  // :208-262
    if (objective_harmony.size() != 1) {
      normalise_cols_l2(Z_corr.data(), d, N, l2_mode());
      dist_R_EO();
    }
    int iter;
    for (iter = 0; iter < max_iter_kmeans; iter++) {
      int st = update_R(); if (st) return st;
      compute_objective();
      if (iter > window_size) { if (check_convergence(0)) { iter++; break; } }
    }
    kmeans_rounds.push_back(iter);
    objective_harmony.push_back(objective_kmeans.back());
    return 0;
  }

  void push_update_order(const int64_t* order) override { injected.emplace_back(order, order + N); }

  int update_R() {  // :269-342
    std::vector<int64_t> update_order;
    if (!injected.empty()) { update_order = std::move(injected.front()); injected.pop_front(); }
    else if (rng_mode == 1) {  // arma::shuffle on R's stream: one randi per element in order, std::sort of (value, index) packets
      ensure_rs();
      struct Pk { int val; int64_t index; };
      std::vector<Pk> pk((size_t)N);
      for (int64_t i = 0; i < N; i++) { pk[i].val = rs_.randi(); pk[i].index = i; }
      std::sort(pk.begin(), pk.end(), [](const Pk& a, const Pk& b) { return a.val < b.val; });
      update_order.resize(N);
      for (int64_t i = 0; i < N; i++) update_order[i] = pk[i].index;
    } else {  // documented generator: cell g sits at position feistel_pos(seed, round, N, g)
      update_order.resize(N);
      for (int64_t g = 0; g < N; g++) update_order[feistel_pos(seed, round_counter, (uint64_t)N, (uint64_t)g)] = g;
    }
    round_counter++;
    const unsigned n_blocks = (unsigned)my_ceil(1.0 / block_size);
    const unsigned cells_per_block = (unsigned)(N * block_size);  // fp32 product, truncated :281
    // physical shuffle :285-291 (the reference gathers R and dist_mat by update_order)
    std::vector<float> Rr((size_t)K * N), Dr((size_t)K * N);
    {
      Timers::Scope t(timers.ms["randomize"]);
      for (int64_t p = 0; p < N; p++) {
        std::memcpy(&Rr[p * K], &R[update_order[p] * K], sizeof(float) * K);
        std::memcpy(&Dr[p * K], &dist[update_order[p] * K], sizeof(float) * K);
      }
    }
    std::vector<float> pen((size_t)K * B); std::vector<ACC> rs(K), tmpO((size_t)K * B);
    for (unsigned blk = 0; blk < n_blocks; blk++) {
      int64_t idx_min = (int64_t)blk * cells_per_block;
      int64_t idx_max = (int64_t)(blk + 1) * cells_per_block - 1;
      if (blk == n_blocks - 1) idx_max = N - 1;
      if (idx_min > idx_max) continue;  // N*block_size rounding can leave trailing empty blocks
      {
        Timers::Scope t(timers.ms["EO_update"]);  // :312-313
        std::fill(rs.begin(), rs.end(), (ACC)0);
        for (int64_t p = idx_min; p <= idx_max; p++) for (int k = 0; k < K; k++) rs[k] += Rr[p * K + k];
        for (int b = 0; b < B; b++) for (int k = 0; k < K; k++) E[(size_t)b * K + k] -= rs[k] * (ACC)Pr_b[b];
        // O -= Rcells * Phi_tcells: the dense x sparse product is formed FIRST (per (k, b) a sequential sum over the block's cells of
        // level b in ascending position -- Armadillo walks the sparse operand's non-zeros column by column, rows ascending), then ONE
        // subtraction per table entry (:313).
        std::fill(tmpO.begin(), tmpO.end(), (ACC)0);
        for (int64_t p = idx_min; p <= idx_max; p++) { int64_t i = update_order[p];
          for (int c = 0; c < C; c++) { size_t b = codes[(size_t)c * N + i]; for (int k = 0; k < K; k++) tmpO[b * K + k] += Rr[p * K + k]; } }
        for (size_t e = 0; e < tmpO.size(); e++) O[e] -= tmpO[e];
      }
      {
        Timers::Scope t(timers.ms["Rcells_update"]);  // :318-323
        for (int b = 0; b < B; b++) for (int k = 0; k < K; k++) {
          float o = (float)O[(size_t)b * K + k], e = (float)E[(size_t)b * K + k];
          pen[(size_t)b * K + k] = std::pow(((2.f * e) + 1.f) / (o + e + 1.f), theta[b]);
        }
        for (int64_t p = idx_min; p <= idx_max; p++) {
          int64_t i = update_order[p]; float* r = &Rr[p * K]; const float* dm = &Dr[p * K];
          for (int k = 0; k < K; k++) r[k] = std::exp(-dm[k] / sigma[k]);
          float s = l1_sum(r, K);
          if (s == 0.f) s = 1.f;
          for (int k = 0; k < K; k++) r[k] /= s;
          for (int k = 0; k < K; k++) {
            float m = 0.f; for (int c = 0; c < C; c++) m += pen[(size_t)codes[(size_t)c * N + i] * K + k];
            r[k] *= m;
          }
          s = l1_sum(r, K);
          if (s == 0.f) s = 1.f;
          for (int k = 0; k < K; k++) r[k] /= s;
        }
      }
      {
        Timers::Scope t(timers.ms["EO_update"]);  // :329-330
        std::fill(rs.begin(), rs.end(), (ACC)0);
        for (int64_t p = idx_min; p <= idx_max; p++) for (int k = 0; k < K; k++) rs[k] += Rr[p * K + k];
        for (int b = 0; b < B; b++) for (int k = 0; k < K; k++) E[(size_t)b * K + k] += rs[k] * (ACC)Pr_b[b];
        std::fill(tmpO.begin(), tmpO.end(), (ACC)0);   // product first, then ONE addition per entry (:330)
        for (int64_t p = idx_min; p <= idx_max; p++) { int64_t i = update_order[p];
          for (int c = 0; c < C; c++) { size_t b = codes[(size_t)c * N + i]; for (int k = 0; k < K; k++) tmpO[b * K + k] += Rr[p * K + k]; } }
        for (size_t e = 0; e < tmpO.size(); e++) O[e] += tmpO[e];
      }
    }
    {
      Timers::Scope t(timers.ms["randomize"]);  // un-shuffle :338-339
      for (int64_t p = 0; p < N; p++) std::memcpy(&R[update_order[p] * K], &Rr[p * K], sizeof(float) * K);
    }
    return 0;
  }

  // ---- moe_correct_ridge_cpp: src/harmony.cpp:345-638 -------------------------
  int moe_correct_ridge() override {
    Timers::Scope tall(timers.ms["correct_ridge_loop"]);
    Z_corr = Z_orig;  // :347
    subset_clusters = skipped_clusters = 0;
    std::vector<char> in_set(N);
    for (int k = 0; k < K; k++) {
      // which levels have support in this cluster :358-402
      std::vector<int> cov_levels(C, 0);
      for (int b = 0, cov = 0; b < B; b++) {
        if (!(b < covariate_bounds[cov])) cov++;
        float rep = (float)O[(size_t)b * K + k] / sizes[b];
        if (rep > batch_proportion_cutoff) cov_levels[cov]++;
      }
      std::vector<int> keep;
      for (int b = 0, cov = 0; b < B; b++) {
        if (cov < C && !(b < covariate_bounds[cov])) cov++;
        float rep = (float)O[(size_t)b * K + k] / sizes[b];
        if (rep > batch_proportion_cutoff && cov_levels[cov] > 1) keep.push_back(b);
      }
      int active = 0; for (int l : cov_levels) if (l > 1) active++;
      const bool full = ((int)keep.size() == B);
      if (!full) { subset_clusters++; if (active == 0) { skipped_clusters++; continue; } }  // :449-452
      const int m = (int)keep.size() + 1;  // design rows incl. intercept
      std::vector<int> row_of(B, -1); for (int a = 0; a < (int)keep.size(); a++) row_of[keep[a]] = a + 1;
      // cells entering the regression = union of kept levels' cells :400,456-460
      if (full) std::fill(in_set.begin(), in_set.end(), 1);
      else { std::fill(in_set.begin(), in_set.end(), 0); for (int b : keep) for (int64_t i : index[b]) in_set[i] = 1; }
      // cov = Phi* diag(R_k) Phi*^T + Lambda ; rhs = Phi* diag(R_k) Z_orig^T   :561-568,592-609
      std::vector<AST> cov((size_t)m * m, 0), rhs((size_t)m * d, 0);
      for (int64_t i = 0; i < N; i++) {
        if (!in_set[i]) continue;
        const AST r = R[i * K + k];
        int rows[16]; int nr = 0; rows[nr++] = 0;
        for (int c = 0; c < C; c++) { int ro = row_of[codes[(size_t)c * N + i]]; if (ro >= 0) rows[nr++] = ro; }
        for (int a = 0; a < nr; a++) for (int b2 = 0; b2 < nr; b2++) cov[(size_t)rows[b2] * m + rows[a]] += r;
        const float* z = &Z_orig[i * d];
        for (int j = 0; j < d; j++) { AST zt = (AST)(z[j] * (float)r);  // Z_tmp = Z_orig % R_k (fp32 product :592)
          for (int a = 0; a < nr; a++) rhs[(size_t)j * m + rows[a]] += zt; }
      }
      for (int a = 1; a < m; a++) {
        float lam = lambda_estimation ? (float)E[(size_t)keep[a - 1] * K + k] * alpha : lambda[keep[a - 1] + 1];  // :434-439,533-544
        cov[(size_t)a * m + a] += (AST)lam;
      }
      // W = inv(cov) * rhs  (:571-609).  Faithful: explicit fp32 inverse (arrowhead closed form for C==1,
      // LU otherwise) then fp32 products; accurate: fp64 LU solve.
      std::vector<ASV> Wk((size_t)m * d);
      bool ok = true;
      if (std::is_same<ASV, float>::value && C == 1) {
        std::vector<float> ac(m), bb(m), acb(m);
        for (int a = 0; a < m; a++) { ac[a] = -(float)cov[(size_t)a * m + 0]; bb[a] = 1.f / (float)cov[(size_t)a * m + a]; }
        ac[0] = 1.f; float b0 = (float)cov[0]; bb[0] = 0.f;
        // arma::accu(square(ac) % b) (:581): Armadillo's linear accumulate runs TWO accumulators over the even / odd elements and adds them at the end
        float u1 = 0.f, u2 = 0.f; int a2 = 0;
        for (; a2 + 1 < m; a2 += 2) { u1 += (ac[a2] * ac[a2]) * bb[a2]; u2 += (ac[a2 + 1] * ac[a2 + 1]) * bb[a2 + 1]; }
        if (a2 < m) u1 += (ac[a2] * ac[a2]) * bb[a2];
        float u = u1 + u2;
        u = b0 - u;
        for (int a = 0; a < m; a++) acb[a] = ac[a] * bb[a];
        acb[0] = 1.f;
        std::vector<float> inv((size_t)m * m);
        for (int c2 = 0; c2 < m; c2++) for (int r2 = 0; r2 < m; r2++) inv[(size_t)c2 * m + r2] = (1.f / u) * (acb[r2] * acb[c2]);
        for (int a = 0; a < m; a++) inv[(size_t)a * m + a] += bb[a];
        for (int j = 0; j < d; j++) for (int r2 = 0; r2 < m; r2++) {
          float s = 0.f; for (int c2 = 0; c2 < m; c2++) s += inv[(size_t)c2 * m + r2] * (float)rhs[(size_t)j * m + c2];
          Wk[(size_t)j * m + r2] = s; }
      } else if (std::is_same<ASV, float>::value) {
        std::vector<float> A(cov.begin(), cov.end()), I((size_t)m * m, 0); for (int a = 0; a < m; a++) I[(size_t)a * m + a] = 1;
        if ((liberty & (32 | 64)) && lapack_inv::ready()) {   // arma::inv through a real LAPACK (lapack_inv.hpp): bit 5 sgetrf + sgetri, bit 6 spotrf + spotri
          I = A; ok = lapack_inv::inv(I.data(), m, (liberty & 64) ? 2 : 1);
        } else
        ok = lu_solve(A, m, I, m);
        for (int j = 0; j < d; j++) for (int r2 = 0; r2 < m; r2++) {
          float s = 0; for (int c2 = 0; c2 < m; c2++) s += I[(size_t)c2 * m + r2] * (float)rhs[(size_t)j * m + c2];
          Wk[(size_t)j * m + r2] = s; }
      } else {
        std::vector<double> A(cov.begin(), cov.end()), X(rhs.begin(), rhs.end()); ok = lu_solve(A, m, X, d);
        for (size_t i = 0; i < Wk.size(); i++) Wk[i] = (ASV)X[i];
      }
      if (!ok) { err = "singular ridge system"; return 4; }
      for (int j = 0; j < d; j++) { Y[(size_t)k * d + j] = (float)Wk[(size_t)j * m]; Wk[(size_t)j * m] = 0; }  // :610-611
      // Z_corr -= W^T Phi* diag(R_k)  :615
      for (int64_t i = 0; i < N; i++) {
        if (!in_set[i]) continue;
        const float r = R[i * K + k]; float* z = &Z_corr[i * d];
        if (liberty & 4) {      // one rounded product per non-zero of Phi_Rk's column, added in row order (Armadillo's dense x sparse), then ONE subtraction
          for (int j = 0; j < d; j++) { float o = 0.f;
            for (int c = 0; c < C; c++) { int ro = row_of[codes[(size_t)c * N + i]]; if (ro >= 0) o += (float)Wk[(size_t)j * m + ro] * r; }
            z[j] -= o; }
          continue;
        }
        for (int j = 0; j < d; j++) { float w = 0.f;
          for (int c = 0; c < C; c++) { int ro = row_of[codes[(size_t)c * N + i]]; if (ro >= 0) w += (float)Wk[(size_t)j * m + ro]; }
          z[j] -= w * r; }
      }
      W.assign((size_t)m * d, 0.f); W_rows = m;
      for (size_t i = 0; i < W.size(); i++) W[i] = (float)Wk[i];
    }
    normalise_cols_l2(Y.data(), d, K, l2_mode());  // :633
    return 0;
  }

  void set_int(const char* what, int64_t v) override {
    std::string w(what);
    if (w == "max_iter_kmeans") max_iter_kmeans = (int)v;
    else if (w == "seed") { seed = (uint64_t)v; rs_seeded = false; }
    else if (w == "rng") { rng_mode = (int)v; rs_seeded = false; }
    else if (w == "liberty") liberty = (int)v;
  }

  template <class T> static int64_t copy_out(const std::vector<T>& v, double* out) {
    if (out) for (size_t i = 0; i < v.size(); i++) out[i] = (double)v[i];
    return (int64_t)v.size();
  }
  int64_t get(const char* what, double* out) override {
    std::string w(what);
    if (w == "Z_corr") return copy_out(Z_corr, out);
    if (w == "Z_orig") return copy_out(Z_orig, out);
    if (w == "R") return copy_out(R, out);
    if (w == "dist") return copy_out(dist, out);
    if (w == "Y") return copy_out(Y, out);
    if (w == "O") return copy_out(O, out);
    if (w == "E") return copy_out(E, out);
    if (w == "W") return copy_out(W, out);
    if (w == "W_rows") { if (out) out[0] = W_rows; return 1; }
    if (w == "Pr_b") return copy_out(Pr_b, out);
    if (w == "theta") return copy_out(theta, out);
    if (w == "sigma") return copy_out(sigma, out);
    if (w == "lambda") return copy_out(lambda, out);
    if (w == "objective_kmeans") return copy_out(objective_kmeans, out);
    if (w == "objective_kmeans_dist") return copy_out(objective_kmeans_dist, out);
    if (w == "objective_kmeans_entropy") return copy_out(objective_kmeans_entropy, out);
    if (w == "objective_kmeans_cross") return copy_out(objective_kmeans_cross, out);
    if (w == "objective_harmony") return copy_out(objective_harmony, out);
    if (w == "kmeans_rounds") return copy_out(kmeans_rounds, out);
    if (w == "seed_cells") return copy_out(seed_cells, out);
    if (w == "subset_clusters") { if (out) out[0] = (double)subset_clusters; return 1; }
    if (w == "skipped_clusters") { if (out) out[0] = (double)skipped_clusters; return 1; }
    if (w == "Lambda") {  // getLambda :657-669  (K x (B+1), column-major)
      if (out) for (int k = 0; k < K; k++) { out[k] = lambda_estimation ? 0.0 : (double)lambda[0];
        for (int b = 0; b < B; b++) out[(size_t)(b + 1) * K + k] = lambda_estimation ? (double)((float)E[(size_t)b * K + k] * alpha) : (double)lambda[b + 1]; }
      return (int64_t)K * (B + 1);
    }
    if (w.rfind("timer:", 0) == 0) { if (out) out[0] = timers.ms[w.substr(6)]; return 1; }
    return -1;
  }
};

}  // namespace

extern "C" {
void* orc_create_mask(unsigned mask) {
  switch (mask & 15u) {
#define ORC_CASE(M) case M: return (OracleBase*)new Oracle<M>();
    ORC_CASE(0) ORC_CASE(1) ORC_CASE(2) ORC_CASE(3) ORC_CASE(4) ORC_CASE(5) ORC_CASE(6) ORC_CASE(7)
    ORC_CASE(8) ORC_CASE(9) ORC_CASE(10) ORC_CASE(11) ORC_CASE(12) ORC_CASE(13) ORC_CASE(14) ORC_CASE(15)
#undef ORC_CASE
  }
  return nullptr;
}
void* orc_create(int accurate) { return orc_create_mask(accurate ? 15u : 0u); }
void orc_destroy(void* h) { delete (OracleBase*)h; }
void orc_set_sgemm(void* fn) { g_sgemm = (sgemm_fn)fn; }
void orc_set_blas1(void* asum, void* nrm2) { blas1::table().asum = (blas1::asum_fn)asum; blas1::table().nrm2 = (blas1::nrm2_fn)nrm2; }   // (liberty bit 7)
void orc_set_lapack(void* getrf, void* getri, void* potrf, void* potri) {     // (liberty bits 5 / 6: lapack_inv.hpp)
  lapack_inv::Table& t = lapack_inv::table();
  t.getrf = (lapack_inv::getrf_fn)getrf; t.getri = (lapack_inv::getri_fn)getri; t.potrf = (lapack_inv::potrf_fn)potrf; t.potri = (lapack_inv::potri_fn)potri;
}
int orc_setup(void* h, const double* Z, int64_t N, int d, const int32_t* phi_i, const int32_t* phi_p, int B,
              const double* sigma, const double* theta, const double* lambda, int n_lambda, double alpha,
              int max_iter_kmeans, double eps_k, double eps_h, int K, double block_size, const int32_t* B_vec,
              int C, double cutoff) {
  return ((OracleBase*)h)->setup(Z, N, d, phi_i, phi_p, B, sigma, theta, lambda, n_lambda, alpha, max_iter_kmeans,
                                 eps_k, eps_h, K, block_size, B_vec, C, cutoff);
}
int orc_init_cluster(void* h, const double* Y0, uint64_t seed) { return ((OracleBase*)h)->init_cluster(Y0, seed); }
int orc_cluster(void* h) { return ((OracleBase*)h)->cluster(); }
int orc_moe_correct_ridge(void* h) { return ((OracleBase*)h)->moe_correct_ridge(); }
int orc_check_convergence(void* h, int type) { return ((OracleBase*)h)->check_convergence(type); }
void orc_compute_objective(void* h) { ((OracleBase*)h)->compute_objective(); }
int64_t orc_get(void* h, const char* what, double* out) { return ((OracleBase*)h)->get(what, out); }
void orc_push_update_order(void* h, const int64_t* order) { ((OracleBase*)h)->push_update_order(order); }
void orc_set_int(void* h, const char* what, int64_t v) { ((OracleBase*)h)->set_int(what, v); }
const char* orc_last_error(void* h) { return ((OracleBase*)h)->err.c_str(); }
// generator spec probes (used by tests to check the product's implementation of the same spec)
uint64_t orc_feistel_pos(uint64_t seed, uint64_t round, uint64_t N, uint64_t g) { return feistel_pos(seed, round, N, g); }
float orc_u01(uint64_t seed, uint64_t stream, uint64_t idx) { return u01(seed, stream, idx); }
// R-compatible stream probes: n uniforms after set.seed(seed); the first n entries of arma::shuffle(0..N-1)
void orc_r_runif(uint32_t seed, int n, double* out) { RStream r; r.set_seed(seed); for (int i = 0; i < n; i++) out[i] = r.unif_rand(); }
void orc_r_shuffle(uint32_t seed, int64_t N, int64_t* out) {
  RStream r; r.set_seed(seed);
  struct Pk { int val; int64_t index; };
  std::vector<Pk> pk((size_t)N);
  for (int64_t i = 0; i < N; i++) { pk[i].val = r.randi(); pk[i].index = i; }
  std::sort(pk.begin(), pk.end(), [](const Pk& a, const Pk& b) { return a.val < b.val; });
  for (int64_t i = 0; i < N; i++) out[i] = pk[i].index;
}
}
