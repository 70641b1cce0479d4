// =============================================================================
// TEST INFRASTRUCTURE ONLY -- arma::inv through a REAL LAPACK, for the oracle (harmony_oracle.cpp, liberty bits 5 / 6) and for the
// stand-in Armadillo header the reference's own sources are compiled over (shim/arma_min.hpp, ref_set_inv_mode).
//
// The several-covariate ridge solve of the reference is `inv_cov = arma::inv(Phi_cov)` (src/harmony.cpp:573).  Armadillo hands that to
// LAPACK; which rounding sequence results depends on the LAPACK / BLAS the package was linked with.  The image has no R and no
// Armadillo, but it does have a LAPACK: OpenBLAS 0.3.28 inside scipy (scipy.libs/libscipy_openblas*.so, symbols scipy_sgetrf_, ... --
// the reference's docs were built on that very OpenBLAS release).  Python injects its entry points (oracle.use_lapack(),
// ref.use_lapack()); nothing links against it.
//
// Call sequences restated from Armadillo's published auxlib (not on disk: written down from knowledge of the published source, so the details
// that cannot be checked here are named -- the workspace size only matters above ~64 rows, where it decides between sgetri's blocked and
// unblocked forms; the ridge systems of this path have B + 1 = 4 .. 201 rows, the subset branch's a few dozen):
//   mode 1  auxlib::inv        sgetrf(n, n, A) ; sgetri(n, A, ipiv, work, lwork) with lwork = max(16, n), raised to the workspace query's
//                              proposal when n > 16
//   mode 2  auxlib::inv_sympd  what inv() takes first when the matrix "looks" symmetric positive definite (sym_helper::guess_sympd;
//                              Phi* diag(R_k) Phi*^T + Lambda is): spotrf('L') ; spotri('L') ; upper triangle = mirror of the lower one.
//                              A matrix spotrf rejects falls through to mode 1, as in Armadillo.
// Which of the two a given RcppArmadillo release takes for this matrix is a property of that release (the sympd shortcut appeared in the
// 9.x series): both are offered, and tools/oracle_liberties.py measures how far each moves a faithful run from the default restatement
// (unblocked LU, harmony_oracle.cpp lu_solve).
// =============================================================================
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

namespace lapack_inv {
typedef void (*getrf_fn)(const int* m, const int* n, float* a, const int* lda, int* ipiv, int* info);
typedef void (*getri_fn)(const int* n, float* a, const int* lda, const int* ipiv, float* work, const int* lwork, int* info);
typedef void (*potrf_fn)(const char* uplo, const int* n, float* a, const int* lda, int* info, size_t uplo_len);
typedef void (*potri_fn)(const char* uplo, const int* n, float* a, const int* lda, int* info, size_t uplo_len);
struct Table { getrf_fn getrf = nullptr; getri_fn getri = nullptr; potrf_fn potrf = nullptr; potri_fn potri = nullptr; };
inline Table& table() { static Table t; return t; }
inline bool ready() { const Table& t = table(); return t.getrf && t.getri && t.potrf && t.potri; }

// A: n x n, column-major, inverted in place.  Returns false if LAPACK reports a singular / non-factorisable matrix.
inline bool general(float* A, int n) {
  const Table& t = table();
  int info = 0, lwork = std::max(16, n);
  std::vector<int> ipiv((size_t)n);
  t.getrf(&n, &n, A, &n, ipiv.data(), &info);
  if (info != 0) return false;
  if (n > 16) {
    float query[2] = {0.f, 0.f}; int minus1 = -1;
    t.getri(&n, A, &n, ipiv.data(), query, &minus1, &info);
    if (info != 0) return false;
    lwork = std::max((int)query[0], lwork);
  }
  std::vector<float> work((size_t)lwork);
  t.getri(&n, A, &n, ipiv.data(), work.data(), &lwork, &info);
  return info == 0;
}
inline bool sympd(float* A, int n) {
  const Table& t = table();
  std::vector<float> keep(A, A + (size_t)n * n);
  int info = 0; const char L = 'L';
  t.potrf(&L, &n, A, &n, &info, 1);
  if (info == 0) t.potri(&L, &n, A, &n, &info, 1);
  if (info != 0) { std::copy(keep.begin(), keep.end(), A); return general(A, n); }       // not sympd after all: the general route
  for (int c = 0; c < n; c++) for (int r = c + 1; r < n; r++) A[(size_t)r * n + c] = A[(size_t)c * n + r];   // symmatl: upper(c, r) = lower(r, c)
  return true;
}
inline bool inv(float* A, int n, int mode) { return mode == 2 ? sympd(A, n) : general(A, n); }
}  // namespace lapack_inv

// norm(col, 1) / norm(col, 2) as Armadillo's op_norm forms them for a contiguous fp32 vector when it is built on a BLAS (restated from the
// published source): fewer than 32 elements -> its own loop with TWO accumulators (elements 0, 2, 4, ... and 1, 3, 5, ..., added at the end);
// 32 and more -> BLAS sasum / snrm2.  The BLAS entry points (cblas form, OpenBLAS 0.3.28 inside scipy) are injected like the LAPACK ones;
// oracle liberty bit 7, ref_set_norm_mode.  normalise(X, p, 0) divides a column by this value (by 1 when it is 0), src/harmony.cpp:42,136,
// 145,150,220,323,326,633.
namespace blas1 {
typedef float (*asum_fn)(int n, const float* x, int incx);
typedef float (*nrm2_fn)(int n, const float* x, int incx);
struct Table { asum_fn asum = nullptr; nrm2_fn nrm2 = nullptr; };
inline Table& table() { static Table t; return t; }
inline bool ready() { return table().asum && table().nrm2; }
inline float norm1(const float* p, long long n) {
  if (n >= 32) return table().asum((int)n, p, 1);
  float a1 = 0.f, a2 = 0.f; long long i = 0;
  for (; i + 1 < n; i += 2) { a1 += std::fabs(p[i]); a2 += std::fabs(p[i + 1]); }
  if (i < n) a1 += std::fabs(p[i]);
  return a1 + a2;
}
inline float norm2(const float* p, long long n) {
  if (n >= 32) return table().nrm2((int)n, p, 1);
  float a1 = 0.f, a2 = 0.f; long long i = 0;
  for (; i + 1 < n; i += 2) { a1 += p[i] * p[i]; a2 += p[i + 1] * p[i + 1]; }
  if (i < n) a1 += p[i] * p[i];
  return std::sqrt(a1 + a2);          // (Armadillo retries with a scaled loop only when this is 0 or not finite: never on this path's unit-scale data)
}
}  // namespace blas1
