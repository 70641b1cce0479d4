// =============================================================================
// TEST INFRASTRUCTURE ONLY -- C entry points around the REFERENCE'S OWN `harmony` class, compiled from the reference's sources where
// they lie (/root/reference/src/harmony.cpp, utils.cpp, timer.cpp -- not copied, not modified) against oracle/shim/ (a minimal stand-in
// for the Armadillo / Rcpp / RcppProgress headers the reference needs and this image lacks; see shim/arma_min.hpp for exactly what that
// stand-in restates).  Built only where /root/reference exists, by `make -C oracle _ref`, into oracle/_ref/libharmony_ref.so.
//
// Purpose: tests/test_oracle_ref.py runs this and the restated oracle (oracle/harmony_oracle.cpp, faithful mode) on the same inputs and
// the same random stream and requires bit-identical state after every call.  That pins the oracle's restatement of the reference's
// control flow and expression order (every line of harmony.cpp / utils.cpp) to the reference's source; what it cannot pin is the
// arithmetic inside Armadillo's own kernels, which both sides restate the same way (the oracle's header, "LIBERTIES").
//
// The entry points mirror oracle/harmony_oracle.cpp's orc_* (same argument lists) so that one Python wrapper shape drives both.
// =============================================================================
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "harmony.h"   // the reference's (found through -I/root/reference/src)

namespace {
struct Handle {
  harmony h;
  std::string err;
  int n_covariates = 0;
};
template <class V> int64_t copy_out(const V* p, size_t n, double* out) {
  if (out) for (size_t i = 0; i < n; i++) out[i] = (double)p[i];
  return (int64_t)n;
}
template <class V> int64_t copy_vec(const std::vector<V>& v, double* out) { return copy_out(v.data(), v.size(), out); }
template <class F> int guarded(Handle* H, F f) {
  try { return f(); }
  catch (const std::exception& e) { H->err = e.what(); return 100; }
  catch (...) { H->err = "unknown exception"; return 101; }
}
}  // namespace

extern "C" {
// arma::inv through a real LAPACK instead of the stand-in's unblocked LU (oracle/lapack_inv.hpp; tests/test_oracle_ref.py)
void ref_set_lapack(void* getrf, void* getri, void* potrf, void* potri) {
  lapack_inv::Table& t = lapack_inv::table();
  t.getrf = (lapack_inv::getrf_fn)getrf; t.getri = (lapack_inv::getri_fn)getri; t.potrf = (lapack_inv::potrf_fn)potrf; t.potri = (lapack_inv::potri_fn)potri;
}
void ref_set_inv_mode(int mode) { arma::shim::inv_mode() = mode; }
void ref_set_blas1(void* asum, void* nrm2) { blas1::table().asum = (blas1::asum_fn)asum; blas1::table().nrm2 = (blas1::nrm2_fn)nrm2; }
void ref_set_norm_mode(int mode) { arma::shim::norm_mode() = mode; }
void ref_set_sgemm(void* fn) { arma::shim::sgemm_ptr() = (arma::shim::sgemm_fn)fn; }     // cblas_sgemm for Y.t() * Z_corr, or NULL
// the reference prints notes through Rcout (= std::cout here): let a caller that owns stdout empty the buffer while it has redirected fd 1
void ref_flush_stdout() { std::cout.flush(); std::fflush(stdout); }
void* ref_create() { return new Handle(); }
void ref_destroy(void* p) { delete (Handle*)p; }
const char* ref_last_error(void* p) { return ((Handle*)p)->err.c_str(); }

// harmony::setup (src/harmony.cpp:29-111).  Phi arrives as the C-hot design's row indices / column pointers (one entry per covariate and
// cell, unit values), as RunHarmony builds it (R/ui.R:210-213).
int ref_setup(void* p, const double* Z, int64_t N, int d, const int32_t* phi_i, const int32_t* phi_p, int B, const double* sigma,
              const double* theta, const double* lambda, int n_lambda, double alpha, int max_iter_kmeans, double eps_k, double eps_h, int K,
              double block_size, const int32_t* B_vec, int C, double cutoff) {
  Handle* H = (Handle*)p;
  return guarded(H, [&]() {
    RMAT Zm((arma::uword)d, (arma::uword)N);
    std::memcpy(Zm.memptr(), Z, sizeof(double) * (size_t)d * (size_t)N);
    const int64_t nnz = phi_p[N];
    arma::uvec rowind((arma::uword)nnz), colptr((arma::uword)N + 1);
    for (int64_t q = 0; q < nnz; q++) rowind[q] = (arma::uword)phi_i[q];
    for (int64_t i = 0; i <= N; i++) colptr[i] = (arma::uword)phi_p[i];
    RSPMAT Phi(rowind, colptr, RVEC((arma::uword)nnz, arma::fill::ones), (arma::uword)B, (arma::uword)N);
    RVEC sg((arma::uword)K), th((arma::uword)B), lm((arma::uword)n_lambda);
    for (int k = 0; k < K; k++) sg[k] = sigma[k];
    for (int b = 0; b < B; b++) th[b] = theta[b];
    for (int i = 0; i < n_lambda; i++) lm[i] = lambda[i];
    std::vector<int> bv(B_vec, B_vec + C);
    H->n_covariates = C;
    H->h.setup(Zm, Phi, sg, th, lm, (float)alpha, max_iter_kmeans, (float)eps_k, (float)eps_h, K, (float)block_size, bv, (float)cutoff, false);
    return 0;
  });
}
// set.seed(seed), then harmony::init_cluster_cpp (src/harmony.cpp:131-156): the k-means++ race and the Lloyd iterations draw from R's stream
int ref_init_cluster(void* p, uint64_t seed) {
  Handle* H = (Handle*)p;
  return guarded(H, [&]() { arma::shim::rng().set_seed((uint32_t)seed); H->h.init_cluster_cpp(); return 0; });
}
// init_cluster_cpp from GIVEN centroids (the parity tests share the initial centroids between backends; the reference has no such entry:
// setY is declared, src/harmony.h:41, and defined nowhere).  Every number still comes out of the reference's own lines: its
// init_cluster_cpp runs first (its own seeds and Lloyd iterations: discarded), then Y = normalise(Y0) (:136), then R, dist_mat, E, O through
// cluster_cpp's cold-start branch (:214-228) entered with no round to run, then compute_objective (:153) and the first objective_harmony
// entry (:154).  The one deviation from a genuine init: the cold-start branch normalises the already normalised Z_corr once more (an ulp
// in single precision -- which is why the bit-for-bit tests never use this entry --, nothing in the double-precision build it exists for).
int ref_init_cluster_from(void* p, const double* Y0, uint64_t seed) {
  Handle* H = (Handle*)p;
  return guarded(H, [&]() {
    harmony& h = H->h;
    arma::shim::rng().set_seed((uint32_t)seed);
    h.init_cluster_cpp();
    MATTYPE Ym(h.d, h.K);
    for (arma::uword i = 0; i < Ym.n_elem; i++) Ym[i] = (SCALAR)Y0[i];
    h.Y = arma::normalise(Ym, 2, 0);
    const unsigned keep = h.max_iter_kmeans;
    h.max_iter_kmeans = 0;
    h.objective_harmony.assign(2, 0.f);
    h.objective_kmeans.assign(1, 0.f);
    const int st = h.cluster_cpp();
    h.max_iter_kmeans = keep;
    h.objective_kmeans.clear(); h.objective_kmeans_dist.clear(); h.objective_kmeans_entropy.clear(); h.objective_kmeans_cross.clear();
    h.objective_harmony.clear(); h.kmeans_rounds.clear();
    h.compute_objective();
    h.objective_harmony.push_back(h.objective_kmeans.back());
    arma::shim::rng().set_seed((uint32_t)seed);          // shuffles drawn from R's stream start at its beginning, as for a backend that drew no seeds
    return st;
  });
}
int ref_scalar_bytes() { return (int)sizeof(SCALAR); }      // 4: the reference as it ships; 8: built with -DHARMONY_SCALAR_DOUBLE (src/types.h:5-9)
void ref_clear_update_orders() { arma::shim::injected_orders().clear(); }
int ref_cluster(void* p) { Handle* H = (Handle*)p; return guarded(H, [&]() { return H->h.cluster_cpp(); }); }
int ref_moe_correct_ridge(void* p) { Handle* H = (Handle*)p; return guarded(H, [&]() { H->h.moe_correct_ridge_cpp(); return 0; }); }
int ref_check_convergence(void* p, int type) { Handle* H = (Handle*)p; return guarded(H, [&]() { return H->h.check_convergence(type) ? 1 : 0; }); }
int ref_compute_objective(void* p) { Handle* H = (Handle*)p; return guarded(H, [&]() { H->h.compute_objective(); return 0; }); }
// the next update_R's arma::shuffle returns this order instead of drawing one
void ref_push_update_order(void* p, const int64_t* order) {
  Handle* H = (Handle*)p;
  std::vector<arma::uword> o((size_t)H->h.N);
  for (size_t i = 0; i < o.size(); i++) o[i] = (arma::uword)order[i];
  arma::shim::injected_orders().push_back(std::move(o));
}
void ref_set_int(void* p, const char* what, int64_t v) {
  Handle* H = (Handle*)p;
  if (std::string(what) == "max_iter_kmeans") H->h.max_iter_kmeans = (unsigned)v;
}
int64_t ref_get(void* p, const char* what, double* out) {
  Handle* H = (Handle*)p; harmony& h = H->h; const std::string w(what);
  try {
    if (w == "Z_corr") return copy_out(h.Z_corr.memptr(), h.Z_corr.n_elem, out);
    if (w == "Z_orig") return copy_out(h.Z_orig.memptr(), h.Z_orig.n_elem, out);
    if (w == "R") return copy_out(h.R.memptr(), h.R.n_elem, out);
    if (w == "dist") return copy_out(h.dist_mat.memptr(), h.dist_mat.n_elem, out);
    if (w == "Y") return copy_out(h.Y.memptr(), h.Y.n_elem, out);
    if (w == "O") return copy_out(h.O.memptr(), h.O.n_elem, out);
    if (w == "E") return copy_out(h.E.memptr(), h.E.n_elem, out);
    if (w == "W") return copy_out(h.W.memptr(), h.W.n_elem, out);
    if (w == "W_rows") { if (out) out[0] = (double)h.W.n_rows; return 1; }
    if (w == "Pr_b") return copy_out(h.Pr_b.memptr(), h.Pr_b.n_elem, out);
    if (w == "theta") return copy_out(h.theta.memptr(), h.theta.n_elem, out);
    if (w == "sigma") return copy_out(h.sigma.memptr(), h.sigma.n_elem, out);
    if (w == "lambda") return copy_out(h.lambda.memptr(), h.lambda.n_elem, out);
    if (w == "block_size") { if (out) out[0] = (double)h.block_size; return 1; }
    if (w == "objective_kmeans") return copy_vec(h.objective_kmeans, out);
    if (w == "objective_kmeans_dist") return copy_vec(h.objective_kmeans_dist, out);
    if (w == "objective_kmeans_entropy") return copy_vec(h.objective_kmeans_entropy, out);
    if (w == "objective_kmeans_cross") return copy_vec(h.objective_kmeans_cross, out);
    if (w == "objective_harmony") return copy_vec(h.objective_harmony, out);
    if (w == "kmeans_rounds") return copy_vec(h.kmeans_rounds, out);
    if (w == "update_order") return copy_out(h.update_order.memptr(), h.update_order.n_elem, out);
    if (w == "Lambda") { RMAT L = h.getLambda(); return copy_out(L.memptr(), L.n_elem, out); }   // getLambda (src/harmony.cpp:657-669)
    if (w == "warnings") { if (out) out[0] = (double)Rcpp::shim_warnings().size(); return 1; }
  } catch (const std::exception& e) { H->err = e.what(); return -2; }
  return -1;
}
}
