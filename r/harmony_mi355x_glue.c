/* harmony_mi355x_glue.c -- .Call glue between R and libharmony_mi355x.so.
 *
 * NEVER BUILT AGAINST R HERE: the build container has no R (no Rinternals.h, no libR).  This file is the
 * binding a maintainer of the reference would add in place of the Rcpp module
 * (/root/reference/src/harmony.cpp:672-709, registered at src/RcppExports.cpp:57-70); it only marshals
 * SEXPs to the plain-pointer C ABI of include/harmony_mi355x.h.  See INTEGRATION.md.
 * What the test-suite does with it: type-checks it against declarations of the R API (tests/stubs/R.h), and EXECUTES it linked with an
 * emulation of the R API calls it makes (tests/stubs/r_emul.c, driven by tests/r_emul.py the way r/harmony_mi355x.R drives it): error
 * paths on the CPU (tests/test_abi_cpu.py), the whole setup -> init -> cluster / correct -> getters sequence, the fp32 seam, R's stream
 * through unif_rand, a user interrupt, the warning and the finalizer on the GPU (tests/test_gpu_parity2.py), bit-identical to the ctypes path.
 *
 *   R CMD SHLIB harmony_mi355x_glue.c -I../include -L../harmony_amd/lib -lharmony_mi355x
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdint.h>
#include "harmony_mi355x.h"

static void hmx_finalizer(SEXP ptr) {
  hmx_ctx* h = (hmx_ctx*)R_ExternalPtrAddr(ptr);
  if (h) { hmx_destroy(h); R_ClearExternalPtr(ptr); }
}
static hmx_ctx* handle(SEXP ptr) {
  hmx_ctx* h = (hmx_ctx*)R_ExternalPtrAddr(ptr);
  if (!h) Rf_error("harmony object has been destroyed");
  return h;
}
static void check(hmx_ctx* h, int status, const char* what) {
  if (status > 0) Rf_error("%s: %s", what, hmx_last_error(h));          /* Rcpp::stop equivalent */
  if (status < 0) Rf_error("%s: terminated by user", what);              /* HMX_ABORTED from a method that returns void in the module */
  const char* w = hmx_last_warning(h);                                   /* one-shot: cleared by the read */
  if (w && w[0]) Rf_warning("%s", w);                                    /* Rcpp::warning equivalent */
}
static int poll_interrupt(void* unused) {                                 /* Progress::check_abort() */
  (void)unused;
  return R_ToplevelExec((void (*)(void*))R_CheckUserInterrupt, NULL) == FALSE;
}

SEXP C_hmx_new(void) {                                                    /* new(harmony), R/ui.R:269 */
  hmx_ctx* h = hmx_create();
  hmx_set_abort_poll(h, poll_interrupt, NULL);
  SEXP ptr = PROTECT(R_MakeExternalPtr(h, R_NilValue, R_NilValue));
  R_RegisterCFinalizerEx(ptr, hmx_finalizer, TRUE);
  UNPROTECT(1);
  return ptr;
}

/* harmonyObj$setup(data_mat, phi, sigma, theta, lambda_vec, alpha, max.iter.cluster, epsilon.cluster,
 *                  epsilon.harmony, nclust, block.size, B_vec, batch.prop.cutoff, verbose)   R/ui.R:271-275
 * phi is a dgCMatrix: pass phi@i, phi@p, phi@x, nrow(phi). */
SEXP C_hmx_setup(SEXP ptr, SEXP Z, SEXP phi_i, SEXP phi_p, SEXP phi_x, SEXP B, SEXP sigma, SEXP theta, SEXP lambda,
                 SEXP alpha, SEXP max_iter_kmeans, SEXP eps_k, SEXP eps_h, SEXP K, SEXP block_size, SEXP B_vec,
                 SEXP cutoff, SEXP verbose) {
  hmx_ctx* h = handle(ptr);
  SEXP dim = Rf_getAttrib(Z, R_DimSymbol);
  const int d = INTEGER(dim)[0];
  const int64_t N = INTEGER(dim)[1];
  int st = hmx_setup(h, REAL(Z), N, d, INTEGER(phi_i), INTEGER(phi_p), REAL(phi_x), Rf_asInteger(B), REAL(sigma),
                     REAL(theta), REAL(lambda), LENGTH(lambda), Rf_asReal(alpha), Rf_asInteger(max_iter_kmeans),
                     Rf_asReal(eps_k), Rf_asReal(eps_h), Rf_asInteger(K), Rf_asReal(block_size), INTEGER(B_vec),
                     LENGTH(B_vec), Rf_asReal(cutoff), Rf_asLogical(verbose));
  check(h, st, "setup");
  return R_NilValue;
}
/* Single-precision seam.  R has no float type; the `float` package carries fp32 matrices as an S4 object whose slot @Data is
 * an INTEGER matrix holding the float bits.  The reference converts the caller's double matrix to fp32 first thing (conv_to<MATTYPE>,
 * src/harmony.cpp:41): a caller who already holds fp32 data hands over Z@Data here and gets Z_corr back the same way
 * (float::float32(.Call("C_hmx_get_matrix_f32", ...))) -- half the bytes over PCIe in both directions and no fp64 blow-up on the host
 * (1M x 50: 200 MB instead of 400 MB each way; egress 9.4 -> ~5 ms). */
SEXP C_hmx_setup_f32(SEXP ptr, SEXP Zbits, SEXP phi_i, SEXP phi_p, SEXP phi_x, SEXP B, SEXP sigma, SEXP theta, SEXP lambda,
                     SEXP alpha, SEXP max_iter_kmeans, SEXP eps_k, SEXP eps_h, SEXP K, SEXP block_size, SEXP B_vec,
                     SEXP cutoff, SEXP verbose) {
  hmx_ctx* h = handle(ptr);
  if (TYPEOF(Zbits) != INTSXP) Rf_error("setup: expected the integer bit matrix of a float32 object (Z@Data)");
  SEXP dim = Rf_getAttrib(Zbits, R_DimSymbol);
  const int d = INTEGER(dim)[0];
  const int64_t N = INTEGER(dim)[1];
  int st = hmx_setup_ex(h, (const void*)INTEGER(Zbits), HMX_F32, HMX_HOST, N, d, INTEGER(phi_i), INTEGER(phi_p), REAL(phi_x), Rf_asInteger(B),
                        REAL(sigma), REAL(theta), REAL(lambda), LENGTH(lambda), Rf_asReal(alpha), Rf_asInteger(max_iter_kmeans),
                        Rf_asReal(eps_k), Rf_asReal(eps_h), Rf_asInteger(K), Rf_asReal(block_size), INTEGER(B_vec),
                        LENGTH(B_vec), Rf_asReal(cutoff), Rf_asLogical(verbose));
  check(h, st, "setup");
  return R_NilValue;
}
/* Z_corr / Z_orig / R as fp32 bits in an INTEGER matrix (rows x cols given by the caller: d or K by N) */
SEXP C_hmx_get_matrix_f32(SEXP ptr, SEXP field, SEXP nrow, SEXP ncol) {
  hmx_ctx* h = handle(ptr);
  const char* f = CHAR(STRING_ELT(field, 0));
  const int nr = Rf_asInteger(nrow), nc = Rf_asInteger(ncol);
  SEXP out = PROTECT(Rf_allocMatrix(INTSXP, nr, nc));
  const int64_t n = (int64_t)nr * nc;
  if (hmx_get_matrix(h, f, (void*)INTEGER(out), HMX_F32, HMX_HOST, n) != n) { UNPROTECT(1); Rf_error("getter '%s': %s", f, hmx_last_error(h)); }
  UNPROTECT(1);
  return out;
}
SEXP C_hmx_set_seed(SEXP ptr, SEXP seed) { hmx_set_int(handle(ptr), "seed", (int64_t)Rf_asReal(seed)); return R_NilValue; }
/* R-compatible randomness (hmx_set_int "rng" = 1): the library draws what the reference draws through RcppArmadillo --
 * randu = Rf_runif(0, 1), randi = int(Rf_runif(0, RAND_MAX)) -- but from R's OWN generator: unif_rand() between
 * GetRNGstate() / PutRNGstate(), so set.seed(), RNGkind() and the position of the stream are all R's. */
static double r_unif(void* unused) { (void)unused; return unif_rand(); }
SEXP C_hmx_use_r_rng(SEXP ptr) {
  hmx_ctx* h = handle(ptr);
  hmx_set_uniform_source(h, r_unif, NULL);
  check(h, hmx_set_int(h, "rng", 1), "rng");
  return R_NilValue;
}
SEXP C_hmx_init_cluster(SEXP ptr) {
  hmx_ctx* h = handle(ptr);
  GetRNGstate();                                                          /* centroid seeds may come from R's stream */
  int st = hmx_init_cluster(h, NULL);
  PutRNGstate();
  check(h, st, "init_cluster_cpp");
  return R_NilValue;
}
SEXP C_hmx_cluster(SEXP ptr) {
  hmx_ctx* h = handle(ptr);
  GetRNGstate();                                                          /* the rounds' shuffles may come from R's stream */
  int st = hmx_cluster(h);
  PutRNGstate();
  if (st > 0) check(h, st, "cluster_cpp");
  return Rf_ScalarInteger(st);                                            /* 0 ok, -1 user interrupt (R/utils.R:26-32) */
}
SEXP C_hmx_moe_correct_ridge(SEXP ptr) { hmx_ctx* h = handle(ptr); int st = hmx_moe_correct_ridge(h); if (st > 0) check(h, st, "moe_correct_ridge_cpp"); return R_NilValue; }
SEXP C_hmx_check_convergence(SEXP ptr, SEXP type) {
  hmx_ctx* h = handle(ptr);
  int r = hmx_check_convergence(h, Rf_asInteger(type));
  if (r < 0) Rf_error("check_convergence: %s", hmx_last_error(h));
  return Rf_ScalarLogical(r);
}
SEXP C_hmx_compute_objective(SEXP ptr) { hmx_ctx* h = handle(ptr); check(h, hmx_compute_objective(h), "compute_objective"); return R_NilValue; }
SEXP C_hmx_set_int(SEXP ptr, SEXP field, SEXP value) {
  hmx_ctx* h = handle(ptr);
  check(h, hmx_set_int(h, CHAR(STRING_ELT(field, 0)), (int64_t)Rf_asReal(value)), "set");
  return R_NilValue;
}
/* every field / getter: returns a numeric vector; the R side sets dim() */
SEXP C_hmx_get(SEXP ptr, SEXP field) {
  hmx_ctx* h = handle(ptr);
  const char* f = CHAR(STRING_ELT(field, 0));
  int64_t n = hmx_get(h, f, NULL, 0);
  if (n < 0) Rf_error("unknown field '%s'", f);
  SEXP out = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)n));
  if (n && hmx_get(h, f, REAL(out), n) != n) { UNPROTECT(1); Rf_error("getter '%s': %s", f, hmx_last_error(h)); }
  UNPROTECT(1);
  return out;
}

static const R_CallMethodDef CallEntries[] = {
  {"C_hmx_new", (DL_FUNC)&C_hmx_new, 0},
  {"C_hmx_setup", (DL_FUNC)&C_hmx_setup, 18},
  {"C_hmx_setup_f32", (DL_FUNC)&C_hmx_setup_f32, 18},
  {"C_hmx_get_matrix_f32", (DL_FUNC)&C_hmx_get_matrix_f32, 4},
  {"C_hmx_set_seed", (DL_FUNC)&C_hmx_set_seed, 2},
  {"C_hmx_use_r_rng", (DL_FUNC)&C_hmx_use_r_rng, 1},
  {"C_hmx_init_cluster", (DL_FUNC)&C_hmx_init_cluster, 1},
  {"C_hmx_cluster", (DL_FUNC)&C_hmx_cluster, 1},
  {"C_hmx_moe_correct_ridge", (DL_FUNC)&C_hmx_moe_correct_ridge, 1},
  {"C_hmx_check_convergence", (DL_FUNC)&C_hmx_check_convergence, 2},
  {"C_hmx_compute_objective", (DL_FUNC)&C_hmx_compute_objective, 1},
  {"C_hmx_set_int", (DL_FUNC)&C_hmx_set_int, 3},
  {"C_hmx_get", (DL_FUNC)&C_hmx_get, 2},
  {NULL, NULL, 0}
};
/* R calls R_init_<name of the DLL> when the package's shared object is loaded.  Default: the glue is the whole DLL of a package named
 * `harmony` (it then replaces RcppExports.cpp's R_init_harmony and with it _rcpp_module_boot_harmony_module).  Recommended instead:
 * the companion package r/make_companion_package.sh assembles (-DHMX_R_PACKAGE=harmonymi355x): the reference package keeps its own
 * DLL -- its other Rcpp exports, scaleRows_dgc / kmeans_centers / find_lambda_cpp (src/RcppExports.cpp:14-62), stay registered -- and
 * R/ui.R:269 asks the companion for the object.  INTEGRATION.md. */
#ifndef HMX_R_PACKAGE
#define HMX_R_PACKAGE harmony
#endif
#define HMX_R_INIT_2(pkg) R_init_##pkg
#define HMX_R_INIT_1(pkg) HMX_R_INIT_2(pkg)
void HMX_R_INIT_1(HMX_R_PACKAGE)(DllInfo* dll) {
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
