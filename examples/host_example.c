/* host_example.c -- driving libharmony_mi355x.so from plain C, exactly as the reference's R code drives its Rcpp
 * module (R/ui.R:269-295, R/utils.R:15-46): new -> setup -> init_cluster_cpp -> harmonize() loop -> getZcorr.
 *
 *   gcc -std=c11 -Iinclude examples/host_example.c -Lharmony_amd/lib -lharmony_mi355x -Wl,-rpath,$PWD/harmony_amd/lib -lm -o host_example
 *   ./host_example            (needs an MI355X: the library has no CPU fallback)
 *
 * Synthetic input: N cells x d PCs (column-major d x N doubles, like the reference's Z), two batches with a shifted mean.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "harmony_mi355x.h"

static double u01(uint64_t* s) { *s = *s * 6364136223846793005ull + 1442695040888963407ull; return (double)(*s >> 11) / 9007199254740992.0; }
static double gauss(uint64_t* s) { return sqrt(-2.0 * log(u01(s) + 1e-300)) * cos(6.283185307179586 * u01(s)); }

int main(void) {
  const int64_t N = 20000; const int32_t d = 20, B = 2, K = 30, C = 1;
  uint64_t rng = 42;
  double* Z = malloc(sizeof(double) * (size_t)N * d);
  int32_t* phi_i = malloc(sizeof(int32_t) * (size_t)N);      /* dgCMatrix slots of the B x N one-hot design */
  int32_t* phi_p = malloc(sizeof(int32_t) * (size_t)(N + 1));
  for (int64_t i = 0; i < N; i++) {
    const int b = (int)(u01(&rng) < 0.4);
    phi_i[i] = b; phi_p[i] = (int32_t)i;
    const int type = (int)(u01(&rng) * 5);
    for (int j = 0; j < d; j++) Z[(size_t)i * d + j] = 3.0 * ((type >> (j % 3)) & 1) + (b ? 0.8 : 0.0) * (j < 4) + gauss(&rng);
  }
  phi_p[N] = (int32_t)N;
  double sigma[30], theta[2] = {2, 2}, lambda = -1;           /* lambda = -1: automatic estimation (R/ui.R:224-231) */
  for (int k = 0; k < K; k++) sigma[k] = 0.1;
  int32_t B_vec[1] = {2};

  hmx_ctx* h = hmx_create();
  if (!h) return 1;
  hmx_set_int(h, "seed", 1);
  int st = hmx_setup(h, Z, N, d, phi_i, phi_p, NULL, B, sigma, theta, &lambda, 1, /*alpha*/ 0.2, /*max_iter_kmeans*/ 20,
                     /*epsilon_kmeans*/ 1e-5, /*epsilon_harmony*/ 1e-4, K, /*block_size*/ 0.05, B_vec, C,
                     /*batch_proportion_cutoff*/ 1e-5, /*verbose*/ 0);
  if (st) { fprintf(stderr, "setup: %s\n", hmx_last_error(h)); return 1; }
  if ((st = hmx_init_cluster(h, NULL))) { fprintf(stderr, "init_cluster: %s\n", hmx_last_error(h)); return 1; }
  int iter;
  for (iter = 1; iter <= 10; iter++) {                         /* harmonize(), R/utils.R:15-46 */
    if ((st = hmx_cluster(h))) { fprintf(stderr, "cluster: %s\n", st < 0 ? "interrupted" : hmx_last_error(h)); return 1; }
    if ((st = hmx_moe_correct_ridge(h))) { fprintf(stderr, "moe_correct_ridge: %s\n", hmx_last_error(h)); return 1; }
    if (hmx_check_convergence(h, 1) == 1) break;
  }
  const int64_t n = hmx_get(h, "Z_corr", NULL, 0);
  double* Zc = malloc(sizeof(double) * (size_t)n);
  hmx_get(h, "Z_corr", Zc, n);
  double obj[64]; const int64_t no = hmx_get(h, "objective_harmony", obj, 64);
  printf("converged after %d harmony iterations; objective_harmony:", iter);
  for (int64_t i = 0; i < no && i < 64; i++) printf(" %.1f", obj[i]);
  printf("\nZ_corr[0..3] of cell 0: %.4f %.4f %.4f %.4f\n", Zc[0], Zc[1], Zc[2], Zc[3]);
  hmx_destroy(h);
  free(Zc); free(Z); free(phi_i); free(phi_p);
  return 0;
}
