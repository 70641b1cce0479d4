/* comm_example.c -- the multi-GPU path of the real host (R / plain C: a process WITHOUT torch): one process per GPU, the
 * built-in RCCL communicator of libharmony_mi355x.so, the 128-byte unique id shipped through a file.
 *
 *   gcc -std=c11 -Iinclude examples/comm_example.c -Lharmony_amd/lib -lharmony_mi355x -Wl,-rpath,$PWD/harmony_amd/lib -lm -o comm_example
 *   for r in 0 1; do ./comm_example $r 2 /tmp/hmx_uid & done; wait        (rank r drives GPU r)
 *   ./comm_example 0 1 /tmp/hmx_uid                                        (1-rank communicator, collectives forced)
 *
 * Every rank generates the same global synthetic data set and keeps its contiguous shard of the cells; O / E / objective /
 * ridge statistics are all-reduced by the library (ncclAllReduce over xGMI, issued from C on the library's stream).
 * Prints "COMM_EXAMPLE_OK" with a checksum that must be identical on every rank and for every world size.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "harmony_mi355x.h"

static double u01(uint64_t* s) { *s = *s * 6364136223846793005ull + 1442695040888963407ull; return (double)(*s >> 11) / 9007199254740992.0; }
static double gauss(uint64_t* s) { return sqrt(-2.0 * log(u01(s) + 1e-300)) * cos(6.283185307179586 * u01(s)); }

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <rank> <world> <unique-id file>\n", argv[0]); return 2; }
  const int rank = atoi(argv[1]), world = atoi(argv[2]);
  const char* idfile = argv[3];
  const int64_t N = 40000; const int32_t d = 20, B = 4, K = 30, C = 1;
  uint64_t rng = 42;
  double* Z = malloc(sizeof(double) * (size_t)N * d);
  int32_t* lev = malloc(sizeof(int32_t) * (size_t)N);
  for (int64_t i = 0; i < N; i++) {                            /* the GLOBAL data set, identical on every rank */
    const int b = (int)(u01(&rng) * B);
    lev[i] = b;
    const int type = (int)(u01(&rng) * 5);
    for (int j = 0; j < d; j++) Z[(size_t)i * d + j] = 3.0 * ((type >> (j % 3)) & 1) + 0.3 * b * (j < 4) + gauss(&rng);
  }
  const int64_t lo = N * rank / world, hi = N * (rank + 1) / world, n = hi - lo;
  int32_t* phi_i = malloc(sizeof(int32_t) * (size_t)n);
  int32_t* phi_p = malloc(sizeof(int32_t) * (size_t)(n + 1));
  for (int64_t i = 0; i < n; i++) { phi_i[i] = lev[lo + i]; phi_p[i] = (int32_t)i; }
  phi_p[n] = (int32_t)n;

  hmx_ctx* h = hmx_create();
  if (!h) return 1;
  hmx_set_int(h, "seed", 1);
  hmx_set_int(h, "device", rank);                              /* one process per GPU */
  uint8_t uid[128];
  if (rank == 0) {
    if (hmx_comm_unique_id(uid)) { fprintf(stderr, "hmx_comm_unique_id failed (librccl not loadable?)\n"); return 1; }
    char tmp[1024]; snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
    FILE* f = fopen(tmp, "wb"); if (!f || fwrite(uid, 1, 128, f) != 128) { perror("unique id file"); return 1; }
    fclose(f); rename(tmp, idfile);                            /* atomic publish */
  } else {
    FILE* f = NULL;
    for (int tries = 0; tries < 600 && !(f = fopen(idfile, "rb")); tries++) { struct timespec ts = {0, 100000000}; nanosleep(&ts, NULL); }
    if (!f || fread(uid, 1, 128, f) != 128) { fprintf(stderr, "rank %d: no unique id in %s\n", rank, idfile); return 1; }
    fclose(f);
  }
  fprintf(stderr, "rank %d/%d: ncclCommInitRank ...\n", rank, world);
  int st = hmx_comm_init(h, rank, world, uid);
  if (st) { fprintf(stderr, "hmx_comm_init: %s\n", hmx_last_error(h)); return 1; }
  fprintf(stderr, "rank %d/%d: communicator up\n", rank, world);
  if ((st = hmx_set_shard(h, rank, world, lo, N, NULL, NULL))) { fprintf(stderr, "set_shard: %s\n", hmx_last_error(h)); return 1; }
  if (world == 1) hmx_set_int(h, "comm_force", 1);             /* exercise the collectives on one rank too */

  double sigma[30], theta[4] = {2, 2, 2, 2}, lambda = -1;
  for (int k = 0; k < K; k++) sigma[k] = 0.1;
  int32_t B_vec[1] = {4};
  st = hmx_setup(h, Z + (size_t)lo * d, n, d, phi_i, phi_p, NULL, B, sigma, theta, &lambda, 1, 0.2, 4, 1e-5, 1e-4, K, 0.05, B_vec, C, 1e-5, 0);
  if (st) { fprintf(stderr, "setup: %s\n", hmx_last_error(h)); return 1; }
  if ((st = hmx_init_cluster(h, NULL))) { fprintf(stderr, "init_cluster: %s\n", hmx_last_error(h)); return 1; }
  int iter;
  for (iter = 1; iter <= 5; iter++) {
    if ((st = hmx_cluster(h))) { fprintf(stderr, "cluster: %s\n", hmx_last_error(h)); return 1; }
    if ((st = hmx_moe_correct_ridge(h))) { fprintf(stderr, "moe_correct_ridge: %s\n", hmx_last_error(h)); return 1; }
    if (hmx_check_convergence(h, 1) == 1) break;
  }
  double obj[64]; const int64_t no = hmx_get(h, "objective_harmony", obj, 64);
  double O[4 * 30]; hmx_get(h, "O", O, 120);
  double chk = 0; for (int i = 0; i < 120; i++) chk += O[i] * (1 + i % 7);
  double calls = 0; hmx_get(h, "comm:calls", &calls, 1);
  printf("COMM_EXAMPLE_OK rank %d/%d iterations %d objective %.4f O-checksum %.6f collectives %.0f\n", rank, world, iter,
         no > 0 ? obj[no - 1] : 0.0, chk, calls);
  printf("block chain over the peers' inboxes: %s\n", hmx_p2p_status(h));   /* set up by hmx_comm_init when world > 1 */
  hmx_destroy(h);
  free(Z); free(lev); free(phi_i); free(phi_p);
  return 0;
}
