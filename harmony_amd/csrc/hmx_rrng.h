// hmx_rrng.h -- R-compatible random numbers for the "rng = R" mode of the library (host side only).
//
// The reference draws its centroid seeds (fill::randu, /root/reference/src/utils.cpp:12,29) and its per-round cell
// shuffle (arma::shuffle, src/harmony.cpp:272-273) from R's RNG: RcppArmadillo replaces Armadillo's generator by
//     randu_val()  = double(::Rf_runif(0, 1))              randi_val() = int(::Rf_runif(0, RAND_MAX))
// (RcppArmadillo's Alt_R_RNG.h; third-party, not vendored under /root/reference), and R's default generator is the
// Mersenne-Twister seeded by set.seed() (R/ui.R:263-266 only makes sure a seed exists).  This header restates
//   * MT19937 (Matsumoto & Nishimura 1998) as R runs it: genrand * 2.3283064365386963e-10, fixed up into (0,1)
//   * set.seed(seed): 50 rounds of the LCG  seed = 69069 * seed + 1  as initial scrambling, then 625 more for
//     dummy[0..624]; FixupSeeds sets dummy[0] = mti = 624                      (R: src/main/RNG.c, RNG_Init / FixupSeeds)
//   * Rf_runif(a, b) = a + (b - a) * unif_rand(), redrawing while u <= 0 or u >= 1   (R: src/nmath/runif.c)
//   * arma::shuffle = draw one randi per element IN ORDER, std::sort the (value, index) packets ascending by value
//     (Armadillo op_shuffle_meat.hpp; std::sort is not stable: ties follow libstdc++'s introsort, as they do in a
//     reference built with GCC)
// Known-answer tests: MT19937's published vector (init_by_array {0x123,0x234,0x345,0x456}: 1067595299 955945823 ...)
// and R's documented streams set.seed(1); runif(3) = 0.2655087 0.3721239 0.5728534, set.seed(42); runif(1) = 0.914806,
// set.seed(123); runif(3) = 0.2875775 0.7883051 0.4089769 (tests/test_abi_cpu.py).  End-to-end equality with a live R
// session needs R, which this image does not have.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace hmx {

class RRng {
 public:
  // a host that owns R's generator (the .Call glue: GetRNGstate / unif_rand / PutRNGstate) can supply the uniforms itself
  typedef double (*unif_fn)(void*);
  void set_source(unif_fn fn, void* user) { src_ = fn; user_ = user; }

  void mt_init_genrand(uint32_t s) {   // MT19937 reference initialisation (only used by the known-answer test)
    mt_[0] = s;
    for (int i = 1; i < 624; i++) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + (uint32_t)i;
    mti_ = 624;
  }
  void mt_init_by_array(const uint32_t* key, int len) {
    mt_init_genrand(19650218u);
    int i = 1, j = 0;
    for (int k = (624 > len ? 624 : len); k; k--) {
      mt_[i] = (mt_[i] ^ ((mt_[i - 1] ^ (mt_[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      i++; j++;
      if (i >= 624) { mt_[0] = mt_[623]; i = 1; }
      if (j >= len) j = 0;
    }
    for (int k = 623; k; k--) {
      mt_[i] = (mt_[i] ^ ((mt_[i - 1] ^ (mt_[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      i++;
      if (i >= 624) { mt_[0] = mt_[623]; i = 1; }
    }
    mt_[0] = 0x80000000u;
    mti_ = 624;
  }
  // R's set.seed(seed) for kind "Mersenne-Twister"
  void set_seed(uint32_t seed) {
    for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;
    uint32_t dummy0 = 0;
    for (int j = 0; j < 625; j++) {
      seed = 69069u * seed + 1u;
      if (j == 0) dummy0 = seed; else mt_[j - 1] = seed;
    }
    (void)dummy0;
    mti_ = 624;   // FixupSeeds: dummy[0] = 624
  }
  uint32_t genrand_int32() {
    if (mti_ >= 624) {
      static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
      int kk;
      for (kk = 0; kk < 624 - 397; kk++) {
        const uint32_t y = (mt_[kk] & 0x80000000u) | (mt_[kk + 1] & 0x7fffffffu);
        mt_[kk] = mt_[kk + 397] ^ (y >> 1) ^ mag01[y & 1u];
      }
      for (; kk < 623; kk++) {
        const uint32_t y = (mt_[kk] & 0x80000000u) | (mt_[kk + 1] & 0x7fffffffu);
        mt_[kk] = mt_[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u];
      }
      const uint32_t y = (mt_[623] & 0x80000000u) | (mt_[0] & 0x7fffffffu);
      mt_[623] = mt_[396] ^ (y >> 1) ^ mag01[y & 1u];
      mti_ = 0;
    }
    uint32_t y = mt_[mti_++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  double unif_rand() {   // R: fixup(MT_genrand())
    if (src_) return src_(user_);
    const double x = (double)genrand_int32() * 2.3283064365386963e-10;
    const double i2_32m1 = 2.328306437080797e-10;
    if (x <= 0.0) return 0.5 * i2_32m1;
    if ((1.0 - x) <= 0.0) return 1.0 - 0.5 * i2_32m1;
    return x;
  }
  double runif(double a, double b) {   // Rf_runif
    if (a == b) return a;
    double u;
    do { u = unif_rand(); } while (u <= 0 || u >= 1);
    return a + (b - a) * u;
  }
  float arma_randu() { return (float)runif(0.0, 1.0); }          // arma_rng::randu<float>
  int arma_randi() { return (int)runif(0.0, (double)RAND_MAX); }  // arma_rng::randi<int>

  // arma::shuffle(linspace<uvec>(0, N-1, N)): out[p] = the cell at position p of the shuffled order
  void arma_shuffle(int64_t N, std::vector<int64_t>& out) {
    struct Packet { int val; int64_t index; };
    std::vector<Packet> pk((size_t)N);
    for (int64_t i = 0; i < N; i++) { pk[(size_t)i].val = arma_randi(); pk[(size_t)i].index = i; }
    std::sort(pk.begin(), pk.end(), [](const Packet& a, const Packet& b) { return a.val < b.val; });
    out.resize((size_t)N);
    for (int64_t i = 0; i < N; i++) out[(size_t)i] = pk[(size_t)i].index;
  }

 private:
  uint32_t mt_[624] = {0};
  int mti_ = 625;
  unif_fn src_ = nullptr; void* user_ = nullptr;
};

}  // namespace hmx
