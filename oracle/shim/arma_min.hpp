// =============================================================================
// TEST INFRASTRUCTURE ONLY -- a minimal stand-in for the Armadillo interface, written from scratch for ONE purpose: to let the
// reference's own engine sources (/root/reference/src/harmony.cpp, utils.cpp, timer.cpp) compile UNMODIFIED, where they lie, into
// oracle/_ref/libharmony_ref.so (oracle/Makefile, target _ref), so that the restated oracle (oracle/harmony_oracle.cpp) can be checked
// against the reference's real control flow bit for bit (tests/test_oracle_ref.py).
//
// What this is NOT: Armadillo.  RcppArmadillo is not on disk (DESCRIPTION:54, unpinned; no network).  Every operation below is an eager,
// single-threaded, fp-sequential restatement of the PUBLISHED semantics of the Armadillo call of the same name, covering exactly the
// calls the three reference sources make.  The reference's algorithm -- every line of setup, init_cluster_cpp, update_R,
// compute_objective, check_convergence, cluster_cpp, moe_correct_ridge_cpp, kmeans_centers, harmony_pow, find_lambda_cpp, getLambda --
// is the reference's; the arithmetic underneath each matrix call is this file's:
//   sum(X, dim), my_accu's operand, products      one fp32 accumulator, ascending index, every product rounded before it is added
//   dense * dense                                 C(m, n) = sum_k A(m, k) B(k, n), k ascending, from 0
//   dense * sparse                                out = 0; for every non-zero (row, col, v) of the sparse operand in storage order:
//                                                 out.col(col) += A.col(row) * v     (Armadillo's classic glue_times_dense_sparse loop)
//   sparse * sparse                               Gustavson: per output column, the right operand's non-zeros in ascending row order
//   normalise(X, p, 0)                            column / norm_p(column), zero norm -> divide by 1; norms with one fp32 accumulator
//   accu(vector)                                  two accumulators (even / odd elements), added at the end
//   inv                                           unblocked fp32 LU with partial pivoting applied to the identity (or, on request of the
//                                                 test harness, a real LAPACK: ../lapack_inv.hpp)
//   kmeans(means, X, K, keep_existing, 1, ...)    one Lloyd iteration, fp64 member sums, an empty cluster keeps its mean
//   randu / randi / shuffle                       R's stream as RcppArmadillo maps it (MT19937 after set.seed's scrambling; shuffle =
//                                                 one randi per element in order + std::sort of (value, index) packets), or an injected
//                                                 permutation (shim::injected_orders())
// i.e. the choices oracle/harmony_oracle.cpp documents for its faithful mode (its header, "LIBERTIES"; the dense * sparse loop is that
// header's bit 2).  Nothing here is shipped, linked into or loaded by the product.
// =============================================================================
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <iostream>
#include <numeric>
#include <set>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../lapack_inv.hpp"

namespace arma {
typedef unsigned long long uword;   // ARMA_64BIT_WORD (src/types.h:2)
typedef long long sword;

template <class T> class Mat;
template <class T> class Col;
template <class T> class SpMat;

namespace shim {
inline void fail(const char* what) { throw std::logic_error(std::string("arma shim: ") + what); }
inline void need(bool ok, const char* what) { if (!ok) fail(what); }
// R's default generator as the reference draws from it through RcppArmadillo (randu = Rf_runif(0, 1), randi = int(Rf_runif(0, RAND_MAX)))
struct RStream {
  uint32_t mt[624]; int mti = 625;
  void set_seed(uint32_t seed) {                                   // set.seed: 50 LCG steps, then 625 words of 69069 x + 1; word 0 is replaced by mti = 624
    for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;
    seed = 69069u * seed + 1u;
    for (int j = 0; j < 624; j++) { seed = 69069u * seed + 1u; mt[j] = seed; }
    mti = 624;
  }
  uint32_t genrand() {
    if (mti >= 624) {
      if (mti == 625) set_seed(4357u);
      for (int k = 0; k < 624; k++) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      mti = 0;
    }
    uint32_t y = mt[mti++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
  double unif_rand() {                                             // MT_genrand * 2^-32, kept inside (0, 1)
    const double x = (double)genrand() * 2.3283064365386963e-10, h = 0.5 * 2.328306437080797e-10;
    return x <= 0.0 ? h : ((1.0 - x) <= 0.0 ? 1.0 - h : x);
  }
  double runif(double a, double b) { if (a == b) return a; double u; do { u = unif_rand(); } while (u <= 0 || u >= 1); return a + (b - a) * u; }
};
inline RStream& rng() { static RStream r; return r; }
inline std::deque<std::vector<uword>>& injected_orders() { static std::deque<std::vector<uword>> q; return q; }
}  // namespace shim

namespace fill {
struct zeros_t {}; struct ones_t {}; struct randu_t {}; struct none_t {};
static const zeros_t zeros = zeros_t();
static const ones_t ones = ones_t();
static const randu_t randu = randu_t();
static const none_t none = none_t();
}  // namespace fill
struct SizeMat { uword n_rows, n_cols; };
enum kmeans_seed_mode { keep_existing, static_subset, random_subset, static_spread, random_spread };

// every dense thing (matrix, view, proxy) can be turned into a Mat
template <class T, class D> struct Base {
  const D& derived() const { return static_cast<const D&>(*this); }
};
template <class T> const Mat<T>& unwrap(const Base<T, Mat<T>>& x) { return static_cast<const Mat<T>&>(x); }
template <class T, class D> Mat<T> unwrap(const Base<T, D>& x) { return x.derived().to_mat(); }

template <class T> struct subview;
template <class T> struct EachCol;
template <class T> struct EachRow;
template <class T> struct ColsProxy;
template <class T> struct RowsProxy;
template <class T> struct ElemProxy;
template <class T> struct DiagView;

// ------------------------------------------------------------------------------------------------------------------ dense matrix
template <class T> class Mat : public Base<T, Mat<T>> {
 public:
  typedef T elem_type;
  uword n_rows = 0, n_cols = 0, n_elem = 0;
  T* mem = nullptr;
  bool owns = true;      // false: a window on somebody else's column (unsafe_col)
  bool is_col = false;   // Col<T>: assignments must keep one column

  Mat() {}
  Mat(uword r, uword c) { init(r, c); zeros(); }
  Mat(uword r, uword c, fill::zeros_t) { init(r, c); zeros(); }
  Mat(uword r, uword c, fill::ones_t) { init(r, c); fill(T(1)); }
  Mat(uword r, uword c, fill::none_t) { init(r, c); }
  Mat(uword r, uword c, fill::randu_t) { init(r, c); randu(); }
  Mat(const SizeMat& s) { init(s.n_rows, s.n_cols); zeros(); }
  Mat(const Mat& o) { init(o.n_rows, o.n_cols); copy_from(o.mem); }
  Mat(Mat&& o) noexcept { steal(o); }
  template <class D> Mat(const Base<T, D>& x) { Mat t = x.derived().to_mat(); steal(t); }
  Mat(const SpMat<T>& s);
  Mat(T* aux, uword r, uword c, bool) : n_rows(r), n_cols(c), n_elem(r * c), mem(aux), owns(false) {}   // window
  ~Mat() { if (owns) delete[] mem; }

  Mat& operator=(const Mat& o) { if (this != &o) assign(o); return *this; }
  Mat& operator=(Mat&& o) noexcept {
    if (this == &o) return *this;
    if (!owns || !o.owns) { assign(o); return *this; }
    check_layout(o.n_rows, o.n_cols);
    delete[] mem; mem = nullptr; steal(o); return *this;
  }
  template <class D> Mat& operator=(const Base<T, D>& x) { Mat t = x.derived().to_mat(); return *this = std::move(t); }
  Mat& operator=(const SpMat<T>& s) { Mat t(s); return *this = std::move(t); }
  Mat to_mat() const { return *this; }

  void set_size(uword r, uword c) { if (r == n_rows && c == n_cols) return; need_own(); check_layout(r, c); delete[] mem; mem = nullptr; init(r, c); }
  void set_size(uword n) { set_size(n, 1); }
  Mat& zeros() { for (uword i = 0; i < n_elem; i++) mem[i] = T(0); return *this; }
  Mat& zeros(uword r, uword c) { set_size(r, c); return zeros(); }
  Mat& ones() { return fill(T(1)); }
  template <class S> Mat& fill(S v) { for (uword i = 0; i < n_elem; i++) mem[i] = T(v); return *this; }
  Mat& randu() { for (uword i = 0; i < n_elem; i++) mem[i] = T(shim::rng().runif(0.0, 1.0)); return *this; }
  T* memptr() { return mem; }
  const T* memptr() const { return mem; }
  T* colptr(uword c) { return mem + c * n_rows; }
  const T* colptr(uword c) const { return mem + c * n_rows; }
  bool is_empty() const { return n_elem == 0; }

  T& operator()(uword i) { shim::need(i < n_elem, "index out of bounds"); return mem[i]; }
  const T& operator()(uword i) const { shim::need(i < n_elem, "index out of bounds"); return mem[i]; }
  T& operator[](uword i) { return mem[i]; }
  const T& operator[](uword i) const { return mem[i]; }
  T& operator()(uword r, uword c) { shim::need(r < n_rows && c < n_cols, "index out of bounds"); return mem[c * n_rows + r]; }
  const T& operator()(uword r, uword c) const { shim::need(r < n_rows && c < n_cols, "index out of bounds"); return mem[c * n_rows + r]; }
  T& at(uword r, uword c) { return mem[c * n_rows + r]; }
  const T& at(uword r, uword c) const { return mem[c * n_rows + r]; }

  Mat t() const { Mat o(n_cols, n_rows, fill::none); for (uword c = 0; c < n_cols; c++) for (uword r = 0; r < n_rows; r++) o.mem[r * n_cols + c] = mem[c * n_rows + r]; return o; }
  Mat as_col() const { Mat o(*this); o.n_rows = n_elem; o.n_cols = 1; return o; }   // column-major order, as Armadillo's vectorisation
  Mat as_row() const { Mat o(*this); o.n_rows = 1; o.n_cols = n_elem; return o; }

  subview<T> row(uword r) const;
  subview<T> col(uword c) const;
  subview<T> submat(uword r0, uword c0, uword r1, uword c1) const;
  subview<T> subvec(uword a, uword b) const;
  Col<T> unsafe_col(uword c) const;
  ColsProxy<T> cols(const Mat<uword>& idx) const;
  RowsProxy<T> rows(const Mat<uword>& idx) const;
  ElemProxy<T> elem(const Mat<uword>& idx) const;
  DiagView<T> diag() const;
  EachCol<T> each_col() const;
  EachRow<T> each_row() const;

  uword index_min() const { shim::need(n_elem > 0, "index_min of an empty object"); uword b = 0; for (uword i = 1; i < n_elem; i++) if (mem[i] < mem[b]) b = i; return b; }
  uword index_max() const { shim::need(n_elem > 0, "index_max of an empty object"); uword b = 0; for (uword i = 1; i < n_elem; i++) if (mem[i] > mem[b]) b = i; return b; }
  T max() const { return mem[index_max()]; }
  T min() const { return mem[index_min()]; }

  template <class D> Mat& operator+=(const Base<T, D>& x) { const Mat<T>& a = unwrap(x); same(a); for (uword i = 0; i < n_elem; i++) mem[i] += a.mem[i]; return *this; }
  template <class D> Mat& operator-=(const Base<T, D>& x) { const Mat<T>& a = unwrap(x); same(a); for (uword i = 0; i < n_elem; i++) mem[i] -= a.mem[i]; return *this; }
  template <class D> Mat& operator%=(const Base<T, D>& x) { const Mat<T>& a = unwrap(x); same(a); for (uword i = 0; i < n_elem; i++) mem[i] *= a.mem[i]; return *this; }
  template <class D> Mat& operator/=(const Base<T, D>& x) { const Mat<T>& a = unwrap(x); same(a); for (uword i = 0; i < n_elem; i++) mem[i] /= a.mem[i]; return *this; }
  template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> Mat& operator+=(S s) { for (uword i = 0; i < n_elem; i++) mem[i] += T(s); return *this; }
  template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> Mat& operator-=(S s) { for (uword i = 0; i < n_elem; i++) mem[i] -= T(s); return *this; }
  template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> Mat& operator*=(S s) { for (uword i = 0; i < n_elem; i++) mem[i] *= T(s); return *this; }
  template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> Mat& operator/=(S s) { for (uword i = 0; i < n_elem; i++) mem[i] /= T(s); return *this; }

 protected:
  void init(uword r, uword c) { n_rows = r; n_cols = c; n_elem = r * c; mem = n_elem ? new T[n_elem] : nullptr; owns = true; }
  void copy_from(const T* src) { if (n_elem) std::memcpy(mem, src, sizeof(T) * n_elem); }
  void need_own() const { shim::need(owns, "resizing a window on another object's memory"); }
  void check_layout(uword, uword c) const { shim::need(!is_col || c == 1, "requested size is not compatible with a column vector"); }
  void same(const Mat& a) const { shim::need(a.n_rows == n_rows && a.n_cols == n_cols, "element-wise operation on objects of different size"); }
  void steal(Mat& o) { n_rows = o.n_rows; n_cols = o.n_cols; n_elem = o.n_elem; mem = o.mem; owns = o.owns; o.mem = nullptr; o.n_rows = o.n_cols = o.n_elem = 0; o.owns = true; }
  void assign(const Mat& o) {
    if (!owns) { shim::need(o.n_elem == n_elem, "assignment through a window must keep its size"); if (n_elem) std::memmove(mem, o.mem, sizeof(T) * n_elem); return; }
    check_layout(o.n_rows, o.n_cols);
    if (o.n_elem != n_elem) { delete[] mem; mem = nullptr; init(o.n_rows, o.n_cols); } else { n_rows = o.n_rows; n_cols = o.n_cols; }
    copy_from(o.mem);
  }
};

template <class T> class Col : public Mat<T> {
 public:
  Col() { this->is_col = true; this->n_cols = 1; }
  explicit Col(uword n) : Mat<T>(n, 1) { this->is_col = true; }
  Col(uword r, uword c) : Mat<T>(r, c) { this->is_col = true; shim::need(c == 1, "column vector with more than one column"); }
  Col(uword n, fill::zeros_t f) : Mat<T>(n, 1, f) { this->is_col = true; }
  Col(uword n, fill::ones_t f) : Mat<T>(n, 1, f) { this->is_col = true; }
  Col(uword n, fill::randu_t f) : Mat<T>(n, 1, f) { this->is_col = true; }
  Col(const SizeMat& s, fill::randu_t f) : Mat<T>(s.n_rows, s.n_cols, f) { this->is_col = true; shim::need(s.n_cols == 1, "column vector with more than one column"); }
  Col(const SizeMat& s, fill::zeros_t f) : Mat<T>(s.n_rows, s.n_cols, f) { this->is_col = true; shim::need(s.n_cols == 1, "column vector with more than one column"); }
  Col(const Col& o) : Mat<T>(static_cast<const Mat<T>&>(o)) { this->is_col = true; }
  Col(Col&& o) noexcept : Mat<T>(std::move(static_cast<Mat<T>&>(o))) { this->is_col = true; }
  template <class D> Col(const Base<T, D>& x) : Mat<T>(x) { this->is_col = true; shim::need(this->n_cols == 1 || this->n_elem == 0, "column vector initialised from an object with several columns"); if (this->n_elem == 0) this->n_cols = 1; }
  Col(const std::vector<T>& v) : Mat<T>(v.size(), 1, fill::none) { this->is_col = true; for (uword i = 0; i < this->n_elem; i++) this->mem[i] = v[i]; }
  Col(T* aux, uword n, bool w) : Mat<T>(aux, n, 1, w) { this->is_col = true; }
  Col& operator=(const Col& o) { Mat<T>::operator=(static_cast<const Mat<T>&>(o)); return *this; }
  Col& operator=(Col&& o) noexcept { Mat<T>::operator=(std::move(static_cast<Mat<T>&>(o))); return *this; }
  template <class D> Col& operator=(const Base<T, D>& x) { Mat<T>::operator=(x); return *this; }
};
template <class T> using Row = Mat<T>;

typedef Mat<double> mat;   typedef Mat<float> fmat;   typedef Mat<uword> umat;
typedef Col<double> vec;   typedef Col<float> fvec;   typedef Col<uword> uvec;
typedef SpMat<double> sp_mat; typedef SpMat<float> sp_fmat;

// ------------------------------------------------------------------------------------------------------------------ views
// a rectangular window (pointer, leading dimension, extent): rows, columns, sub-matrices and sub-vectors of a Mat
template <class T> struct subview : public Base<T, subview<T>> {
  T* p; uword ld, n_rows, n_cols, n_elem;
  subview(T* p_, uword ld_, uword r, uword c) : p(p_), ld(ld_), n_rows(r), n_cols(c), n_elem(r * c) {}
  subview(const subview&) = default;
  T& at(uword r, uword c) const { return p[c * ld + r]; }
  T& operator()(uword r, uword c) const { shim::need(r < n_rows && c < n_cols, "index out of bounds"); return at(r, c); }
  T& operator()(uword i) const { shim::need(i < n_elem, "index out of bounds"); return n_rows == 1 ? at(0, i) : at(i % n_rows, i / n_rows); }
  T& operator[](uword i) const { return n_rows == 1 ? at(0, i) : at(i % n_rows, i / n_rows); }
  Mat<T> to_mat() const { Mat<T> o(n_rows, n_cols, fill::none); for (uword c = 0; c < n_cols; c++) for (uword r = 0; r < n_rows; r++) o.mem[c * n_rows + r] = at(r, c); return o; }
  void put(const Mat<T>& a) const { shim::need(a.n_rows == n_rows && a.n_cols == n_cols, "assignment to a view of a different size"); for (uword c = 0; c < n_cols; c++) for (uword r = 0; r < n_rows; r++) at(r, c) = a.mem[c * n_rows + r]; }
  const subview& operator=(const subview& o) const { Mat<T> t = o.to_mat(); put(t); return *this; }
  subview& operator=(const subview& o) { Mat<T> t = o.to_mat(); put(t); return *this; }
  template <class D> const subview& operator=(const Base<T, D>& x) const { const Mat<T>& a = unwrap(x); put(a); return *this; }
  template <class D> const subview& operator+=(const Base<T, D>& x) const { const Mat<T>& a = unwrap(x); Mat<T> t = to_mat(); t += a; put(t); return *this; }
  template <class D> const subview& operator-=(const Base<T, D>& x) const { const Mat<T>& a = unwrap(x); Mat<T> t = to_mat(); t -= a; put(t); return *this; }
  void zeros() const { for (uword c = 0; c < n_cols; c++) for (uword r = 0; r < n_rows; r++) at(r, c) = T(0); }
  template <class S> void fill(S v) const { for (uword c = 0; c < n_cols; c++) for (uword r = 0; r < n_rows; r++) at(r, c) = T(v); }
  Mat<T> t() const { return to_mat().t(); }
  Mat<T> as_col() const { return to_mat().as_col(); }
  Mat<T> as_row() const { return to_mat().as_row(); }
  EachCol<T> each_col() const;
  EachRow<T> each_row() const;
};
template <class T> subview<T> Mat<T>::row(uword r) const { shim::need(r < n_rows, "row index out of bounds"); return subview<T>(mem + r, n_rows, 1, n_cols); }
template <class T> subview<T> Mat<T>::col(uword c) const { shim::need(c < n_cols, "column index out of bounds"); return subview<T>(mem + c * n_rows, n_rows, n_rows, 1); }
template <class T> subview<T> Mat<T>::submat(uword r0, uword c0, uword r1, uword c1) const {
  shim::need(r0 <= r1 && c0 <= c1 && r1 < n_rows && c1 < n_cols, "submat: indices out of bounds or incorrectly used");
  return subview<T>(mem + c0 * n_rows + r0, n_rows, r1 - r0 + 1, c1 - c0 + 1);
}
template <class T> subview<T> Mat<T>::subvec(uword a, uword b) const {
  shim::need(a <= b && b < n_elem && (n_rows == 1 || n_cols == 1), "subvec: indices out of bounds or incorrectly used");
  return n_cols == 1 ? subview<T>(mem + a, n_rows, b - a + 1, 1) : subview<T>(mem + a * n_rows, n_rows, 1, b - a + 1);
}
template <class T> Col<T> Mat<T>::unsafe_col(uword c) const { shim::need(c < n_cols, "column index out of bounds"); return Col<T>(mem + c * n_rows, n_rows, false); }

// X.each_col() op= v : every column with the vector v (one element per row); X.each_row() likewise with one element per column
template <class T> struct EachCol {
  subview<T> v;
  template <class F> void apply(const Mat<T>& x, F f) const { shim::need(x.n_elem == v.n_rows, "each_col(): incompatible size"); for (uword c = 0; c < v.n_cols; c++) for (uword r = 0; r < v.n_rows; r++) f(v.at(r, c), x.mem[r]); }
  template <class D> void operator+=(const Base<T, D>& x) const { apply(unwrap(x), [](T& a, T b) { a += b; }); }
  template <class D> void operator-=(const Base<T, D>& x) const { apply(unwrap(x), [](T& a, T b) { a -= b; }); }
  template <class D> void operator%=(const Base<T, D>& x) const { apply(unwrap(x), [](T& a, T b) { a *= b; }); }
  template <class D> void operator/=(const Base<T, D>& x) const { apply(unwrap(x), [](T& a, T b) { a /= b; }); }
};
template <class T> struct EachRow {
  subview<T> v;
  template <class F> void apply(const Mat<T>& x, F f) const { shim::need(x.n_elem == v.n_cols, "each_row(): incompatible size"); for (uword c = 0; c < v.n_cols; c++) for (uword r = 0; r < v.n_rows; r++) f(v.at(r, c), x.mem[c]); }
  template <class D> void operator+=(const Base<T, D>& x) const { apply(unwrap(x), [](T& a, T b) { a += b; }); }
  template <class D> void operator-=(const Base<T, D>& x) const { apply(unwrap(x), [](T& a, T b) { a -= b; }); }
  template <class D> void operator%=(const Base<T, D>& x) const { apply(unwrap(x), [](T& a, T b) { a *= b; }); }
  template <class D> void operator/=(const Base<T, D>& x) const { apply(unwrap(x), [](T& a, T b) { a /= b; }); }
};
template <class T> EachCol<T> Mat<T>::each_col() const { return EachCol<T>{subview<T>(mem, n_rows, n_rows, n_cols)}; }
template <class T> EachRow<T> Mat<T>::each_row() const { return EachRow<T>{subview<T>(mem, n_rows, n_rows, n_cols)}; }
template <class T> EachCol<T> subview<T>::each_col() const { return EachCol<T>{*this}; }
template <class T> EachRow<T> subview<T>::each_row() const { return EachRow<T>{*this}; }
#define ARMA_SHIM_EACH_BIN(OP, OPEQ)                                                                                                             \
  template <class T, class D> Mat<T> operator OP(const EachCol<T>& e, const Base<T, D>& x) { Mat<T> o = e.v.to_mat(); o.each_col() OPEQ x; return o; } \
  template <class T, class D> Mat<T> operator OP(const EachRow<T>& e, const Base<T, D>& x) { Mat<T> o = e.v.to_mat(); o.each_row() OPEQ x; return o; }
ARMA_SHIM_EACH_BIN(+, +=) ARMA_SHIM_EACH_BIN(-, -=) ARMA_SHIM_EACH_BIN(%, %=) ARMA_SHIM_EACH_BIN(/, /=)
#undef ARMA_SHIM_EACH_BIN

// X.cols(indices) / X.rows(indices) / X.elem(indices): gather on read, scatter on assignment
template <class T> struct ColsProxy : public Base<T, ColsProxy<T>> {
  Mat<T>* m = nullptr; Mat<uword> idx;
  ColsProxy() = default; ColsProxy(const ColsProxy&) = default;
  Mat<T> to_mat() const { Mat<T> o(m->n_rows, idx.n_elem, fill::none); for (uword j = 0; j < idx.n_elem; j++) { shim::need(idx.mem[j] < m->n_cols, "cols(): index out of bounds"); std::memcpy(o.colptr(j), m->colptr(idx.mem[j]), sizeof(T) * m->n_rows); } return o; }
  template <class D> void operator=(const Base<T, D>& x) const { const Mat<T>& a = unwrap(x); shim::need(a.n_rows == m->n_rows && a.n_cols == idx.n_elem, "cols(): assignment of a different size");
    for (uword j = 0; j < idx.n_elem; j++) { shim::need(idx.mem[j] < m->n_cols, "cols(): index out of bounds"); std::memcpy(m->colptr(idx.mem[j]), a.colptr(j), sizeof(T) * m->n_rows); } }
  void operator=(const ColsProxy& o) const { Mat<T> t = o.to_mat(); *this = t; }
};
template <class T> struct RowsProxy : public Base<T, RowsProxy<T>> {
  Mat<T>* m = nullptr; Mat<uword> idx;
  RowsProxy() = default; RowsProxy(const RowsProxy&) = default;
  Mat<T> to_mat() const { Mat<T> o(idx.n_elem, m->n_cols, fill::none); for (uword c = 0; c < m->n_cols; c++) for (uword j = 0; j < idx.n_elem; j++) { shim::need(idx.mem[j] < m->n_rows, "rows(): index out of bounds"); o.at(j, c) = m->at(idx.mem[j], c); } return o; }
  template <class D> void operator=(const Base<T, D>& x) const { const Mat<T>& a = unwrap(x); shim::need(a.n_rows == idx.n_elem && a.n_cols == m->n_cols, "rows(): assignment of a different size");
    for (uword c = 0; c < m->n_cols; c++) for (uword j = 0; j < idx.n_elem; j++) { shim::need(idx.mem[j] < m->n_rows, "rows(): index out of bounds"); m->at(idx.mem[j], c) = a.at(j, c); } }
  void operator=(const RowsProxy& o) const { Mat<T> t = o.to_mat(); *this = t; }
};
template <class T> struct ElemProxy : public Base<T, ElemProxy<T>> {
  Mat<T>* m; Mat<uword> idx;
  Mat<T> to_mat() const { Mat<T> o(idx.n_elem, 1, fill::none); for (uword j = 0; j < idx.n_elem; j++) o.mem[j] = (*m)(idx.mem[j]); return o; }
  template <class S> void fill(S v) const { for (uword j = 0; j < idx.n_elem; j++) (*m)(idx.mem[j]) = T(v); }
  void zeros() const { fill(0); }
};
template <class T> struct DiagView : public Base<T, DiagView<T>> {
  Mat<T>* m;
  uword len() const { return m->n_rows < m->n_cols ? m->n_rows : m->n_cols; }
  Mat<T> to_mat() const { Mat<T> o(len(), 1, fill::none); for (uword i = 0; i < len(); i++) o.mem[i] = m->at(i, i); return o; }
  template <class D> void operator=(const Base<T, D>& x) const { const Mat<T>& a = unwrap(x); shim::need(a.n_elem == len(), "diag(): assignment of a different size"); for (uword i = 0; i < len(); i++) m->at(i, i) = a.mem[i]; }
  template <class D> void operator+=(const Base<T, D>& x) const { const Mat<T>& a = unwrap(x); shim::need(a.n_elem == len(), "diag(): operand of a different size"); for (uword i = 0; i < len(); i++) m->at(i, i) += a.mem[i]; }
  template <class D> void operator-=(const Base<T, D>& x) const { const Mat<T>& a = unwrap(x); shim::need(a.n_elem == len(), "diag(): operand of a different size"); for (uword i = 0; i < len(); i++) m->at(i, i) -= a.mem[i]; }
};
template <class T> ColsProxy<T> Mat<T>::cols(const Mat<uword>& idx) const { ColsProxy<T> p; p.m = const_cast<Mat<T>*>(this); p.idx = idx; return p; }
template <class T> RowsProxy<T> Mat<T>::rows(const Mat<uword>& idx) const { RowsProxy<T> p; p.m = const_cast<Mat<T>*>(this); p.idx = idx; return p; }
template <class T> ElemProxy<T> Mat<T>::elem(const Mat<uword>& idx) const { ElemProxy<T> p; p.m = const_cast<Mat<T>*>(this); p.idx = idx; return p; }
template <class T> DiagView<T> Mat<T>::diag() const { DiagView<T> p; p.m = const_cast<Mat<T>*>(this); return p; }

// ------------------------------------------------------------------------------------------------------------------ element-wise expressions
#define ARMA_SHIM_ARITH(S) class = typename std::enable_if<std::is_arithmetic<S>::value>::type
#define ARMA_SHIM_BINOP(OP)                                                                                                                       \
  template <class T, class A, class B> Mat<T> operator OP(const Base<T, A>& x, const Base<T, B>& y) {                                              \
    const Mat<T>& a = unwrap(x); const Mat<T>& b = unwrap(y);                                                                                     \
    shim::need(a.n_rows == b.n_rows && a.n_cols == b.n_cols, "element-wise operation on objects of different size");                              \
    Mat<T> o(a.n_rows, a.n_cols, fill::none); for (uword i = 0; i < a.n_elem; i++) o.mem[i] = a.mem[i] OP b.mem[i]; return o; }                    \
  template <class T, class A, class S, ARMA_SHIM_ARITH(S)> Mat<T> operator OP(const Base<T, A>& x, S s) {                                         \
    const Mat<T>& a = unwrap(x); const T v = T(s); Mat<T> o(a.n_rows, a.n_cols, fill::none); for (uword i = 0; i < a.n_elem; i++) o.mem[i] = a.mem[i] OP v; return o; } \
  template <class T, class A, class S, ARMA_SHIM_ARITH(S)> Mat<T> operator OP(S s, const Base<T, A>& x) {                                         \
    const Mat<T>& a = unwrap(x); const T v = T(s); Mat<T> o(a.n_rows, a.n_cols, fill::none); for (uword i = 0; i < a.n_elem; i++) o.mem[i] = v OP a.mem[i]; return o; }
ARMA_SHIM_BINOP(+) ARMA_SHIM_BINOP(-) ARMA_SHIM_BINOP(/)
#undef ARMA_SHIM_BINOP
template <class T, class A, class B> Mat<T> operator%(const Base<T, A>& x, const Base<T, B>& y) {          // Schur product
  const Mat<T>& a = unwrap(x); const Mat<T>& b = unwrap(y);
  shim::need(a.n_rows == b.n_rows && a.n_cols == b.n_cols, "element-wise multiplication of objects of different size");
  Mat<T> o(a.n_rows, a.n_cols, fill::none); for (uword i = 0; i < a.n_elem; i++) o.mem[i] = a.mem[i] * b.mem[i]; return o;
}
template <class T, class A, class S, ARMA_SHIM_ARITH(S)> Mat<T> operator*(const Base<T, A>& x, S s) {
  const Mat<T>& a = unwrap(x); const T v = T(s); Mat<T> o(a.n_rows, a.n_cols, fill::none); for (uword i = 0; i < a.n_elem; i++) o.mem[i] = a.mem[i] * v; return o; }
template <class T, class A, class S, ARMA_SHIM_ARITH(S)> Mat<T> operator*(S s, const Base<T, A>& x) {
  const Mat<T>& a = unwrap(x); const T v = T(s); Mat<T> o(a.n_rows, a.n_cols, fill::none); for (uword i = 0; i < a.n_elem; i++) o.mem[i] = v * a.mem[i]; return o; }
template <class T, class A> Mat<T> operator-(const Base<T, A>& x) { const Mat<T>& a = unwrap(x); Mat<T> o(a.n_rows, a.n_cols, fill::none); for (uword i = 0; i < a.n_elem; i++) o.mem[i] = -a.mem[i]; return o; }
template <class T, class A, class S, ARMA_SHIM_ARITH(S)> Mat<uword> operator>(const Base<T, A>& x, S s) {
  const Mat<T>& a = unwrap(x); Mat<uword> o(a.n_rows, a.n_cols, fill::none); for (uword i = 0; i < a.n_elem; i++) o.mem[i] = a.mem[i] > T(s) ? 1 : 0; return o; }
template <class T, class A, class S, ARMA_SHIM_ARITH(S)> Mat<uword> operator<(const Base<T, A>& x, S s) {
  const Mat<T>& a = unwrap(x); Mat<uword> o(a.n_rows, a.n_cols, fill::none); for (uword i = 0; i < a.n_elem; i++) o.mem[i] = a.mem[i] < T(s) ? 1 : 0; return o; }

// dense * dense: every entry one sequential dot product (k ascending, from 0) -- or, when the test harness injected one (ref_set_sgemm), a real
// BLAS sgemm for the matrix-matrix case (M > 1: Y.t() * Z_corr, src/harmony.cpp:141,221 -- Armadillo's glue_times hands it to sgemm('T', 'N')
// on Y itself; the row-vector products of the seeding, utils.cpp:28, are gemv calls in Armadillo and stay this loop)
namespace shim {
typedef void (*sgemm_fn)(int order, int transa, int transb, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc);
inline sgemm_fn& sgemm_ptr() { static sgemm_fn f = nullptr; return f; }
}
template <class T, class A, class B> Mat<T> operator*(const Base<T, A>& x, const Base<T, B>& y) {
  const Mat<T>& a = unwrap(x); const Mat<T>& b = unwrap(y);
  shim::need(a.n_cols == b.n_rows, "matrix multiplication: incompatible dimensions");
  const uword M = a.n_rows, N = b.n_cols, Kd = a.n_cols;
  const Mat<T> at = a.t();                                   // rows of A contiguous: the same sums, friendlier strides
  Mat<T> o(M, N, fill::none);
  if constexpr (std::is_same<T, float>::value) {
    if (shim::sgemm_ptr() && M > 1 && Kd > 1) {
      shim::sgemm_ptr()(102 /*ColMajor*/, 112 /*Trans*/, 111 /*NoTrans*/, (int)M, (int)N, (int)Kd, 1.0f, at.mem, (int)Kd, b.mem, (int)Kd, 0.0f, o.mem, (int)M);
      return o;
    }
  }
  for (uword n = 0; n < N; n++) { const T* bc = b.colptr(n);
    for (uword m = 0; m < M; m++) { const T* ar = at.colptr(m); T s = T(0); for (uword k = 0; k < Kd; k++) s += ar[k] * bc[k]; o.mem[n * M + m] = s; } }
  return o;
}

#define ARMA_SHIM_MAP(NAME, EXPR)                                                                                                                 \
  template <class T, class A> Mat<T> NAME(const Base<T, A>& x) { const Mat<T>& a = unwrap(x); Mat<T> o(a.n_rows, a.n_cols, fill::none);           \
    for (uword i = 0; i < a.n_elem; i++) { const T v = a.mem[i]; o.mem[i] = (EXPR); } return o; }
ARMA_SHIM_MAP(exp, std::exp(v))
ARMA_SHIM_MAP(log, std::log(v))
ARMA_SHIM_MAP(sqrt, std::sqrt(v))
ARMA_SHIM_MAP(square, v * v)
ARMA_SHIM_MAP(abs, std::abs(v))
ARMA_SHIM_MAP(floor, std::floor(v))
// trunc_log: log with its argument clamped to the smallest / largest positive finite value
ARMA_SHIM_MAP(trunc_log, (!(v > T(0)) ? std::log(std::numeric_limits<T>::min()) : (std::isinf(v) ? std::log(std::numeric_limits<T>::max()) : std::log(v))))
#undef ARMA_SHIM_MAP
template <class T, class A, class S, ARMA_SHIM_ARITH(S)> Mat<T> pow(const Base<T, A>& x, S e) {
  const Mat<T>& a = unwrap(x); const T ev = T(e); Mat<T> o(a.n_rows, a.n_cols, fill::none); for (uword i = 0; i < a.n_elem; i++) o.mem[i] = std::pow(a.mem[i], ev); return o; }

namespace shim { inline int& norm_mode() { static int m = 0; return m; } }    // 1: op_norm on a real BLAS (../lapack_inv.hpp, blas1; ref_set_norm_mode)
// sum(X, 0): column sums (a row); sum(X, 1): row sums (a column), the columns added in order
template <class T, class A> Mat<T> sum(const Base<T, A>& x, uword dim = 0) {
  const Mat<T>& a = unwrap(x);
  shim::need(dim <= 1, "sum(): dimension must be 0 or 1");
  if (dim == 0) {
    Mat<T> o(1, a.n_cols);
    for (uword c = 0; c < a.n_cols; c++) {
      const T* p = a.colptr(c);
      if (shim::norm_mode()) {        // (ref_set_norm_mode: arrayops::accumulate -- two accumulators over the elements 0, 2, ... / 1, 3, ...)
        T s1 = T(0), s2 = T(0); uword r = 0;
        for (; r + 1 < a.n_rows; r += 2) { s1 += p[r]; s2 += p[r + 1]; }
        if (r < a.n_rows) s1 += p[r];
        o.mem[c] = s1 + s2;
      } else { T s = T(0); for (uword r = 0; r < a.n_rows; r++) s += p[r]; o.mem[c] = s; }
    }
    return o;
  }
  Mat<T> o(a.n_rows, 1);
  for (uword c = 0; c < a.n_cols; c++) { const T* p = a.colptr(c); for (uword r = 0; r < a.n_rows; r++) o.mem[r] += p[r]; }
  return o;
}
template <class T, class A> T accu(const Base<T, A>& x) {   // two accumulators over the even / odd elements
  const Mat<T>& a = unwrap(x); T s1 = T(0), s2 = T(0); uword i = 0;
  for (; i + 1 < a.n_elem; i += 2) { s1 += a.mem[i]; s2 += a.mem[i + 1]; }
  if (i < a.n_elem) s1 += a.mem[i];
  return s1 + s2;
}
template <class T, class A> T as_scalar(const Base<T, A>& x) { const Mat<T>& a = unwrap(x); shim::need(a.n_elem == 1, "as_scalar(): expression must evaluate to exactly one element"); return a.mem[0]; }
template <class S, ARMA_SHIM_ARITH(S)> S as_scalar(S s) { return s; }
template <class T> T shim_norm(const T* p, uword n, int pp) {
  if constexpr (std::is_same<T, float>::value) {
    if (shim::norm_mode() && blas1::ready() && (pp == 1 || pp == 2)) return pp == 1 ? blas1::norm1(p, (long long)n) : blas1::norm2(p, (long long)n);
  }
  T s = T(0);
  if (pp == 1) { for (uword i = 0; i < n; i++) s += std::abs(p[i]); return s; }
  shim::need(pp == 2, "norm(): only p = 1 and p = 2");
  for (uword i = 0; i < n; i++) s += p[i] * p[i];
  return std::sqrt(s);
}
template <class T, class A> T norm(const Base<T, A>& x, int p = 2) { const Mat<T>& a = unwrap(x); shim::need(a.n_rows == 1 || a.n_cols == 1 || a.n_elem == 0, "norm(): vectors only"); return shim_norm(a.mem, a.n_elem, p); }
template <class T, class A> Mat<T> normalise(const Base<T, A>& x, int p = 2, uword dim = 0) {
  Mat<T> o = unwrap(x);
  shim::need(dim == 0, "normalise(): columns only");
  for (uword c = 0; c < o.n_cols; c++) { T* q = o.colptr(c); T nv = shim_norm(q, o.n_rows, p); if (nv == T(0)) nv = T(1); for (uword r = 0; r < o.n_rows; r++) q[r] /= nv; }
  return o;
}
template <class T, class A> Mat<T> repmat(const Base<T, A>& x, uword nr, uword nc) {
  const Mat<T>& a = unwrap(x); Mat<T> o(a.n_rows * nr, a.n_cols * nc, fill::none);
  for (uword c = 0; c < o.n_cols; c++) for (uword r = 0; r < o.n_rows; r++) o.at(r, c) = a.at(r % a.n_rows, c % a.n_cols);
  return o;
}
template <class T, class A> Col<uword> find(const Base<T, A>& x) {
  const Mat<T>& a = unwrap(x); std::vector<uword> v; for (uword i = 0; i < a.n_elem; i++) if (a.mem[i] != T(0)) v.push_back(i); return Col<uword>(v); }
template <class T, class A> SizeMat size(const Base<T, A>& x) { const Mat<T>& a = unwrap(x); return SizeMat{a.n_rows, a.n_cols}; }

template <class MT> MT zeros(uword r, uword c) { MT m(r, c); m.zeros(); return m; }
template <class MT> MT zeros(uword n) { MT m(n); m.zeros(); return m; }
template <class MT> MT ones(uword r, uword c) { MT m(r, c); m.ones(); return m; }
template <class MT> MT ones(uword n) { MT m(n); m.ones(); return m; }
template <class VT> VT linspace(double a, double b, uword n) {
  VT o(n);
  if (n == 1) { o.mem[0] = (typename VT::elem_type)b; return o; }
  const double delta = (b - a) / double(n - 1);
  for (uword i = 0; i < n; i++) o.mem[i] = (typename VT::elem_type)(a + double(i) * delta);
  if (n > 1) o.mem[n - 1] = (typename VT::elem_type)b;
  return o;
}
// shuffle(vector): an injected permutation when one is queued, otherwise R's stream -- one randi per element in order, then a sort of the
// (value, index) packets by value
template <class T, class A> Col<T> shuffle(const Base<T, A>& x) {
  const Mat<T>& a = unwrap(x);
  shim::need(a.n_rows == 1 || a.n_cols == 1 || a.n_elem == 0, "shuffle(): vectors only");
  Col<T> o(a.n_elem);
  if (!shim::injected_orders().empty()) {
    std::vector<uword> ord = std::move(shim::injected_orders().front()); shim::injected_orders().pop_front();
    shim::need(ord.size() == a.n_elem, "injected permutation of the wrong length");
    for (uword i = 0; i < a.n_elem; i++) o.mem[i] = a.mem[ord[i]];
    return o;
  }
  struct Pk { int val; uword index; };
  std::vector<Pk> pk(a.n_elem);
  for (uword i = 0; i < a.n_elem; i++) { pk[i].val = (int)shim::rng().runif(0.0, (double)RAND_MAX); pk[i].index = i; }
  std::sort(pk.begin(), pk.end(), [](const Pk& p, const Pk& q) { return p.val < q.val; });
  for (uword i = 0; i < a.n_elem; i++) o.mem[i] = a.mem[pk[i].index];
  return o;
}
// inv: unblocked LU with partial pivoting on [A | I], row by row (the restatement the oracle documents for arma::inv) -- or, when the
// test harness asked for it (shim::inv_mode(), ref_set_inv_mode) and injected one, through a real LAPACK in Armadillo's own call
// sequences (../lapack_inv.hpp: 1 = sgetrf + sgetri, 2 = spotrf + spotri + mirror; fp32 only)
namespace shim { inline int& inv_mode() { static int m = 0; return m; } }
template <class T, class A> Mat<T> inv(const Base<T, A>& x) {
  Mat<T> a = unwrap(x);
  shim::need(a.n_rows == a.n_cols, "inv(): matrix must be square");
  if constexpr (std::is_same<T, float>::value) {
    if (shim::inv_mode() && lapack_inv::ready()) {
      if (!lapack_inv::inv(a.mem, (int)a.n_rows, shim::inv_mode())) throw std::runtime_error("inv(): matrix is singular");
      return a;
    }
  }
  const int n = (int)a.n_rows; Mat<T> b(n, n); for (int i = 0; i < n; i++) b.at(i, i) = T(1);
  T* Am = a.mem; T* Bm = b.mem;
  for (int c = 0; c < n; c++) {
    int p = c; T best = std::fabs(Am[c * n + c]);
    for (int r = c + 1; r < n; r++) if (std::fabs(Am[c * n + r]) > best) { best = std::fabs(Am[c * n + r]); p = r; }
    if (best == T(0)) throw std::runtime_error("inv(): matrix is singular");
    if (p != c) { for (int j = 0; j < n; j++) { std::swap(Am[j * n + c], Am[j * n + p]); std::swap(Bm[j * n + c], Bm[j * n + p]); } }
    const T piv = T(1) / Am[c * n + c];
    for (int r = c + 1; r < n; r++) {
      const T f = Am[c * n + r] * piv; if (f == T(0)) continue;
      Am[c * n + r] = f;
      for (int j = c + 1; j < n; j++) Am[j * n + r] -= f * Am[j * n + c];
      for (int j = 0; j < n; j++) Bm[j * n + r] -= f * Bm[j * n + c];
    }
  }
  for (int j = 0; j < n; j++)
    for (int r = n - 1; r >= 0; r--) { T s = Bm[j * n + r]; for (int c = r + 1; c < n; c++) s -= Am[c * n + r] * Bm[j * n + c]; Bm[j * n + r] = s / Am[r * n + r]; }
  return b;
}
// kmeans(means, data, K, keep_existing, n_iter, print): n_iter Lloyd iterations from the given means -- nearest mean by squared Euclidean
// distance (|y|^2 - 2 x.y, first minimum), new mean = the members' average (fp64 sums), an empty cluster keeps its mean
template <class T> bool kmeans(Mat<T>& means, const Mat<T>& X, uword K, kmeans_seed_mode mode, uword n_iter, bool) {
  shim::need(mode == keep_existing && means.n_cols == K && means.n_rows == X.n_rows, "kmeans(): only keep_existing with K given means");
  const uword d = X.n_rows, N = X.n_cols;
  std::vector<double> sums(d * K); std::vector<long long> cnt(K); std::vector<T> ynorm(K);
  for (uword it = 0; it < n_iter; it++) {
    std::fill(sums.begin(), sums.end(), 0.0); std::fill(cnt.begin(), cnt.end(), 0);
    for (uword k = 0; k < K; k++) { T s = T(0); const T* y = means.colptr(k); for (uword j = 0; j < d; j++) s += y[j] * y[j]; ynorm[k] = s; }
    for (uword n = 0; n < N; n++) {
      const T* xx = X.colptr(n); uword bk = 0; T bs = std::numeric_limits<T>::max();
      for (uword k = 0; k < K; k++) { T dot = T(0); const T* y = means.colptr(k); for (uword j = 0; j < d; j++) dot += y[j] * xx[j]; const T sc = ynorm[k] - T(2) * dot; if (sc < bs) { bs = sc; bk = k; } }
      cnt[bk]++;
      for (uword j = 0; j < d; j++) sums[bk * d + j] += xx[j];
    }
    for (uword k = 0; k < K; k++) if (cnt[k] > 0) for (uword j = 0; j < d; j++) means.at(j, k) = (T)(sums[k * d + j] / (double)cnt[k]);
  }
  return true;
}

// ------------------------------------------------------------------------------------------------------------------ sparse (CSC)
template <class T> struct SpDiag;
template <class T> class SpMat {
 public:
  typedef T elem_type;
  uword n_rows = 0, n_cols = 0, n_nonzero = 0, n_elem = 0;
  std::vector<T> vv; std::vector<uword> rr, cc;            // values, row of each value, start of each column (n_cols + 1)
  const T* values = nullptr; const uword* row_indices = nullptr; const uword* col_ptrs = nullptr;

  SpMat() { cc.assign(1, 0); sync(); }
  SpMat(uword r, uword c) : n_rows(r), n_cols(c) { cc.assign(c + 1, 0); sync(); }
  SpMat(const SpMat& o) : n_rows(o.n_rows), n_cols(o.n_cols), vv(o.vv), rr(o.rr), cc(o.cc) { sync(); }
  SpMat(SpMat&& o) noexcept : n_rows(o.n_rows), n_cols(o.n_cols), vv(std::move(o.vv)), rr(std::move(o.rr)), cc(std::move(o.cc)) { sync(); }
  // (row indices, column pointers, values, n_rows, n_cols)
  template <class D> SpMat(const Mat<uword>& rowind, const Mat<uword>& colptr, const Base<T, D>& vals, uword r, uword c) : n_rows(r), n_cols(c) {
    const Mat<T>& v = unwrap(vals);
    shim::need(colptr.n_elem == c + 1 && rowind.n_elem == v.n_elem && colptr.mem[c] == v.n_elem, "SpMat(rowind, colptr, values): inconsistent sizes");
    vv.assign(v.mem, v.mem + v.n_elem); rr.assign(rowind.mem, rowind.mem + rowind.n_elem); cc.assign(colptr.mem, colptr.mem + colptr.n_elem);
    for (uword j = 0; j < c; j++) for (uword q = cc[j]; q < cc[j + 1]; q++) shim::need(rr[q] < r && (q == cc[j] || rr[q - 1] < rr[q]), "SpMat(rowind, colptr, values): row indices out of bounds or not ascending");
    sync();
  }
  SpMat(const Mat<T>& m) : n_rows(m.n_rows), n_cols(m.n_cols) { from_dense(m); }
  SpMat& operator=(const SpMat& o) { if (this != &o) { n_rows = o.n_rows; n_cols = o.n_cols; vv = o.vv; rr = o.rr; cc = o.cc; sync(); } return *this; }
  SpMat& operator=(SpMat&& o) noexcept { if (this != &o) { n_rows = o.n_rows; n_cols = o.n_cols; vv = std::move(o.vv); rr = std::move(o.rr); cc = std::move(o.cc); sync(); } return *this; }
  SpMat& operator=(const Mat<T>& m) { n_rows = m.n_rows; n_cols = m.n_cols; from_dense(m); return *this; }
  void zeros() { vv.clear(); rr.clear(); cc.assign(n_cols + 1, 0); sync(); }
  void sync() { n_nonzero = vv.size(); n_elem = n_rows * n_cols; values = vv.data(); row_indices = rr.data(); col_ptrs = cc.data(); }

  struct const_iterator {
    const SpMat* m; uword pos, c;
    void settle() { while (c < m->n_cols && m->cc[c + 1] <= pos) c++; }
    uword row() const { return m->rr[pos]; }
    uword col() const { return c; }
    T operator*() const { return m->vv[pos]; }
    const_iterator& operator++() { pos++; settle(); return *this; }
    bool operator!=(const const_iterator& o) const { return pos != o.pos; }
    bool operator==(const const_iterator& o) const { return pos == o.pos; }
  };
  const_iterator begin() const { const_iterator it{this, 0, 0}; it.settle(); return it; }
  const_iterator end() const { return const_iterator{this, (uword)vv.size(), n_cols}; }

  SpMat t() const {
    SpMat o(n_cols, n_rows); o.vv.resize(vv.size()); o.rr.resize(rr.size());
    for (uword q = 0; q < rr.size(); q++) o.cc[rr[q] + 1]++;
    for (uword j = 0; j < n_rows; j++) o.cc[j + 1] += o.cc[j];
    std::vector<uword> at(o.cc.begin(), o.cc.end() - 1);
    for (uword j = 0; j < n_cols; j++) for (uword q = cc[j]; q < cc[j + 1]; q++) { const uword w = at[rr[q]]++; o.rr[w] = j; o.vv[w] = vv[q]; }
    o.sync(); return o;
  }
  SpMat cols(const Mat<uword>& idx) const {
    SpMat o(n_rows, idx.n_elem);
    for (uword j = 0; j < idx.n_elem; j++) { const uword s = idx.mem[j]; shim::need(s < n_cols, "cols(): index out of bounds"); for (uword q = cc[s]; q < cc[s + 1]; q++) { o.rr.push_back(rr[q]); o.vv.push_back(vv[q]); } o.cc[j + 1] = o.rr.size(); }
    o.sync(); return o;
  }
  SpMat submat(uword r0, uword c0, uword r1, uword c1) const {
    shim::need(r0 <= r1 && c0 <= c1 && r1 < n_rows && c1 < n_cols, "submat: indices out of bounds or incorrectly used");
    SpMat o(r1 - r0 + 1, c1 - c0 + 1);
    for (uword j = c0; j <= c1; j++) { for (uword q = cc[j]; q < cc[j + 1]; q++) if (rr[q] >= r0 && rr[q] <= r1) { o.rr.push_back(rr[q] - r0); o.vv.push_back(vv[q]); } o.cc[j - c0 + 1] = o.rr.size(); }
    o.sync(); return o;
  }
  SpDiag<T> diag() { return SpDiag<T>{this}; }
  SpDiag<T> diag() const { return SpDiag<T>{const_cast<SpMat*>(this)}; }
  T get(uword r, uword c) const { for (uword q = cc[c]; q < cc[c + 1]; q++) if (rr[q] == r) return vv[q]; return T(0); }

 private:
  void from_dense(const Mat<T>& m) {
    vv.clear(); rr.clear(); cc.assign(n_cols + 1, 0);
    for (uword c = 0; c < n_cols; c++) { for (uword r = 0; r < n_rows; r++) if (m.at(r, c) != T(0)) { rr.push_back(r); vv.push_back(m.at(r, c)); } cc[c + 1] = rr.size(); }
    sync();
  }
};
template <class T> Mat<T>::Mat(const SpMat<T>& s) { init(s.n_rows, s.n_cols); zeros(); for (uword c = 0; c < s.n_cols; c++) for (uword q = s.cc[c]; q < s.cc[c + 1]; q++) mem[c * n_rows + s.rr[q]] = s.vv[q]; }

template <class T> struct SpDiag : public Base<T, SpDiag<T>> {
  SpMat<T>* m;
  explicit SpDiag(SpMat<T>* mm) : m(mm) {}
  uword len() const { return m->n_rows < m->n_cols ? m->n_rows : m->n_cols; }
  Mat<T> to_mat() const { Mat<T> o(len(), 1); for (uword i = 0; i < len(); i++) o.mem[i] = m->get(i, i); return o; }
  // the diagonal takes the given values (zeros are not stored); everything off the diagonal stays
  template <class D> void operator=(const Base<T, D>& x) const {
    const Mat<T>& a = unwrap(x);
    shim::need(a.n_elem == len(), "diag(): assignment of a different size");
    SpMat<T> o(m->n_rows, m->n_cols);
    for (uword c = 0; c < m->n_cols; c++) {
      bool placed = (c >= len());
      for (uword q = m->cc[c]; q < m->cc[c + 1]; q++) {
        const uword r = m->rr[q];
        if (r == c) continue;
        if (!placed && r > c) { if (a.mem[c] != T(0)) { o.rr.push_back(c); o.vv.push_back(a.mem[c]); } placed = true; }
        o.rr.push_back(r); o.vv.push_back(m->vv[q]);
      }
      if (!placed && a.mem[c] != T(0)) { o.rr.push_back(c); o.vv.push_back(a.mem[c]); }
      o.cc[c + 1] = o.rr.size();
    }
    o.sync(); *m = std::move(o);
  }
};

// sum(sparse, dim) as a dense vector (the reference converts it at once: VECTYPE(sum(Phi, 1)), sum(Phi, 1) / N)
template <class T> Mat<T> sum(const SpMat<T>& s, uword dim = 0) {
  shim::need(dim <= 1, "sum(): dimension must be 0 or 1");
  if (dim == 0) { Mat<T> o(1, s.n_cols); for (uword c = 0; c < s.n_cols; c++) { T a = T(0); for (uword q = s.cc[c]; q < s.cc[c + 1]; q++) a += s.vv[q]; o.mem[c] = a; } return o; }
  Mat<T> o(s.n_rows, 1); for (uword q = 0; q < s.rr.size(); q++) o.mem[s.rr[q]] += s.vv[q]; return o;
}
template <class T, class S, ARMA_SHIM_ARITH(S)> Mat<T> operator+(const SpMat<T>& s, S v) { Mat<T> o(s); o += v; return o; }
template <class T> SpMat<T> operator+(const SpMat<T>& a, const SpMat<T>& b) {
  shim::need(a.n_rows == b.n_rows && a.n_cols == b.n_cols, "addition of sparse matrices of different size");
  SpMat<T> o(a.n_rows, a.n_cols);
  for (uword c = 0; c < a.n_cols; c++) {
    uword p = a.cc[c], q = b.cc[c];
    while (p < a.cc[c + 1] || q < b.cc[c + 1]) {
      const uword ra = p < a.cc[c + 1] ? a.rr[p] : ~uword(0), rb = q < b.cc[c + 1] ? b.rr[q] : ~uword(0);
      if (ra == rb) { o.rr.push_back(ra); o.vv.push_back(a.vv[p] + b.vv[q]); p++; q++; }
      else if (ra < rb) { o.rr.push_back(ra); o.vv.push_back(a.vv[p]); p++; }
      else { o.rr.push_back(rb); o.vv.push_back(b.vv[q]); q++; }
    }
    o.cc[c + 1] = o.rr.size();
  }
  o.sync(); return o;
}
// sparse * sparse, Gustavson: column j of the result accumulates A.col(i) * B(i, j) over the non-zeros (i ascending) of B's column j
template <class T> SpMat<T> operator*(const SpMat<T>& a, const SpMat<T>& b) {
  shim::need(a.n_cols == b.n_rows, "matrix multiplication: incompatible dimensions");
  SpMat<T> o(a.n_rows, b.n_cols);
  std::vector<T> acc(a.n_rows, T(0)); std::vector<char> hit(a.n_rows, 0); std::vector<uword> rows;
  for (uword j = 0; j < b.n_cols; j++) {
    rows.clear();
    for (uword q = b.cc[j]; q < b.cc[j + 1]; q++) { const uword i = b.rr[q]; const T bv = b.vv[q];
      for (uword p = a.cc[i]; p < a.cc[i + 1]; p++) { const uword r = a.rr[p]; if (!hit[r]) { hit[r] = 1; rows.push_back(r); } acc[r] += a.vv[p] * bv; } }
    std::sort(rows.begin(), rows.end());
    for (uword r : rows) { if (acc[r] != T(0)) { o.rr.push_back(r); o.vv.push_back(acc[r]); } acc[r] = T(0); hit[r] = 0; }
    o.cc[j + 1] = o.rr.size();
  }
  o.sync(); return o;
}
// dense * sparse: out = 0, then one rounded product per non-zero of the sparse operand, in its storage order
template <class T, class A> Mat<T> operator*(const Base<T, A>& x, const SpMat<T>& s) {
  const Mat<T>& a = unwrap(x);
  shim::need(a.n_cols == s.n_rows, "matrix multiplication: incompatible dimensions");
  Mat<T> o(a.n_rows, s.n_cols);
  for (uword c = 0; c < s.n_cols; c++) { T* oc = o.colptr(c);
    for (uword q = s.cc[c]; q < s.cc[c + 1]; q++) { const T* ac = a.colptr(s.rr[q]); const T v = s.vv[q]; for (uword r = 0; r < a.n_rows; r++) oc[r] += ac[r] * v; } }
  return o;
}
template <class T> SpMat<T> join_cols(const SpMat<T>& a, const SpMat<T>& b) {    // a on top of b
  shim::need(a.n_cols == b.n_cols, "join_cols(): number of columns must be the same");
  SpMat<T> o(a.n_rows + b.n_rows, a.n_cols);
  for (uword c = 0; c < a.n_cols; c++) {
    for (uword q = a.cc[c]; q < a.cc[c + 1]; q++) { o.rr.push_back(a.rr[q]); o.vv.push_back(a.vv[q]); }
    for (uword q = b.cc[c]; q < b.cc[c + 1]; q++) { o.rr.push_back(b.rr[q] + a.n_rows); o.vv.push_back(b.vv[q]); }
    o.cc[c + 1] = o.rr.size();
  }
  o.sync(); return o;
}

// ------------------------------------------------------------------------------------------------------------------ conv_to
template <class X> struct shim_is_sparse : std::false_type {};
template <class T> struct shim_is_sparse<SpMat<T>> : std::true_type {};
template <class Out> struct conv_to {
  typedef typename Out::elem_type OT;
  template <class T2, class D> static Out from(const Base<T2, D>& x) {
    const Mat<T2>& a = unwrap(x); Out o(a.n_rows, a.n_cols); for (uword i = 0; i < a.n_elem; i++) o.mem[i] = (OT)a.mem[i]; return o; }
  template <class T2> static Out from(const SpMat<T2>& s) {
    static_assert(shim_is_sparse<Out>::value, "conv_to: sparse to sparse only");
    Out o(s.n_rows, s.n_cols); o.vv.resize(s.vv.size()); for (uword q = 0; q < s.vv.size(); q++) o.vv[q] = (OT)s.vv[q]; o.rr = s.rr; o.cc = s.cc; o.sync(); return o; }
  template <class T2> static Out from(const std::vector<T2>& v) { Out o(v.size(), 1); for (uword i = 0; i < v.size(); i++) o.mem[i] = (OT)v[i]; return o; }
};
#undef ARMA_SHIM_ARITH
}  // namespace arma
