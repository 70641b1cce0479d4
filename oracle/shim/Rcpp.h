// TEST INFRASTRUCTURE ONLY -- stands where <Rcpp.h> would, for the handful of names the reference's engine sources use
// (/root/reference/src/harmony.cpp, utils.cpp, timer.cpp): stop / warning, the two streams, and the module-registration macros, which
// expand to nothing that runs (oracle/ref_driver.cpp drives the class directly).  See arma_min.hpp.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <map>
#include <numeric>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace Rcpp {
static std::ostream& Rcout = std::cout;
static std::ostream& Rcerr = std::cerr;
inline std::vector<std::string>& shim_warnings() { static std::vector<std::string> w; return w; }
[[noreturn]] inline void stop(const std::string& msg) { throw std::runtime_error(msg); }
inline void warning(const std::string& msg) { shim_warnings().push_back(msg); }
// RCPP_MODULE(name) { class_<T>("T").constructor().field(...).method(...); } -- accepted, never called
template <class C> struct class_ {
  explicit class_(const char*) {}
  class_& constructor() { return *this; }
  template <class F> class_& field(const char*, F) { return *this; }
  template <class M> class_& method(const char*, M) { return *this; }
};
}  // namespace Rcpp
#define RCPP_EXPOSED_CLASS(NAME)
#define RCPP_MODULE(NAME) [[maybe_unused]] static void rcpp_module_##NAME##_never_called()
