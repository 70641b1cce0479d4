// TEST INFRASTRUCTURE ONLY -- stands where <RcppArmadillo.h> would: see arma_min.hpp (what it is, what it is not).
#pragma once
#include "arma_min.hpp"
#include "Rcpp.h"
