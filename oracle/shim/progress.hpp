// TEST INFRASTRUCTURE ONLY -- stands where RcppProgress's <progress.hpp> would: a progress bar that shows nothing and is never aborted.
#pragma once
class Progress {
 public:
  Progress(unsigned long, bool) {}
  void increment(unsigned long = 1) {}
  static bool check_abort() { return false; }
};
