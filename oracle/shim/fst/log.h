// Project-owned stand-in for <fst/log.h>: LOG(FATAL) prints and aborts, which
// is what the reference's VALID_CHECK macros rely on (decoder_utils.h:17-29).
// TEST INFRASTRUCTURE ONLY (see fst/fstlib.h in this directory).
#pragma once
#include <cstdlib>
#include <iostream>

namespace shim_log {
struct FatalStream {
  ~FatalStream() { std::cerr << std::endl; std::abort(); }
  template <class T> FatalStream &operator<<(const T &x) { std::cerr << x; return *this; }
};
}  // namespace shim_log
#define LOG(severity) ::shim_log::FatalStream()
