// Error text of the C ABI (kge_last_error): a thread-local message every failing entry point leaves behind.  A translation unit of its
// own -- host-only, no HIP -- so that the sanitizer build of the host objects (`make asan`: sampler.cpp, frontend.cpp, mtjump.cpp) links
// the same code as libkge_hip.so.
#include <cstdarg>
#include <cstdio>

#include "kge_host.hpp"

namespace kge {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace kge

extern "C" const char* kge_last_error(void) { return kge::err_buf(); }
