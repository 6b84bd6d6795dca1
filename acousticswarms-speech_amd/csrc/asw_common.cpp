#include "asw_common.h"

namespace asw {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace asw

extern "C" const char* asw_last_error(void) { return asw::err_buf(); }
extern "C" int asw_abi_version(void) { return 1; }
