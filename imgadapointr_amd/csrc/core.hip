// libptops: ABI version and the thread-local error message.
#include <stdarg.h>
#include <stdio.h>

#include "ptops_common.h"

namespace ptops {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ptops

PTOPS_API int ptops_abi_version(void) { return 1; }
PTOPS_API const char* ptops_last_error(void) { return ptops::g_err; }
