// TEST INFRASTRUCTURE: the elementwise kernel files compiled for the host (see hip/hip_runtime.h).  Exports the same extern "C"
// entry points as libptcore.so for those files; pointers are host pointers.
#include <stdarg.h>
#include <stdio.h>
static char g_err[512];
void ptc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* emu_last_error() { return g_err; }
#include "../../pointcept_amd/csrc/rope.hip"
