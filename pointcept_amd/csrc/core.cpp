// core.cpp -- version string and thread-local error message of libptcore.so.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/ptcore.h"

static thread_local char g_err[512] = "";

void ptc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ptc_last_error(void) { return g_err; }
#ifndef PTC_ABI_HASH
#define PTC_ABI_HASH "unknown"
#endif
// "abi <crc32 of include/ptcore.h at build time>": pointcept_amd/_lib.py compares it with the header it binds
extern "C" const char* ptc_version(void) { return "ptcore 0.2 (gfx950, abi " PTC_ABI_HASH ")"; }
