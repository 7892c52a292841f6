// abi.cpp — version / error plumbing of the C ABI (include/mumpy_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

namespace mumpy {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mumpy

extern "C" int mumpy_abi_version(void) { return MUMPY_ABI_VERSION; }
extern "C" const char* mumpy_last_error(void) { return mumpy::g_err; }
