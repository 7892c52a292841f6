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
#ifdef MUMPY_TUNING
#include <stdlib.h>
int tune_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
const char* tune_str(const char* name) { return getenv(name); }
#endif
}  // namespace mumpy

// 1 when this library was built with the tuning hooks (reads MUMPY_* environment variables), 0 for the shipped build
extern "C" int mumpy_tuning_build(void) {
#ifdef MUMPY_TUNING
    return 1;
#else
    return 0;
#endif
}

extern "C" int mumpy_abi_version(void) { return MUMPY_ABI_VERSION; }
extern "C" const char* mumpy_last_error(void) { return mumpy::g_err; }
