// Error reporting + ABI version for libwgs_hip.so.
#include "wgs_common.h"
#include "../../include/wgs.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void wgs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
const char* wgs_last_error(void) { return g_err; }
int wgs_abi_version(void) { return 1; }
}
