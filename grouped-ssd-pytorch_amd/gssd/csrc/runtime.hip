// Error string + ABI identification for libgssd_hip.so.
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void gssd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int gssd_abi_version(void) { return 8; }
extern "C" int gssd_conv_desc_size(void) { return (int)sizeof(gssd_conv_desc); }
extern "C" const char* gssd_last_error(void) { return g_err; }
extern "C" const char* gssd_build_arch(void) { return "gfx950"; }
