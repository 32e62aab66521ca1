// api.cpp — library identification and thread-local error text for libsph3d.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/sph3d.h"

namespace sph3d {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace sph3d

extern "C" int sph3d_abi_version(void) { return 1; }
extern "C" const char* sph3d_last_error(void) { return sph3d::g_err; }
extern "C" const char* sph3d_build_info(void)
{
    return "libsph3d gfx950 (" __VERSION__ ") -ffp-contract=off -munsafe-fp-atomics";
}
