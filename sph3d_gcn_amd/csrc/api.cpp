// api.cpp — library identification and thread-local error text for libsph3d.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <map>
#include <mutex>
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

// A library-owned device buffer per stream for entry points whose reference signature has no workspace argument
// (the neighbour search's cell grid, nngrid.hip).  Work on one stream is ordered, so the buffer is reused from call to call;
// streams do not share it.  Grown on demand: the old buffer is released with hipFree, which waits for the device.
namespace {
struct Scratch { void* p; size_t bytes; };
std::mutex g_scratch_mu;
std::map<hipStream_t, Scratch> g_scratch;
}  // namespace

void* stream_scratch(hipStream_t stream, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    Scratch& s = g_scratch[stream];
    if (s.bytes >= bytes && s.p != nullptr) return s.p;
    if (s.p != nullptr) {
        (void)hipFree(s.p);
        s.p = nullptr;
        s.bytes = 0;
    }
    const size_t want = bytes + bytes / 4;            // head-room: the plans' levels differ by small factors
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    s.p = p;
    s.bytes = want;
    return p;
}
}  // namespace sph3d

extern "C" int sph3d_abi_version(void) { return 1; }
extern "C" const char* sph3d_last_error(void) { return sph3d::g_err; }
extern "C" const char* sph3d_build_info(void)
{
    return "libsph3d gfx950 (" __VERSION__ ") -ffp-contract=off -munsafe-fp-atomics";
}
