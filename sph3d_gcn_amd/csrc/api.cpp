// api.cpp — library identification and thread-local error text for libsph3d.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <map>
#include <tuple>
#include <mutex>
#include <utility>
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

// A library-owned device buffer per (device, stream) for the CONVENIENCE entry points whose reference signature has no workspace
// argument (sph3d_build_sphere_neighbor, sph3d_build_sphere_graph[_ocml]: the neighbour search's cell grid, nngrid.hip).  The
// `_ws` twins of those entry points take the buffer from the caller and never come here.  Work on one stream is ordered, so the
// buffer is reused from call to call; streams and devices do not share one (the null stream of two devices are two keys).
// Grown on demand: the old buffer is released with hipFree, which waits for the device.  Under stream capture nothing may
// be allocated: the call then runs without the grid (nullptr), which is always valid.  sph3d_release_stream_scratch drops
// the entry of a stream that is about to be destroyed (a recycled handle would otherwise inherit it — harmless for
// correctness, the buffer holds no state between calls, but it would never be freed).
namespace {
struct Scratch { void* p; size_t bytes; };
std::mutex g_scratch_mu;
std::map<std::tuple<int, hipStream_t, int>, Scratch> g_scratch;
}  // namespace

// kind 0: plain scratch (no state between calls).  kind 1: zero-initialised when (re)allocated, and every kernel that uses it
// leaves its first 16 KB zero again (the arrival counters of the in-kernel split-K exchange, gemm.hip): state that survives
// from call to call on the stream, so it has its own buffer.
void* stream_scratch(hipStream_t stream, size_t bytes, int kind)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    Scratch& s = g_scratch[std::make_tuple(dev, stream, kind)];
    if (s.bytes >= bytes && s.p != nullptr) return s.p;
    if (cap != hipStreamCaptureStatusNone) return nullptr;      // no hipMalloc / hipFree while a graph is being captured
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
    if (kind == 1 && hipMemset(p, 0, want) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return nullptr;
    }
    s.p = p;
    s.bytes = want;
    return p;
}

// -> number of buffers released.  all_streams: every entry of the current device
int release_scratch(hipStream_t stream, bool all_streams)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    int n = 0;
    for (auto it = g_scratch.begin(); it != g_scratch.end();) {
        if (std::get<0>(it->first) == dev && (all_streams || std::get<1>(it->first) == stream)) {
            if (it->second.p != nullptr) {
                (void)hipFree(it->second.p);
                n++;
            }
            it = g_scratch.erase(it);
        } else {
            ++it;
        }
    }
    return n;
}
}  // namespace sph3d

extern "C" int sph3d_abi_version(void) { return SPH3D_ABI_VERSION; }
extern "C" const char* sph3d_last_error(void) { return sph3d::g_err; }
extern "C" int sph3d_release_stream_scratch(sph3d_stream_t stream) { return sph3d::release_scratch((hipStream_t)stream, false); }
extern "C" int sph3d_release_all_scratch(void) { return sph3d::release_scratch(nullptr, true); }
extern "C" const char* sph3d_build_info(void)
{
    return "libsph3d gfx950 (" __VERSION__ ") -ffp-contract=off -munsafe-fp-atomics";
}
