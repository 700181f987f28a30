// orbx_rt.h — thin runtime layer: HIP (product) or plain libc (tests/emu build of the same sources).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include "orbx_platform.h"

namespace orbx { namespace rt {

// what the library holds right now, process-wide: device allocations, page-locked allocations, streams, events (orbx_debug_live_resources:
// the lifetime tests create and destroy every kind of handle and expect these to return to where they started)
struct Live { std::atomic<long long> dev{0}, pinned{0}, streams{0}, events{0}; };
inline Live& live() { static Live l; return l; }

#ifdef ORBX_EMU
typedef int stream_t;
struct event_s { double t; };
typedef event_s* event_t;
inline double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
inline int set_device(int) { return 0; }
inline int set_host_wait(int, int) { return 0; }
inline bool memory_is_host() { return true; }
// ORBX_EMU_DEVICES: the tests let the emulator report several "GPUs" (all of them host memory) to exercise the device plumbing of multi-GPU hosts
inline int device_count() { const char* e = getenv("ORBX_EMU_DEVICES"); const int n = e ? atoi(e) : 1; return n > 0 ? n : 1; }
inline void* dmalloc(size_t n) { void* p = calloc(n ? n : 1, 1); if (p) live().dev++; return p; }
inline void dfree(void* p) { if (p) { live().dev--; free(p); } }
inline void* hmalloc(size_t n) { void* p = calloc(n ? n : 1, 1); if (p) live().pinned++; return p; }
inline void hfree(void* p) { if (p) { live().pinned--; free(p); } }
inline int copy_h2d(void* d, const void* s, size_t n, stream_t) { memcpy(d, s, n); return 0; }
inline int copy_d2h(void* d, const void* s, size_t n, stream_t) { memcpy(d, s, n); return 0; }
inline int copy_d2d(void* d, const void* s, size_t n, stream_t) { memmove(d, s, n); return 0; }
inline int copy2d(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int /*kind*/, stream_t) {
    for (size_t y = 0; y < h; y++) memcpy((char*)d + y * dp, (const char*)s + y * sp, w);
    return 0;
}
inline int memset_async(void* d, int v, size_t n, stream_t) { memset(d, v, n); return 0; }
inline int stream_create(stream_t* s) { *s = 0; live().streams++; return 0; }
inline void stream_destroy(stream_t) { live().streams--; }
inline int stream_sync(stream_t) { return 0; }
inline int event_create(event_t* e) { *e = new event_s{0}; live().events++; return 0; }
inline void event_destroy(event_t e) { if (e) { live().events--; delete e; } }
inline int event_record(event_t e, stream_t) { e->t = now_ms(); return 0; }
inline int stream_wait_event(stream_t, event_t) { return 0; }
inline int event_sync(event_t) { return 0; }
inline float event_elapsed_ms(event_t a, event_t b) { return (float)(b->t - a->t); }
inline const char* last_error() { return "emu"; }
inline int check_launch() { return 0; }
// ORBX_EMU_LDS_LIMIT: the tests give the emulator the device's LDS size (163840 on gfx950) to reach the capacity refusals on the CPU
inline size_t lds_limit(int) { const char* e = getenv("ORBX_EMU_LDS_LIMIT"); const long long v = e ? atoll(e) : 0; return v > 0 ? (size_t)v : (size_t)1 << 30; }
#else
typedef hipStream_t stream_t;
typedef hipEvent_t event_t;
// A failed HIP call leaves a sticky per-thread "last error" in the runtime.  When the process shares that runtime with another
// library (PyTorch in bench.py --gpus N), the other library's next hipGetLastError() check would trip over it, so every wrapper
// consumes the error it has seen and keeps its own copy for last_error().
inline hipError_t& last_code() { static thread_local hipError_t e = hipSuccess; return e; }
inline int hip_ok(hipError_t e) { if (e == hipSuccess) return 0; last_code() = e; (void)hipGetLastError(); return -1; }
#define ORBX_HIP_OK(x) ::orbx::rt::hip_ok(x)
inline int set_device(int d) { return ORBX_HIP_OK(hipSetDevice(d)); }
// how a host thread waits for the device in hipStreamSynchronize / hipEventSynchronize: 0 = the runtime's choice (it spins: lowest latency, one core per waiting
// thread), 1 = blocking (sleeps on the completion interrupt: ~10 us later, no core), 2 = spin, 3 = yield
inline int set_host_wait(int d, int mode) {
    if (ORBX_HIP_OK(hipSetDevice(d))) return -1;
    const unsigned f = mode == 1 ? hipDeviceScheduleBlockingSync : mode == 2 ? hipDeviceScheduleSpin : mode == 3 ? hipDeviceScheduleYield : hipDeviceScheduleAuto;
    return ORBX_HIP_OK(hipSetDeviceFlags(f));
}
inline bool memory_is_host() { return false; }
inline int device_count() { int n = 0; if (hip_ok(hipGetDeviceCount(&n))) return 0; return n; }
inline void* dmalloc(size_t n) { void* p = nullptr; if (hip_ok(hipMalloc(&p, n ? n : 1))) return nullptr; live().dev++; return p; }
inline void dfree(void* p) { if (p) { live().dev--; (void)hip_ok(hipFree(p)); } }
inline void* hmalloc(size_t n) { void* p = nullptr; if (hip_ok(hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault))) return nullptr; live().pinned++; return p; }
inline void hfree(void* p) { if (p) { live().pinned--; (void)hip_ok(hipHostFree(p)); } }
inline int copy_h2d(void* d, const void* s, size_t n, stream_t st) { return ORBX_HIP_OK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st)); }
inline int copy_d2h(void* d, const void* s, size_t n, stream_t st) { return ORBX_HIP_OK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st)); }
inline int copy_d2d(void* d, const void* s, size_t n, stream_t st) { return ORBX_HIP_OK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st)); }
// kind: 0 host->device, 1 device->device
inline int copy2d(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int kind, stream_t st) {
    return ORBX_HIP_OK(hipMemcpy2DAsync(d, dp, s, sp, w, h, kind ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
}
inline int memset_async(void* d, int v, size_t n, stream_t st) { return ORBX_HIP_OK(hipMemsetAsync(d, v, n, st)); }
inline int stream_create(stream_t* s) { if (ORBX_HIP_OK(hipStreamCreateWithFlags(s, hipStreamNonBlocking))) return -1; live().streams++; return 0; }
inline void stream_destroy(stream_t s) { if (s) { live().streams--; (void)hip_ok(hipStreamDestroy(s)); } }
inline int stream_sync(stream_t s) { return ORBX_HIP_OK(hipStreamSynchronize(s)); }
inline int event_create(event_t* e) { if (ORBX_HIP_OK(hipEventCreate(e))) return -1; live().events++; return 0; }
inline void event_destroy(event_t e) { if (e) { live().events--; (void)hip_ok(hipEventDestroy(e)); } }
inline int event_record(event_t e, stream_t s) { return ORBX_HIP_OK(hipEventRecord(e, s)); }
inline int stream_wait_event(stream_t s, event_t e) { return ORBX_HIP_OK(hipStreamWaitEvent(s, e, 0)); }
inline int event_sync(event_t e) { return ORBX_HIP_OK(hipEventSynchronize(e)); }
inline float event_elapsed_ms(event_t a, event_t b) { float ms = 0; if (hip_ok(hipEventElapsedTime(&ms, a, b))) ms = 0; return ms; }   // unrecorded events: 0
inline const char* last_error() { hipError_t e = hipGetLastError(); if (e == hipSuccess) e = last_code(); return hipGetErrorString(e); }
inline int check_launch() { return ORBX_HIP_OK(hipGetLastError()); }
// largest LDS allocation (static + dynamic) one workgroup may ask for on this device
inline size_t lds_limit(int dev) {
    static thread_local int cached_dev = -1; static thread_local size_t cached = 0;
    if (dev != cached_dev) { int v = 0; if (hip_ok(hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev))) v = 65536; cached = (size_t)v; cached_dev = dev; }
    return cached;
}
#endif

}}  // namespace orbx::rt
