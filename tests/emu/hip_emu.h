// tests/emu/hip_emu.h — TEST INFRASTRUCTURE ONLY.
//
// A tiny SIMT emulator that lets the *same kernel sources* under
// orb_slam3_detailed_comments_amd/csrc/*.hip be compiled with g++ and executed on CPU cores, so the
// `-m "not gpu"` suite can check kernel logic (barriers, wave ballots/shuffles, LDS carving, ordered
// compaction, the quadtree) against the oracle without a GPU.  It is NOT a product path: the Python
// package only ever loads the hipcc-built liborbx_hip.so and fails loudly when it is missing; the
// emulator build (tests/emu/liborbx_emu.so) is loaded explicitly by tests only.
//
// Model: one workgroup = N cooperative fibers (ucontext) on one OS thread, wave size 64.
// __syncthreads / wave exchanges yield round-robin until every fiber of the group arrives.
// Workgroups of a launch are spread over a small pool of OS threads.
#pragma once
#include <ucontext.h>
#include <sys/mman.h>
#include <algorithm>
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned _x = 1, unsigned _y = 1, unsigned _z = 1) : x(_x), y(_y), z(_z) {}
};

// The two flavours of the fiber switch (below) live in different inline namespaces: function-local statics of inline functions (the per-thread fiber pool, the
// worker pool) are unified process-wide by the dynamic linker, and a process may hold two emulator builds at once (tests/test_emu_variants.py builds its
// own; the sanitizer runs load a ucontext build beside them) - with one name for both, builds with different Fiber layouts shared one pool and crashed.
#if !defined(__x86_64__) && !defined(HIPEMU_UCONTEXT)
#define HIPEMU_UCONTEXT
#endif
#ifdef HIPEMU_UCONTEXT
#define HIPEMU_FLAVOUR uctx
#else
#define HIPEMU_FLAVOUR stackswitch
#endif
namespace hipemu {
inline namespace HIPEMU_FLAVOUR {

constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

// Switching fibers: by default a six-register stack switch (x86-64 SysV: rbx, rbp, r12-r15 + the stack pointer; the kernels never touch the FP control
// words) - swapcontext() saves and restores the signal mask with a system call per switch, which was a third of the CPU time of the emulated suite.
// -DHIPEMU_UCONTEXT keeps ucontext (the sanitizer builds: ASan knows swapcontext, not a hand-made switch; other architectures).
#ifdef HIPEMU_UCONTEXT
typedef ucontext_t Ctx;
#else
struct Ctx { void* sp = nullptr; };
static __attribute__((naked, noinline)) void switch_ctx(Ctx* /*from: rdi*/, Ctx* /*to: rsi*/) {
    __asm__ volatile(
        "pushq %rbp\n\t" "pushq %rbx\n\t" "pushq %r12\n\t" "pushq %r13\n\t" "pushq %r14\n\t" "pushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq (%rsi), %rsp\n\t"
        "popq %r15\n\t" "popq %r14\n\t" "popq %r13\n\t" "popq %r12\n\t" "popq %rbx\n\t" "popq %rbp\n\t"
        "ret\n\t");
}
#endif

struct Block;
struct Fiber {
    Ctx ctx;
    dim3 tid;
    int flat = 0, lane = 0, wave = 0;
    bool done = false;
    void* stack = nullptr;
    Block* blk = nullptr;
};
struct WaveState {
    int count = 0, gen = 0, lanes = 0;
    uint64_t slot[kWave];
};
struct Block {
    dim3 bid, bdim, gdim;
    int nthreads = 0, alive = 0;
    int bar_count = 0, bar_gen = 0;
    std::vector<WaveState> waves;
    unsigned char* dyn_smem = nullptr;
    Ctx sched;
    const std::function<void()>* fn = nullptr;
};
struct Tls {
    Fiber* cur = nullptr;
    std::vector<Fiber> pool;
    std::vector<unsigned char> smem;
    // the workers of a launch are threads of that launch: their fiber stacks go with them (they were left mapped until round 5 - a few MB of touched
    // pages per launch, tens of GB over a sanitizer run of the suite)
    ~Tls() { for (Fiber& f : pool) if (f.stack) munmap(f.stack, kStack); }
};
inline Tls& tls() { static thread_local Tls t; return t; }
inline Fiber& cur() { return *tls().cur; }

#ifdef HIPEMU_UCONTEXT
inline void switch_ctx(Ctx* from, Ctx* to) { swapcontext(from, to); }
#endif
inline void yield() { Fiber* f = tls().cur; switch_ctx(&f->ctx, &f->blk->sched); }

inline void block_barrier() {
    Block* b = cur().blk;
    const int gen = b->bar_gen;
    if (++b->bar_count >= b->alive) { b->bar_count = 0; b->bar_gen++; return; }
    while (b->bar_gen == gen) yield();
}
inline void wave_barrier() {
    Fiber& f = cur();
    WaveState& w = f.blk->waves[f.wave];
    const int gen = w.gen;
    if (++w.count >= w.lanes) { w.count = 0; w.gen++; return; }
    while (w.gen == gen) yield();
}
// every lane of the wave contributes one 64-bit value; all lanes get the whole vector
inline void wave_exchange(uint64_t mine, uint64_t out[kWave]) {
    Fiber& f = cur();
    WaveState& w = f.blk->waves[f.wave];
    w.slot[f.lane] = mine;
    wave_barrier();
    for (int i = 0; i < kWave; i++) out[i] = i < w.lanes ? w.slot[i] : 0;
    wave_barrier();
}

inline void trampoline() {
    Fiber* f = tls().cur;
    (*f->blk->fn)();
    f->done = true;
    f->blk->alive--;
    // a thread that exits while others wait at a barrier: re-evaluate the barrier condition
    Block* b = f->blk;
    if (b->bar_count > 0 && b->bar_count >= b->alive) { b->bar_count = 0; b->bar_gen++; }
    WaveState& w = b->waves[f->wave];
    w.lanes--;
    if (w.count > 0 && w.count >= w.lanes) { w.count = 0; w.gen++; }
    switch_ctx(&f->ctx, &b->sched);
    __builtin_unreachable();            // a finished fiber is never resumed
}

inline void run_block(const std::function<void()>& fn, dim3 bid, dim3 bdim, dim3 gdim, size_t smem_bytes) {
    Tls& t = tls();
    const int n = (int)(bdim.x * bdim.y * bdim.z);
    if ((int)t.pool.size() < n) {
        size_t old = t.pool.size();
        t.pool.resize(n);
        for (size_t i = old; i < t.pool.size(); i++) {
            t.pool[i].stack = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            assert(t.pool[i].stack != MAP_FAILED);
        }
    }
    if (t.smem.size() < smem_bytes + 64) t.smem.resize(smem_bytes + 64);
    Block b;
    b.bid = bid; b.bdim = bdim; b.gdim = gdim; b.nthreads = n; b.alive = n; b.fn = &fn;
    b.dyn_smem = (unsigned char*)(((uintptr_t)t.smem.data() + 63) & ~(uintptr_t)63);
    b.waves.resize((n + kWave - 1) / kWave);
    for (int i = 0; i < n; i++) {
        Fiber& f = t.pool[i];
        f.flat = i; f.lane = i % kWave; f.wave = i / kWave; f.done = false; f.blk = &b;
        f.tid = dim3(i % bdim.x, (i / bdim.x) % bdim.y, i / (bdim.x * bdim.y));
        b.waves[f.wave].lanes++;
#ifdef HIPEMU_UCONTEXT
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
#else
        // first switch into the fiber: six zeroed callee-saved registers are popped, `ret` enters trampoline() with the stack pointer where a call
        // would have left it (8 below a 16-byte boundary), above it a null return address (trampoline never returns)
        uint64_t* sp = (uint64_t*)(((uintptr_t)f.stack + kStack) & ~(uintptr_t)15);
        *--sp = 0;
        *--sp = (uint64_t)(uintptr_t)(void (*)())trampoline;
        for (int r = 0; r < 6; r++) *--sp = 0;
        f.ctx.sp = sp;
#endif
    }
    while (b.alive > 0) {
        for (int i = 0; i < n; i++) {
            if (t.pool[i].done) continue;
            t.cur = &t.pool[i];
            switch_ctx(&b.sched, &t.pool[i].ctx);
        }
    }
    t.cur = nullptr;
}

inline int& num_workers() { static int n = std::max(1u, std::min(8u, std::thread::hardware_concurrency())); return n; }

// The workers are threads that live as long as the process (round 5; a launch used to create and join its own, with their fiber stacks mapped anew -
// most launches of the test suite are small, and that was most of their cost).  One launch runs at a time; a launch from another host thread waits.
struct Pool {
    std::mutex launch_m, m;
    std::condition_variable cv_job, cv_done;
    const std::function<void()>* job = nullptr;
    long gen = 0;
    int want = 0, running = 0, nthreads = 0;
    pid_t pid = 0;
};
inline void pool_thread(Pool* p, int index) {
    long seen = 0;
    for (;;) {
        const std::function<void()>* job;
        {
            std::unique_lock<std::mutex> l(p->m);
            p->cv_job.wait(l, [&] { return p->gen != seen; });
            seen = p->gen;
            if (index >= p->want) continue;
            job = p->job;
        }
        (*job)();
        {
            std::lock_guard<std::mutex> l(p->m);
            if (--p->running == 0) p->cv_done.notify_all();
        }
    }
}
inline Pool* pool() {
    static std::mutex m;
    static Pool* p = nullptr;
    std::lock_guard<std::mutex> l(m);
    if (!p || p->pid != getpid()) {                 // (a forked child has none of the parent's threads: it gets a pool of its own; the old one is left alone)
        p = new Pool();
        p->pid = getpid();
        p->nthreads = num_workers();
        for (int i = 0; i < p->nthreads; i++) std::thread(pool_thread, p, i).detach();
    }
    return p;
}

inline void launch(dim3 grid, dim3 block, size_t smem, std::function<void()> fn) {
    const long total = (long)grid.x * grid.y * grid.z;
    if (total == 0) return;
    std::atomic<long> next(0);
    const std::function<void()> worker = [&]() {
        for (;;) {
            long i = next.fetch_add(1);
            if (i >= total) break;
            dim3 bid((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((long)grid.x * grid.y)));
            run_block(fn, bid, block, grid, smem);
        }
    };
    const int nw = (int)std::min<long>(num_workers(), total);
    if (nw <= 1) { worker(); return; }
    Pool* p = pool();
    std::lock_guard<std::mutex> one(p->launch_m);
    std::unique_lock<std::mutex> l(p->m);
    p->job = &worker; p->want = std::min(nw, p->nthreads); p->running = p->want; p->gen++;
    p->cv_job.notify_all();
    p->cv_done.wait(l, [&] { return p->running == 0; });
}

}  // inline namespace
}  // namespace hipemu

// ---- the HIP device-side vocabulary the kernels use ------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define threadIdx (hipemu::cur().tid)
#define blockIdx (hipemu::cur().blk->bid)
#define blockDim (hipemu::cur().blk->bdim)
#define gridDim (hipemu::cur().blk->gdim)

static inline void __syncthreads() { hipemu::block_barrier(); }
static inline unsigned long long __ballot(int pred) {
    uint64_t v[hipemu::kWave]; hipemu::wave_exchange(pred ? 1 : 0, v);
    unsigned long long m = 0; for (int i = 0; i < hipemu::kWave; i++) if (v[i]) m |= 1ull << i; return m;
}
template <typename T> static inline T __emu_shfl_idx(T val, int src) {
    static_assert(sizeof(T) <= 8, "shfl value too wide");
    uint64_t mine = 0; memcpy(&mine, &val, sizeof(T));
    uint64_t v[hipemu::kWave]; hipemu::wave_exchange(mine, v);
    if (src < 0 || src >= hipemu::kWave) src = hipemu::cur().lane;
    T out; memcpy(&out, &v[src], sizeof(T)); return out;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) { int l = hipemu::cur().lane; return __emu_shfl_idx(v, (l / width) * width + (src % width)); }
template <typename T> static inline T __shfl_xor(T v, int m, int width = 64) { int l = hipemu::cur().lane; int s = l ^ m; return __emu_shfl_idx(v, (s / width == l / width) ? s : l); }
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) { int l = hipemu::cur().lane; int s = l + (int)d; return __emu_shfl_idx(v, (s / width == l / width) ? s : l); }
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) { int l = hipemu::cur().lane; int s = l - (int)d; return __emu_shfl_idx(v, (s >= 0 && s / width == l / width) ? s : l); }

static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }   // v_mul_u32_u24: low 24 bits of each factor, low 32 bits of the product
static inline int __mul24(int a, int b) { return a * b; }     // device: v_mul_i32_i24; the callers keep both factors below 2^23
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }

template <typename T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicMax(T* p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o > v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// IEEE single/double ops with explicit rounding (the emulator is compiled -ffp-contract=off)
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dsub_rn(double a, double b) { return a - b; }
static inline int __float2int_rn(float a) { return (int)lrintf(a); }
static inline int __float2int_rz(float a) { return (int)a; }
static inline int __double2int_rz(double a) { return (int)a; }
static inline float __int2float_rn(int a) { return (float)a; }
static inline float __double2float_rn(double a) { return (float)a; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
