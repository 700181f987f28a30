// orbx_comm.cpp — the one exchange step of the multi-GPU mode (BASELINE.json configs[4]) behind the C ABI: an all-gather of the
// descriptor blocks [B][cap][32] and counts [B] of every rank's last batch, for a host that matches globally (loop closing / place recognition over
// all camera streams).  The hot path itself has no collective: streams are independent and one process (or thread) drives one GPU.
//
// Product build: RCCL (ncclAllGather over xGMI) on a stream of the communicator's own, behind a device-to-device snapshot of the handle's results,
// so that the handle's next extraction overlaps the collective.  librccl is loaded with dlopen() on the first orbx_comm_* call: a host that never
// exchanges descriptors does not load it (a live RCCL communicator costs the extraction streams hardware queues).
// Emulator build (tests): "ranks" are threads of one process that meet in a mutex-and-condition-variable rendezvous keyed by the unique id; the
// data movement is memcpy.  Same entry points, same layouts, no RCCL.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>
#include "orbx_internal.h"
#include "../../include/orbx.h"
#ifdef ORBX_EMU
#include <condition_variable>
#include <map>
#include <memory>
#else
#include <dlfcn.h>
#include <string>
#include <vector>
#endif

using namespace orbx;

namespace {
#ifndef ORBX_EMU
// the slice of the NCCL API this file uses (rccl/rccl.h: ncclUniqueId is 128 bytes; ncclUint8 = 1; ncclSuccess = 0)
struct NcclId { char internal[128]; };
typedef void* nccl_comm_t;
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string err;                       // why it is not ok: dlerror() read once, where the failure happened
    std::string path;                      // the file that was loaded
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // RCCL has to sit on the SAME HIP runtime as this library: its streams and buffers are handed to it.  A process may hold two runtimes - this
        // library bound to /opt/rocm's, then `import torch` maps the copy bundled in torch/lib together with its own librccl.so.1 - and a dlopen by
        // soname then returns the RCCL that is already there, on the other runtime (ncclCommInitRank: "unhandled cuda error", found by
        // tests/test_lifetime.py on the GPU).  So the first candidates are the librccl files next to the libamdhip64 this library resolved.
        std::vector<std::string> names;
        Dl_info di;
        if (dladdr((const void*)&hipDeviceSynchronize, &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) { dir.resize(slash + 1); names.push_back(dir + "librccl.so.1"); names.push_back(dir + "librccl.so"); }
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) names.push_back(name);
        for (const std::string& name : names) {
            r.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.lib) { r.path = name; break; }
            const char* e = dlerror();
            if (e && r.err.empty()) r.err = e;
        }
        if (!r.lib) return;
        r.GetUniqueId = (int (*)(NcclId*))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (int (*)(nccl_comm_t*, int, NcclId, int))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (int (*)(nccl_comm_t))dlsym(r.lib, "ncclCommDestroy");
        r.AllGather = (int (*)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t))dlsym(r.lib, "ncclAllGather");
        r.GroupStart = (int (*)())dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (int (*)())dlsym(r.lib, "ncclGroupEnd");
        r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd;
        r.err = r.ok ? "" : "symbols missing";
    });
    return r;
}
const char* nccl_err(int e) { Rccl& r = rccl(); return r.GetErrorString ? r.GetErrorString(e) : "?"; }
#else
// in-process stand-in: every rank of a communicator is a thread of this process
struct Meeting {
    std::mutex m; std::condition_variable cv;
    int world = 0, arrived = 0, generation = 0, joined = 0;
    std::vector<const void*> src; std::vector<size_t> bytes;
};
std::mutex g_reg_m;
std::map<std::vector<uint8_t>, std::shared_ptr<Meeting>> g_reg;
int g_next_id = 1;
#endif
}  // namespace

struct orbx_comm {
    int world = 1, rank = 0, device = 0;
    bool owned = true;                      // created here (destroyed here) or adopted from the host
    rt::stream_t stream{}; rt::event_t ev_snap{}, ev_done{}; bool have_stream = false;
    DevBuf<uint8_t> snap, all2[2];          // this rank's snapshot [B * cap * 32 | B * 4]; the gathered blocks [world][B * cap * 32] | [world][B * 4], two of them:
    int cur = 0;                            // exchanges alternate, so that what exchange k returned stays untouched until exchange k + 2 is enqueued
    int B = 0, cap = 0; size_t desc_bytes = 0; bool pending = false;
#ifndef ORBX_EMU
    nccl_comm_t nccl = nullptr;
#else
    std::shared_ptr<Meeting> meet;
#endif
};

extern "C" {

int orbx_comm_unique_id(uint8_t id[ORBX_COMM_ID_BYTES]) {
    if (!id) return fail(ORBX_E_ARG, "null id");
    memset(id, 0, ORBX_COMM_ID_BYTES);
#ifndef ORBX_EMU
    Rccl& r = rccl();
    if (!r.ok) return fail(ORBX_E_DEVICE, "librccl could not be loaded (%s)", r.err.c_str());
    NcclId nid; const int e = r.GetUniqueId(&nid);
    if (e) return fail(ORBX_E_DEVICE, "ncclGetUniqueId: %s", nccl_err(e));
    static_assert(sizeof(NcclId) == ORBX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id, &nid, sizeof nid);
#else
    std::lock_guard<std::mutex> l(g_reg_m);
    const int v = g_next_id++;
    memcpy(id, &v, sizeof v); memcpy(id + 8, "orbx-emu", 8);
#endif
    return ORBX_OK;
}

static int comm_common_init(orbx_comm* c) {
    if (rt::set_device(c->device)) return fail(ORBX_E_DEVICE, "hipSetDevice(%d) failed", c->device);
    if (rt::stream_create(&c->stream) | rt::event_create(&c->ev_snap) | rt::event_create(&c->ev_done)) return fail(ORBX_E_DEVICE, "stream / event creation failed");
    c->have_stream = true;
    return ORBX_OK;
}

int orbx_comm_create(orbx_comm** out, int world, int rank, const uint8_t id[ORBX_COMM_ID_BYTES], int device_id) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail(ORBX_E_ARG, "bad communicator arguments");
    *out = nullptr;
    if (device_id < 0 || device_id >= rt::device_count()) return fail(ORBX_E_DEVICE, "no usable GPU %d (HIP reports %d devices)", device_id, rt::device_count());
    orbx_comm* c = new orbx_comm();
    c->world = world; c->rank = rank; c->device = device_id;
    int rc = comm_common_init(c);
    if (rc) { orbx_comm_destroy(c); return rc; }
#ifndef ORBX_EMU
    Rccl& r = rccl();
    if (!r.ok) { orbx_comm_destroy(c); return fail(ORBX_E_DEVICE, "librccl could not be loaded (%s)", r.err.c_str()); }
    NcclId nid; memcpy(&nid, id, sizeof nid);
    const int e = r.CommInitRank(&c->nccl, world, nid, rank);           // collective over the ranks: every rank calls it with the same id
    if (e) { orbx_comm_destroy(c); return fail(ORBX_E_DEVICE, "ncclCommInitRank(rank %d of %d, GPU %d): %s [%s]", rank, world, device_id, nccl_err(e), rccl().path.c_str()); }
#else
    std::vector<uint8_t> key(id, id + ORBX_COMM_ID_BYTES);
    {
        std::lock_guard<std::mutex> l(g_reg_m);
        auto& slot = g_reg[key];
        if (!slot) { slot = std::make_shared<Meeting>(); slot->world = world; slot->src.assign(world, nullptr); slot->bytes.assign(world, 0); }
        if (slot->world != world) { orbx_comm_destroy(c); return fail(ORBX_E_ARG, "ranks disagree about the world size"); }
        c->meet = slot;
        if (++slot->joined == world) g_reg.erase(key);                  // complete: a later communicator with a recycled id starts afresh
    }
#endif
    *out = c;
    return ORBX_OK;
}

int orbx_comm_adopt(orbx_comm** out, void* nccl_comm, int world, int rank, int device_id) {
    if (!out || !nccl_comm || world < 1 || rank < 0 || rank >= world) return fail(ORBX_E_ARG, "bad communicator arguments");
    *out = nullptr;
#ifdef ORBX_EMU
    return fail(ORBX_E_ARG, "the emulator build has no RCCL communicators to adopt");
#else
    if (device_id < 0 || device_id >= rt::device_count()) return fail(ORBX_E_DEVICE, "no usable GPU %d", device_id);
    if (!rccl().ok) return fail(ORBX_E_DEVICE, "librccl could not be loaded");
    orbx_comm* c = new orbx_comm();
    c->world = world; c->rank = rank; c->device = device_id; c->owned = false; c->nccl = nccl_comm;
    const int rc = comm_common_init(c);
    if (rc) { orbx_comm_destroy(c); return rc; }
    *out = c;
    return ORBX_OK;
#endif
}

void orbx_comm_destroy(orbx_comm* c) {
    if (!c) return;
    rt::set_device(c->device);
    if (c->have_stream) { rt::stream_sync(c->stream); }
#ifndef ORBX_EMU
    if (c->nccl && c->owned) rccl().CommDestroy(c->nccl);
#endif
    if (c->have_stream) { rt::stream_destroy(c->stream); rt::event_destroy(c->ev_snap); rt::event_destroy(c->ev_done); }
    c->snap.release(); for (auto& x : c->all2) x.release();        // (DevBuf has no destructor: found by tests/test_lifetime.py)
    delete c;
}

int orbx_comm_world(const orbx_comm* c) { return c ? c->world : ORBX_E_ARG; }
int orbx_comm_rank(const orbx_comm* c) { return c ? c->rank : ORBX_E_ARG; }

int orbx_allgather_descriptors(orbx_extractor* h, orbx_comm* c, void** desc_all, void** n_all, int* B_out, int* cap_out) {
    if (!h || !c || h->lastB <= 0) return fail(ORBX_E_ARG, "nothing extracted yet / null");
    if (h->device != c->device) return fail(ORBX_E_ARG, "the extractor lives on GPU %d, the communicator on GPU %d", h->device, c->device);
    rt::set_device(c->device);
    const int B = h->lastB, cap = h->kp_total_cap;
    const size_t db = (size_t)B * cap * 32, nb = sizeof(int) * (size_t)B, W = (size_t)c->world;
    if (c->pending) {
        // the previous exchange still reads the snapshot and writes the gathered blocks: this call's snapshot copies wait for it ON THE DEVICE (the
        // handle's stream waits for ev_done), the host goes on; only a reallocation of the buffers needs the previous exchange to have finished
        DevBuf<uint8_t>& nxt = c->all2[c->cur ^ 1];
        const bool grows = !(c->snap.p && db + nb + 64 <= c->snap.n && nxt.p && W * (db + nb) + 64 <= nxt.n);
        if (grows) { rt::event_sync(c->ev_done); c->pending = false; }
        else if (rt::stream_wait_event(h->s0, c->ev_done)) return fail(ORBX_E_DEVICE, "waiting for the previous exchange failed: %s", rt::last_error());
    }
    c->B = 0;                                                                      // nothing to fetch until this exchange has been enqueued completely
    c->cur ^= 1;                                                                   // the other gathered block: the previous exchange's result is not written by this one
    DevBuf<uint8_t>& all = c->all2[c->cur];
    if (c->snap.ensure(db + nb + 64) || all.ensure(W * (db + nb) + 64)) return fail(ORBX_E_DEVICE, "allocation failed (%zu bytes gathered)", W * (db + nb));
    // snapshot on the HANDLE's stream (behind the extraction that is producing the block), everything after it on the communicator's stream: the handle
    // is free for its next batch as soon as the two copies have run
    if (rt::copy_d2d(c->snap.p, h->d_desc.p, db, h->s0) || rt::copy_d2d(c->snap.p + db, h->d_nm.p, nb, h->s0) || rt::event_record(c->ev_snap, h->s0) ||
        rt::stream_wait_event(c->stream, c->ev_snap))
        return fail(ORBX_E_DEVICE, "snapshot failed: %s", rt::last_error());
    uint8_t* all_desc = all.p; uint8_t* all_n = all.p + W * db;
#ifndef ORBX_EMU
    Rccl& r = rccl();
    int e = r.GroupStart();
    if (!e) e = r.AllGather(c->snap.p, all_desc, db, /*ncclUint8*/ 1, c->nccl, c->stream);         // rank r's block lands at [r * db, (r + 1) * db)
    if (!e) e = r.AllGather(c->snap.p + db, all_n, nb, 1, c->nccl, c->stream);
    const int e2 = r.GroupEnd();
    if (e || e2) return fail(ORBX_E_DEVICE, "ncclAllGather: %s", nccl_err(e ? e : e2));
#else
    {
        Meeting& m = *c->meet;
        std::unique_lock<std::mutex> l(m.m);
        const int gen = m.generation;
        m.src[c->rank] = c->snap.p; m.bytes[c->rank] = db + nb;
        if (++m.arrived == m.world) { m.arrived = 0; m.generation++; m.cv.notify_all(); }
        else m.cv.wait(l, [&] { return m.generation != gen; });
        int odd = -1;
        for (int r = 0; r < m.world; r++) if (m.bytes[r] != db + nb) odd = r;
        for (int r = 0; r < m.world && odd < 0; r++) { memcpy(all_desc + (size_t)r * db, m.src[r], db); memcpy(all_n + (size_t)r * nb, (const uint8_t*)m.src[r] + db, nb); }
        // nobody may overwrite its snapshot before everybody has copied it
        const int gen2 = m.generation;
        if (++m.arrived == m.world) { m.arrived = 0; m.generation++; m.cv.notify_all(); }
        else m.cv.wait(l, [&] { return m.generation != gen2; });
        if (odd >= 0) return fail(ORBX_E_ARG, "rank %d gathers blocks of another shape", odd);        // after the second rendezvous: no rank is left waiting
    }
#endif
    if (rt::event_record(c->ev_done, c->stream)) return fail(ORBX_E_DEVICE, "event record failed");
    c->pending = true;
    c->B = B; c->cap = cap; c->desc_bytes = db;
    if (desc_all) *desc_all = all_desc;
    if (n_all) *n_all = all_n;
    if (B_out) *B_out = B;
    if (cap_out) *cap_out = cap;
    return ORBX_OK;
}

int orbx_comm_wait(orbx_comm* c) {
    if (!c) return fail(ORBX_E_ARG, "null");
    rt::set_device(c->device);
    if (c->pending) { if (rt::event_sync(c->ev_done)) return fail(ORBX_E_DEVICE, "exchange failed: %s", rt::last_error()); c->pending = false; }
    return ORBX_OK;
}

int orbx_comm_fetch(orbx_comm* c, uint8_t* desc_all_host, int* n_all_host) {
    if (!c || c->B <= 0) return fail(ORBX_E_ARG, "nothing gathered yet");
    rt::set_device(c->device);
    const size_t W = (size_t)c->world, db = c->desc_bytes, nb = sizeof(int) * (size_t)c->B;
    int e = 0;
    if (desc_all_host) e |= rt::copy_d2h(desc_all_host, c->all2[c->cur].p, W * db, c->stream);
    if (n_all_host) e |= rt::copy_d2h(n_all_host, c->all2[c->cur].p + W * db, W * nb, c->stream);
    if (e || rt::stream_sync(c->stream)) return fail(ORBX_E_DEVICE, "D2H failed: %s", rt::last_error());
    c->pending = false;
    return ORBX_OK;
}

}  // extern "C"
