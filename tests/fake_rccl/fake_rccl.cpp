// fake_rccl -- a stand-in for librccl.so.1.  TEST INFRASTRUCTURE ONLY.
//
// Why it exists: the device-group code of the product (sponge_amd/csrc/pmx_mgpu.cpp) binds RCCL with
// dlopen("librccl.so.1").  The development boxes have ONE GPU and RCCL refuses two ranks on one device, so every
// world > 1 branch of that file - shard offsets, the `mine` test of the ragged gather, the d_top layout of the sharded
// tree, the host fan-out with more than one worker thread - could not execute there.  This library exports the twelve
// entry points the product binds, with communicators that are plain structs whose "ranks" may all live on the same
// device: a collective is performed as device-to-device hipMemcpyAsync on the streams the caller gave, ordered with
// events exactly as far as RCCL's stream semantics promise (a rank's receive buffer is complete when ITS stream reaches
// the point after the call; a send buffer may be reused by its owner's stream after the call).
//
// What it proves and what it does not: the product's OWN bookkeeping (which pointer, which offset, which count, which
// root, which stream, the same sequence of calls on every rank) is executed and checked - the stand-in verifies that all
// ranks of a communicator post matching operations and that every buffer range lies inside one device allocation.  It
// says nothing about RCCL itself, xGMI, or inter-process rendezvous.
//
// It is reached only when a test puts this directory first on LD_LIBRARY_PATH of a fresh child process; nothing under
// sponge_amd/, bench.py or __graft_entry__.smoke() knows it exists.  The reference has no counterpart: it has no
// multi-device code (src/poseidon/mod.rs:62-183 - independence of the states is the whole contract).
//
// Semantics implemented (RCCL 2.27 header, /opt/rocm/include/rccl/rccl.h):
//   ncclAllGather(send, recv, count)   recv[r * count .. (r+1) * count) on every rank = rank r's send; in place when
//                                      send == recv + rank * count
//   ncclBroadcast(send, recv, count, root)   every rank's recv = the root's send; the non-roots' send is ignored; in
//                                      place on the root when send == recv
//   ncclGroupStart / ncclGroupEnd      calls inside a group are queued per communicator and performed at the outermost
//                                      GroupEnd; the k-th queued call of every rank of a communicator is one collective
//   ncclCommInitAll                    one thread owns all ranks;   ncclCommInitRank   ranks are joined by id, inside ONE
//                                      process (threads), and the call blocks until all of them have arrived, as RCCL does
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

enum Kind { kAllGather = 0, kBroadcast = 1 };

struct Op {
    Kind kind;
    const void *send;
    void *recv;
    size_t count;
    ncclDataType_t type;
    int root;
    hipStream_t stream;
};

struct Clique {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    std::vector<ncclComm *> member;          // by rank
    int joined = 0, alive = 0;
    // one round = every rank has posted the calls of its group
    uint64_t epoch = 0;
    int arrived = 0;
    std::vector<std::vector<Op>> posted;     // by rank
    ncclResult_t last_result = ncclSuccess;  // result of the round that ended at `epoch`
    std::vector<hipEvent_t> events;          // every event a collective made; destroyed with the last communicator
};

}  // namespace

struct ncclComm {
    Clique *clique = nullptr;
    int rank = 0;
    int device = 0;
    std::vector<Op> pending;                 // queued inside a group by the owning thread
    uint64_t post_epoch = 0;
};

namespace {

// ---- statistics and fault injection (read by the tests through fake_rccl_* below) --------------------------------------
struct Stats {
    std::atomic<long> all_gathers{0}, broadcasts{0}, groups{0}, copies{0}, inplace_skips{0}, comms_created{0}, comms_destroyed{0};
    std::atomic<long long> bytes{0};
};
Stats g_stats;
// fail the n-th next call (1 = the very next) of one entry point: 0 none, 1 AllGather, 2 Broadcast, 3 GroupEnd,
// 4 CommInitAll, 5 CommInitRank, 6 GetUniqueId, 7 GroupStart
std::atomic<int> g_fail_which{0}, g_fail_countdown{0};
bool injected(int which) {
    if (g_fail_which.load() != which) return false;
    if (g_fail_countdown.fetch_sub(1) == 1) {
        g_fail_which.store(0);
        return true;
    }
    return false;
}

std::mutex g_detail_lock;
std::string g_detail;   // why the last ncclInvalidArgument / ncclInternalError was returned
ncclResult_t fail(ncclResult_t code, const std::string &why) {
    std::lock_guard<std::mutex> l(g_detail_lock);
    g_detail = why;
    return code;
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

// [p, p + bytes) must lie inside ONE device allocation (a span that runs past the end of its buffer is exactly the kind
// of slip in the product's offset arithmetic this library is here to catch)
ncclResult_t check_range(const void *p, size_t bytes, const char *what, int rank) {
    if (bytes == 0) return ncclSuccess;
    if (!p) return fail(ncclInvalidArgument, std::string(what) + ": null pointer on rank " + std::to_string(rank));
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess || attr.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(ncclInvalidArgument, std::string(what) + ": not a device pointer on rank " + std::to_string(rank));
    }
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ncclInvalidArgument, std::string(what) + ": no allocation behind the pointer on rank " + std::to_string(rank));
    }
    const char *b = (const char *)base, *q = (const char *)p;
    if (q < b || q + bytes > b + size) {
        char msg[256];
        std::snprintf(msg, sizeof msg, "%s: rank %d touches [%zu, %zu) of an allocation of %zu bytes", what, rank, (size_t)(q - b),
                      (size_t)(q - b) + bytes, size);
        return fail(ncclInvalidArgument, msg);
    }
    return ncclSuccess;
}

struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != d) (void)hipSetDevice(d);
        else prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

#define FAKE_HIP(expr)                                                                                      \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(ncclUnhandledCudaError, std::string(#expr ": ") + hipGetErrorString(e_)); \
    } while (0)

// one collective: op[r] is what rank r posted.  Called with the clique's lock held by whichever thread arrived last.
ncclResult_t perform(Clique *c, const std::vector<Op> &op) {
    const int W = c->world;
    const Op &o0 = op[0];
    const size_t esz = type_bytes(o0.type);
    if (esz == 0) return fail(ncclInvalidArgument, "unsupported datatype");
    for (int r = 1; r < W; ++r) {
        const Op &o = op[r];
        if (o.kind != o0.kind || o.count != o0.count || o.type != o0.type || o.root != o0.root) {
            char msg[256];
            std::snprintf(msg, sizeof msg, "ranks 0 and %d posted different collectives (kind %d/%d, count %zu/%zu, root %d/%d)", r,
                          (int)o0.kind, (int)o.kind, o0.count, o.count, o0.root, o.root);
            return fail(ncclInvalidArgument, msg);
        }
    }
    const size_t bytes = o0.count * esz;
    if (o0.kind == kBroadcast && (o0.root < 0 || o0.root >= W)) return fail(ncclInvalidArgument, "broadcast root out of range");
    // argument checks before anything is enqueued
    for (int r = 0; r < W; ++r) {
        DeviceScope ds(c->member[r]->device);
        ncclResult_t rc = ncclSuccess;
        if (o0.kind == kAllGather) {
            rc = check_range(op[r].send, bytes, "ncclAllGather sendbuff", r);
            if (rc == ncclSuccess) rc = check_range(op[r].recv, bytes * (size_t)W, "ncclAllGather recvbuff", r);
        } else {
            if (r == o0.root) rc = check_range(op[r].send, bytes, "ncclBroadcast sendbuff (root)", r);
            if (rc == ncclSuccess) rc = check_range(op[r].recv, bytes, "ncclBroadcast recvbuff", r);
        }
        if (rc != ncclSuccess) return rc;
    }
    if (bytes == 0) return ncclSuccess;
    // "ready": everything rank r enqueued before the call has happened (its send buffer holds the data, its receive buffer
    // may be overwritten)
    // (the events are kept until the communicator goes: a stream may still be waiting on one when this call returns)
    std::vector<hipEvent_t> ready(W, nullptr), done(W, nullptr);
    struct Keep {
        Clique *c; std::vector<hipEvent_t> &a, &b;
        ~Keep() { for (auto *v : {&a, &b}) for (hipEvent_t e : *v) if (e) c->events.push_back(e); }
    } keep{c, ready, done};
    for (int r = 0; r < W; ++r) {
        DeviceScope ds(c->member[r]->device);
        FAKE_HIP(hipEventCreateWithFlags(&ready[r], hipEventDisableTiming));
        FAKE_HIP(hipEventCreateWithFlags(&done[r], hipEventDisableTiming));
        FAKE_HIP(hipEventRecord(ready[r], op[r].stream));
    }
    for (int i = 0; i < W; ++i) {   // receiver
        DeviceScope ds(c->member[i]->device);
        hipStream_t st = op[i].stream;
        for (int j = 0; j < W; ++j)
            if (j != i) FAKE_HIP(hipStreamWaitEvent(st, ready[j], 0));
        if (o0.kind == kAllGather) {
            for (int j = 0; j < W; ++j) {
                char *dst = (char *)op[i].recv + (size_t)j * bytes;
                if ((const void *)dst == op[j].send) { g_stats.inplace_skips++; continue; }   // rank i's own slot, in place
                FAKE_HIP(hipMemcpyAsync(dst, op[j].send, bytes, hipMemcpyDeviceToDevice, st));
                g_stats.copies++;
                g_stats.bytes += (long long)bytes;
            }
        } else {
            const void *src = op[o0.root].send;
            if (op[i].recv == src) { g_stats.inplace_skips++; }
            else {
                FAKE_HIP(hipMemcpyAsync(op[i].recv, src, bytes, hipMemcpyDeviceToDevice, st));
                g_stats.copies++;
                g_stats.bytes += (long long)bytes;
            }
        }
        FAKE_HIP(hipEventRecord(done[i], st));
    }
    // a sender's stream may not run ahead (and overwrite its send buffer) before every reader has copied from it
    for (int j = 0; j < W; ++j) {
        DeviceScope ds(c->member[j]->device);
        for (int i = 0; i < W; ++i)
            if (i != j) FAKE_HIP(hipStreamWaitEvent(op[j].stream, done[i], 0));
    }
    (o0.kind == kAllGather ? g_stats.all_gathers : g_stats.broadcasts)++;
    return ncclSuccess;
}

// every rank has posted: the k-th call of each rank is one collective
ncclResult_t run_round(Clique *c) {
    const size_t n_ops = c->posted[0].size();
    for (int r = 1; r < c->world; ++r) {
        if (c->posted[r].size() != n_ops) {
            char msg[160];
            std::snprintf(msg, sizeof msg, "rank 0 posted %zu collectives in this group, rank %d posted %zu", n_ops, r, c->posted[r].size());
            return fail(ncclInvalidArgument, msg);
        }
    }
    std::vector<Op> op((size_t)c->world);
    for (size_t k = 0; k < n_ops; ++k) {
        for (int r = 0; r < c->world; ++r) op[(size_t)r] = c->posted[(size_t)r][k];
        ncclResult_t rc = perform(c, op);
        if (rc != ncclSuccess) return rc;
    }
    return ncclSuccess;
}

thread_local int t_depth = 0;
thread_local bool t_group_failed = false;         // a call inside the open group was refused: the group is void
thread_local std::vector<ncclComm *> t_touched;   // communicators with queued calls, in first-use order

ncclResult_t flush_group() {
    std::vector<ncclComm *> touched;
    touched.swap(t_touched);
    // post everything this thread queued (a thread that owns all ranks of a communicator completes the round itself)
    for (ncclComm *comm : touched) {
        Clique *c = comm->clique;
        std::unique_lock<std::mutex> l(c->m);
        c->posted[(size_t)comm->rank] = std::move(comm->pending);
        comm->pending.clear();
        comm->post_epoch = c->epoch;
        if (++c->arrived == c->world) {
            c->last_result = run_round(c);
            for (auto &p : c->posted) p.clear();
            c->arrived = 0;
            c->epoch++;
            c->cv.notify_all();
        }
    }
    ncclResult_t result = ncclSuccess;
    for (ncclComm *comm : touched) {
        Clique *c = comm->clique;
        std::unique_lock<std::mutex> l(c->m);
        c->cv.wait(l, [&] { return c->epoch > comm->post_epoch; });
        if (c->last_result != ncclSuccess) result = c->last_result;
    }
    g_stats.groups++;
    return result;
}

ncclResult_t enqueue(ncclComm *comm, const Op &op) {
    if (!comm || !comm->clique) return fail(ncclInvalidArgument, "null communicator");
    bool seen = false;
    for (ncclComm *c : t_touched) seen = seen || c == comm;
    if (!seen) t_touched.push_back(comm);
    comm->pending.push_back(op);
    if (t_depth == 0) return flush_group();   // a call outside a group is a group of its own
    return ncclSuccess;
}

std::mutex g_registry_lock;
std::map<std::string, Clique *> g_registry;   // communicators being formed by ncclCommInitRank, by id
std::atomic<unsigned> g_next_id{1};

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) {
    if (!version) return ncclInvalidArgument;
    *version = NCCL_VERSION_CODE;
    return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    if (injected(6)) return fail(ncclSystemError, "injected failure (ncclGetUniqueId)");
    std::memset(id->internal, 0, sizeof id->internal);
    std::snprintf(id->internal, sizeof id->internal, "fake_rccl:%u", g_next_id.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
    if (!comms || ndev <= 0) return fail(ncclInvalidArgument, "ncclCommInitAll: bad argument");
    if (injected(4)) return fail(ncclSystemError, "injected failure (ncclCommInitAll)");
    Clique *c = new Clique();
    c->world = ndev;
    c->member.resize((size_t)ndev);
    c->posted.resize((size_t)ndev);
    c->joined = c->alive = ndev;
    for (int r = 0; r < ndev; ++r) {
        ncclComm *m = new ncclComm();
        m->clique = c;
        m->rank = r;
        m->device = devlist ? devlist[r] : r;   // (the same device may appear several times: that is the point)
        c->member[(size_t)r] = m;
        comms[r] = m;
        g_stats.comms_created++;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks <= 0 || rank < 0 || rank >= nranks) return fail(ncclInvalidArgument, "ncclCommInitRank: bad argument");
    if (injected(5)) return fail(ncclSystemError, "injected failure (ncclCommInitRank)");
    const std::string key(id.internal, sizeof id.internal);
    Clique *c = nullptr;
    {
        std::lock_guard<std::mutex> l(g_registry_lock);
        auto it = g_registry.find(key);
        if (it == g_registry.end()) {
            c = new Clique();
            c->world = nranks;
            c->member.assign((size_t)nranks, nullptr);
            c->posted.resize((size_t)nranks);
            g_registry[key] = c;
        } else {
            c = it->second;
        }
    }
    ncclComm *m = new ncclComm();
    m->clique = c;
    m->rank = rank;
    if (hipGetDevice(&m->device) != hipSuccess) m->device = 0;
    std::unique_lock<std::mutex> l(c->m);
    if (c->world != nranks || c->member[(size_t)rank]) {
        l.unlock();
        delete m;
        return fail(ncclInvalidArgument, "ncclCommInitRank: rank joined twice or the ranks disagree about the world size");
    }
    c->member[(size_t)rank] = m;
    c->alive++;
    if (++c->joined == nranks) {
        std::lock_guard<std::mutex> rl(g_registry_lock);
        g_registry.erase(key);
        c->cv.notify_all();
    } else {
        c->cv.wait(l, [&] { return c->joined == nranks; });   // RCCL blocks here until every rank has arrived, too
    }
    *comm = m;
    g_stats.comms_created++;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    Clique *c = comm->clique;
    bool last = false;
    {
        std::lock_guard<std::mutex> l(c->m);
        c->member[(size_t)comm->rank] = nullptr;
        last = --c->alive == 0;
    }
    delete comm;
    if (last) {
        for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
        delete c;
    }
    g_stats.comms_destroyed++;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->clique->world;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = comm->rank;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    if (injected(1)) { t_group_failed = t_depth > 0; return fail(ncclInternalError, "injected failure (ncclAllGather)"); }
    return enqueue(comm, Op{kAllGather, sendbuff, recvbuff, sendcount, datatype, 0, stream});
}

#ifndef FAKE_RCCL_OMIT_BROADCAST   // (the "broken library" build of tests/fake_rccl/Makefile: a symbol the product needs is missing)
ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream) {
    if (injected(2)) { t_group_failed = t_depth > 0; return fail(ncclInternalError, "injected failure (ncclBroadcast)"); }
    return enqueue(comm, Op{kBroadcast, sendbuff, recvbuff, count, datatype, root, stream});
}
#endif

ncclResult_t ncclGroupStart() {
    if (injected(7)) return fail(ncclInternalError, "injected failure (ncclGroupStart)");
    ++t_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return fail(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart");
    if (--t_depth > 0) return ncclSuccess;
    // a group one of whose calls was refused performs nothing (the caller has the refused call's code already); the
    // ranks that did queue must not be left waiting for the one that could not
    const bool refused = t_group_failed, inject = injected(3);
    t_group_failed = false;
    if (refused || inject) {
        for (ncclComm *c : t_touched) c->pending.clear();
        t_touched.clear();
        return inject ? fail(ncclInternalError, "injected failure (ncclGroupEnd)") : ncclInternalError;
    }
    return flush_group();
}

const char *ncclGetErrorString(ncclResult_t result) {
    static thread_local std::string text;
    const char *base = "unknown result code";
    switch (result) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: base = "unhandled cuda error"; break;
        case ncclSystemError: base = "unhandled system error"; break;
        case ncclInternalError: base = "internal error"; break;
        case ncclInvalidArgument: base = "invalid argument"; break;
        case ncclInvalidUsage: base = "invalid usage"; break;
        default: break;
    }
    std::lock_guard<std::mutex> l(g_detail_lock);
    text = std::string(base) + " [fake_rccl: " + g_detail + "]";
    return text.c_str();
}

// ---- for the tests only (not RCCL entry points) ------------------------------------------------------------------------
// proof that THIS library is the one the product bound
int fake_rccl_marker(void) { return 0x5EED; }
// out[0..7] = all-gathers, broadcasts, groups, copies enqueued, in-place slots skipped, bytes copied, communicators
// created, communicators destroyed
void fake_rccl_stats(long long out[8]) {
    out[0] = g_stats.all_gathers; out[1] = g_stats.broadcasts; out[2] = g_stats.groups; out[3] = g_stats.copies;
    out[4] = g_stats.inplace_skips; out[5] = g_stats.bytes; out[6] = g_stats.comms_created; out[7] = g_stats.comms_destroyed;
}
// the `countdown`-th next call of entry point `which` (see g_fail_which) fails; which = 0 clears
void fake_rccl_fail(int which, int countdown) {
    g_fail_countdown.store(countdown);
    g_fail_which.store(which);
}

}  // extern "C"
