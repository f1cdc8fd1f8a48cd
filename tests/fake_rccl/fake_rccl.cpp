// fake_rccl -- a stand-in for librccl.so.1.  TEST INFRASTRUCTURE ONLY.
//
// Why it exists: the device-group code of the product (sponge_amd/csrc/pmx_mgpu.cpp) binds RCCL with
// dlopen("librccl.so.1").  The development boxes have ONE GPU and RCCL refuses two ranks on one device, so every
// world > 1 branch of that file - shard offsets, the `mine` test of the ragged gather, the d_top layout of the sharded
// tree, the host fan-out with more than one worker thread - could not execute there.  This library exports the fourteen
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
//   ncclSend(send, count, peer) / ncclRecv(recv, count, peer)   point to point: the i-th message rank a ever sends to rank b is the
//                                      i-th rank b ever receives from rank a (same count and type, or both calls are refused).  Only the
//                                      two ranks of a message take part - a rank with nothing to send or receive posts nothing, as with
//                                      RCCL.  A rank's group returns when all its messages have been matched and enqueued; a message
//                                      whose other half does not appear within 30 s is an error where RCCL would hang.  A group holds
//                                      collectives or point-to-point calls, not both (the product never mixes them)
//   ncclGroupStart / ncclGroupEnd      calls inside a group are queued per communicator and performed at the outermost
//                                      GroupEnd; the k-th queued call of every rank of a communicator is one collective
//   ncclCommInitAll                    one thread owns all ranks;   ncclCommInitRank   ranks are joined by id, inside ONE
//                                      process (threads), and the call blocks until all of them have arrived, as RCCL does
//
// Ranks in DIFFERENT processes (FAKE_RCCL_XPROC=1 in the process that calls ncclGetUniqueId; the id then names a POSIX
// shared-memory object): what `bench.py --gpus N` under torch.distributed.run and tests/mgpu_rank_worker.py do.  The ranks
// meet in the shared object (descriptor table + a sense-reversing barrier, two-minute time-outs instead of hangs), compare
// what they posted exactly as above, and the payload goes through one shared-memory object per rank: owner's stream
// drained, device -> object, barrier, object -> every reader's receive buffer on the reader's stream, drained, barrier.
// Slow and fully synchronous - it is a rehearsal stage, not a transport.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

enum Kind { kAllGather = 0, kBroadcast = 1, kSend = 2, kRecv = 3 };   // (kSend / kRecv: `root` is the peer)
inline bool is_p2p(Kind k) { return k == kSend || k == kRecv; }

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
    // point to point: halves of messages whose other half has not been posted yet, by (source, destination), oldest first
    std::map<std::pair<int, int>, std::deque<struct Half *>> open_sends, open_recvs;
};

// one posted ncclSend / ncclRecv of a rank of a clique (owned by the posting communicator until matched or given up)
struct Half {
    Op op;
    int rank = 0;
    bool matched = false;
    ncclResult_t rc = ncclSuccess;
};
constexpr double kP2PTimeout = 30.0;   // seconds a posted half waits for the other one

}  // namespace

namespace {
struct XComm;
}
struct ncclComm {
    Clique *clique = nullptr;                // ranks of one process
    XComm *x = nullptr;                      // ranks of several processes (exactly one of the two is set)
    int rank = 0, world = 0;
    int device = 0;
    std::vector<Op> pending;                 // queued inside a group by the owning thread
    uint64_t post_epoch = 0;
    std::vector<std::unique_ptr<Half>> halves;   // the point-to-point calls of the group being flushed
};

namespace {

// ---- statistics and fault injection (read by the tests through fake_rccl_* below) --------------------------------------
struct Stats {
    std::atomic<long> all_gathers{0}, broadcasts{0}, groups{0}, copies{0}, inplace_skips{0}, comms_created{0}, comms_destroyed{0}, messages{0};
    std::atomic<long long> bytes{0};
};
Stats g_stats;
// fail the n-th next call (1 = the very next) of one entry point: 0 none, 1 AllGather, 2 Broadcast, 3 GroupEnd,
// 4 CommInitAll, 5 CommInitRank, 6 GetUniqueId, 7 GroupStart, 8 Send, 9 Recv
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

// one message: both halves are posted.  Called with the clique's lock held, by whichever rank posted second.
void deliver(Clique *c, Half *sh, Half *rh) {
    const Op &so = sh->op, &ro = rh->op;
    const int s = sh->rank, d = rh->rank;
    auto both = [&](ncclResult_t rc) { sh->rc = rh->rc = rc; sh->matched = rh->matched = true; };
    if (so.count != ro.count || so.type != ro.type) {
        char msg[200];
        std::snprintf(msg, sizeof msg, "a message from rank %d to rank %d: sent as %zu elements, received as %zu", s, d, so.count, ro.count);
        return both(fail(ncclInvalidArgument, msg));
    }
    const size_t esz = type_bytes(so.type), bytes = so.count * esz;
    if (esz == 0) return both(fail(ncclInvalidArgument, "unsupported datatype"));
    ncclResult_t rc;
    { DeviceScope ds(c->member[(size_t)s]->device); rc = check_range(so.send, bytes, "ncclSend sendbuff", s); }
    if (rc == ncclSuccess) { DeviceScope ds(c->member[(size_t)d]->device); rc = check_range(ro.recv, bytes, "ncclRecv recvbuff", d); }
    if (rc != ncclSuccess) return both(rc);
    if (bytes == 0) return both(ncclSuccess);
    auto enqueue_copy = [&]() -> ncclResult_t {
        hipEvent_t ready = nullptr, done = nullptr;
        {
            DeviceScope ds(c->member[(size_t)s]->device);
            FAKE_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
            c->events.push_back(ready);
            FAKE_HIP(hipEventRecord(ready, so.stream));          // the send buffer holds the data (the sender's thread is inside its group: nothing newer is on its stream)
        }
        {
            DeviceScope ds(c->member[(size_t)d]->device);
            FAKE_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
            c->events.push_back(done);
            FAKE_HIP(hipStreamWaitEvent(ro.stream, ready, 0));
            FAKE_HIP(hipMemcpyAsync(ro.recv, so.send, bytes, hipMemcpyDeviceToDevice, ro.stream));
            FAKE_HIP(hipEventRecord(done, ro.stream));
        }
        {
            DeviceScope ds(c->member[(size_t)s]->device);
            FAKE_HIP(hipStreamWaitEvent(so.stream, done, 0));     // the sender's stream may not overwrite the buffer before it was read
        }
        return ncclSuccess;
    };
    rc = enqueue_copy();
    if (rc == ncclSuccess) {
        g_stats.copies++;
        g_stats.messages++;
        g_stats.bytes += (long long)bytes;
    }
    both(rc);
}

// a rank posts the point-to-point calls of its group: every half whose other half is already there is delivered now
ncclResult_t post_halves(ncclComm *comm, std::vector<Op> &&ops) {
    Clique *c = comm->clique;
    std::lock_guard<std::mutex> l(c->m);
    for (const Op &o : ops) {
        if (o.root < 0 || o.root >= c->world) return fail(ncclInvalidArgument, "peer " + std::to_string(o.root) + " out of range on rank " + std::to_string(comm->rank));
        comm->halves.emplace_back(new Half{o, comm->rank});
        Half *h = comm->halves.back().get();
        const bool send = o.kind == kSend;
        const std::pair<int, int> key = send ? std::make_pair(comm->rank, o.root) : std::make_pair(o.root, comm->rank);
        auto &mine = send ? c->open_sends[key] : c->open_recvs[key];
        auto &other = send ? c->open_recvs[key] : c->open_sends[key];
        if (mine.empty() && !other.empty()) {      // (mine not empty: an older half of this pair is still waiting - order is kept)
            Half *o2 = other.front();
            other.pop_front();
            deliver(c, send ? h : o2, send ? o2 : h);
        } else {
            mine.push_back(h);
        }
    }
    c->cv.notify_all();
    return ncclSuccess;
}

// ... and waits until all of them have been (a half nobody answers within kP2PTimeout is given up: RCCL would hang there)
ncclResult_t wait_halves(ncclComm *comm) {
    Clique *c = comm->clique;
    std::unique_lock<std::mutex> l(c->m);
    auto all_matched = [&] { for (auto &h : comm->halves) if (!h->matched) return false; return true; };
    const bool in_time = c->cv.wait_for(l, std::chrono::duration<double>(kP2PTimeout), all_matched);
    ncclResult_t result = ncclSuccess;
    if (!in_time) {
        int left = 0;
        for (auto &h : comm->halves) {
            if (h->matched) continue;
            ++left;
            const bool send = h->op.kind == kSend;
            auto &q = send ? c->open_sends[{comm->rank, h->op.root}] : c->open_recvs[{h->op.root, comm->rank}];
            for (auto it = q.begin(); it != q.end(); ++it) if (*it == h.get()) { q.erase(it); break; }
        }
        result = fail(ncclInvalidUsage, "rank " + std::to_string(comm->rank) + ": " + std::to_string(left) + " point-to-point calls of a group found no partner within 30 s (RCCL would hang here)");
    }
    for (auto &h : comm->halves) if (h->matched && h->rc != ncclSuccess && result == ncclSuccess) result = h->rc;
    comm->halves.clear();
    return result;
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

ncclResult_t x_perform(ncclComm *c, const std::vector<Op> &ops);   // ranks in different processes, below

thread_local int t_depth = 0;
thread_local bool t_group_failed = false;         // a call inside the open group was refused: the group is void
thread_local std::vector<ncclComm *> t_touched;   // communicators with queued calls, in first-use order

ncclResult_t flush_group() {
    std::vector<ncclComm *> all, touched;
    all.swap(t_touched);
    ncclResult_t x_result = ncclSuccess;
    for (ncclComm *comm : all) {
        if (!comm->x) { touched.push_back(comm); continue; }
        std::vector<Op> ops;
        ops.swap(comm->pending);
        const ncclResult_t rc = x_perform(comm, ops);
        if (rc != ncclSuccess) x_result = rc;
    }
    // point-to-point groups: only the ranks of a message take part - every communicator of this thread posts its halves, then waits for them
    // (a thread that owns both ranks of a message has delivered it by the time it waits)
    ncclResult_t p2p_result = ncclSuccess;
    {
        std::vector<ncclComm *> collectives, p2p;
        for (ncclComm *comm : touched) {
            size_t n = 0;
            for (const Op &o : comm->pending) n += is_p2p(o.kind);
            if (n && n != comm->pending.size()) {
                comm->pending.clear();
                p2p_result = fail(ncclInvalidUsage, "a group with collectives AND point-to-point calls: not something the stand-in models");
            } else (n ? p2p : collectives).push_back(comm);
        }
        for (ncclComm *comm : p2p) {
            std::vector<Op> ops;
            ops.swap(comm->pending);
            const ncclResult_t rc = post_halves(comm, std::move(ops));
            if (rc != ncclSuccess) p2p_result = rc;
        }
        for (ncclComm *comm : p2p) {
            const ncclResult_t rc = wait_halves(comm);
            if (rc != ncclSuccess) p2p_result = rc;
        }
        touched.swap(collectives);
    }
    if (p2p_result != ncclSuccess) x_result = p2p_result;
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
    ncclResult_t result = x_result;
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
    if (!comm || (!comm->clique && !comm->x)) return fail(ncclInvalidArgument, "null communicator");
    bool seen = false;
    for (ncclComm *c : t_touched) seen = seen || c == comm;
    if (!seen) t_touched.push_back(comm);
    comm->pending.push_back(op);
    if (t_depth == 0) return flush_group();   // a call outside a group is a group of its own
    return ncclSuccess;
}

// ---- ranks in different processes ---------------------------------------------------------------------------------------
constexpr int kXMaxWorld = 64;
constexpr unsigned kXMagic = 0xFA4ECC17u;
constexpr double kXTimeout = 120.0;   // seconds a rank waits for the others before it gives up (a hang helps nobody)

struct XDesc {
    int kind, root, type, rc;
    unsigned long long count, n_ops;
};
constexpr int kXRing = 32;           // messages one rank may have in flight to one peer
struct XMessage {
    unsigned long long count, offset;   // offset into the sender's staging object
    int type, rc;                       // rc: the receiver's verdict, written before it counts the message as taken
};
struct XPair {                          // source -> destination
    std::atomic<unsigned long long> sent, taken;   // messages ever posted by the source / consumed by the destination
    XMessage ring[kXRing];
};
struct XShared {
    std::atomic<unsigned> magic;
    int world;
    std::atomic<int> joined, left;
    std::atomic<int> bar_count, bar_sense;
    XDesc desc[kXMaxWorld];
    XPair pair[kXMaxWorld][kXMaxWorld]; // [source][destination] (pages nobody touches are never backed)
};
struct XComm {
    XShared *sh = nullptr;
    std::string token;                        // the objects are /fake_rccl_<token> and /fake_rccl_<token>_r<rank>
    int sense = 0;
    int data_fd = -1;
    void *data = nullptr;
    size_t data_bytes = 0;
};

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

ncclResult_t x_barrier(ncclComm *c, const char *where) {
    XShared *sh = c->x->sh;
    const int s = (c->x->sense ^= 1);
    if (sh->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == sh->world) {
        sh->bar_count.store(0, std::memory_order_relaxed);
        sh->bar_sense.store(s, std::memory_order_release);
        return ncclSuccess;
    }
    const double t0 = now_s();
    for (unsigned spin = 0; sh->bar_sense.load(std::memory_order_acquire) != s; ++spin) {
        if (spin > 1000) usleep(100);
        if ((spin & 1023) == 1023 && now_s() - t0 > kXTimeout)
            return fail(ncclSystemError, std::string("rank ") + std::to_string(c->rank) + " waited two minutes for the other ranks at " + where);
    }
    return ncclSuccess;
}

std::string x_data_name(const std::string &token, int rank) { return "/fake_rccl_" + token + "_r" + std::to_string(rank); }

// this rank's staging object, at least `bytes` long
ncclResult_t x_stage(ncclComm *c, size_t bytes) {
    XComm *x = c->x;
    if (x->data_bytes >= bytes) return ncclSuccess;
    if (x->data) munmap(x->data, x->data_bytes);
    x->data = nullptr;
    x->data_bytes = 0;
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    if (ftruncate(x->data_fd, (off_t)want) != 0) return fail(ncclSystemError, std::string("ftruncate of the staging object: ") + std::strerror(errno));
    void *m = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_SHARED, x->data_fd, 0);
    if (m == MAP_FAILED) return fail(ncclSystemError, std::string("mmap of the staging object: ") + std::strerror(errno));
    x->data = m;
    x->data_bytes = want;
    return ncclSuccess;
}

// copy `bytes` of rank j's staging object to dst (device) on st
ncclResult_t x_fetch(ncclComm *c, int j, void *dst, size_t bytes, hipStream_t st, size_t offset = 0) {
    XComm *x = c->x;
    if (j == c->rank) {
        FAKE_HIP(hipMemcpyAsync(dst, (const char *)x->data + offset, bytes, hipMemcpyHostToDevice, st));
        FAKE_HIP(hipStreamSynchronize(st));
        return ncclSuccess;
    }
    const int fd = shm_open(x_data_name(x->token, j).c_str(), O_RDONLY, 0);
    if (fd < 0) return fail(ncclSystemError, "rank " + std::to_string(j) + "'s staging object cannot be opened: " + std::strerror(errno));
    void *m = mmap(nullptr, offset + bytes, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail(ncclSystemError, std::string("mmap of a peer's staging object: ") + std::strerror(errno));
    hipError_t e = hipMemcpyAsync(dst, (const char *)m + offset, bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    munmap(m, offset + bytes);
    if (e != hipSuccess) return fail(ncclUnhandledCudaError, std::string("copy from a peer's staging object: ") + hipGetErrorString(e));
    return ncclSuccess;
}

// a group of point-to-point calls, ranks in different processes.  Only the two ranks of a message meet: the sender stages the data in its
// own object and lists the message in the pair's ring; the receiver waits for the ring entry, takes the data out of the sender's object,
// leaves its verdict and counts the message as taken; the sender returns when all its messages were taken.
ncclResult_t x_p2p(ncclComm *c, const std::vector<Op> &ops) {
    XShared *sh = c->x->sh;
    const int W = c->world, me = c->rank;
    ncclResult_t mine = ncclSuccess;
    size_t total = 0;
    std::vector<size_t> offset(ops.size(), 0);
    for (size_t k = 0; k < ops.size() && mine == ncclSuccess; ++k) {
        const Op &o = ops[k];
        const size_t esz = type_bytes(o.type), bytes = o.count * esz;
        if (esz == 0) mine = fail(ncclInvalidArgument, "unsupported datatype");
        else if (o.root < 0 || o.root >= W) mine = fail(ncclInvalidArgument, "peer " + std::to_string(o.root) + " out of range on rank " + std::to_string(me));
        else if (o.kind == kSend) {
            mine = check_range(o.send, bytes, "ncclSend sendbuff", me);
            offset[k] = total;
            total += (bytes + 15) & ~(size_t)15;
        } else {
            mine = check_range(o.recv, bytes, "ncclRecv recvbuff", me);
        }
    }
    if (mine != ncclSuccess) return mine;
    if (total && (mine = x_stage(c, total)) != ncclSuccess) return mine;
    // post the sends
    std::vector<std::pair<int, unsigned long long>> posted;   // (destination, sequence number)
    for (size_t k = 0; k < ops.size(); ++k) {
        const Op &o = ops[k];
        if (o.kind != kSend) continue;
        const size_t bytes = o.count * type_bytes(o.type);
        hipError_t e = hipStreamSynchronize(o.stream);            // everything enqueued before the call has happened
        if (e == hipSuccess && bytes) e = hipMemcpy((char *)c->x->data + offset[k], o.send, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(ncclUnhandledCudaError, std::string("staging a send buffer: ") + hipGetErrorString(e));
        XPair &p = sh->pair[me][o.root];
        const unsigned long long seq = p.sent.load(std::memory_order_relaxed);
        if (seq - p.taken.load(std::memory_order_acquire) >= (unsigned long long)kXRing)
            return fail(ncclInvalidUsage, "more messages in flight to one peer than the stand-in's ring holds");
        p.ring[seq % kXRing] = XMessage{o.count, offset[k], (int)o.type, 0};
        p.sent.store(seq + 1, std::memory_order_release);
        posted.emplace_back(o.root, seq);
    }
    auto wait_for = [&](std::atomic<unsigned long long> &counter, unsigned long long above, const std::string &what) -> ncclResult_t {
        const double t0 = now_s();
        for (unsigned spin = 0; counter.load(std::memory_order_acquire) <= above; ++spin) {
            if (spin > 1000) usleep(100);
            if ((spin & 1023) == 1023 && now_s() - t0 > 30.0) return fail(ncclInvalidUsage, "rank " + std::to_string(me) + ": " + what + " within 30 s (RCCL would hang here)");
        }
        return ncclSuccess;
    };
    // take what the receives name
    for (const Op &o : ops) {
        if (o.kind != kRecv) continue;
        XPair &p = sh->pair[o.root][me];
        const unsigned long long seq = p.taken.load(std::memory_order_relaxed);
        ncclResult_t rc = wait_for(p.sent, seq, "a receive from rank " + std::to_string(o.root) + " found no send");
        if (rc != ncclSuccess) { mine = rc; break; }
        XMessage &m = p.ring[seq % kXRing];
        ncclResult_t verdict = ncclSuccess;
        if (m.count != o.count || m.type != (int)o.type) {
            char msg[200];
            std::snprintf(msg, sizeof msg, "a message from rank %d to rank %d: sent as %llu elements, received as %zu", o.root, me, m.count, o.count);
            verdict = fail(ncclInvalidArgument, msg);
        } else if (o.count) {
            const size_t bytes = o.count * type_bytes(o.type);
            if (hipStreamSynchronize(o.stream) != hipSuccess) verdict = fail(ncclUnhandledCudaError, "the receiver's stream failed");
            else verdict = x_fetch(c, o.root, o.recv, bytes, o.stream, (size_t)m.offset);
            if (verdict == ncclSuccess) {
                g_stats.copies++;
                g_stats.messages++;
                g_stats.bytes += (long long)bytes;
            }
        }
        m.rc = (int)verdict;
        p.taken.store(seq + 1, std::memory_order_release);
        if (verdict != ncclSuccess && mine == ncclSuccess) mine = verdict;
    }
    // the staging object may be reused once every message was taken
    for (const auto &ps : posted) {
        XPair &p = sh->pair[me][ps.first];
        ncclResult_t rc = wait_for(p.taken, ps.second, "a send to rank " + std::to_string(ps.first) + " found no receive");
        if (rc != ncclSuccess) { if (mine == ncclSuccess) mine = rc; continue; }
        const int verdict = p.ring[ps.second % kXRing].rc;
        if (verdict != 0 && mine == ncclSuccess) mine = fail((ncclResult_t)verdict, "rank " + std::to_string(ps.first) + " refused a message (its process has the reason)");
    }
    return mine;
}

// the calls one rank queued in a group (or one call outside a group); every rank of the communicator is in here with its own
ncclResult_t x_perform(ncclComm *c, const std::vector<Op> &ops) {
    XShared *sh = c->x->sh;
    const int W = c->world, me = c->rank;
    ncclResult_t rc;
    size_t p2p = 0;
    for (const Op &o : ops) p2p += is_p2p(o.kind);
    if (p2p && p2p != ops.size()) return fail(ncclInvalidUsage, "a group with collectives AND point-to-point calls: not something the stand-in models");
    if (p2p) return x_p2p(c, ops);      // (only the ranks of a message meet: no barrier over the communicator)
    sh->desc[me].n_ops = ops.size();
    if ((rc = x_barrier(c, "the start of a group")) != ncclSuccess) return rc;
    ncclResult_t verdict = ncclSuccess;
    for (int r = 0; r < W; ++r)
        if (sh->desc[r].n_ops != ops.size() && verdict == ncclSuccess) {
            char msg[160];
            std::snprintf(msg, sizeof msg, "rank %d posted %zu collectives in this group, rank %d posted %llu", me, ops.size(), r, sh->desc[r].n_ops);
            verdict = fail(ncclInvalidArgument, msg);
        }
    if ((rc = x_barrier(c, "the group's size check")) != ncclSuccess) return rc;
    if (verdict != ncclSuccess) return verdict;
    for (const Op &o : ops) {
        const size_t esz = type_bytes(o.type);
        const size_t bytes = o.count * esz;
        const bool sender = o.kind == kAllGather || me == o.root;
        ncclResult_t mine = ncclSuccess;
        if (esz == 0) mine = fail(ncclInvalidArgument, "unsupported datatype");
        else if (o.kind == kBroadcast && (o.root < 0 || o.root >= W)) mine = fail(ncclInvalidArgument, "broadcast root out of range");
        else if (o.kind == kAllGather) {
            mine = check_range(o.send, bytes, "ncclAllGather sendbuff", me);
            if (mine == ncclSuccess) mine = check_range(o.recv, bytes * (size_t)W, "ncclAllGather recvbuff", me);
        } else {
            if (sender) mine = check_range(o.send, bytes, "ncclBroadcast sendbuff (root)", me);
            if (mine == ncclSuccess) mine = check_range(o.recv, bytes, "ncclBroadcast recvbuff", me);
        }
        if (mine == ncclSuccess && bytes) {
            hipError_t e = hipStreamSynchronize(o.stream);            // everything enqueued before the call has happened
            if (e == hipSuccess && sender) {
                mine = x_stage(c, bytes);
                if (mine == ncclSuccess) e = hipMemcpy(c->x->data, o.send, bytes, hipMemcpyDeviceToHost);
            }
            if (e != hipSuccess) mine = fail(ncclUnhandledCudaError, std::string("staging the send buffer: ") + hipGetErrorString(e));
        }
        sh->desc[me].kind = (int)o.kind;
        sh->desc[me].root = o.root;
        sh->desc[me].type = (int)o.type;
        sh->desc[me].count = o.count;
        sh->desc[me].rc = (int)mine;
        if ((rc = x_barrier(c, "a collective (posted)")) != ncclSuccess) return rc;
        verdict = mine;
        for (int r = 0; r < W && verdict == ncclSuccess; ++r) {
            const XDesc &d = sh->desc[r];
            if (d.rc != 0) verdict = fail((ncclResult_t)d.rc, "rank " + std::to_string(r) + " refused the call (its process has the reason)");
            else if (d.kind != (int)o.kind || d.count != o.count || d.type != (int)o.type || d.root != o.root) {
                char msg[256];
                std::snprintf(msg, sizeof msg, "ranks %d and %d posted different collectives (kind %d/%d, count %zu/%llu, root %d/%d)", me, r,
                              (int)o.kind, d.kind, o.count, d.count, o.root, d.root);
                verdict = fail(ncclInvalidArgument, msg);
            }
        }
        if (verdict == ncclSuccess && bytes) {
            if (o.kind == kAllGather) {
                for (int j = 0; j < W && verdict == ncclSuccess; ++j) {
                    char *dst = (char *)o.recv + (size_t)j * bytes;
                    if (j == me && (const void *)dst == o.send) { g_stats.inplace_skips++; continue; }
                    verdict = x_fetch(c, j, dst, bytes, o.stream);
                    g_stats.copies++;
                    g_stats.bytes += (long long)bytes;
                }
            } else if (me == o.root && o.recv == o.send) {
                g_stats.inplace_skips++;
            } else {
                verdict = x_fetch(c, o.root, o.recv, bytes, o.stream);
                g_stats.copies++;
                g_stats.bytes += (long long)bytes;
            }
        }
        if ((rc = x_barrier(c, "a collective (delivered)")) != ncclSuccess) return rc;   // the staging objects may be reused
        if (verdict != ncclSuccess) return verdict;
        (o.kind == kAllGather ? g_stats.all_gathers : g_stats.broadcasts)++;
    }
    return ncclSuccess;
}

ncclResult_t x_join(ncclComm *m, int nranks, const std::string &token) {
    if (nranks > kXMaxWorld) return fail(ncclInvalidArgument, "the cross-process stand-in holds at most 64 ranks");
    XComm *x = new XComm();
    x->token = token;
    m->x = x;
    const std::string name = "/fake_rccl_" + token;
    bool creator = true;
    int fd = shm_open(name.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 && errno == EEXIST) {
        creator = false;
        fd = shm_open(name.c_str(), O_RDWR, 0600);
    }
    if (fd < 0) return fail(ncclSystemError, "shm_open(" + name + "): " + std::strerror(errno));
    if (creator && ftruncate(fd, (off_t)sizeof(XShared)) != 0) {
        close(fd);
        return fail(ncclSystemError, std::string("ftruncate of the rendezvous object: ") + std::strerror(errno));
    }
    const double t0 = now_s();
    if (!creator) {   // the creator may not have sized it yet
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < sizeof(XShared)) {
            if (now_s() - t0 > kXTimeout) { close(fd); return fail(ncclSystemError, "the rendezvous object was never sized"); }
            usleep(1000);
        }
    }
    void *mem = mmap(nullptr, sizeof(XShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) return fail(ncclSystemError, std::string("mmap of the rendezvous object: ") + std::strerror(errno));
    XShared *sh = static_cast<XShared *>(mem);
    x->sh = sh;
    if (creator) {   // a fresh object is all zeros: the atomics start at 0
        sh->world = nranks;
        sh->magic.store(kXMagic, std::memory_order_release);
    } else {
        while (sh->magic.load(std::memory_order_acquire) != kXMagic) {
            if (now_s() - t0 > kXTimeout) return fail(ncclSystemError, "the rendezvous object was never initialised");
            usleep(1000);
        }
    }
    if (sh->world != nranks) return fail(ncclInvalidArgument, "ncclCommInitRank: the ranks disagree about the world size");
    x->data_fd = shm_open(x_data_name(token, m->rank).c_str(), O_RDWR | O_CREAT, 0600);
    if (x->data_fd < 0) return fail(ncclSystemError, std::string("shm_open of the staging object: ") + std::strerror(errno));
    sh->joined.fetch_add(1, std::memory_order_acq_rel);
    while (sh->joined.load(std::memory_order_acquire) < nranks) {   // RCCL blocks here until every rank has arrived, too
        if (now_s() - t0 > kXTimeout) return fail(ncclSystemError, "ncclCommInitRank: not every rank arrived within two minutes");
        usleep(200);
    }
    return ncclSuccess;
}

void x_leave(ncclComm *m) {
    XComm *x = m->x;
    if (x->data) munmap(x->data, x->data_bytes);
    if (x->data_fd >= 0) {
        close(x->data_fd);
        shm_unlink(x_data_name(x->token, m->rank).c_str());
    }
    if (x->sh) {
        if (x->sh->left.fetch_add(1, std::memory_order_acq_rel) + 1 >= x->sh->world) shm_unlink(("/fake_rccl_" + x->token).c_str());
        munmap(x->sh, sizeof(XShared));
    }
    delete x;
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
    const char *xproc = std::getenv("FAKE_RCCL_XPROC");
    if (xproc && xproc[0] == '1')   // the ranks will be processes: the id names the shared-memory object they meet in
        std::snprintf(id->internal, sizeof id->internal, "fake_rccl_x:%d_%u_%llx", (int)getpid(), g_next_id.fetch_add(1),
                      (unsigned long long)(now_s() * 1e6));
    else
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
        m->world = ndev;
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
    if (std::strncmp(id.internal, "fake_rccl_x:", 12) == 0) {
        ncclComm *m = new ncclComm();
        m->rank = rank;
        m->world = nranks;
        if (hipGetDevice(&m->device) != hipSuccess) m->device = 0;
        const ncclResult_t rc = x_join(m, nranks, std::string(id.internal + 12, strnlen(id.internal + 12, sizeof id.internal - 12)));
        if (rc != ncclSuccess) {
            if (m->x) x_leave(m);
            delete m;
            return rc;
        }
        *comm = m;
        g_stats.comms_created++;
        return ncclSuccess;
    }
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
    m->world = nranks;
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
    if (comm->x) {
        x_leave(comm);
        delete comm;
        g_stats.comms_destroyed++;
        return ncclSuccess;
    }
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
    *count = comm->world;
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

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (injected(8)) { t_group_failed = t_depth > 0; return fail(ncclInternalError, "injected failure (ncclSend)"); }
    return enqueue(comm, Op{kSend, sendbuff, nullptr, count, datatype, peer, stream});
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (injected(9)) { t_group_failed = t_depth > 0; return fail(ncclInternalError, "injected failure (ncclRecv)"); }
    return enqueue(comm, Op{kRecv, nullptr, recvbuff, count, datatype, peer, stream});
}

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
long long fake_rccl_messages(void) { return g_stats.messages; }   // point-to-point messages delivered
// the `countdown`-th next call of entry point `which` (see g_fail_which) fails; which = 0 clears
void fake_rccl_fail(int which, int countdown) {
    g_fail_countdown.store(countdown);
    g_fail_which.store(which);
}

}  // extern "C"
