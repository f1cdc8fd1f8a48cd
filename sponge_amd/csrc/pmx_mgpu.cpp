// Device groups: the batch sharded over the GPUs of one node, RCCL (over xGMI) for the final gather.
//
// Sponge states are independent - the reference has no cross-state data flow anywhere in src/poseidon/mod.rs:62-183 -
// so the batch is cut into contiguous shards [g n / G, (g+1) n / G), one per GPU, and the data path needs NO
// collective.  RCCL moves data in exactly two places: the final gather of the result shards - to every rank (ncclAllGather, or a
// group of ncclBroadcasts when the shards are ragged) or to one (grouped ncclSend / ncclRecv: 1 / world of the bytes per link),
// as one call behind the last step or piece by piece behind the pieces of that step - and the 32-byte subtree roots of the
// sharded Merkle reduction.
//
// A group is either every GPU of one process (pmx_mgpu_create: ncclCommInitAll, one host thread drives all devices)
// or one rank of a multi-process job (pmx_mgpu_create_rank: ncclCommInitRank with an id made by pmx_mgpu_unique_id
// and carried to the other processes by the caller).  Either way a group holds `n_local` (device, pmx_ctx, stream,
// communicator) slots with consecutive ranks starting at `first_rank`.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/poseidon_mi355x.h"
#include "pmx_ctx.hpp"

using namespace pmx;

static_assert(PMX_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the ABI's id size is RCCL's");

struct pmx_mgpu {
    int world = 0;        // ranks in the communicator
    int first_rank = 0;   // rank of local slot 0
    std::vector<int> device;
    std::vector<pmx_ctx *> ctx;       // owned (pmx_ctx_create)
    std::vector<ncclComm_t> comm;
    uint32_t t = 0;
};

// RCCL is bound when the first group is formed, not when the library is loaded: everything that runs on one GPU - the
// whole single-device ABI - neither needs librccl nor pays for mapping it.  dlopen by SONAME: a process that already
// holds a copy (PyTorch bundles one) gets that copy.
namespace {
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl &rccl_lib() {
    static Rccl *lib = [] {
        Rccl *r = new Rccl();
        const char *chosen = "librccl.so.1";
#ifdef PMX_TEST_HOOKS
        // (test build only) PMX_RCCL_LIBRARY=<path>: bind THAT collective library - the tests' stand-in for ranks that are separate
        // processes on one GPU (tests/fake_rccl).  Nothing else is tried when it is set.  The shipped library reads no environment
        // variable here: it binds RCCL by SONAME, and a site's own build is chosen the way any shared library is (the loader's search path).
        const char *named = std::getenv("PMX_RCCL_LIBRARY");
        if (named && *named) {
            chosen = named;
            r->handle = dlopen(named, RTLD_NOW | RTLD_LOCAL);
        } else
#endif
        {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r->handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r->handle) break;
            }
        }
        if (!r->handle) {
            const char *e = dlerror();
            r->error = std::string(chosen) + " could not be loaded: " + (e ? e : "unknown error");
            return r;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(r->handle, name);
            if (!p && r->error.empty()) r->error = std::string("librccl has no symbol ") + name;
            return p;
        };
        r->GetVersion = (decltype(r->GetVersion))sym("ncclGetVersion");
        r->GetUniqueId = (decltype(r->GetUniqueId))sym("ncclGetUniqueId");
        r->CommInitAll = (decltype(r->CommInitAll))sym("ncclCommInitAll");
        r->CommInitRank = (decltype(r->CommInitRank))sym("ncclCommInitRank");
        r->CommDestroy = (decltype(r->CommDestroy))sym("ncclCommDestroy");
        r->CommCount = (decltype(r->CommCount))sym("ncclCommCount");
        r->CommUserRank = (decltype(r->CommUserRank))sym("ncclCommUserRank");
        r->AllGather = (decltype(r->AllGather))sym("ncclAllGather");
        r->Broadcast = (decltype(r->Broadcast))sym("ncclBroadcast");
        r->Send = (decltype(r->Send))sym("ncclSend");
        r->Recv = (decltype(r->Recv))sym("ncclRecv");
        r->GroupStart = (decltype(r->GroupStart))sym("ncclGroupStart");
        r->GroupEnd = (decltype(r->GroupEnd))sym("ncclGroupEnd");
        r->GetErrorString = (decltype(r->GetErrorString))sym("ncclGetErrorString");
        return r;
    }();
    return *lib;
}
}  // namespace

// PMX_OK, or PMX_ERR_RCCL with the loader's message
static int rccl_ready() {
    const Rccl &r = rccl_lib();
    if (!r.error.empty()) return set_error(PMX_ERR_RCCL, "%s", r.error.c_str());
    return PMX_OK;
}

static int rccl_fail(ncclResult_t r, const char *what) {
    return set_error(PMX_ERR_RCCL, "%s: %s", what, rccl_lib().GetErrorString(r));
}

#define PMX_RCCL(expr)                                        \
    do {                                                      \
        ncclResult_t r_ = (expr);                             \
        if (r_ != ncclSuccess) return rccl_fail(r_, #expr);   \
    } while (0)

extern "C" int pmx_shard_bounds(size_t n, int world, int rank, size_t *start, size_t *count) {
    if (world <= 0 || rank < 0 || rank >= world || !start || !count) return set_error(PMX_ERR_ARG, "pmx_shard_bounds: bad argument");
    const size_t base = n / (size_t)world, extra = n % (size_t)world, r = (size_t)rank;
    *start = r * base + (r < extra ? r : extra);
    *count = base + (r < extra ? 1 : 0);
    return PMX_OK;
}

extern "C" int pmx_mgpu_unique_id(uint8_t id[PMX_UNIQUE_ID_BYTES]) {
    PMX_ABI_BEGIN("pmx_mgpu_unique_id")
    if (!id) return set_error(PMX_ERR_ARG, "pmx_mgpu_unique_id: null pointer");
    if (int rc = rccl_ready()) return rc;
    ncclUniqueId u;
    PMX_RCCL(rccl_lib().GetUniqueId(&u));
    std::memcpy(id, u.internal, PMX_UNIQUE_ID_BYTES);
    return PMX_OK;
    PMX_ABI_END
}

static void group_free(pmx_mgpu *g) {
    for (size_t l = 0; l < g->comm.size(); ++l) {
        if (g->comm[l]) {
            DeviceGuard guard(g->device[l]);
            (void)rccl_lib().CommDestroy(g->comm[l]);
        }
    }
    for (pmx_ctx *c : g->ctx)
        if (c) (void)pmx_ctx_destroy(c);
    delete g;
}

// owns a group under construction: whatever way the constructor leaves (error code or exception), the half-built group
// is torn down unless release() was reached
struct GroupHolder {
    pmx_mgpu *g;
    ~GroupHolder() { if (g) group_free(g); }
    pmx_mgpu *release() { pmx_mgpu *r = g; g = nullptr; return r; }
};

static int group_contexts(pmx_mgpu *g, const pmx_config *cfg) {
    for (size_t l = 0; l < g->device.size(); ++l) {
        pmx_ctx *c = nullptr;
        int rc = pmx_ctx_create(cfg, g->device[l], &c);
        if (rc) return rc;
        g->ctx[l] = c;
    }
    g->t = (uint32_t)pmx_ctx_width(g->ctx[0]);
    return PMX_OK;
}

// Test hooks (include/poseidon_mi355x_testing.h; not part of the product ABI, not in the Rust binding) exist ONLY in the test build of
// this file (-DPMX_TEST_HOOKS: build/pmx_mgpu_test.o -> libposeidon_mi355x_test.so, every other object shared with the shipped library).
// In libposeidon_mi355x.so the three queries below are compile-time constants and no hook symbol is exported.
//   fault:          the host fan-out of the next calls fails on one local slot and / or behaves as if no worker thread
//                   could be started (error carry-back and the serial path, on any box);
//   shared device:  pmx_mgpu_create accepts the same device ordinal in several slots and more slots than visible devices,
//                   so that every world > 1 branch below runs on a ONE-GPU box behind tests/fake_rccl (RCCL itself
//                   refuses two ranks on one device).
// Process-wide, read by the fan-out's worker threads: atomics.
#ifdef PMX_TEST_HOOKS
namespace {
struct Hooks {
    std::atomic<int> fail_local{-1};
    std::atomic<bool> no_threads{false};
    std::atomic<bool> shared_device{false};
};
Hooks g_hooks;
inline int hook_fail_local() { return g_hooks.fail_local.load(); }
inline bool hook_no_threads() { return g_hooks.no_threads.load(); }
inline bool hook_shared_device() { return g_hooks.shared_device.load(); }
}  // namespace
extern "C" int pmx_test_hooks_enabled(void) { return 1; }
extern "C" int pmx_mgpu_test_fault(int fail_local, int no_threads) {
    g_hooks.fail_local.store(fail_local);
    g_hooks.no_threads.store(no_threads != 0);
    return PMX_OK;
}
extern "C" int pmx_mgpu_test_shared_device(int allow) {
    g_hooks.shared_device.store(allow != 0);
    return PMX_OK;
}
#else
namespace {
constexpr bool hook_no_threads() { return false; }
constexpr bool hook_shared_device() { return false; }
}  // namespace
#endif

extern "C" int pmx_mgpu_create(const pmx_config *cfg, int n_devices, const int *devices, pmx_mgpu **out) {
    PMX_ABI_BEGIN("pmx_mgpu_create")
    if (!cfg || !out) return set_error(PMX_ERR_ARG, "pmx_mgpu_create: null pointer");
    *out = nullptr;
    const int visible = pmx_device_count();
    if (visible == 0) return set_error(PMX_ERR_HIP, "no HIP device available; this library has no CPU fallback");
    if (int rc = rccl_ready()) return rc;
    const bool shared = hook_shared_device();   // test build only: slots may share a device (constant false in the shipped library)
    const int max_slots = shared || visible > PMX_MAX_LOCAL_DEVICES ? PMX_MAX_LOCAL_DEVICES : visible;
    if (n_devices <= 0 || n_devices > max_slots) return set_error(PMX_ERR_ARG, "n_devices %d out of range [1,%d]", n_devices, max_slots);
    if (shared && !devices) return set_error(PMX_ERR_ARG, "shared-device groups (test hook) name their devices explicitly");
    GroupHolder hold{new (std::nothrow) pmx_mgpu()};
    pmx_mgpu *g = hold.g;
    if (!g) return set_error(PMX_ERR_HOST, "out of host memory");
    g->world = n_devices;
    g->first_rank = 0;
    g->device.resize(n_devices);
    g->ctx.assign(n_devices, nullptr);
    g->comm.assign(n_devices, nullptr);
    for (int l = 0; l < n_devices; ++l) {
        const int d = devices ? devices[l] : l;
        g->device[l] = d;
        if (d < 0 || d >= visible) return set_error(PMX_ERR_ARG, "device %d out of range [0,%d)", d, visible);
        for (int k = 0; k < l && !shared; ++k)
            if (g->device[k] == d) return set_error(PMX_ERR_ARG, "device %d listed twice", d);
    }
    int rc = group_contexts(g, cfg);
    if (rc) return rc;
    ncclResult_t r = rccl_lib().CommInitAll(g->comm.data(), n_devices, g->device.data());
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitAll");
    *out = hold.release();
    return PMX_OK;
    PMX_ABI_END
}

extern "C" int pmx_mgpu_create_rank(const pmx_config *cfg, int device, int rank, int world,
                                    const uint8_t id[PMX_UNIQUE_ID_BYTES], pmx_mgpu **out) {
    PMX_ABI_BEGIN("pmx_mgpu_create_rank")
    if (!cfg || !out || !id) return set_error(PMX_ERR_ARG, "pmx_mgpu_create_rank: null pointer");
    *out = nullptr;
    if (world <= 0 || rank < 0 || rank >= world) return set_error(PMX_ERR_ARG, "rank %d / world %d out of range", rank, world);
    const int visible = pmx_device_count();
    if (visible == 0) return set_error(PMX_ERR_HIP, "no HIP device available; this library has no CPU fallback");
    if (device < 0 || device >= visible) return set_error(PMX_ERR_ARG, "device %d out of range [0,%d)", device, visible);
    if (int rc = rccl_ready()) return rc;
    GroupHolder hold{new (std::nothrow) pmx_mgpu()};
    pmx_mgpu *g = hold.g;
    if (!g) return set_error(PMX_ERR_HOST, "out of host memory");
    g->world = world;
    g->first_rank = rank;
    g->device.assign(1, device);
    g->ctx.assign(1, nullptr);
    g->comm.assign(1, nullptr);
    int rc = group_contexts(g, cfg);
    if (rc) return rc;
    ncclUniqueId u;
    std::memcpy(u.internal, id, PMX_UNIQUE_ID_BYTES);
    {
        DeviceGuard guard(device);
        if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
        ncclResult_t r = rccl_lib().CommInitRank(&g->comm[0], world, u, rank);
        if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
    }
    *out = hold.release();
    return PMX_OK;
    PMX_ABI_END
}

extern "C" int pmx_mgpu_destroy(pmx_mgpu *g) {
    PMX_ABI_BEGIN("pmx_mgpu_destroy")
    if (!g) return PMX_OK;
    (void)pmx_mgpu_synchronize(g);
    group_free(g);
    return PMX_OK;
    PMX_ABI_END
}

extern "C" int pmx_mgpu_get_info(const pmx_mgpu *g, pmx_mgpu_info *info) {
    PMX_ABI_BEGIN("pmx_mgpu_get_info")
    if (!g || !info) return set_error(PMX_ERR_ARG, "pmx_mgpu_get_info: null pointer");
    std::memset(info, 0, sizeof *info);
    info->world = g->world;
    info->n_local = (int)g->device.size();
    info->first_rank = g->first_rank;
    info->width = (int)g->t;
    PMX_RCCL(rccl_lib().GetVersion(&info->rccl_version));
    // what the LIVE communicator says about itself (not what the group was asked for)
    PMX_RCCL(rccl_lib().CommCount(g->comm[0], &info->comm_ranks));
    PMX_RCCL(rccl_lib().CommUserRank(g->comm[0], &info->comm_first_rank));
    for (size_t l = 0; l < g->device.size() && l < PMX_MAX_LOCAL_DEVICES; ++l) info->devices[l] = g->device[l];
    return PMX_OK;
    PMX_ABI_END
}

extern "C" void *pmx_mgpu_stream(const pmx_mgpu *g, int local) {
    if (!g || local < 0 || local >= (int)g->ctx.size()) return nullptr;
    return (void *)g->ctx[local]->stream;
}

extern "C" pmx_ctx *pmx_mgpu_ctx(const pmx_mgpu *g, int local) {
    if (!g || local < 0 || local >= (int)g->ctx.size()) return nullptr;
    return g->ctx[local];
}

extern "C" int pmx_mgpu_synchronize(pmx_mgpu *g) {
    PMX_ABI_BEGIN("pmx_mgpu_synchronize")
    if (!g) return set_error(PMX_ERR_ARG, "pmx_mgpu_synchronize: null pointer");
    for (size_t l = 0; l < g->ctx.size(); ++l) {
        if (!g->ctx[l]) continue;
        PMX_BIND(g->ctx[l]);
        PMX_HIP(hipStreamSynchronize(g->ctx[l]->stream));
        PMX_HIP(hipStreamSynchronize(g->ctx[l]->stream2));   // (pmx_mgpu_permute_gather_dev's transfers: joined into `stream` by the call itself unless it failed half way)
    }
    return PMX_OK;
    PMX_ABI_END
}

static int local_span(const pmx_mgpu *g, size_t n_total, size_t l, size_t *start, size_t *count) {
    return pmx_shard_bounds(n_total, g->world, g->first_rank + (int)l, start, count);
}

// ---- permutation ---------------------------------------------------------------------------------------------------
extern "C" int pmx_mgpu_permute_shards_dev(pmx_mgpu *g, uint64_t *const *d_shards, size_t n_total) {
    PMX_ABI_BEGIN("pmx_mgpu_permute_shards_dev")
    if (!g || !d_shards) return set_error(PMX_ERR_ARG, "pmx_mgpu_permute_shards_dev: null pointer");
    for (size_t l = 0; l < g->ctx.size(); ++l) {
        size_t start = 0, count = 0;
        int rc = local_span(g, n_total, l, &start, &count);
        if (rc) return rc;
        if ((rc = pmx_permute_batch_dev(g->ctx[l], d_shards[l], count, g->ctx[l]->stream))) return rc;   // no collective on the data path
    }
    return PMX_OK;
    PMX_ABI_END
}

// Host batches, single-process groups: shard l goes through device l's own host path (pinned memory: chunked H2D /
// kernel / D2H pipeline on two streams), one host thread per device so that all devices copy and compute concurrently -
// also for pageable memory, whose copies block the calling thread.  `work(l, start, count)` runs on device l's thread.
template <class Work>
static int fan_out(pmx_mgpu *g, size_t n, const char *who, Work work) {
    if ((int)g->ctx.size() != g->world)
        return set_error(PMX_ERR_ARG, "%s needs a single-process group (this one holds %zu of %d ranks)", who, g->ctx.size(), g->world);
    if (n == 0) return PMX_OK;
    const size_t L = g->ctx.size();
    std::vector<int> rcs(L, PMX_OK);
    std::vector<std::string> msgs(L);
    // runs on worker threads: nothing may leave it by exception (an exception that escapes a thread function is
    // std::terminate, and the caller's PMX_ABI_BEGIN guard is on another stack)
    auto run = [&](size_t l) noexcept {
        int rc = PMX_OK;
        try {
            size_t start = 0, count = 0;
            rc = local_span(g, n, l, &start, &count);
            if (!rc && count) {
#ifdef PMX_TEST_HOOKS
                if (hook_fail_local() == (int)l) rc = set_error(PMX_ERR_HIP, "injected failure (pmx_mgpu_test_fault)");
                else
#endif
                    rc = work(l, start, count);
            }
            if (rc) msgs[l] = pmx_last_error();   // the error text is thread-local: carry it back to the caller's thread
        } catch (...) {
            rc = PMX_ERR_HOST;                    // (msgs[l] may be what could not be allocated: the code alone goes back)
        }
        rcs[l] = rc;
    };
    // one host thread per further device; a thread that cannot be started (std::system_error) is not fatal: its shard
    // runs on the calling thread instead, after the ones that did start.  The workers are joined on every way out of
    // this frame (destroying a joinable std::thread is std::terminate).
    struct Workers {
        std::vector<std::thread> th;
        void join() { for (auto &w : th) if (w.joinable()) w.join(); }
        ~Workers() { join(); }
    } workers;
    std::vector<size_t> serial;
    workers.th.reserve(L);
    serial.reserve(L);
    for (size_t l = 1; l < L; ++l) {
        try {
            if (hook_no_threads()) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
            workers.th.emplace_back(run, l);
        } catch (const std::system_error &) {
            serial.push_back(l);
        }
    }
    run(0);
    for (size_t l : serial) run(l);
    workers.join();
    for (size_t l = 0; l < L; ++l)
        if (rcs[l]) return set_error(rcs[l], "device %d (slot %zu): %s", g->device[l], l, msgs[l].empty() ? "host failure on the worker thread" : msgs[l].c_str());
    return PMX_OK;
}

extern "C" int pmx_mgpu_permute_batch(pmx_mgpu *g, uint64_t *states, size_t n) {
    PMX_ABI_BEGIN("pmx_mgpu_permute_batch")
    if (!g || (!states && n)) return set_error(PMX_ERR_ARG, "pmx_mgpu_permute_batch: null pointer");
    return fan_out(g, n, "pmx_mgpu_permute_batch",
                   [&](size_t l, size_t start, size_t count) { return pmx_permute_batch(g->ctx[l], states + start * g->t * 4, count); });
    PMX_ABI_END
}

extern "C" int pmx_mgpu_hash_batch(pmx_mgpu *g, const uint64_t *in, size_t in_len, uint64_t *out, size_t out_len, size_t n) {
    PMX_ABI_BEGIN("pmx_mgpu_hash_batch")
    if (!g || (!in && n && in_len) || (!out && n && out_len)) return set_error(PMX_ERR_ARG, "pmx_mgpu_hash_batch: null pointer");
    return fan_out(g, n, "pmx_mgpu_hash_batch", [&](size_t l, size_t start, size_t count) {
        return pmx_hash_batch(g->ctx[l], in ? in + start * in_len * 4 : nullptr, in_len, out ? out + start * out_len * 4 : nullptr, out_len, count);
    });
    PMX_ABI_END
}

// ---- the final gather ------------------------------------------------------------------------------------------------
// Every rank ends with all shards in rank order.  Equal shards: one ncclAllGather.  Ragged shards (n_total not a
// multiple of the world size): one ncclBroadcast per rank inside a group call, each into its own span of the output.
extern "C" int pmx_mgpu_all_gather_dev(pmx_mgpu *g, const uint64_t *const *d_shards, uint64_t *const *d_all, size_t n_total,
                                       size_t row_elems) {
    PMX_ABI_BEGIN("pmx_mgpu_all_gather_dev")
    if (!g || !d_shards || !d_all) return set_error(PMX_ERR_ARG, "pmx_mgpu_all_gather_dev: null pointer");
    if (row_elems == 0 || n_total == 0) return PMX_OK;
    if (n_total > SIZE_MAX / (row_elems * 32)) return set_error(PMX_ERR_ARG, "gather byte size overflows size_t");
    const size_t words = row_elems * 4;   // u64 words per unit
    const bool equal = n_total % (size_t)g->world == 0;
    PMX_RCCL(rccl_lib().GroupStart());
    ncclResult_t r = ncclSuccess;
    for (size_t l = 0; l < g->ctx.size() && r == ncclSuccess; ++l) {
        DeviceGuard guard(g->device[l]);
        hipStream_t st = g->ctx[l]->stream;
        if (equal) {
            r = rccl_lib().AllGather(d_shards[l], d_all[l], (n_total / (size_t)g->world) * words, ncclUint64, g->comm[l], st);
        } else {
            for (int root = 0; root < g->world && r == ncclSuccess; ++root) {
                size_t start = 0, count = 0;
                (void)pmx_shard_bounds(n_total, g->world, root, &start, &count);
                if (count == 0) continue;
                const bool mine = root == g->first_rank + (int)l;
                r = rccl_lib().Broadcast(mine ? (const void *)d_shards[l] : (const void *)(d_all[l] + start * words), d_all[l] + start * words,
                                  count * words, ncclUint64, root, g->comm[l], st);
            }
        }
    }
    ncclResult_t e = rccl_lib().GroupEnd();
    if (r != ncclSuccess) return rccl_fail(r, equal ? "ncclAllGather" : "ncclBroadcast");
    if (e != ncclSuccess) return rccl_fail(e, "ncclGroupEnd");
    return PMX_OK;
    PMX_ABI_END
}

// ---- gather to one rank, and the gather piece by piece behind the last step ---------------------------------------------
// piece i of `chunks` of a shard of `count` units
static void piece_span(size_t count, int chunks, int i, size_t *first, size_t *cnt) {
    const size_t lo = count / (size_t)chunks * (size_t)i + count % (size_t)chunks * (size_t)i / (size_t)chunks;
    const size_t hi = count / (size_t)chunks * (size_t)(i + 1) + count % (size_t)chunks * (size_t)(i + 1) / (size_t)chunks;
    *first = lo;
    *cnt = hi - lo;
}

// Inside an open group: the transfers of piece i of every shard, on the slots' first or second stream.  root >= 0: every other
// rank's piece goes to `root` (one ncclSend per rank, world - 1 ncclRecv on the root); root < 0: to every rank.  A rank that receives
// also copies its own piece into its copy of the result.  Every rank derives every other rank's piece from (n_total, world, chunks, i)
// alone, so the sends and the receives of a pair always agree.
static int post_piece(pmx_mgpu *g, const uint64_t *const *d_shards, uint64_t *const *d_all, size_t n_total, size_t words, int root, int chunks, int i,
                      bool second_stream, ncclResult_t *nr, const char **what) {
    for (size_t l = 0; l < g->ctx.size(); ++l) {
        const int me = g->first_rank + (int)l;
        const bool i_receive = root < 0 || root == me;
        DeviceGuard guard(g->device[l]);
        if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
        hipStream_t st = second_stream ? g->ctx[l]->stream2 : g->ctx[l]->stream;
        size_t start = 0, count = 0, pf = 0, pc = 0;
        (void)pmx_shard_bounds(n_total, g->world, me, &start, &count);
        piece_span(count, chunks, i, &pf, &pc);
        const uint64_t *mine = d_shards[l] + pf * words;
        if (i_receive && pc && d_all[l] + (start + pf) * words != mine)
            PMX_HIP(hipMemcpyAsync(d_all[l] + (start + pf) * words, mine, pc * words * 8, hipMemcpyDeviceToDevice, st));
        for (int peer = 0; peer < g->world; ++peer) {
            if (peer == me) continue;
            if ((root < 0 || root == peer) && pc) {
                *nr = rccl_lib().Send(mine, pc * words, ncclUint64, peer, g->comm[l], st);
                if (*nr != ncclSuccess) { *what = "ncclSend"; return PMX_OK; }
            }
            if (i_receive) {
                size_t qs = 0, qn = 0, qf = 0, qc = 0;
                (void)pmx_shard_bounds(n_total, g->world, peer, &qs, &qn);
                piece_span(qn, chunks, i, &qf, &qc);
                if (qc) {
                    *nr = rccl_lib().Recv(d_all[l] + (qs + qf) * words, qc * words, ncclUint64, peer, g->comm[l], st);
                    if (*nr != ncclSuccess) { *what = "ncclRecv"; return PMX_OK; }
                }
            }
        }
    }
    return PMX_OK;
}

// one group: GroupStart, the piece, GroupEnd - with the error of whichever call failed first
static int gather_piece(pmx_mgpu *g, const uint64_t *const *d_shards, uint64_t *const *d_all, size_t n_total, size_t words, int root, int chunks, int i,
                        bool second_stream) {
    if (g->world == 1) {   // nobody to talk to: the copy of the own piece is all there is (and RCCL is not entered with an empty group)
        ncclResult_t nr = ncclSuccess;
        const char *what = "";
        return post_piece(g, d_shards, d_all, n_total, words, root, chunks, i, second_stream, &nr, &what);
    }
    PMX_RCCL(rccl_lib().GroupStart());
    ncclResult_t nr = ncclSuccess;
    const char *what = "";
    const int rc = post_piece(g, d_shards, d_all, n_total, words, root, chunks, i, second_stream, &nr, &what);
    const ncclResult_t e = rccl_lib().GroupEnd();
    if (rc) return rc;
    if (nr != ncclSuccess) return rccl_fail(nr, what);
    if (e != ncclSuccess) return rccl_fail(e, "ncclGroupEnd");
    return PMX_OK;
}

static int gather_args(const pmx_mgpu *g, const void *d_shards, uint64_t *const *d_all, int root, const char *who) {
    if (!g || !d_shards || !d_all) return set_error(PMX_ERR_ARG, "%s: null pointer", who);
    if (root >= g->world) return set_error(PMX_ERR_ARG, "%s: root %d out of range [0,%d) (negative: every rank)", who, root, g->world);
    for (size_t l = 0; l < g->ctx.size(); ++l)
        if ((root < 0 || root == g->first_rank + (int)l) && !d_all[l]) return set_error(PMX_ERR_ARG, "%s: slot %zu receives the result and has no buffer for it", who, l);
    return PMX_OK;
}

extern "C" int pmx_mgpu_gather_dev(pmx_mgpu *g, const uint64_t *const *d_shards, uint64_t *const *d_all, size_t n_total, size_t row_elems, int root) {
    PMX_ABI_BEGIN("pmx_mgpu_gather_dev")
    if (int rc = gather_args(g, d_shards, d_all, root, "pmx_mgpu_gather_dev")) return rc;
    if (root < 0) return set_error(PMX_ERR_ARG, "pmx_mgpu_gather_dev: root %d out of range [0,%d) (pmx_mgpu_all_gather_dev gathers to every rank)", root, g->world);
    if (row_elems == 0 || n_total == 0) return PMX_OK;
    if (n_total > SIZE_MAX / (row_elems * 32)) return set_error(PMX_ERR_ARG, "gather byte size overflows size_t");
    return gather_piece(g, d_shards, d_all, n_total, row_elems * 4, root, 1, 0, false);
    PMX_ABI_END
}

// The last step of a job and its gather in one call.  Every local shard is permuted in `chunks` pieces on the slot's stream; piece i's
// transfers are posted on the slot's SECOND stream behind an event of piece i's kernel, so that the links carry piece i while the
// kernels of pieces i + 1 ... run; at the end the slot's stream waits for its second stream - to the caller's stream order the call
// is pmx_mgpu_permute_shards_dev followed by the gather.  The transfers are point to point whatever the root: a piece lands inside its
// rank's span, which one ncclAllGather cannot do (its layout is rank-major per call).
extern "C" int pmx_mgpu_permute_gather_dev(pmx_mgpu *g, uint64_t *const *d_shards, uint64_t *const *d_all, size_t n_total, int root, int chunks) {
    PMX_ABI_BEGIN("pmx_mgpu_permute_gather_dev")
    if (int rc = gather_args(g, d_shards, d_all, root, "pmx_mgpu_permute_gather_dev")) return rc;
    if (chunks < 1 || chunks > pmx_ctx::kPipeChunks) return set_error(PMX_ERR_ARG, "pmx_mgpu_permute_gather_dev: chunks %d out of range [1,%d]", chunks, pmx_ctx::kPipeChunks);
    if (n_total == 0) return PMX_OK;
    if (n_total > SIZE_MAX / ((size_t)g->t * 32)) return set_error(PMX_ERR_ARG, "gather byte size overflows size_t");
    const size_t words = (size_t)g->t * 4;
    for (int i = 0; i < chunks; ++i) {
        for (size_t l = 0; l < g->ctx.size(); ++l) {
            size_t start = 0, count = 0, pf = 0, pc = 0;
            int rc = local_span(g, n_total, l, &start, &count);
            if (rc) return rc;
            piece_span(count, chunks, i, &pf, &pc);
            pmx_ctx *c = g->ctx[l];
            if (pc && (rc = pmx_permute_batch_dev(c, d_shards[l] + pf * words, pc, c->stream))) return rc;
            PMX_BIND(c);
            PMX_HIP(hipEventRecord(c->pipe_done[i], c->stream));
            PMX_HIP(hipStreamWaitEvent(c->stream2, c->pipe_done[i], 0));
        }
        if (int rc = gather_piece(g, d_shards, d_all, n_total, words, root, chunks, i, true)) return rc;
    }
    for (size_t l = 0; l < g->ctx.size(); ++l) {
        pmx_ctx *c = g->ctx[l];
        PMX_BIND(c);
        PMX_HIP(hipEventRecord(c->pipe_up[0], c->stream2));
        PMX_HIP(hipStreamWaitEvent(c->stream, c->pipe_up[0], 0));
    }
    return PMX_OK;
    PMX_ABI_END
}

// ---- Merkle 2-to-1 ---------------------------------------------------------------------------------------------------
// Each rank reduces its own contiguous subtree of m = n_leaves / world leaves (level by level, pmx_merkle_2to1_dev), the
// `world` subtree roots - 32 bytes each - are all-gathered, and every rank finishes the top log2(world) levels itself.
extern "C" int pmx_mgpu_merkle_2to1_dev(pmx_mgpu *g, uint64_t *const *d_nodes, uint64_t *const *d_top, size_t n_leaves) {
    PMX_ABI_BEGIN("pmx_mgpu_merkle_2to1_dev")
    if (!g || !d_nodes || !d_top) return set_error(PMX_ERR_ARG, "pmx_mgpu_merkle_2to1_dev: null pointer");
    const size_t W = (size_t)g->world;
    if (W & (W - 1)) return set_error(PMX_ERR_ARG, "the sharded tree needs a power-of-two number of ranks (have %d)", g->world);
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1)) || n_leaves < W) return set_error(PMX_ERR_ARG, "n_leaves must be a power of two >= the number of ranks");
    const size_t m = n_leaves / W;
    for (size_t l = 0; l < g->ctx.size(); ++l) {
        int rc = pmx_merkle_2to1_dev(g->ctx[l], d_nodes[l], m, g->ctx[l]->stream);
        if (rc) return rc;
    }
    if (W == 1) {
        PMX_BIND(g->ctx[0]);
        PMX_HIP(hipMemcpyAsync(d_top[0], d_nodes[0] + (2 * m - 2) * 4, 32, hipMemcpyDeviceToDevice, g->ctx[0]->stream));
        return PMX_OK;
    }
    PMX_RCCL(rccl_lib().GroupStart());
    ncclResult_t r = ncclSuccess;
    for (size_t l = 0; l < g->ctx.size() && r == ncclSuccess; ++l) {
        DeviceGuard guard(g->device[l]);
        r = rccl_lib().AllGather(d_nodes[l] + (2 * m - 2) * 4, d_top[l], 4, ncclUint64, g->comm[l], g->ctx[l]->stream);
    }
    ncclResult_t e = rccl_lib().GroupEnd();
    if (r != ncclSuccess) return rccl_fail(r, "ncclAllGather");
    if (e != ncclSuccess) return rccl_fail(e, "ncclGroupEnd");
    for (size_t l = 0; l < g->ctx.size(); ++l) {
        int rc = pmx_merkle_2to1_dev(g->ctx[l], d_top[l], W, g->ctx[l]->stream);
        if (rc) return rc;
    }
    return PMX_OK;
    PMX_ABI_END
}

// Host leaves, single-process groups.  root: [4].
extern "C" int pmx_mgpu_merkle_2to1(pmx_mgpu *g, const uint64_t *leaves, size_t n_leaves, uint64_t *root) {
    PMX_ABI_BEGIN("pmx_mgpu_merkle_2to1")
    if (!g || !leaves || !root) return set_error(PMX_ERR_ARG, "pmx_mgpu_merkle_2to1: null pointer");
    if ((int)g->ctx.size() != g->world)
        return set_error(PMX_ERR_ARG, "pmx_mgpu_merkle_2to1 needs a single-process group (this one holds %zu of %d ranks)", g->ctx.size(), g->world);
    const size_t W = (size_t)g->world, L = g->ctx.size();
    if (W & (W - 1)) return set_error(PMX_ERR_ARG, "the sharded tree needs a power-of-two number of ranks (have %d)", g->world);
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1)) || n_leaves < W) return set_error(PMX_ERR_ARG, "n_leaves must be a power of two >= the number of ranks");
    if (n_leaves > SIZE_MAX / 64) return set_error(PMX_ERR_ARG, "tree byte size overflows size_t");
    const size_t m = n_leaves / W;
    std::vector<uint64_t *> d_nodes(L, nullptr), d_top(L, nullptr);
    int rc = PMX_OK;
    auto cleanup = [&]() {
        (void)pmx_mgpu_synchronize(g);
        for (size_t l = 0; l < L; ++l) {
            DeviceGuard guard(g->device[l]);
            if (d_nodes[l]) (void)hipFree(d_nodes[l]);
            if (d_top[l]) (void)hipFree(d_top[l]);
        }
    };
    for (size_t l = 0; l < L && !rc; ++l) {
        DeviceGuard guard(g->device[l]);
        hipError_t e = hipMalloc((void **)&d_nodes[l], (2 * m - 1) * 32);
        if (e == hipSuccess) e = hipMalloc((void **)&d_top[l], (2 * W - 1) * 32);
        if (e == hipSuccess) e = hipMemcpyAsync(d_nodes[l], leaves + (size_t)l * m * 4, m * 32, hipMemcpyHostToDevice, g->ctx[l]->stream);
        if (e != hipSuccess) rc = hip_fail(e, "pmx_mgpu_merkle_2to1: device staging");
    }
    if (!rc) rc = pmx_mgpu_merkle_2to1_dev(g, d_nodes.data(), d_top.data(), n_leaves);
    if (!rc) {
        DeviceGuard guard(g->device[0]);
        hipError_t e = hipMemcpyAsync(root, d_top[0] + (W == 1 ? 0 : (2 * W - 2) * 4), 32, hipMemcpyDeviceToHost, g->ctx[0]->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g->ctx[0]->stream);
        if (e != hipSuccess) rc = hip_fail(e, "pmx_mgpu_merkle_2to1: root copy");
    }
    cleanup();
    return rc;
    PMX_ABI_END
}
