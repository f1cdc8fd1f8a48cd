// C ABI of libposeidon_mi355x.so: contexts, validation, host<->device staging, kernel dispatch.
// See include/poseidon_mi355x.h for the contract and the reference interfaces each entry replaces.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/poseidon_mi355x.h"
#include "pmx_ctx.hpp"
#include "pmx_internal.hpp"
#include "pmx_prepare.hpp"
#include "pmx_launch.hpp"
#include "pmx_sponge_plan.hpp"

namespace pmx {

static thread_local char g_error[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    return set_error(PMX_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace pmx

using namespace pmx;

static int ctx_scratch(pmx_ctx *ctx, int slot, size_t bytes, void **out) {
    if (bytes == 0) bytes = 16;
    if (ctx->scratch_bytes[slot] < bytes) {
        void *old = ctx->scratch[slot];
        ctx->scratch[slot] = nullptr;
        ctx->scratch_bytes[slot] = 0;
        if (old) PMX_HIP(hipFree(old));
        PMX_HIP(hipMalloc(&ctx->scratch[slot], bytes));
        ctx->scratch_bytes[slot] = bytes;
    }
    *out = ctx->scratch[slot];
    return PMX_OK;
}

// Host-buffer entry points queue asynchronous copies that read and write the CALLER's memory: whatever way they
// leave (including an error half way through), nothing may still be in flight on the context's streams.
struct StreamDrain {
    pmx_ctx *ctx;
    ~StreamDrain() {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamSynchronize(ctx->stream3);
    }
};

// The host-buffer pipeline of a batch whose buffers are page-locked: chunk i is uploaded on `stream`, computed on `stream2` behind the
// upload's event and downloaded on `stream3` behind the kernel's - the two copy directions have a stream (and a DMA engine) each, so chunk
// i + 1 goes up while chunk i - 1 comes down (with a stream per LANE, round 5, each lane's download stood in front of its next upload and
// the link carried one direction at a time: 61 GB/s for both together, profiles/r05/z_host_path.txt).  Pageable buffers: one chunk.
template <class Up, class Run, class Down>
static int host_pipeline(pmx_ctx *ctx, size_t n, size_t step, Up &&up, Run &&run, Down &&down) {
    int rc = PMX_OK;
    size_t first = 0;
    for (int i = 0; first < n; first += step, ++i) {
        const size_t cnt = n - first < step ? n - first : step;
        const int slot = i % pmx_ctx::kPipeChunks;   // (at most kPipeChunks chunks: pipeline_rows)
        if ((rc = up(first, cnt, ctx->stream))) return rc;
        PMX_HIP(hipEventRecord(ctx->pipe_up[slot], ctx->stream));
        PMX_HIP(hipStreamWaitEvent(ctx->stream2, ctx->pipe_up[slot], 0));
        if ((rc = run(first, cnt, ctx->stream2))) return rc;
        PMX_HIP(hipEventRecord(ctx->pipe_done[slot], ctx->stream2));
        PMX_HIP(hipStreamWaitEvent(ctx->stream3, ctx->pipe_done[slot], 0));
        if ((rc = down(first, cnt, ctx->stream3))) return rc;
    }
    PMX_HIP(hipStreamSynchronize(ctx->stream3));
    PMX_HIP(hipStreamSynchronize(ctx->stream2));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

// n * elems_per_row * 32 bytes, or an error if that does not fit size_t / the launch grid
static int batch_bytes(size_t n, size_t elems_per_row, size_t *bytes) {
    if (n > (size_t)0x7fffffff * 64) return set_error(PMX_ERR_ARG, "batch too large");
    if (elems_per_row && n > (SIZE_MAX / 32) / elems_per_row) return set_error(PMX_ERR_ARG, "batch byte size overflows size_t");
    *bytes = n * elems_per_row * 32;
    return PMX_OK;
}

extern "C" int pmx_abi_version(void) { return PMX_ABI_VERSION; }

extern "C" const char *pmx_last_error(void) { return g_error; }

extern "C" int pmx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int pmx_ctx_create(const pmx_config *cfg, int device, pmx_ctx **out) {
    PMX_ABI_BEGIN("pmx_ctx_create")
    if (!cfg || !out) return set_error(PMX_ERR_ARG, "pmx_ctx_create: null pointer");
    *out = nullptr;
    if (!cfg->ark || !cfg->mds) return set_error(PMX_ERR_ARG, "pmx_ctx_create: null ark/mds");
    Prepared pp;
    std::string err;
    int rc = prepare(cfg, pp, err);   // the asserts of PoseidonConfig::new + limits of this build
    if (rc) return set_error(rc, "%s", err.c_str());
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return set_error(PMX_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return set_error(PMX_ERR_ARG, "device %d out of range [0,%d)", device, ndev);

    pmx_ctx *ctx = new (std::nothrow) pmx_ctx();
    if (!ctx) return set_error(PMX_ERR_HOST, "pmx_ctx_create: out of host memory");
    ctx->device = device;
    ctx->t = pp.t;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) { delete ctx; return hip_fail(guard.err, "hipSetDevice"); }

    const size_t bytes = pp.consts.size() * 4;
    e = hipMalloc((void **)&ctx->d_consts, bytes);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_consts, pp.consts.data(), bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking);
    for (int i = 0; i < pmx_ctx::kPipeChunks && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&ctx->pipe_up[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->pipe_done[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        for (hipStream_t st : {ctx->stream, ctx->stream2, ctx->stream3})
            if (st) (void)hipStreamDestroy(st);
        for (int i = 0; i < pmx_ctx::kPipeChunks; ++i) {
            if (ctx->pipe_up[i]) (void)hipEventDestroy(ctx->pipe_up[i]);
            if (ctx->pipe_done[i]) (void)hipEventDestroy(ctx->pipe_done[i]);
        }
        if (ctx->d_consts) (void)hipFree(ctx->d_consts);
        delete ctx;
        return hip_fail(e, "pmx_ctx_create: device setup");
    }
    DevConfig &d = ctx->dev;
    d.consts = ctx->d_consts;
    d.n_const_words = (uint32_t)pp.consts.size();
    d.mds_offset = (uint32_t)pp.mds_offset;
    d.opt_offset = (uint32_t)pp.opt_offset;
    d.coop_offset = (uint32_t)pp.coop_offset;
    d.mfma_offset = (uint32_t)pp.mfma_offset;
    d.mfma_dense = pp.mfma_dense ? 1u : 0u;
    d.win_offset = (uint32_t)pp.win_offset;
    d.io_offset = (uint32_t)pp.io_offset;
    d.has_opt = pp.has_opt ? 1u : 0u;
    {
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || lds <= 0) lds = 64 * 1024;
        d.max_lds_bytes = (uint32_t)lds;
    }
    d.rounds = pp.c;
    d.field = pp.f;
    d.field.io = nullptr;   // engines point it at consts + io_offset on the device
    d.one = pp.one;
    *out = ctx;
    return PMX_OK;
    PMX_ABI_END
}

static int ctx_free(pmx_ctx *ctx) {
    DeviceGuard guard(ctx->device);
    for (hipStream_t st : {ctx->stream, ctx->stream2, ctx->stream3}) {
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    }
    for (int i = 0; i < pmx_ctx::kPipeChunks; ++i) {
        if (ctx->pipe_up[i]) (void)hipEventDestroy(ctx->pipe_up[i]);
        if (ctx->pipe_done[i]) (void)hipEventDestroy(ctx->pipe_done[i]);
    }
    for (int i = 0; i < 4; ++i)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    // (the caller's streams are the caller's to drain before it destroys the context, as with every *_dev call)
    for (pmx_ctx::PassBlock &b : ctx->pass_pool) {
        if (b.done) (void)hipEventDestroy(b.done);
        if (b.ptr) (void)hipFree(b.ptr);
    }
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->d_consts) (void)hipFree(ctx->d_consts);
    delete ctx;
    return PMX_OK;
}

extern "C" int pmx_ctx_destroy(pmx_ctx *ctx) {
    if (!ctx) return PMX_OK;
    if (ctx->cache_key) return set_error(PMX_ERR_ARG, "pmx_ctx_destroy: this context came from pmx_ctx_acquire; hand it back with pmx_ctx_release");
    return ctx_free(ctx);
}

// ---- shared contexts ---------------------------------------------------------------------------------
// CryptographicSponge::new clones the parameters into every sponge (src/poseidon/mod.rs:219-230) and downstream code
// calls it once per transcript.  Creating a device context per sponge would re-derive the sparse-round tables, allocate
// and upload every time, so the bindings take their context from this process-wide cache instead: one context per
// (config contents, device), reference-counted; idle contexts stay resident (at most kMaxIdle of them, oldest first
// out) so that  new -> drop -> new  does not rebuild either.
namespace {
struct CtxCache {
    std::mutex lock;
    std::unordered_map<uint64_t, std::vector<pmx_ctx *>> by_key;
    std::deque<pmx_ctx *> idle;   // refs == 0, oldest first
};
CtxCache &ctx_cache() {
    static CtxCache *c = new CtxCache();   // never destroyed: contexts may outlive static destructors' order
    return *c;
}
constexpr size_t kMaxIdle = 8;

std::string config_blob(const pmx_config *cfg, int device) {
    const size_t t = (size_t)cfg->rate + cfg->capacity, rounds = (size_t)cfg->full_rounds + cfg->partial_rounds;
    std::string b;
    auto put = [&](const void *p, size_t n) { b.append((const char *)p, n); };
    put(&device, sizeof device);
    put(&cfg->full_rounds, 4); put(&cfg->partial_rounds, 4); put(&cfg->alpha, 8); put(&cfg->rate, 4); put(&cfg->capacity, 4);
    put(cfg->modulus, 32);
    put(cfg->ark, rounds * t * 32);
    put(cfg->mds, t * t * 32);
    return b;
}
uint64_t fnv1a(const std::string &b) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : b) { h ^= c; h *= 1099511628211ull; }
    return h ? h : 1;
}
}  // namespace

// The cache lock is held only for the bookkeeping: a context is created (table derivation, hipMalloc, upload) and freed
// (stream synchronise, hipFree) outside it, so one thread building or evicting a context never stalls another thread's
// acquire or release.  Two threads that miss on the same new config both build one; the second to come back finds the
// first one's entry, takes a reference on it and frees its own.
static pmx_ctx *cache_find(CtxCache &cc, uint64_t key, const std::string &blob) {
    auto it = cc.by_key.find(key);
    if (it == cc.by_key.end()) return nullptr;
    for (pmx_ctx *c : it->second) {
        if (c->cache_blob == blob) {
            if (c->cache_refs++ == 0) {
                for (auto d = cc.idle.begin(); d != cc.idle.end(); ++d)
                    if (*d == c) { cc.idle.erase(d); break; }
            }
            return c;
        }
    }
    return nullptr;
}

extern "C" int pmx_ctx_acquire(const pmx_config *cfg, int device, pmx_ctx **out) {
    PMX_ABI_BEGIN("pmx_ctx_acquire")
    if (!cfg || !out) return set_error(PMX_ERR_ARG, "pmx_ctx_acquire: null pointer");
    *out = nullptr;
    if (!cfg->ark || !cfg->mds) return set_error(PMX_ERR_ARG, "pmx_ctx_acquire: null ark/mds");
    const uint64_t t64 = (uint64_t)cfg->rate + cfg->capacity, rounds = (uint64_t)cfg->full_rounds + cfg->partial_rounds;
    if (t64 == 0 || t64 > PMX_MAX_WIDTH || rounds == 0 || rounds > 4096) return pmx_ctx_create(cfg, device, out);   // let create() report it
    std::string blob = config_blob(cfg, device);
    const uint64_t key = fnv1a(blob);
    CtxCache &cc = ctx_cache();
    {
        std::lock_guard<std::mutex> lock(cc.lock);
        if (pmx_ctx *c = cache_find(cc, key, blob)) { *out = c; return PMX_OK; }
    }
    pmx_ctx *fresh = nullptr;
    int rc = pmx_ctx_create(cfg, device, &fresh);   // not under the lock
    if (rc) return rc;
    pmx_ctx *winner = nullptr;
    try {
        std::lock_guard<std::mutex> lock(cc.lock);
        winner = cache_find(cc, key, blob);
        if (!winner) {
            fresh->cache_blob = std::move(blob);
            try {
                cc.by_key[key].push_back(fresh);
            } catch (...) {
                // still under the lock: the map belongs to every thread (a bucket the failed insert may have left empty)
                auto it = cc.by_key.find(key);
                if (it != cc.by_key.end() && it->second.empty()) cc.by_key.erase(it);
                throw;
            }
            fresh->cache_key = key;      // set last: from here on the context belongs to the cache
            fresh->cache_refs = 1;
        }
    } catch (...) {
        (void)ctx_free(fresh);           // the lock is released by now; `fresh` never reached the cache
        throw;
    }
    if (winner) {
        (void)ctx_free(fresh);
        *out = winner;
    } else {
        *out = fresh;
    }
    return PMX_OK;
    PMX_ABI_END
}

// unlinks a context from its bucket (the caller holds the lock and frees it after letting go)
static void cache_unlink(CtxCache &cc, pmx_ctx *c) {
    auto b = cc.by_key.find(c->cache_key);
    if (b == cc.by_key.end()) return;
    for (auto it = b->second.begin(); it != b->second.end(); ++it)
        if (*it == c) { b->second.erase(it); break; }
    if (b->second.empty()) cc.by_key.erase(b);
}

extern "C" int pmx_ctx_release(pmx_ctx *ctx) {
    PMX_ABI_BEGIN("pmx_ctx_release")
    if (!ctx) return PMX_OK;
    if (!ctx->cache_key) return set_error(PMX_ERR_ARG, "pmx_ctx_release: this context came from pmx_ctx_create; use pmx_ctx_destroy");
    CtxCache &cc = ctx_cache();
    std::vector<pmx_ctx *> evicted;
    evicted.reserve(2);   // (nothing below may fail between unlinking a context and remembering it)
    {
        std::lock_guard<std::mutex> lock(cc.lock);
        if (ctx->cache_refs <= 0) return set_error(PMX_ERR_ARG, "pmx_ctx_release: released more often than acquired");
        if (--ctx->cache_refs == 0) {
            cc.idle.push_back(ctx);
            while (cc.idle.size() > kMaxIdle) {
                pmx_ctx *old = cc.idle.front();
                cc.idle.pop_front();
                cache_unlink(cc, old);
                evicted.push_back(old);
            }
        }
    }
    for (pmx_ctx *old : evicted) (void)ctx_free(old);   // stream synchronise + hipFree: not under the lock
    return PMX_OK;
    PMX_ABI_END
}

extern "C" int pmx_ctx_cache_clear(void) {
    PMX_ABI_BEGIN("pmx_ctx_cache_clear")
    CtxCache &cc = ctx_cache();
    std::vector<pmx_ctx *> evicted;
    {
        std::lock_guard<std::mutex> lock(cc.lock);
        evicted.reserve(cc.idle.size());
        while (!cc.idle.empty()) {
            pmx_ctx *old = cc.idle.front();
            cc.idle.pop_front();
            cache_unlink(cc, old);
            evicted.push_back(old);
        }
    }
    for (pmx_ctx *old : evicted) (void)ctx_free(old);
    return PMX_OK;
    PMX_ABI_END
}

extern "C" int pmx_ctx_width(const pmx_ctx *ctx) { return ctx ? (int)ctx->t : 0; }

extern "C" int pmx_ctx_engine_info(const pmx_ctx *ctx, int op, size_t n, size_t len, pmx_engine_info *out) {
    if (!ctx || !out) return set_error(PMX_ERR_ARG, "pmx_ctx_engine_info: null pointer");
    if (describe_launch(ctx->dev, ctx->t, op, n, len, out) != hipSuccess) return set_error(PMX_ERR_ARG, "pmx_ctx_engine_info: unknown op %d", op);
    return PMX_OK;
}

static bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

// ---- pinned host memory ----------------------------------------------------------------------------
extern "C" int pmx_host_alloc(void **ptr, size_t bytes) {
    if (!ptr) return set_error(PMX_ERR_ARG, "pmx_host_alloc: null pointer");
    *ptr = nullptr;
    PMX_HIP(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return PMX_OK;
}

extern "C" int pmx_host_free(void *ptr) {
    if (ptr) PMX_HIP(hipHostFree(ptr));
    return PMX_OK;
}

// ---- device memory for callers without HIP bindings ------------------------------------------------------
static int device_ok(int device) {
    const int ndev = pmx_device_count();
    if (ndev == 0) return set_error(PMX_ERR_HIP, "no HIP device available; this library has no CPU fallback");
    if (device < 0 || device >= ndev) return set_error(PMX_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    return PMX_OK;
}

extern "C" int pmx_device_alloc(int device, void **d_ptr, size_t bytes) {
    if (!d_ptr) return set_error(PMX_ERR_ARG, "pmx_device_alloc: null pointer");
    *d_ptr = nullptr;
    if (int rc = device_ok(device)) return rc;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    PMX_HIP(hipMalloc(d_ptr, bytes ? bytes : 16));
    return PMX_OK;
}

extern "C" int pmx_device_free(int device, void *d_ptr) {
    if (!d_ptr) return PMX_OK;
    if (int rc = device_ok(device)) return rc;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    PMX_HIP(hipFree(d_ptr));
    return PMX_OK;
}

static int device_copy(int device, void *dst, const void *src, size_t bytes, void *stream, hipMemcpyKind kind) {
    if ((!dst || !src) && bytes) return set_error(PMX_ERR_ARG, "pmx_device_upload / download: null pointer");
    if (bytes == 0) return PMX_OK;
    if (int rc = device_ok(device)) return rc;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    PMX_HIP(hipMemcpyAsync(dst, src, bytes, kind, (hipStream_t)stream));
    return PMX_OK;
}

extern "C" int pmx_device_upload(int device, void *d_dst, const void *h_src, size_t bytes, void *stream) {
    return device_copy(device, d_dst, h_src, bytes, stream, hipMemcpyHostToDevice);
}

extern "C" int pmx_device_download(int device, void *h_dst, const void *d_src, size_t bytes, void *stream) {
    return device_copy(device, h_dst, d_src, bytes, stream, hipMemcpyDeviceToHost);
}

extern "C" int pmx_stream_synchronize(int device, void *stream) {
    if (int rc = device_ok(device)) return rc;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    PMX_HIP(hipStreamSynchronize((hipStream_t)stream));
    return PMX_OK;
}

static bool is_pinned(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();   // an ordinary (unregistered) host pointer: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// A pageable buffer of a large call is page-locked for the length of the call (hipHostRegister: 4.2 ms for the 96 MiB of 2^20 t = 3
// states, unregister 0.1 ms - profiles/r06/g_host_path_probe.txt) so that it takes the same pipeline: 7 ms instead of the 8 ... 20 ms of the
// runtime's own staging through bounce buffers.  A registration that fails (locked-memory limit, a range that is not plain memory) leaves
// the buffer as it is and the call takes the single-chunk path.
struct ScopedPin {
    void *ptr = nullptr;
    ScopedPin(const void *p, size_t bytes) {
        if (p && bytes >= kMinBytes && !is_pinned(p)) {
            if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) == hipSuccess) ptr = const_cast<void *>(p);
            else (void)hipGetLastError();
        }
    }
    ~ScopedPin() {
        if (ptr && hipHostUnregister(ptr) != hipSuccess) (void)hipGetLastError();
    }
    ScopedPin(const ScopedPin &) = delete;
    ScopedPin &operator=(const ScopedPin &) = delete;
    static constexpr size_t kMinBytes = (size_t)16 << 20;   // below this the registration costs more than it saves
};

// chunks of a batch for the pinned pipeline: rows per chunk (a multiple of 256).  A chunk is a kernel launch of its own: at least 65536
// rows (below that a launch is bound by one permutation's latency, at t = 3 on the lone-permutation kernels), at most kPipeChunks chunks
// (2^20 t = 3 states: 16 chunks 2.50 ms, 8 chunks 2.56, 32 chunks 4.2 - profiles/r06/g_host_path_probe_graded_chunks_and_zero_copy_not_kept.txt)
static size_t pipeline_rows(size_t n) {
    size_t chunks = n >> 16;
    if (chunks <= 1) return n;
    if (chunks > (size_t)pmx_ctx::kPipeChunks) chunks = pmx_ctx::kPipeChunks;
    const size_t rows = (n + chunks - 1) / chunks;
    return (rows + 255) / 256 * 256;
}

// ---- permutation ---------------------------------------------------------------------------------
extern "C" int pmx_permute_batch_dev(pmx_ctx *ctx, uint64_t *d_states, size_t n, void *stream) {
    if (!ctx || (!d_states && n)) return set_error(PMX_ERR_ARG, "pmx_permute_batch_dev: null pointer");
    if (n == 0) return PMX_OK;
    if (!aligned16(d_states)) return set_error(PMX_ERR_ARG, "device pointers must be 16-byte aligned");
    if (n > (size_t)0x7fffffff * 64) return set_error(PMX_ERR_ARG, "batch too large");
    PMX_BIND(ctx);
    PMX_HIP(launch_permute(ctx->dev, ctx->t, d_states, n, (hipStream_t)stream));
    return PMX_OK;
}

extern "C" int pmx_permute_batch(pmx_ctx *ctx, uint64_t *states, size_t n) {
    PMX_ABI_BEGIN("pmx_permute_batch")
    if (!ctx || (!states && n)) return set_error(PMX_ERR_ARG, "pmx_permute_batch: null pointer");
    if (n == 0) return PMX_OK;
    PMX_BIND(ctx);
    int rc = PMX_OK;
    std::lock_guard<std::mutex> lock(ctx->host_lock);
    const size_t row = (size_t)ctx->t * 32;
    size_t bytes = 0;
    if ((rc = batch_bytes(n, ctx->t, &bytes))) return rc;
    void *d = nullptr;
    if ((rc = ctx_scratch(ctx, 0, bytes, &d))) return rc;
    ScopedPin pin(states, bytes);   // (declared in front of the drain: unregistered only after the drain has found the streams idle)
    StreamDrain drain{ctx};
    const size_t step = is_pinned(states) ? pipeline_rows(n) : n;   // pinned: upload / kernel / download overlap, both directions at once
    return host_pipeline(
        ctx, n, step,
        [&](size_t first, size_t cnt, hipStream_t st) -> int {
            PMX_HIP(hipMemcpyAsync((char *)d + first * row, (char *)states + first * row, cnt * row, hipMemcpyHostToDevice, st));
            return PMX_OK;
        },
        [&](size_t first, size_t cnt, hipStream_t st) -> int { return pmx_permute_batch_dev(ctx, (uint64_t *)((char *)d + first * row), cnt, st); },
        [&](size_t first, size_t cnt, hipStream_t st) -> int {
            PMX_HIP(hipMemcpyAsync((char *)states + first * row, (char *)d + first * row, cnt * row, hipMemcpyDeviceToHost, st));
            return PMX_OK;
        });
    PMX_ABI_END
}

// ---- hash ----------------------------------------------------------------------------------------
extern "C" int pmx_hash_batch_dev(pmx_ctx *ctx, const uint64_t *d_in, size_t in_len, uint64_t *d_out, size_t out_len,
                                  size_t n, void *stream) {
    if (!ctx || (!d_in && n && in_len) || (!d_out && n && out_len)) return set_error(PMX_ERR_ARG, "pmx_hash_batch_dev: null pointer");
    if (n == 0) return PMX_OK;
    if (!aligned16(d_in) || !aligned16(d_out)) return set_error(PMX_ERR_ARG, "device pointers must be 16-byte aligned");
    if (n > (size_t)0x7fffffff * 64) return set_error(PMX_ERR_ARG, "batch too large");
    PMX_BIND(ctx);
    // two elements in, one out, rate >= 2: this IS the 2-to-1 compression (same memory layout as one tree level), whose
    // launcher has the quad kernel for small batches - a batch of authentication paths advances one level per call
    if (in_len == 2 && out_len == 1 && ctx->dev.rounds.rate >= 2) {
        PMX_HIP(launch_compress(ctx->dev, ctx->t, d_in, d_out, n, (hipStream_t)stream));
        return PMX_OK;
    }
    PMX_HIP(launch_hash(ctx->dev, ctx->t, d_in, in_len, d_out, out_len, n, (hipStream_t)stream));
    return PMX_OK;
}

extern "C" int pmx_hash_batch(pmx_ctx *ctx, const uint64_t *in, size_t in_len, uint64_t *out, size_t out_len, size_t n) {
    PMX_ABI_BEGIN("pmx_hash_batch")
    if (!ctx || (!in && n && in_len) || (!out && n && out_len)) return set_error(PMX_ERR_ARG, "pmx_hash_batch: null pointer");
    if (n == 0) return PMX_OK;
    PMX_BIND(ctx);
    int rc = PMX_OK;
    std::lock_guard<std::mutex> lock(ctx->host_lock);
    size_t in_bytes = 0, out_bytes = 0;
    if ((rc = batch_bytes(n, in_len, &in_bytes)) || (rc = batch_bytes(n, out_len, &out_bytes))) return rc;
    void *d_in = nullptr, *d_out = nullptr;
    if ((rc = ctx_scratch(ctx, 0, in_bytes, &d_in))) return rc;
    if ((rc = ctx_scratch(ctx, 1, out_bytes, &d_out))) return rc;
    ScopedPin pin_in(in, in_bytes), pin_out(out, out_bytes);   // (in front of the drain: see pmx_permute_batch)
    StreamDrain drain{ctx};
    const size_t in_row = in_len * 32, out_row = out_len * 32;
    const size_t step = ((!in_bytes || is_pinned(in)) && (!out_bytes || is_pinned(out))) ? pipeline_rows(n) : n;
    return host_pipeline(
        ctx, n, step,
        [&](size_t first, size_t cnt, hipStream_t st) -> int {
            if (in_bytes) PMX_HIP(hipMemcpyAsync((char *)d_in + first * in_row, (const char *)in + first * in_row, cnt * in_row, hipMemcpyHostToDevice, st));
            return PMX_OK;
        },
        [&](size_t first, size_t cnt, hipStream_t st) -> int {
            return pmx_hash_batch_dev(ctx, (const uint64_t *)((char *)d_in + first * in_row), in_len, (uint64_t *)((char *)d_out + first * out_row), out_len, cnt, st);
        },
        [&](size_t first, size_t cnt, hipStream_t st) -> int {
            if (out_bytes) PMX_HIP(hipMemcpyAsync((char *)out + first * out_row, (char *)d_out + first * out_row, cnt * out_row, hipMemcpyDeviceToHost, st));
            return PMX_OK;
        });
    PMX_ABI_END
}

// ---- duplex sponge driver ------------------------------------------------------------------------
// The pass drivers launch one kernel per permutation a sponge of the call can need: a call is limited to kMaxPasses of them
// (65536 rates of elements per sponge and call - 16 MiB at rate 8; longer inputs are absorbed in several calls, which is the same
// thing to a duplex sponge).  The limit is the same for EVERY absorb and squeeze call, whichever engine the width and the batch
// size select (the per-lane kernels loop inside one launch and would not need it: a caller's length limit must not depend on
// its batch size).
static constexpr size_t kMaxPasses = 65536;  // launches = permutations a sponge of the call can need
static int check_pass_count(const pmx_ctx *ctx, int op, size_t /*n*/, size_t len, const char *who) {
    const uint32_t rate = ctx->dev.rounds.rate;
    const size_t passes = rate == 0 ? 0 : (op == PMX_OP_SQUEEZE ? squeeze_passes(len, rate) : absorb_passes(len, rate));
    if (passes > kMaxPasses + 1)     // (passes - 1 permutation launches)
        return set_error(PMX_ERR_ARG, "%s: %zu elements per sponge is more than 65536 rates (%u) in one call; split the call", who, len, rate);
    return PMX_OK;
}

// PassScratch::get / done for a context (called with ctx->pass_lock held).  Nothing on this path frees memory (hipFree waits for
// the whole device) or asks a CALLER's stream anything (the handle may belong to a stream that is being captured, or that is gone):
// a block is released by the context's own event, recorded behind the last launch that uses it.
static hipError_t ctx_pass_scratch(void *owner, hipStream_t st, size_t bytes, uint32_t **out) {
    pmx_ctx *ctx = static_cast<pmx_ctx *>(owner);
    pmx_ctx::PassBlock *pick = nullptr;
    for (pmx_ctx::PassBlock &b : ctx->pass_pool)            // the block this stream used last: ordered behind that use by the stream itself
        if (b.recorded && b.stream == st && b.bytes >= bytes) { pick = &b; break; }
    if (!pick) {
        for (pmx_ctx::PassBlock &b : ctx->pass_pool) {      // the smallest idle block that is large enough
            if (b.poisoned || b.bytes < bytes || (pick && pick->bytes <= b.bytes)) continue;
            if (b.recorded && hipEventQuery(b.done) != hipSuccess) {
                (void)hipGetLastError();                    // not ready - or not queryable (recorded into a capture): not idle
                continue;
            }
            pick = &b;
        }
    }
    if (!pick) {
        pmx_ctx::PassBlock fresh;
        size_t want = (bytes + 4095) & ~(size_t)4095;
        for (const pmx_ctx::PassBlock &b : ctx->pass_pool)  // grow in doublings: an outgrown block stays in the pool for smaller calls
            if (b.bytes * 2 > want && b.bytes < bytes) want = b.bytes * 2;
        hipError_t e = hipMalloc(&fresh.ptr, want);
        if (e != hipSuccess) return e;
        e = hipEventCreateWithFlags(&fresh.done, hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipFree(fresh.ptr);
            return e;
        }
        fresh.bytes = want;
        ctx->pass_pool.push_back(fresh);
        pick = &ctx->pass_pool.back();
    }
    pick->stream = st;
    pick->recorded = false;      // in use by a call that is being enqueued: `done` records the event
    *out = static_cast<uint32_t *>(pick->ptr);
    return hipSuccess;
}
static void ctx_pass_done(void *owner, hipStream_t st, uint32_t *block) {
    pmx_ctx *ctx = static_cast<pmx_ctx *>(owner);
    for (pmx_ctx::PassBlock &b : ctx->pass_pool) {
        if (b.ptr != block) continue;
        // a record that failed leaves `done` unrecorded (or holding an older, completed record): a query would call the block idle while
        // this call's launches may still read it - such a block is never handed to another stream again (its own stream orders the reuse)
        if (hipEventRecord(b.done, st) != hipSuccess) {
            (void)hipGetLastError();
            b.poisoned = true;
        }
        b.recorded = true;
        return;
    }
}

extern "C" int pmx_sponge_absorb_batch_dev(pmx_ctx *ctx, uint64_t *d_states, uint32_t *d_tag, uint32_t *d_index,
                                           const uint64_t *d_in, size_t in_len, size_t n, void *stream) {
    if (!ctx || ((!d_states || !d_tag || !d_index) && n) || (!d_in && n && in_len))
        return set_error(PMX_ERR_ARG, "pmx_sponge_absorb_batch_dev: null pointer");
    if (n == 0 || in_len == 0) return PMX_OK;  // absorbing an empty input changes nothing (mod.rs:234-236)
    if (!aligned16(d_states) || !aligned16(d_in)) return set_error(PMX_ERR_ARG, "device pointers must be 16-byte aligned");
    if (n > (size_t)0x7fffffff * 64) return set_error(PMX_ERR_ARG, "batch too large");
    if (int rc = check_pass_count(ctx, PMX_OP_ABSORB, n, in_len, "pmx_sponge_absorb_batch_dev")) return rc;
    PMX_ABI_BEGIN("pmx_sponge_absorb_batch_dev")
    PMX_BIND(ctx);
    std::lock_guard<std::mutex> lock(ctx->pass_lock);
    PMX_HIP(launch_absorb(ctx->dev, ctx->t, d_states, d_tag, d_index, d_in, in_len, n, (hipStream_t)stream, PassScratch{ctx, ctx_pass_scratch, ctx_pass_done}));
    return PMX_OK;
    PMX_ABI_END
}

extern "C" int pmx_sponge_squeeze_batch_dev(pmx_ctx *ctx, uint64_t *d_states, uint32_t *d_tag, uint32_t *d_index,
                                            uint64_t *d_out, size_t out_len, size_t n, void *stream) {
    if (!ctx || ((!d_states || !d_tag || !d_index) && n) || (!d_out && n && out_len))
        return set_error(PMX_ERR_ARG, "pmx_sponge_squeeze_batch_dev: null pointer");
    if (n == 0) return PMX_OK;
    if (!aligned16(d_states) || !aligned16(d_out)) return set_error(PMX_ERR_ARG, "device pointers must be 16-byte aligned");
    if (n > (size_t)0x7fffffff * 64) return set_error(PMX_ERR_ARG, "batch too large");
    if (int rc = check_pass_count(ctx, PMX_OP_SQUEEZE, n, out_len, "pmx_sponge_squeeze_batch_dev")) return rc;
    PMX_ABI_BEGIN("pmx_sponge_squeeze_batch_dev")
    PMX_BIND(ctx);
    std::lock_guard<std::mutex> lock(ctx->pass_lock);
    PMX_HIP(launch_squeeze(ctx->dev, ctx->t, d_states, d_tag, d_index, d_out, out_len, n, (hipStream_t)stream, PassScratch{ctx, ctx_pass_scratch, ctx_pass_done}));
    return PMX_OK;
    PMX_ABI_END
}

static constexpr size_t kSmallCallBytes = 64 * 1024;

static int check_modes(const pmx_ctx *ctx, const uint32_t *tag, const uint32_t *index, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        if (tag[i] > PMX_MODE_SQUEEZING) return set_error(PMX_ERR_ARG, "sponge %zu: mode tag %u is neither Absorbing nor Squeezing", i, tag[i]);
        if (index[i] > ctx->dev.rounds.rate) return set_error(PMX_ERR_ARG, "sponge %zu: mode index %u > rate %u", i, index[i], ctx->dev.rounds.rate);
    }
    return PMX_OK;
}

static int sponge_host(pmx_ctx *ctx, uint64_t *states, uint32_t *tag, uint32_t *index, const uint64_t *in, uint64_t *out,
                       size_t len, size_t n, bool absorb) {
    PMX_ABI_BEGIN("pmx_sponge_*_batch")
    if (!ctx || ((!states || !tag || !index) && n)) return set_error(PMX_ERR_ARG, "pmx_sponge_*_batch: null pointer");
    if (n == 0) return PMX_OK;
    if (absorb && len == 0) return PMX_OK;
    if ((absorb && !in) || (!absorb && !out && len)) return set_error(PMX_ERR_ARG, "pmx_sponge_*_batch: null data pointer");
    int rc = check_modes(ctx, tag, index, n);
    if (rc) return rc;
    // The reference takes any length (mod.rs:232-254, 321-341); a device call moves at most kMaxPasses rates per sponge (check_pass_count).
    // To a duplex sponge `absorb(a ++ b)` is `absorb(a); absorb(b)`, and `squeeze(k1 + k2)` is `squeeze(k1); squeeze(k2)` unless the
    // second call asks for exactly `rate` elements from a sponge that stands inside its rate (the test of mod.rs:175 then skips a
    // permutation the long call performs): a longer call is cut into pieces of kMaxPasses rates, never leaving a last piece of one rate.
    const size_t rate = ctx->dev.rounds.rate, max_len = kMaxPasses * rate;
    if (rate != 0 && len > max_len) {
        std::vector<uint64_t> rows;   // n > 1: the piece of every sponge, packed [n][piece]
        for (size_t done = 0; done < len;) {
            size_t piece = len - done < max_len ? len - done : max_len;
            if (len - done - piece == rate) piece -= rate;   // (kMaxPasses > 1: the piece stays positive and is not `rate` itself)
            const uint64_t *in_piece = in ? in + done * 4 : nullptr;
            uint64_t *out_piece = out ? out + done * 4 : nullptr;
            if (n > 1) {
                if (n > (SIZE_MAX / 32) / piece) return set_error(PMX_ERR_ARG, "batch byte size overflows size_t");
                rows.resize(n * piece * 4);
                if (absorb)
                    for (size_t i = 0; i < n; ++i) std::memcpy(rows.data() + i * piece * 4, in + (i * len + done) * 4, piece * 32);
                in_piece = out_piece = rows.data();
            }
            if ((rc = sponge_host(ctx, states, tag, index, absorb ? in_piece : nullptr, absorb ? nullptr : out_piece, piece, n, absorb))) return rc;
            if (!absorb && n > 1)
                for (size_t i = 0; i < n; ++i) std::memcpy(out + (i * len + done) * 4, rows.data() + i * piece * 4, piece * 32);
            done += piece;
        }
        return PMX_OK;
    }
    PMX_BIND(ctx);
    std::lock_guard<std::mutex> lock(ctx->host_lock);
    size_t st_bytes = 0, io_bytes = 0;
    if ((rc = batch_bytes(n, ctx->t, &st_bytes)) || (rc = batch_bytes(n, len, &io_bytes))) return rc;
    // A handful of sponges (the single PoseidonSponge of the trait shims is n = 1): the call is all latency, and eight
    // separate copies from pageable memory cost as much as the permutation itself.  Pack [states | io | tag | index] into
    // the context's page-locked block: one copy in, the kernel, one copy out.
    const size_t words = (n * 4 + 15) / 16 * 16, packed = st_bytes + io_bytes + 2 * words;
    if (packed <= kSmallCallBytes) {
        if (!ctx->pinned) PMX_HIP(hipHostMalloc(&ctx->pinned, kSmallCallBytes, hipHostMallocDefault));
        void *d = nullptr;
        if ((rc = ctx_scratch(ctx, 0, packed, &d))) return rc;
        char *h = (char *)ctx->pinned, *dd = (char *)d;
        const size_t o_io = st_bytes, o_tag = st_bytes + io_bytes, o_idx = o_tag + words;
        std::memcpy(h, states, st_bytes);
        if (absorb) std::memcpy(h + o_io, in, io_bytes);
        std::memcpy(h + o_tag, tag, n * 4);
        std::memcpy(h + o_idx, index, n * 4);
        StreamDrain drain{ctx};
        PMX_HIP(hipMemcpyAsync(dd, h, packed, hipMemcpyHostToDevice, ctx->stream));
        if (absorb) rc = pmx_sponge_absorb_batch_dev(ctx, (uint64_t *)dd, (uint32_t *)(dd + o_tag), (uint32_t *)(dd + o_idx), (const uint64_t *)(dd + o_io), len, n, ctx->stream);
        else rc = pmx_sponge_squeeze_batch_dev(ctx, (uint64_t *)dd, (uint32_t *)(dd + o_tag), (uint32_t *)(dd + o_idx), (uint64_t *)(dd + o_io), len, n, ctx->stream);
        if (rc) return rc;
        PMX_HIP(hipMemcpyAsync(h, dd, packed, hipMemcpyDeviceToHost, ctx->stream));
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        std::memcpy(states, h, st_bytes);
        std::memcpy(tag, h + o_tag, n * 4);
        std::memcpy(index, h + o_idx, n * 4);
        if (!absorb && io_bytes) std::memcpy(out, h + o_io, io_bytes);
        return PMX_OK;
    }
    void *d_st = nullptr, *d_io = nullptr, *d_tag = nullptr, *d_idx = nullptr;
    if ((rc = ctx_scratch(ctx, 0, st_bytes, &d_st))) return rc;
    if ((rc = ctx_scratch(ctx, 1, io_bytes, &d_io))) return rc;
    if ((rc = ctx_scratch(ctx, 2, n * 4, &d_tag))) return rc;
    if ((rc = ctx_scratch(ctx, 3, n * 4, &d_idx))) return rc;
    StreamDrain drain{ctx};
    PMX_HIP(hipMemcpyAsync(d_st, states, st_bytes, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_tag, tag, n * 4, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_idx, index, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (absorb) {
        PMX_HIP(hipMemcpyAsync(d_io, in, io_bytes, hipMemcpyHostToDevice, ctx->stream));
        rc = pmx_sponge_absorb_batch_dev(ctx, (uint64_t *)d_st, (uint32_t *)d_tag, (uint32_t *)d_idx, (const uint64_t *)d_io, len, n, ctx->stream);
    } else {
        rc = pmx_sponge_squeeze_batch_dev(ctx, (uint64_t *)d_st, (uint32_t *)d_tag, (uint32_t *)d_idx, (uint64_t *)d_io, len, n, ctx->stream);
    }
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(states, d_st, st_bytes, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(tag, d_tag, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(index, d_idx, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (!absorb && io_bytes) PMX_HIP(hipMemcpyAsync(out, d_io, io_bytes, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_ABI_END
}

extern "C" int pmx_sponge_absorb_batch(pmx_ctx *ctx, uint64_t *states, uint32_t *mode_tag, uint32_t *mode_index,
                                       const uint64_t *in, size_t in_len, size_t n) {
    return sponge_host(ctx, states, mode_tag, mode_index, in, nullptr, in_len, n, true);
}

extern "C" int pmx_sponge_squeeze_batch(pmx_ctx *ctx, uint64_t *states, uint32_t *mode_tag, uint32_t *mode_index,
                                        uint64_t *out, size_t out_len, size_t n) {
    return sponge_host(ctx, states, mode_tag, mode_index, nullptr, out, out_len, n, false);
}

// ---- Merkle 2-to-1 -------------------------------------------------------------------------------
// Level by level on the caller's stream: level l reads the n_leaves >> (l-1) nodes of level l-1 and writes
// n_leaves >> l parents.  (Cutting the tree into subtrees on concurrent streams was measured SLOWER - 7.0 ms ->
// 12.5 ms at 8 streams for 2^21 leaves - because the narrow levels are latency-bound, not launch-bound.)
extern "C" int pmx_merkle_2to1_dev(pmx_ctx *ctx, uint64_t *d_nodes, size_t n_leaves, void *stream) {
    if (!ctx || !d_nodes) return set_error(PMX_ERR_ARG, "pmx_merkle_2to1_dev: null pointer");
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) return set_error(PMX_ERR_ARG, "n_leaves must be a power of two");
    if (ctx->dev.rounds.rate < 2) return set_error(PMX_ERR_CONFIG, "2-to-1 compression needs rate >= 2");
    if (!aligned16(d_nodes)) return set_error(PMX_ERR_ARG, "device pointers must be 16-byte aligned");
    PMX_BIND(ctx);
    size_t src = 0, width = n_leaves;
    while (width > 1) {
        PMX_HIP(launch_compress(ctx->dev, ctx->t, d_nodes + src * 4, d_nodes + (src + width) * 4, width / 2, (hipStream_t)stream));
        src += width;
        width /= 2;
    }
    return PMX_OK;
}

// ---- many trees at once ------------------------------------------------------------------------------
// The levels of ONE tree narrow down to a single compression, and a level of at most 16384 compressions costs the same 65 us
// whatever its width (one permutation's dependent chain): 15 such levels are a quarter of a 2^21-leaf tree's
// time.  n_trees trees of the same size advance TOGETHER, level by level: with the leaves laid out tree after tree, level l of
// every tree is one contiguous array, its pairs never straddle two trees (leaves_per_tree is a power of two), and the narrowest
// level launched is n_trees compressions wide.  It is the first log2(leaves_per_tree) levels of one tree over all the leaves.
static int forest_shape(size_t n_trees, size_t leaves_per_tree, size_t *total_leaves, size_t *total_nodes) {
    if (n_trees == 0 || leaves_per_tree == 0 || (leaves_per_tree & (leaves_per_tree - 1)))
        return set_error(PMX_ERR_ARG, "a forest needs n_trees >= 1 and leaves_per_tree a power of two");
    if (n_trees > (SIZE_MAX / 128) / leaves_per_tree) return set_error(PMX_ERR_ARG, "forest byte size overflows size_t");
    *total_leaves = n_trees * leaves_per_tree;
    *total_nodes = n_trees * (2 * leaves_per_tree - 1);
    return PMX_OK;
}

extern "C" int pmx_merkle_2to1_forest_dev(pmx_ctx *ctx, uint64_t *d_nodes, size_t n_trees, size_t leaves_per_tree, void *stream) {
    if (!ctx || !d_nodes) return set_error(PMX_ERR_ARG, "pmx_merkle_2to1_forest_dev: null pointer");
    size_t total = 0, n_nodes = 0;
    if (int rc = forest_shape(n_trees, leaves_per_tree, &total, &n_nodes)) return rc;
    if (ctx->dev.rounds.rate < 2) return set_error(PMX_ERR_CONFIG, "2-to-1 compression needs rate >= 2");
    if (!aligned16(d_nodes)) return set_error(PMX_ERR_ARG, "device pointers must be 16-byte aligned");
    PMX_BIND(ctx);
    size_t src = 0, width = total;
    while (width > n_trees) {      // level l of all trees: [src, src + width) -> [src + width, src + width + width / 2)
        PMX_HIP(launch_compress(ctx->dev, ctx->t, d_nodes + src * 4, d_nodes + (src + width) * 4, width / 2, (hipStream_t)stream));
        src += width;
        width /= 2;
    }
    return PMX_OK;
}

extern "C" int pmx_merkle_2to1_forest(pmx_ctx *ctx, const uint64_t *leaves, size_t n_trees, size_t leaves_per_tree, uint64_t *nodes,
                                      uint64_t *roots) {
    PMX_ABI_BEGIN("pmx_merkle_2to1_forest")
    if (!ctx || !leaves) return set_error(PMX_ERR_ARG, "pmx_merkle_2to1_forest: null pointer");
    size_t total = 0, n_nodes = 0;
    int rc = forest_shape(n_trees, leaves_per_tree, &total, &n_nodes);
    if (rc) return rc;
    PMX_BIND(ctx);
    std::lock_guard<std::mutex> lock(ctx->host_lock);
    void *d = nullptr;
    if ((rc = ctx_scratch(ctx, 0, n_nodes * 32, &d))) return rc;
    StreamDrain drain{ctx};
    PMX_HIP(hipMemcpyAsync(d, leaves, total * 32, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pmx_merkle_2to1_forest_dev(ctx, (uint64_t *)d, n_trees, leaves_per_tree, ctx->stream))) return rc;
    if (nodes) PMX_HIP(hipMemcpyAsync(nodes, d, n_nodes * 32, hipMemcpyDeviceToHost, ctx->stream));
    if (roots) PMX_HIP(hipMemcpyAsync(roots, (uint64_t *)d + (n_nodes - n_trees) * 4, n_trees * 32, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_ABI_END
}

extern "C" int pmx_merkle_2to1(pmx_ctx *ctx, const uint64_t *leaves, size_t n_leaves, uint64_t *nodes, uint64_t *root) {
    PMX_ABI_BEGIN("pmx_merkle_2to1")
    if (!ctx || !leaves) return set_error(PMX_ERR_ARG, "pmx_merkle_2to1: null pointer");
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) return set_error(PMX_ERR_ARG, "n_leaves must be a power of two");
    PMX_BIND(ctx);
    int rc = PMX_OK;
    if (n_leaves > SIZE_MAX / 64) return set_error(PMX_ERR_ARG, "tree byte size overflows size_t");
    std::lock_guard<std::mutex> lock(ctx->host_lock);
    const size_t n_nodes = 2 * n_leaves - 1;
    void *d = nullptr;
    if ((rc = ctx_scratch(ctx, 0, n_nodes * 32, &d))) return rc;
    StreamDrain drain{ctx};
    PMX_HIP(hipMemcpyAsync(d, leaves, n_leaves * 32, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pmx_merkle_2to1_dev(ctx, (uint64_t *)d, n_leaves, ctx->stream))) return rc;
    if (nodes) PMX_HIP(hipMemcpyAsync(nodes, d, n_nodes * 32, hipMemcpyDeviceToHost, ctx->stream));
    if (root) PMX_HIP(hipMemcpyAsync(root, (uint64_t *)d + (n_nodes - 1) * 4, 32, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_ABI_END
}

// ---- authentication paths ---------------------------------------------------------------------------------------------
extern "C" int pmx_merkle_paths(const uint64_t *nodes, size_t n_leaves, const uint64_t *indices, size_t k, uint64_t *paths_out) {
    if ((!nodes || !indices || !paths_out) && k) return set_error(PMX_ERR_ARG, "pmx_merkle_paths: null pointer");
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) return set_error(PMX_ERR_ARG, "n_leaves must be a power of two");
    size_t depth = 0;
    while (((size_t)1 << depth) < n_leaves) ++depth;
    for (size_t i = 0; i < k; ++i) {
        if (indices[i] >= n_leaves) return set_error(PMX_ERR_ARG, "leaf index %llu out of range", (unsigned long long)indices[i]);
        size_t idx = (size_t)indices[i], first = 0, width = n_leaves;   // first node of the current level, its width
        for (size_t level = 0; level < depth; ++level) {
            std::memcpy(paths_out + (i * depth + level) * 4, nodes + (first + (idx ^ 1)) * 4, 32);
            first += width;
            width /= 2;
            idx >>= 1;
        }
    }
    return PMX_OK;
}

// k authentication paths advance one level per launch, everything resident on the device: cur[i] starts as leaves[i];
// per level a gather kernel lays (cur[i], sibling) out as the pair array one tree level has - left / right by bit `level`
// of indices[i] - and the 2-to-1 compression launcher (quad kernel for k <= 32768) writes the parents back into cur.
// d_work: [k][12] u64 of scratch (cur [k][4], then pairs [k][8]).  d_ok[i] = 1 iff cur[i] == root after `depth` levels and
// indices[i] < 2^depth (an index with bits at or above `depth` names no leaf of this tree).
extern "C" int pmx_merkle_verify_paths_dev(pmx_ctx *ctx, const uint64_t *d_leaves, const uint64_t *d_indices, const uint64_t *d_paths,
                                           size_t depth, size_t k, const uint64_t *d_root, uint8_t *d_ok, uint64_t *d_work, void *stream) {
    if (!ctx || ((!d_leaves || !d_indices || !d_ok || !d_work) && k) || (!d_paths && k && depth) || !d_root)
        return set_error(PMX_ERR_ARG, "pmx_merkle_verify_paths_dev: null pointer");
    if (ctx->dev.rounds.rate < 2) return set_error(PMX_ERR_CONFIG, "2-to-1 compression needs rate >= 2");
    if (depth >= 64) return set_error(PMX_ERR_ARG, "depth out of range");
    if (k == 0) return PMX_OK;
    if (k > (size_t)0x7fffffff * 64 || (depth && k > (SIZE_MAX / 32) / depth)) return set_error(PMX_ERR_ARG, "batch too large");
    if (!aligned16(d_leaves) || !aligned16(d_paths) || !aligned16(d_work) || !aligned16(d_root)) return set_error(PMX_ERR_ARG, "device pointers must be 16-byte aligned");
    PMX_BIND(ctx);
    hipStream_t st = (hipStream_t)stream;
    uint64_t *cur = d_work, *pairs = d_work + k * 4;
    PMX_HIP(hipMemcpyAsync(cur, d_leaves, k * 32, hipMemcpyDeviceToDevice, st));
    for (size_t level = 0; level < depth; ++level) {
        PMX_HIP(launch_path_pairs(cur, d_paths, d_indices, depth, level, pairs, k, st));
        PMX_HIP(launch_compress(ctx->dev, ctx->t, pairs, cur, k, st));
    }
    PMX_HIP(launch_path_check(cur, d_root, d_indices, depth, d_ok, k, st));
    return PMX_OK;
}

// Host buffers: one upload of (leaves, indices, paths, root), `depth` level steps on the device, one download of ok.
extern "C" int pmx_merkle_verify_paths(pmx_ctx *ctx, const uint64_t *leaves, const uint64_t *indices, const uint64_t *paths,
                                       size_t depth, size_t k, const uint64_t root[PMX_LIMBS], uint8_t *ok_out) {
    PMX_ABI_BEGIN("pmx_merkle_verify_paths")
    if (!ctx || ((!leaves || !indices || !ok_out) && k) || (!paths && k && depth) || !root)
        return set_error(PMX_ERR_ARG, "pmx_merkle_verify_paths: null pointer");
    if (ctx->dev.rounds.rate < 2) return set_error(PMX_ERR_CONFIG, "2-to-1 compression needs rate >= 2");
    if (depth >= 64) return set_error(PMX_ERR_ARG, "depth out of range");
    if (k == 0) return PMX_OK;
    if (k > SIZE_MAX / 128 || (depth && k > (SIZE_MAX / 32) / depth)) return set_error(PMX_ERR_ARG, "batch byte size overflows size_t");
    PMX_BIND(ctx);
    int rc = PMX_OK;
    std::lock_guard<std::mutex> lock(ctx->host_lock);
    // slot 0: leaves [k][4] | work [k][12];  slot 1: paths [k][depth][4];  slot 2: root [4] | indices [k];  slot 3: ok [k]
    void *d0 = nullptr, *d1 = nullptr, *d2 = nullptr, *d3 = nullptr;
    if ((rc = ctx_scratch(ctx, 0, k * 128, &d0))) return rc;
    if ((rc = ctx_scratch(ctx, 1, k * depth * 32, &d1))) return rc;
    if ((rc = ctx_scratch(ctx, 2, 32 + k * 8, &d2))) return rc;
    if ((rc = ctx_scratch(ctx, 3, k, &d3))) return rc;
    uint64_t *d_leaves = (uint64_t *)d0, *d_work = d_leaves + k * 4, *d_root = (uint64_t *)d2, *d_idx = d_root + 4;
    StreamDrain drain{ctx};
    PMX_HIP(hipMemcpyAsync(d_leaves, leaves, k * 32, hipMemcpyHostToDevice, ctx->stream));
    if (depth) PMX_HIP(hipMemcpyAsync(d1, paths, k * depth * 32, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_root, root, 32, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_idx, indices, k * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pmx_merkle_verify_paths_dev(ctx, d_leaves, d_idx, (const uint64_t *)d1, depth, k, d_root, (uint8_t *)d3, d_work, ctx->stream))) return rc;
    PMX_HIP(hipMemcpyAsync(ok_out, d3, k, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_ABI_END
}
