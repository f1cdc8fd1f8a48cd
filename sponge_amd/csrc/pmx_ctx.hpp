// Private definition of pmx_ctx and the helpers shared by pmx_api.cpp (one device) and pmx_mgpu.cpp (device groups).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/poseidon_mi355x.h"
#include "pmx_internal.hpp"

struct pmx_ctx {
    int device = 0;
    uint32_t t = 0;
    pmx::DevConfig dev{};            // kernel-argument block (points at d_consts)
    uint32_t *d_consts = nullptr;    // device: constant table (pmx_prepare.hpp layout)
    hipStream_t stream = nullptr;    // used by the host-buffer entry points (and by a device group for this device)
    hipStream_t stream2 = nullptr;   // the pinned-memory pipeline: `stream` only uploads, `stream2` only computes, `stream3` only downloads
    hipStream_t stream3 = nullptr;
    static constexpr int kPipeChunks = 16;
    hipEvent_t pipe_up[kPipeChunks] = {}, pipe_done[kPipeChunks] = {};   // chunk i uploaded / computed (created with the context)
    void *scratch[4] = {nullptr, nullptr, nullptr, nullptr};   // grow-only device staging for the host-buffer entry points
    size_t scratch_bytes[4] = {0, 0, 0, 0};
    void *pinned = nullptr;          // page-locked host block for small host-buffer calls (allocated on first use)
    // The host-buffer entry points use the staging buffers and the two streams above: one caller at a time.  Contexts
    // handed out by pmx_ctx_acquire are shared between sponges (and threads), so those entry points take this lock;
    // the *_dev entry points only read the immutable fields and enqueue on the caller's stream.
    std::mutex host_lock;
    // Pass lists of the absorb / squeeze drivers that run as passes (pmx_device.hip: sponge_passes): a pool of device blocks, each
    // with an event the CONTEXT owns, recorded on the caller's stream behind the last launch that uses the block.  A call takes the
    // block its stream used last (stream order makes that safe while the earlier call is still running), else any block whose event
    // has completed, else a new one; nothing is ever freed or queried through a caller's stream handle on the enqueue path - the
    // blocks go when the context goes.  pass_lock is held while a driver call enqueues, which also keeps two threads from
    // interleaving their launches on one stream of this context.
    struct PassBlock { void *ptr = nullptr; size_t bytes = 0; hipStream_t stream = nullptr; hipEvent_t done = nullptr; bool recorded = false; bool poisoned = false; };
    std::mutex pass_lock;
    std::vector<PassBlock> pass_pool;
    // pmx_ctx_acquire / pmx_ctx_release bookkeeping (0 for contexts made by pmx_ctx_create)
    uint64_t cache_key = 0;
    long cache_refs = 0;
    std::string cache_blob;          // the config bytes the key was computed from (collision check)
};

namespace pmx {

int hip_fail(hipError_t e, const char *what);

#define PMX_HIP(expr)                                             \
    do {                                                          \
        hipError_t e_ = (expr);                                   \
        if (e_ != hipSuccess) return ::pmx::hip_fail(e_, #expr);  \
    } while (0)

// Every entry point runs with its context's device current and puts the caller's device back on the way out: in a
// process that drives several GPUs (torch, a device group) a call must not leave the thread's current device changed
// behind the caller's back.
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
        else prev = -1;   // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

#define PMX_BIND(ctx)                                                                \
    ::pmx::DeviceGuard device_guard_((ctx)->device);                                 \
    if (device_guard_.err != hipSuccess) return ::pmx::hip_fail(device_guard_.err, "hipSetDevice")

}  // namespace pmx
