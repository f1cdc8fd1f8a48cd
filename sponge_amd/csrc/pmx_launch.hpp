// Launcher entry points of pmx_device.hip (host-callable; they only enqueue on `st`).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/poseidon_mi355x.h"
#include "pmx_internal.hpp"

namespace pmx {

using EngineInfo = ::pmx_engine_info;
// pmx_ctx_engine_info: the engine `op` over n units would be launched on, decided by the launchers' own conditions
hipError_t describe_launch(const DevConfig &c, uint32_t t, int op, size_t n, size_t len, EngineInfo *o);

hipError_t launch_permute(const DevConfig &c, uint32_t t, uint64_t *states, size_t n, hipStream_t st);
hipError_t launch_hash(const DevConfig &c, uint32_t t, const uint64_t *in, size_t in_len, uint64_t *out, size_t out_len,
                       size_t n, hipStream_t st);
// out[i] = 2-to-1 compression of in[2i], in[2i+1]   (rate >= 2)
hipError_t launch_compress(const DevConfig &c, uint32_t t, const uint64_t *in, uint64_t *out, size_t n, hipStream_t st);
// Device scratch for the pass lists of the drivers that run as passes (pmx_device.hip: sponge_passes): `get` hands out at least
// `bytes` bytes that stay valid for everything enqueued on `st` by this call, `done` is called once behind the call's last launch
// (pmx_api.cpp: a pool of blocks owned by the context, each released by an event recorded there).  Engines that need no lists
// call neither.
struct PassScratch {
    void *owner = nullptr;
    hipError_t (*get)(void *owner, hipStream_t st, size_t bytes, uint32_t **out) = nullptr;
    void (*done)(void *owner, hipStream_t st, uint32_t *block) = nullptr;
};
hipError_t launch_absorb(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                         const uint64_t *in, size_t in_len, size_t n, hipStream_t st, const PassScratch &scratch);
hipError_t launch_squeeze(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                          uint64_t *out, size_t out_len, size_t n, hipStream_t st, const PassScratch &scratch);

// Authentication paths, one level per step (pmx_merkle_verify_paths_dev): pairs[i] = (cur[i], sibling) or (sibling, cur[i])
// by bit `level` of indices[i], sibling = paths[i][level]; then ok[i] = (cur[i] == root) && indices[i] < 2^depth.
hipError_t launch_path_pairs(const uint64_t *cur, const uint64_t *paths, const uint64_t *indices, size_t depth, size_t level,
                             uint64_t *pairs, size_t k, hipStream_t st);
hipError_t launch_path_check(const uint64_t *cur, const uint64_t *root, const uint64_t *indices, size_t depth, uint8_t *ok,
                             size_t k, hipStream_t st);

}  // namespace pmx
