// Launcher entry points of pmx_device.hip (host-callable; they only enqueue on `st`).
#pragma once
#include <hip/hip_runtime.h>

#include "pmx_internal.hpp"

namespace pmx {

hipError_t launch_permute(const DevConfig &c, uint32_t t, uint64_t *states, size_t n, hipStream_t st);
hipError_t launch_hash(const DevConfig &c, uint32_t t, const uint64_t *in, size_t in_len, uint64_t *out, size_t out_len,
                       size_t n, hipStream_t st);
// out[i] = 2-to-1 compression of in[2i], in[2i+1]   (rate >= 2)
hipError_t launch_compress(const DevConfig &c, uint32_t t, const uint64_t *in, uint64_t *out, size_t n, hipStream_t st);
hipError_t launch_absorb(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                         const uint64_t *in, size_t in_len, size_t n, hipStream_t st);
hipError_t launch_squeeze(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                          uint64_t *out, size_t out_len, size_t n, hipStream_t st);

}  // namespace pmx
