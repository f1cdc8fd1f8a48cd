// gfx950 kernels of the batched Poseidon permutation and duplex-sponge driver, plus their launchers.
//
// Work decomposition: one lane owns one sponge state; a wavefront owns 64 contiguous states of the
// [n][t][4]u64 batch.  State I/O goes through LDS so that every global access is a full-width
// 16-B-per-lane contiguous stream, whatever t is.  Round constants and the MDS matrix are wave-uniform.
//
// Two engines implement the same interface:
//   RegEngine<T, ALPHA>  state in VGPRs, loops over elements unrolled, ARK+MDS staged in LDS.   (t = 3)
//   LdsEngine<ALPHA>     any width at run time: state kept in LDS as [element][lane] (conflict-free
//                        16-B accesses), element loops rolled, constants through the scalar cache.
//
// Reference semantics implemented here (file:line in /root/reference):
//   permute        src/poseidon/mod.rs:95-118   (apply_ark :76-80, apply_s_box :63-74, apply_mds :82-93)
//   absorb         src/poseidon/mod.rs:232-254 + absorb_internal :121-150
//   squeeze        src/poseidon/mod.rs:321-341 + squeeze_internal :153-182
#include <hip/hip_runtime.h>

#include "../../include/poseidon_mi355x.h"
#include "pmx_field.hpp"
#include "pmx_internal.hpp"
#include "pmx_launch.hpp"

namespace pmx {

extern __shared__ uint4 pmx_lds[];  // dynamic LDS, 16-byte granules

__device__ __forceinline__ void init_field(FieldRt &f, const DevConfig &c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f.p[i] = c.p[i];
    f.inv32 = c.inv32;
}

// The wave-uniform scalars of a config, copied out of the kernel-argument block by value so that they
// stay in SGPRs (holding a reference to the by-value kernel argument sends it to scratch).
struct Rounds {
    const uint32_t *consts;
    uint32_t rate, capacity, half_full, partial_rounds, total_rounds;
    uint64_t alpha;
    __device__ __forceinline__ explicit Rounds(const DevConfig &d)
        : consts(d.consts), rate(d.rate), capacity(d.capacity), half_full(d.half_full),
          partial_rounds(d.partial_rounds), total_rounds(d.total_rounds),
          alpha(((uint64_t)d.alpha_hi << 32) | d.alpha_lo) {}
};

__device__ __forceinline__ bool is_full_round(uint32_t r, const Rounds &c) {
    return r < c.half_full || r >= c.half_full + c.partial_rounds;
}

// ------------------------------------------------------------------------------------------------
// RegEngine: t known at compile time, state in registers.
// LDS: [constants: n_const_words u32][staging: kThreads * T * 2 uint4]
// ------------------------------------------------------------------------------------------------
template <int T, int ALPHA>
struct RegEngine {
    static constexpr int kThreads = 256;
    static constexpr int kChunks = 2 * T;  // 16-byte chunks per state

    Fe s[T];
    Rounds c;
    FieldRt f;
    const uint32_t *lconst;  // LDS copy of ark|mds
    uint4 *stage;            // LDS staging for coalesced state I/O
    Fe one;

    static size_t lds_bytes(const DevConfig &c, uint32_t /*t*/) {
        return (size_t)((c.n_const_words + 3) / 4) * 16 + (size_t)kThreads * kChunks * 16;
    }

    __device__ __forceinline__ explicit RegEngine(const DevConfig &cfg) : c(cfg) {
        init_field(f, cfg);
        const uint32_t const_chunks = (cfg.n_const_words + 3) / 4;
        const uint4 *g = reinterpret_cast<const uint4 *>(cfg.consts);
        for (uint32_t q = threadIdx.x; q < const_chunks; q += kThreads) pmx_lds[q] = g[q];
        lconst = reinterpret_cast<const uint32_t *>(pmx_lds);
        stage = pmx_lds + const_chunks;
#pragma unroll
        for (int i = 0; i < 8; ++i) one.l[i] = cfg.one[i];
        __syncthreads();
    }

    __device__ __forceinline__ uint32_t width() const { return T; }

    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < T; ++i) s[i] = fe_zero();
    }

    // The block's kThreads states are contiguous in global memory: copy them as one linear stream.
    __device__ __forceinline__ void load_states(const uint64_t *g_states, size_t n) {
        const size_t first = (size_t)blockIdx.x * kThreads;
        const size_t valid = n > first ? (n - first < (size_t)kThreads ? n - first : (size_t)kThreads) : 0;
        const uint4 *g = reinterpret_cast<const uint4 *>(g_states) + first * kChunks;
        const uint32_t n_chunks = (uint32_t)valid * kChunks;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kChunks; ++k) {
            const uint32_t q = threadIdx.x + k * kThreads;
            if (q < n_chunks) stage[q] = g[q];
        }
        __syncthreads();
        if (threadIdx.x < valid) {
#pragma unroll
            for (int i = 0; i < T; ++i)
                s[i] = fe_from_u4(stage[threadIdx.x * kChunks + 2 * i], stage[threadIdx.x * kChunks + 2 * i + 1]);
        } else {
            zero();
        }
    }

    __device__ __forceinline__ void store_states(uint64_t *g_states, size_t n) {
        const size_t first = (size_t)blockIdx.x * kThreads;
        const size_t valid = n > first ? (n - first < (size_t)kThreads ? n - first : (size_t)kThreads) : 0;
        uint4 *g = reinterpret_cast<uint4 *>(g_states) + first * kChunks;
        const uint32_t n_chunks = (uint32_t)valid * kChunks;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < T; ++i) {
            stage[threadIdx.x * kChunks + 2 * i] = fe_lo(s[i]);
            stage[threadIdx.x * kChunks + 2 * i + 1] = fe_hi(s[i]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kChunks; ++k) {
            const uint32_t q = threadIdx.x + k * kThreads;
            if (q < n_chunks) g[q] = stage[q];
        }
    }

    // element i of this lane's state; i may differ between lanes
    __device__ __forceinline__ Fe get(uint32_t i) const {
        Fe r = s[0];
#pragma unroll
        for (int k = 1; k < T; ++k) {
#pragma unroll
            for (int w = 0; w < 8; ++w) r.l[w] = (i == (uint32_t)k) ? s[k].l[w] : r.l[w];
        }
        return r;
    }
    __device__ __forceinline__ void set(uint32_t i, const Fe &v) {
#pragma unroll
        for (int k = 0; k < T; ++k) {
#pragma unroll
            for (int w = 0; w < 8; ++w) s[k].l[w] = (i == (uint32_t)k) ? v.l[w] : s[k].l[w];
        }
    }

    __device__ __forceinline__ void permute() {
        const uint32_t *ark = lconst;
        const uint32_t *mds = lconst + (size_t)c.total_rounds * T * 8;
        const uint64_t alpha = c.alpha;
        for (uint32_t r = 0; r < c.total_rounds; ++r) {
#pragma unroll
            for (int i = 0; i < T; ++i) s[i] = fe_add(s[i], fe_load(ark + ((size_t)r * T + i) * 8), f);
            s[0] = fe_sbox<ALPHA>(s[0], alpha, one, f);
            if (is_full_round(r, c)) {
#pragma unroll
                for (int i = 1; i < T; ++i) s[i] = fe_sbox<ALPHA>(s[i], alpha, one, f);
            }
            Fe ns[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                Fe acc = fe_mul(s[0], fe_load(mds + ((size_t)i * T) * 8), f);
#pragma unroll
                for (int j = 1; j < T; ++j) acc = fe_add(acc, fe_mul(s[j], fe_load(mds + ((size_t)i * T + j) * 8), f), f);
                ns[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < T; ++i) s[i] = ns[i];
        }
    }
};

// ------------------------------------------------------------------------------------------------
// LdsEngine: width is a run-time value.  Each wave keeps its 64 states in LDS as
// cur[(element*2 + half) * 64 + lane] (16-byte granules), with a second buffer for the MDS output.
// LDS: per wave 2 buffers x t x 2 x 64 uint4  (t KiB each).
// ------------------------------------------------------------------------------------------------
template <int ALPHA>
struct LdsEngine {
    static constexpr int kThreads = 128;

    Rounds c;
    FieldRt f;
    uint32_t t;
    uint32_t lane;
    uint4 *cur;
    uint4 *nxt;
    Fe one;

    static size_t lds_bytes(const DevConfig & /*c*/, uint32_t t) {
        return (size_t)(kThreads / 64) * 2 * t * 2 * 64 * 16;
    }

    __device__ __forceinline__ explicit LdsEngine(const DevConfig &cfg) : c(cfg) {
        init_field(f, cfg);
        t = c.rate + c.capacity;
        lane = threadIdx.x & 63;
        const uint32_t wave = threadIdx.x >> 6;
        cur = pmx_lds + (size_t)wave * 2 * (t * 2 * 64);
        nxt = cur + t * 2 * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) one.l[i] = cfg.one[i];
    }

    __device__ __forceinline__ uint32_t width() const { return t; }

    __device__ __forceinline__ Fe get(uint32_t i) const { return fe_from_u4(cur[(i * 2) * 64 + lane], cur[(i * 2 + 1) * 64 + lane]); }
    __device__ __forceinline__ void set(uint32_t i, const Fe &v) {
        cur[(i * 2) * 64 + lane] = fe_lo(v);
        cur[(i * 2 + 1) * 64 + lane] = fe_hi(v);
    }

    __device__ __forceinline__ void zero() {
        for (uint32_t i = 0; i < t; ++i) set(i, fe_zero());
    }

    // wave-level: 64 contiguous states = 64*2t contiguous 16-B chunks in global memory
    __device__ __forceinline__ void load_states(const uint64_t *g_states, size_t n) {
        const size_t first = ((size_t)blockIdx.x * kThreads + (threadIdx.x & ~63u));
        const size_t valid = n > first ? (n - first < 64 ? n - first : 64) : 0;
        const uint32_t chunks = 2 * t;
        const uint4 *g = reinterpret_cast<const uint4 *>(g_states) + first * chunks;
        const uint32_t n_chunks = (uint32_t)valid * chunks;
        __syncthreads();
        for (uint32_t k = 0; k < chunks; ++k) {
            const uint32_t q = lane + k * 64;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < n_chunks) v = g[q];
            cur[(q % chunks) * 64 + q / chunks] = v;  // q/chunks = state within wave, q%chunks = chunk of state
        }
        __syncthreads();
    }

    __device__ __forceinline__ void store_states(uint64_t *g_states, size_t n) {
        const size_t first = ((size_t)blockIdx.x * kThreads + (threadIdx.x & ~63u));
        const size_t valid = n > first ? (n - first < 64 ? n - first : 64) : 0;
        const uint32_t chunks = 2 * t;
        uint4 *g = reinterpret_cast<uint4 *>(g_states) + first * chunks;
        const uint32_t n_chunks = (uint32_t)valid * chunks;
        __syncthreads();
        for (uint32_t k = 0; k < chunks; ++k) {
            const uint32_t q = lane + k * 64;
            if (q < n_chunks) g[q] = cur[(q % chunks) * 64 + q / chunks];
        }
        __syncthreads();
    }

    __device__ __forceinline__ void permute() {
        const uint32_t *ark = c.consts;
        const uint32_t *mds = c.consts + (size_t)c.total_rounds * t * 8;
        const uint64_t alpha = c.alpha;
        uint4 *const home = cur;
        for (uint32_t r = 0; r < c.total_rounds; ++r) {
            const uint32_t n_sbox = is_full_round(r, c) ? t : 1;
            for (uint32_t i = 0; i < t; ++i) {
                Fe x = fe_add(get(i), fe_load(ark + ((size_t)r * t + i) * 8), f);
                if (i < n_sbox) x = fe_sbox<ALPHA>(x, alpha, one, f);
                set(i, x);
            }
            for (uint32_t i = 0; i < t; ++i) {
                Fe acc = fe_mul(get(0), fe_load(mds + ((size_t)i * t) * 8), f);
                for (uint32_t j = 1; j < t; ++j)
                    acc = fe_add(acc, fe_mul(get(j), fe_load(mds + ((size_t)i * t + j) * 8), f), f);
                nxt[(i * 2) * 64 + lane] = fe_lo(acc);
                nxt[(i * 2 + 1) * 64 + lane] = fe_hi(acc);
            }
            uint4 *tmp = cur;
            cur = nxt;
            nxt = tmp;
        }
        // Lanes may permute a different number of times (per-sponge modes) and other lanes read this
        // lane's slots in store_states: always leave the state in the buffer it started in.
        if (cur != home) {
            for (uint32_t q = 0; q < 2 * t; ++q) nxt[q * 64 + lane] = cur[q * 64 + lane];
            tmp_swap();
        }
    }

    __device__ __forceinline__ void tmp_swap() {
        uint4 *tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
};

// ------------------------------------------------------------------------------------------------
// Kernels (identical for both engines)
// ------------------------------------------------------------------------------------------------
template <class Engine>
__global__ void __launch_bounds__(Engine::kThreads) permute_kernel(const DevConfig c, uint64_t *states, size_t n) {
    Engine e(c);
    e.load_states(states, n);
    e.permute();
    e.store_states(states, n);
}

// absorb `in_len` elements into this lane's sponge; idx is the next absorb index, or `rate` to force the
// permutation a Squeezing sponge performs first (mod.rs:247-252).  Returns the final next_absorb_index.
template <class Engine>
__device__ __forceinline__ uint32_t absorb_elements(Engine &e, const uint64_t *row, size_t in_len, uint32_t idx,
                                                    bool active) {
    const Rounds &c = e.c;
    for (size_t k = 0; k < in_len; ++k) {
        // rate filled and more input remains -> permute (mod.rs:137-148; also the :241-244 case)
        const bool need = active && idx == c.rate;
        if (__builtin_amdgcn_ballot_w64(need)) {
            if (need) {
                e.permute();
                idx = 0;
            }
        }
        if (active) {
            const Fe x = fe_load(reinterpret_cast<const uint32_t *>(row + 4 * k));
            const uint32_t pos = c.capacity + idx;
            e.set(pos, fe_add(e.get(pos), x, e.f));  // state[capacity + idx] += element (mod.rs:128,143)
            idx += 1;
        }
    }
    return idx;
}

// squeeze_internal (mod.rs:153-182) preceded by the mode handling of mod.rs:323-338.
// `need` = permute before the first copy.  Returns the final next_squeeze_index.
template <class Engine>
__device__ __forceinline__ uint32_t squeeze_elements(Engine &e, uint64_t *row, size_t out_len, uint32_t idx, bool need,
                                                     bool active) {
    const Rounds &c = e.c;
    size_t rem = out_len;
    size_t pos = 0;
    bool done = !active;
    while (__builtin_amdgcn_ballot_w64(!done)) {
        const bool do_perm = !done && need;
        if (__builtin_amdgcn_ballot_w64(do_perm)) {
            if (do_perm) e.permute();
        }
        if (!done) {
            const bool last = idx + rem <= c.rate;
            const uint32_t take = last ? (uint32_t)rem : c.rate - idx;
            for (uint32_t k = 0; k < take; ++k)
                fe_store(reinterpret_cast<uint32_t *>(row + 4 * (pos + k)), e.get(c.capacity + idx + k));
            if (last) {
                idx += take;
                done = true;
            } else {
                need = rem != c.rate;  // mod.rs:175, tested before the slice is advanced
                rem -= take;
                pos += take;
                idx = 0;
            }
        }
    }
    return idx;
}

template <class Engine>
__global__ void __launch_bounds__(Engine::kThreads)
    hash_kernel(const DevConfig c, const uint64_t *in, size_t in_len, uint64_t *out, size_t out_len, size_t n) {
    Engine e(c);
    const size_t gid = (size_t)blockIdx.x * Engine::kThreads + threadIdx.x;
    const bool active = gid < n;
    e.zero();                                                  // CryptographicSponge::new, mod.rs:219-230
    const uint64_t *row_in = in + (active ? gid : 0) * in_len * 4;
    uint64_t *row_out = out + (active ? gid : 0) * out_len * 4;
    (void)absorb_elements(e, row_in, in_len, 0, active);
    // the sponge is Absorbing here, so the squeeze always permutes first (mod.rs:324-328)
    (void)squeeze_elements(e, row_out, out_len, 0, true, active);
}

template <class Engine>
__global__ void __launch_bounds__(Engine::kThreads)
    absorb_kernel(const DevConfig c, uint64_t *states, uint32_t *mode_tag, uint32_t *mode_index, const uint64_t *in,
                  size_t in_len, size_t n) {
    Engine e(c);
    const size_t gid = (size_t)blockIdx.x * Engine::kThreads + threadIdx.x;
    const bool active = gid < n;
    e.load_states(states, n);
    uint32_t idx = 0;
    if (active) idx = (mode_tag[gid] == PMX_MODE_ABSORBING) ? mode_index[gid] : e.c.rate;
    idx = absorb_elements(e, in + (active ? gid : 0) * in_len * 4, in_len, idx, active);
    e.store_states(states, n);
    if (active) {
        mode_tag[gid] = PMX_MODE_ABSORBING;                    // mod.rs:130-132
        mode_index[gid] = idx;
    }
}

template <class Engine>
__global__ void __launch_bounds__(Engine::kThreads)
    squeeze_kernel(const DevConfig c, uint64_t *states, uint32_t *mode_tag, uint32_t *mode_index, uint64_t *out,
                   size_t out_len, size_t n) {
    Engine e(c);
    const size_t gid = (size_t)blockIdx.x * Engine::kThreads + threadIdx.x;
    const bool active = gid < n;
    e.load_states(states, n);
    uint32_t idx = 0;
    bool need = true;                                          // Absorbing -> permute, start at 0 (mod.rs:324-328)
    if (active && mode_tag[gid] == PMX_MODE_SQUEEZING) {       // mod.rs:330-336
        idx = mode_index[gid];
        need = idx == e.c.rate;
        if (need) idx = 0;
    }
    idx = squeeze_elements(e, out + (active ? gid : 0) * out_len * 4, out_len, idx, need, active);
    e.store_states(states, n);
    if (active) {
        mode_tag[gid] = PMX_MODE_SQUEEZING;                    // mod.rs:162-164
        mode_index[gid] = idx;
    }
}

// ------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------
template <class Engine>
struct Launch {
    static int grid(size_t n) { return (int)((n + Engine::kThreads - 1) / Engine::kThreads); }

    static hipError_t permute(const DevConfig &c, uint32_t t, uint64_t *states, size_t n, hipStream_t st) {
        hipLaunchKernelGGL(permute_kernel<Engine>, dim3(grid(n)), dim3(Engine::kThreads), Engine::lds_bytes(c, t), st, c,
                           states, n);
        return hipGetLastError();
    }
    static hipError_t hash(const DevConfig &c, uint32_t t, const uint64_t *in, size_t in_len, uint64_t *out,
                           size_t out_len, size_t n, hipStream_t st) {
        hipLaunchKernelGGL(hash_kernel<Engine>, dim3(grid(n)), dim3(Engine::kThreads), Engine::lds_bytes(c, t), st, c, in,
                           in_len, out, out_len, n);
        return hipGetLastError();
    }
    static hipError_t absorb(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                             const uint64_t *in, size_t in_len, size_t n, hipStream_t st) {
        hipLaunchKernelGGL(absorb_kernel<Engine>, dim3(grid(n)), dim3(Engine::kThreads), Engine::lds_bytes(c, t), st, c,
                           states, tag, index, in, in_len, n);
        return hipGetLastError();
    }
    static hipError_t squeeze(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                              uint64_t *out, size_t out_len, size_t n, hipStream_t st) {
        hipLaunchKernelGGL(squeeze_kernel<Engine>, dim3(grid(n)), dim3(Engine::kThreads), Engine::lds_bytes(c, t), st, c,
                           states, tag, index, out, out_len, n);
        return hipGetLastError();
    }
};

// Engine choice: width 3 runs from registers; every other width uses the LDS-resident engine.
// alpha 5 and 17 have dedicated addition chains, other exponents share the generic S-box.
#define PMX_DISPATCH(CALL)                                                                  \
    do {                                                                                    \
        const uint64_t alpha = ((uint64_t)c.alpha_hi << 32) | c.alpha_lo;                   \
        if (t == 3) {                                                                       \
            if (alpha == 5) return Launch<RegEngine<3, 5>>::CALL;                           \
            if (alpha == 17) return Launch<RegEngine<3, 17>>::CALL;                         \
            return Launch<RegEngine<3, 0>>::CALL;                                           \
        }                                                                                   \
        if (alpha == 5) return Launch<LdsEngine<5>>::CALL;                                  \
        if (alpha == 17) return Launch<LdsEngine<17>>::CALL;                                \
        return Launch<LdsEngine<0>>::CALL;                                                  \
    } while (0)

hipError_t launch_permute(const DevConfig &c, uint32_t t, uint64_t *states, size_t n, hipStream_t st) {
    PMX_DISPATCH(permute(c, t, states, n, st));
}
hipError_t launch_hash(const DevConfig &c, uint32_t t, const uint64_t *in, size_t in_len, uint64_t *out, size_t out_len,
                       size_t n, hipStream_t st) {
    PMX_DISPATCH(hash(c, t, in, in_len, out, out_len, n, st));
}
hipError_t launch_absorb(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                         const uint64_t *in, size_t in_len, size_t n, hipStream_t st) {
    PMX_DISPATCH(absorb(c, t, states, tag, index, in, in_len, n, st));
}
hipError_t launch_squeeze(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                          uint64_t *out, size_t out_len, size_t n, hipStream_t st) {
    PMX_DISPATCH(squeeze(c, t, states, tag, index, out, out_len, n, st));
}

}  // namespace pmx
