// gfx950 kernels of the batched Poseidon permutation and duplex-sponge driver, plus their launchers.
//
// Work decomposition: one lane owns one sponge state; a wavefront owns 64 contiguous states of the
// [n][t][4]u64 batch.  State I/O goes through LDS so that every global access is a full-width
// 16-B-per-lane contiguous stream, whatever t is.  Round constants and the MDS matrix are wave-uniform.
// Arithmetic: pmx_field.hpp (unsaturated 9 x 29-bit Montgomery form); round schedule: pmx_permute.hpp.
//
// Engines (same interface: load_states / store_states / get / set / zero / permute), one per regime:
//   QuadEngine<ALPHA>          t = 3, launches of <= 32768 units (latency-bound: narrow tree levels, a handful of sponges):
//                              one state per quad of lanes, sparse rounds three multiplications deep.
//   HybridEngine<T, ALPHA>     t = 3..9 on the optimised schedule: state in VGPRs, the element loop of the full rounds' S-boxes rolled
//                              through one LDS scratch array per wave, EVERY product by a constant a layer on the matrix cores
//                              (pmx_mfma.hpp: the dense layers, and the partial rounds as windows closed by one layer each).
//   LdsEngine<ALPHA>           any width at run time (t = 2, 10..16, no partial section, a zero in the schedule's algebra): the
//                              reference's dense schedule, state kept in LDS as [element][limb][lane], element loops rolled.
// (Rounds 1-5 also had a register engine for t = 3 and VALU-row forms of the hybrid engine; they went in round 6, when every modulus
// got its int8 tables.)
//
// Reference semantics implemented here (file:line in /root/reference):
//   permute        src/poseidon/mod.rs:95-118   (apply_ark :76-80, apply_s_box :63-74, apply_mds :82-93)
//   absorb         src/poseidon/mod.rs:232-254 + absorb_internal :121-150
//   squeeze        src/poseidon/mod.rs:321-341 + squeeze_internal :153-182
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "../../include/poseidon_mi355x.h"
#include "pmx_field.hpp"
#include "pmx_internal.hpp"
#include "pmx_launch.hpp"
#include "pmx_permute.hpp"
#include "pmx_sponge_plan.hpp"

// The file is compiled five times (Makefile, in parallel): PMX_TU = 0 holds the t = 3 register engine, the run-time-width
// engine, the cooperative kernel and the public launchers; PMX_TU = 1 / 3 hold the hybrid engines for alpha = 5 (widths up to 6 /
// from 7), PMX_TU = 2 / 4 the same for the generic S-box (each width x {VALU rows, matrix cores} x 7 kernels - by far the longest compile).
#ifndef PMX_TU
#define PMX_TU 0
#endif

namespace pmx {

extern __shared__ uint4 pmx_lds[];  // dynamic LDS, 16-byte granules

// ------------------------------------------------------------------------------------------------
// HybridEngine: t = 3..9 on the optimised schedule, every product by a constant on the matrix cores (pmx_mfma.hpp): the dense layers
// and, as windows of t S-boxes per layer, the linear part of the partial rounds.  State in registers (9 t VGPRs); every wave has one
// LDS scratch array [element][limb][lane] (2.25 (t - 1) KiB) that gives the rolled S-box loop of the full rounds and the layers' output
// rows their dynamic indexing (pmx_permute.hpp: permute_hybrid) and doubles as the staging area of the coalesced ABI store.
// Wave-uniform kernels only (permute, hash, compress, and the passes of the absorb / squeeze driver, which ARE permutation launches:
// inside a per-lane loop not every lane is active, and the lane exchange of the rows needs both lanes of a pair).
// (Rounds 1-4 also had this engine with VALU rows and sparse partial rounds, and a register engine for t = 3; since round 6 every config
// that has the optimised schedule has the window tables - any modulus - and the rest runs on the run-time-width engine.)
// ------------------------------------------------------------------------------------------------
// Register bounds (their byte strings and sums want registers), measured per width:
#ifndef PMX_MFMA_4WAVE_MAX_T
#define PMX_MFMA_4WAVE_MAX_T 3   // (t = 3: 120 VGPRs - four waves per SIMD without a spill)
#endif
#ifndef PMX_MFMA_3WAVE_MAX_T
#define PMX_MFMA_3WAVE_MAX_T 5
#endif
// Four waves per workgroup (one per SIMD), two workgroups per CU at t = 9 (8 x 18 KiB of scratch).  Since round 6 the rows' tables are
// streamed by every wave for itself (pmx_mfma.hpp: no LDS tile, no workgroup barrier inside the permutation); the workgroup only shares
// the barriers of the state store staging.  (One wave per workgroup was measured: +-0, profiles/r06/b_ab_*.)
constexpr int kMfmaWaves = 4;
template <int T, int ALPHA>
struct HybridEngine {
    static_assert(T >= PMX_MFMA_MIN_T && T <= PMX_MFMA_MAX_T && mfma_window_for(T) > 0, "the window engines cover t = 3 .. 9");
    static constexpr int kWaves = kMfmaWaves;
    static constexpr int kThreads = 64 * kWaves;
    // waves per SIMD the register allocation must allow (4: <= 128 VGPRs, 3: <= 168, 2: <= 256)
    static constexpr int kMinWaves = T <= PMX_MFMA_4WAVE_MAX_T ? 4 : T <= PMX_MFMA_3WAVE_MAX_T ? 3 : 2;
    static constexpr int kMinWavesDriver = 2;   // (the pass kernels carry the driver's walk around the permutation: held to two waves)
    // the rows exchange operands between the lanes of a pair (l, l + 32): permute() must be reached by every lane of the wave - never
    // from a per-lane loop (absorb_kernel / squeeze_kernel static_assert on this; the drivers run as passes, sponge_first_kernel)
    static constexpr bool kWaveUniformOnly = true;
    static constexpr int kChunks = 2 * T;
    // one wave's LDS region: scratch slots for elements 0..T-2 (2304 B each) or the ABI staging of its 64 states
    // (2048 T B), whichever is larger
    static constexpr size_t kScratchBytes = (size_t)(T - 1) * kN * 64 * 4, kStageBytes = (size_t)64 * kChunks * 16;
    static constexpr size_t kWaveBytes = kScratchBytes > kStageBytes ? kScratchBytes : kStageBytes;

    struct Scratch {
        uint32_t *base;   // + lane
        __device__ __forceinline__ Fe get(uint32_t i) const {
            Fe r;
#pragma unroll
            for (int w = 0; w < kN; ++w) r.l[w] = base[(i * kN + w) * 64];
            return r;
        }
        __device__ __forceinline__ void set(uint32_t i, const Fe &v) {
#pragma unroll
            for (int w = 0; w < kN; ++w) base[(i * kN + w) * 64] = v.l[w];
        }
    };

    Fe s[T];
    Rounds c;
    FieldRt f;
    Fe one;
    OptTables tb;
    Scratch sc;
    uint4 *region;    // this wave's LDS region
    uint32_t lane;

    static size_t lds_bytes(const DevConfig & /*d*/, uint32_t /*t*/) { return kWaves * kWaveBytes; }

    __device__ __forceinline__ HybridEngine(const DevConfig &d, const uint32_t *consts) : c(d.rounds), f(d.field), one(d.one) {
        f.io = consts + d.io_offset;
        tb.ark = consts + d.opt_offset;
        tb.mfma = consts + d.mfma_offset;
        tb.win = consts + d.win_offset;
        lane = threadIdx.x & 63;
        region = pmx_lds + (threadIdx.x >> 6) * (kWaveBytes / 16);
        sc.base = reinterpret_cast<uint32_t *>(region) + lane;
    }

    __device__ __forceinline__ void zero() {
        static_for<0, T>([&](auto i) { s[i] = fe_zero(); });
    }
    __device__ __forceinline__ Fe from_abi(const Abi &x) const { return fe_from_abi_scaled(x); }      // the optimised schedule carries the ABI residue as its internal form (pmx_field.hpp: fe_from_abi_scaled)
    __device__ __forceinline__ Abi to_abi(const Fe &x) const { return fe_to_abi_scaled(x, f); }

    // widths whose permute kernel otherwise keeps spills inside the rounds read their elements in a rolled loop (t = 6 at three waves per
    // SIMD: 76 -> 12 bytes of scratch; t = 9: 76 -> 0); t = 7, 8 have none and lose 1 %
    static constexpr bool kRolledLoad = T == 6 || T >= 9;
    // Every lane reads and writes its own 32 T contiguous bytes with 16-byte accesses.  Across the lanes of a wave these are
    // strided, but every cache line is used in full within the 2 T accesses of the lane that owns it, and a wide permutation
    // moves 64 T bytes in ~2 ms of arithmetic: nothing to gain from staging the wave's span through LDS, and without the
    // staging code (two barriers, the T-element gather) the register allocation of the permute kernel comes out like that of
    // the hash kernel (no spills inside the rounds).
    // the T elements at g into the registers; `adjust(i, element)` may replace an element on its way in (the absorb driver
    // adds its input there, on the ABI residues, before the bit-slicing - absorb_adjust below)
    template <class Adjust>
    __device__ __forceinline__ void load_elements(const uint4 *g, const Adjust &adjust) {
        if constexpr (kRolledLoad) {
            zero();
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (uint32_t i = 0; i < (uint32_t)T; ++i) set(i, from_abi(adjust(i, abi_from_u4(g[2 * i], g[2 * i + 1]))));
        } else {
            static_for<0, T>([&](auto i) { s[i] = from_abi(adjust((uint32_t)i, abi_from_u4(g[2 * i], g[2 * i + 1]))); });
        }
    }
    struct NoAdjust {
        __device__ __forceinline__ Abi operator()(uint32_t, const Abi &a) const { return a; }
    };
    // state[pos + j] += row[j] for j < count (mod.rs:128,143), as the state comes in: both fully reduced residues, one
    // 256-bit add and one conditional subtraction (abi_add_mod); count is per lane, 0 for a lane that absorbs nothing here
    struct AbsorbAdjust {
        const uint32_t *row, *p32;
        uint32_t pos, count;
        __device__ __forceinline__ Abi operator()(uint32_t i, const Abi &a) const {
            const uint32_t j = i - pos;
            if (j < count) return abi_add_mod(a, abi_load(row + 8 * j), p32);
            return a;
        }
    };
    __device__ __forceinline__ void load_states(const uint64_t *g_states, size_t n) {
        const size_t gid = (size_t)blockIdx.x * kThreads + threadIdx.x;
        load_elements(reinterpret_cast<const uint4 *>(g_states) + (gid < n ? gid : 0) * kChunks, NoAdjust{});
    }
    __device__ __forceinline__ void load_states(const uint64_t *g_states, size_t n, const AbsorbAdjust &add) {
        const size_t gid = (size_t)blockIdx.x * kThreads + threadIdx.x;
        load_elements(reinterpret_cast<const uint4 *>(g_states) + (gid < n ? gid : 0) * kChunks, add);
    }

    // the store goes through the wave's LDS region so that every 16-byte write instruction covers 1 KiB of contiguous memory:
    // written lane by lane (stride 32 T bytes) the partial lines are not all merged before they leave the L2 - 1.57 x the
    // bytes at t = 9 (WRITE_SIZE, profiles/r03)
    __device__ __forceinline__ void store_states(uint64_t *g_states, size_t n) {
        const size_t first = (size_t)blockIdx.x * kThreads + (threadIdx.x & ~63u);
        const size_t valid = n > first ? (n - first < (size_t)64 ? n - first : (size_t)64) : 0;
        uint4 *g = reinterpret_cast<uint4 *>(g_states) + first * kChunks;
        const uint32_t n_chunks = (uint32_t)valid * kChunks;
        __syncthreads();
        static_for<0, T>([&](auto i) {
            const Abi a = to_abi(s[i]);
            region[lane * kChunks + 2 * i] = abi_lo(a);
            region[lane * kChunks + 2 * i + 1] = abi_hi(a);
            PMX_SCHED_FENCE();   // one conversion at a time: interleaved, the T exact reductions are the widest point of the kernel
        });
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kChunks; ++k) {
            const uint32_t q = lane + k * 64;
            if (q < n_chunks) g[q] = region[q];
        }
        __syncthreads();
    }

    // One state per lane at a per-lane address (permute_listed_kernel: the sponges of a pass, gathered through an index
    // list).  Every lane reads and writes its own 32 T contiguous bytes with 16-byte accesses: each line is used in full by
    // the lane that owns it.
    __device__ __forceinline__ void load_state_at(const uint64_t *mine, const AbsorbAdjust &add) {
        load_elements(reinterpret_cast<const uint4 *>(mine), add);
    }
    // store_states for the sponges of this wave whose `keep` bit is set (sponge_first_kernel: the permutation ran for the whole
    // workgroup, only the sponges that needed it take its result).  Staged through the wave's region like store_states: a
    // 16-byte write instruction covers whole states, so the predicate travels as the wave's ballot.
    // rate elements copied out of the state that is being stored (squeeze, mod.rs:159-170): out[j] = state[pos + j], j < count,
    // read from the wave's LDS staging, where every element already sits fully reduced in ABI form; count is per lane
    struct CopyOut {
        uint32_t *row;
        uint32_t pos, count;
    };
    __device__ __forceinline__ void copy_out_staged(const CopyOut &out) const {
        for (uint32_t j = 0; j < out.count; ++j) {
            uint4 *dst = reinterpret_cast<uint4 *>(out.row + 8 * j);
            dst[0] = region[lane * kChunks + 2 * (out.pos + j)];
            dst[1] = region[lane * kChunks + 2 * (out.pos + j) + 1];
        }
    }
    __device__ __forceinline__ void store_states_where(uint64_t *g_states, size_t n, uint64_t keep_mask, const CopyOut &out) {
        const size_t first = (size_t)blockIdx.x * kThreads + (threadIdx.x & ~63u);
        const size_t valid = n > first ? (n - first < (size_t)64 ? n - first : (size_t)64) : 0;
        uint4 *g = reinterpret_cast<uint4 *>(g_states) + first * kChunks;
        const uint32_t n_chunks = (uint32_t)valid * kChunks;
        __syncthreads();
        static_for<0, T>([&](auto i) {
            const Abi a = to_abi(s[i]);
            region[lane * kChunks + 2 * i] = abi_lo(a);
            region[lane * kChunks + 2 * i + 1] = abi_hi(a);
            PMX_SCHED_FENCE();
        });
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kChunks; ++k) {
            const uint32_t q = lane + k * 64;
            if (q < n_chunks && ((keep_mask >> (q / kChunks)) & 1)) g[q] = region[q];
        }
        copy_out_staged(out);
        __syncthreads();
    }

    // `together`: the wave's active lanes hold CONSECUTIVE states starting at wave_base (a prefix of the lanes) - then the
    // wave's span goes out as whole kilobytes like store_states does; otherwise every lane writes its own state.  Either
    // way the T exact reductions happen once, into the wave's LDS region.
    __device__ __forceinline__ void store_state_at(uint64_t *mine, bool keep, bool together, uint64_t *wave_base, uint32_t valid, const CopyOut &out) {
        __syncthreads();
        static_for<0, T>([&](auto i) {
            const Abi a = to_abi(s[i]);
            region[lane * kChunks + 2 * i] = abi_lo(a);
            region[lane * kChunks + 2 * i + 1] = abi_hi(a);
            PMX_SCHED_FENCE();   // one conversion at a time (see store_states)
        });
        __syncthreads();
        if (together) {          // wave-uniform
            uint4 *g = reinterpret_cast<uint4 *>(wave_base);
            const uint32_t n_chunks = valid * kChunks;
#pragma unroll
            for (int k = 0; k < kChunks; ++k) {
                const uint32_t q = lane + k * 64;
                if (q < n_chunks) g[q] = region[q];
            }
        } else if (keep) {
            uint4 *g = reinterpret_cast<uint4 *>(mine);
#pragma clang loop unroll(disable)
            for (int k = 0; k < kChunks; ++k) g[k] = region[lane * kChunks + k];
        }
        copy_out_staged(out);
        __syncthreads();
    }

    static void describe(EngineInfo &o) {
        std::snprintf(o.engine, sizeof o.engine, "HybridEngine<%d,%d,mfma,windows of %d>", T, ALPHA, mfma_window_for(T));
        o.threads = kThreads;
        o.optimised = 1;
        o.row_tables = mfma_hist_tab(T) ? 1 : 2;   // how the history terms of the S-box inputs are formed: 1 shifted table on the VALU (t = 3), 2 rows on the matrix cores
        o.lane_tables = 0;
        o.mfma_dense = 1;
        o.partial_window = mfma_window_for(T);
    }

    __device__ __forceinline__ Fe get(uint32_t i) const {
        Fe r = s[0];
        static_for<1, T>([&](auto k) {
#pragma unroll
            for (int w = 0; w < kN; ++w) r.l[w] = (i == (uint32_t)k) ? s[k].l[w] : r.l[w];
        });
        return r;
    }
    __device__ __forceinline__ void set(uint32_t i, const Fe &v) {
        static_for<0, T>([&](auto k) {
#pragma unroll
            for (int w = 0; w < kN; ++w) s[k].l[w] = (i == (uint32_t)k) ? v.l[w] : s[k].l[w];
        });
    }

    // lane0_zero: the caller knows lane 0 of the state is zero (pmx_permute.hpp: the window engines skip its round-0 S-box)
    __device__ __forceinline__ void permute(uint32_t want_lo = 0, uint32_t want_hi = T, bool lane0_zero = false) {
        permute_hybrid<T, ALPHA, Scratch, mfma_window_for(T)>(s, sc, tb, c, one, f, want_lo, want_hi, lane0_zero);
    }
};

// ------------------------------------------------------------------------------------------------
// LdsEngine: width is a run-time value.  Each wave keeps its 64 states in LDS as
// cur[(element * 9 + limb) * 64 + lane] u32, with a second buffer for the MDS output.
// LDS per wave: 2 buffers x t x 9 x 64 x 4 B  (2.25 t KiB each); the second buffer doubles as the
// staging area of the coalesced ABI load/store (t x 32 B x 64 fits one buffer).
// ------------------------------------------------------------------------------------------------
template <int ALPHA>
struct LdsEngine {
    static constexpr int kThreads = 128;
    static constexpr int kMinWaves = 1, kMinWavesDriver = 1;
    static constexpr bool kWaveUniformOnly = false;

    Rounds c;
    FieldRt f;
    Fe one;
    const uint32_t *ark;
    const uint32_t *mds;
    uint32_t t;
    uint32_t lane;
    uint32_t *cur;
    uint32_t *nxt;

    static size_t lds_bytes(const DevConfig & /*d*/, uint32_t t) { return (size_t)(kThreads / 64) * 2 * t * kN * 64 * 4; }

    __device__ __forceinline__ LdsEngine(const DevConfig &d, const uint32_t *consts) : c(d.rounds), f(d.field), one(d.one) {
        f.io = consts + d.io_offset;
        ark = consts;
        mds = consts + d.mds_offset;
        t = c.rate + c.capacity;
        lane = threadIdx.x & 63;
        const uint32_t wave = threadIdx.x >> 6;
        cur = reinterpret_cast<uint32_t *>(pmx_lds) + (size_t)wave * 2 * (t * kN * 64);
        nxt = cur + t * kN * 64;
    }

    __device__ __forceinline__ Fe get(uint32_t i) const {
        Fe r;
#pragma unroll
        for (int w = 0; w < kN; ++w) r.l[w] = cur[(i * kN + w) * 64 + lane];
        return r;
    }
    __device__ __forceinline__ void set(uint32_t i, const Fe &v) {
#pragma unroll
        for (int w = 0; w < kN; ++w) cur[(i * kN + w) * 64 + lane] = v.l[w];
    }
    __device__ __forceinline__ void set_next(uint32_t i, const Fe &v) {
#pragma unroll
        for (int w = 0; w < kN; ++w) nxt[(i * kN + w) * 64 + lane] = v.l[w];
    }
    __device__ __forceinline__ void swap() {
        uint32_t *tmp = cur;
        cur = nxt;
        nxt = tmp;
    }

    __device__ __forceinline__ void zero() {
        for (uint32_t i = 0; i < t; ++i) set(i, fe_zero());
    }
    __device__ __forceinline__ Fe from_abi(const Abi &x) const { return fe_from_abi(x, f); }            // dense schedule: exact conversions
    __device__ __forceinline__ Abi to_abi(const Fe &x) const { return fe_to_abi(x, f); }

    // wave-level: 64 contiguous ABI states = 64*2t contiguous 16-B chunks in global memory, staged through
    // the `nxt` buffer (chunk q of the wave at uint4 index q), then converted element by element
    __device__ __forceinline__ void load_states(const uint64_t *g_states, size_t n) {
        const size_t first = ((size_t)blockIdx.x * kThreads + (threadIdx.x & ~63u));
        const size_t valid = n > first ? (n - first < 64 ? n - first : 64) : 0;
        const uint32_t chunks = 2 * t;
        const uint4 *g = reinterpret_cast<const uint4 *>(g_states) + first * chunks;
        const uint32_t n_chunks = (uint32_t)valid * chunks;
        uint4 *st = reinterpret_cast<uint4 *>(nxt);
        __syncthreads();
        for (uint32_t k = 0; k < chunks; ++k) {
            const uint32_t q = lane + k * 64;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < n_chunks) v = g[q];
            st[q] = v;
        }
        __syncthreads();
        for (uint32_t i = 0; i < t; ++i) set(i, from_abi(abi_from_u4(st[lane * chunks + 2 * i], st[lane * chunks + 2 * i + 1])));
        __syncthreads();
    }

    __device__ __forceinline__ void store_states(uint64_t *g_states, size_t n) {
        const size_t first = ((size_t)blockIdx.x * kThreads + (threadIdx.x & ~63u));
        const size_t valid = n > first ? (n - first < 64 ? n - first : 64) : 0;
        const uint32_t chunks = 2 * t;
        uint4 *g = reinterpret_cast<uint4 *>(g_states) + first * chunks;
        const uint32_t n_chunks = (uint32_t)valid * chunks;
        uint4 *st = reinterpret_cast<uint4 *>(nxt);
        __syncthreads();
        for (uint32_t i = 0; i < t; ++i) {
            const Abi a = to_abi(get(i));
            st[lane * chunks + 2 * i] = abi_lo(a);
            st[lane * chunks + 2 * i + 1] = abi_hi(a);
        }
        __syncthreads();
        for (uint32_t k = 0; k < chunks; ++k) {
            const uint32_t q = lane + k * 64;
            if (q < n_chunks) g[q] = st[q];
        }
        __syncthreads();
    }

    static void describe(EngineInfo &o) {
        std::snprintf(o.engine, sizeof o.engine, "LdsEngine<%d>", ALPHA);
        o.threads = kThreads;
        o.optimised = o.row_tables = o.lane_tables = o.mfma_dense = 0;
    }

    __device__ __forceinline__ void permute(uint32_t /*want_lo*/ = 0, uint32_t /*want_hi*/ = PMX_MAX_WIDTH, bool /*lane0_zero*/ = false) {
        uint32_t *const home = cur;
        permute_dense_rt<ALPHA>(*this, t, ark, mds, c, one, f);
        // Lanes may permute a different number of times (per-sponge modes) while load/store_states use the
        // wave-uniform `nxt` as staging: always leave the state in the buffer it started in.
        if (cur != home) {
            for (uint32_t q = 0; q < t * kN; ++q) nxt[q * 64 + lane] = cur[q * 64 + lane];
            swap();
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Kernels (identical for both engines)
// ------------------------------------------------------------------------------------------------
template <class Engine>
__global__ void __launch_bounds__(Engine::kThreads, Engine::kMinWaves) permute_kernel(const DevConfig d, const uint32_t *__restrict__ consts, uint64_t *__restrict__ states, size_t n) {
    Engine e(d, consts);
    e.load_states(states, n);
    // (all lanes of the result are wanted; the width is passed as the run-time value it also is in the other kernels: with the
    // compile-time constant the wide engines' last round is specialised and the register allocation of the whole kernel
    // comes out worse - 132 instead of 0..20 bytes of scratch per lane at t = 9, spills inside the sparse rounds)
    e.permute(0, e.c.rate + e.c.capacity);
    e.store_states(states, n);
}

// absorb `in_len` elements into this lane's sponge; idx is the next absorb index, or `rate` to force the
// permutation a Squeezing sponge performs first (mod.rs:247-252).  Returns the final next_absorb_index.
// The reference walks the input element by element and permutes whenever the rate is full and more input remains
// (mod.rs:137-148).  Sponges of one wave may stand at different positions, and a permutation costs the wave the same
// whether one lane needs it or all 64: so every lane keeps its OWN input cursor, and a pass of the loop is "every lane
// absorbs until its rate is full or its input ends, then the lanes that are full and have input left permute together".
// A wave then executes max-over-lanes permutations (2 for absorb(4) at rate 2 whatever the positions) instead of one per
// input position at which some lane happens to be full (4).
template <class Engine>
__device__ __forceinline__ uint32_t absorb_elements(Engine &e, const uint64_t *row, size_t in_len, uint32_t idx,
                                                    bool active) {
    const Rounds &c = e.c;
    size_t k = active ? 0 : in_len;   // this lane's cursor into its input row
    while (__builtin_amdgcn_ballot_w64(k < in_len)) {
        for (uint32_t j = 0; j < c.rate; ++j) {   // wave-uniform trip count; lanes drop out as they fill up or run dry
            if (k < in_len && idx < c.rate) {
                const Fe x = e.from_abi(abi_load(reinterpret_cast<const uint32_t *>(row + 4 * k)));
                const uint32_t pos = c.capacity + idx;
                // state[capacity + idx] += element (mod.rs:128,143); normalised so the permutation's own lazy
                // round-constant add stays within the limb bounds
                e.set(pos, fe_normalize(fe_add_lazy(e.get(pos), x)));
                idx += 1;
                k += 1;
            }
        }
        // rate filled and more input remains -> permute (mod.rs:137-148; also the :241-252 cases)
        const bool need = k < in_len && idx == c.rate;
        if (__builtin_amdgcn_ballot_w64(need)) {
            if (need) {
                e.permute();
                idx = 0;
            }
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
                abi_store(reinterpret_cast<uint32_t *>(row + 4 * (pos + k)), e.to_abi(e.get(c.capacity + idx + k)));
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

// Fixed-shape hash: every row runs  new; absorb(in_len elements); squeeze_native(out_len)  with the same lengths,
// so the whole state machine is wave-uniform.  It is written as ONE loop around ONE permutation call site (the
// permutation is the bulk of the kernel's code; two inlined copies overflow the instruction cache at t = 9).
template <class Engine>
__global__ void __launch_bounds__(Engine::kThreads, Engine::kMinWaves)
    hash_kernel(const DevConfig d, const uint32_t *__restrict__ consts, const uint64_t *__restrict__ in, size_t in_len,
                uint64_t *__restrict__ out, size_t out_len, size_t n) {
    Engine e(d, consts);
    const Rounds &c = e.c;
    const size_t gid = (size_t)blockIdx.x * Engine::kThreads + threadIdx.x;
    const bool active = gid < n;
    e.zero();                                                  // CryptographicSponge::new, mod.rs:219-230
    const uint64_t *row_in = in + (active ? gid : 0) * in_len * 4;
    uint64_t *row_out = out + (active ? gid : 0) * out_len * 4;
    size_t k_in = 0, rem = out_len, pos = 0;
    uint32_t idx = 0;
    bool squeezing = false, need = false;
    bool fresh = c.capacity >= 1;   // until the first permutation the capacity lanes of the new sponge - lane 0 among them - are zero
    // The sponge is dropped when the row is done, so the LAST permutation only has to produce the lanes the last squeeze
    // copies out: [capacity, capacity + rem) once rem <= rate elements are left (want_hi; otherwise the whole state).
    const uint32_t t_all = c.rate + c.capacity;
    uint32_t want_hi = t_all;
    for (;;) {
        if (need) {
            e.permute(want_hi < t_all ? c.capacity : 0, want_hi, fresh);
            need = false;
            fresh = false;
        }
        if (k_in < in_len) {                                   // absorb_internal, mod.rs:121-150
            if (idx == c.rate) {                               // rate full and more input remains
                need = true;
                idx = 0;
                continue;
            }
            Fe x = fe_zero();
            if (active) x = e.from_abi(abi_load(reinterpret_cast<const uint32_t *>(row_in + 4 * k_in)));
            const uint32_t at = c.capacity + idx;
            e.set(at, fe_normalize(fe_add_lazy(e.get(at), x)));
            ++idx;
            ++k_in;
            continue;
        }
        if (!squeezing) {                                      // Absorbing -> permute, squeeze from 0 (mod.rs:324-328)
            squeezing = true;
            need = true;
            idx = 0;
            if (rem <= c.rate) want_hi = c.capacity + (uint32_t)rem;
            continue;
        }
        const bool last = idx + rem <= c.rate;                 // squeeze_internal, mod.rs:153-182
        const uint32_t take = last ? (uint32_t)rem : c.rate - idx;
        for (uint32_t k = 0; k < take; ++k) {
            const Abi v = e.to_abi(e.get(c.capacity + idx + k));
            if (active) abi_store(reinterpret_cast<uint32_t *>(row_out + 4 * (pos + k)), v);
        }
        if (last) break;
        need = rem != c.rate;                                  // mod.rs:175, tested before the slice is advanced
        rem -= take;
        pos += take;
        idx = 0;
        if (need && rem <= c.rate) want_hi = c.capacity + (uint32_t)rem;
    }
}

// 2-to-1 compression, the Merkle-tree primitive:  out = (new; absorb([l, r]); squeeze_native(1))[0]
//   = permute(state with state[capacity] = l, state[capacity+1] = r, rest 0)[capacity]      (rate >= 2),
// because absorbing rate-many-or-fewer elements into a fresh sponge permutes exactly once, at the squeeze
// (src/poseidon/mod.rs:126-135, 324-328).  One permutation site, 64 contiguous bytes in and 32 out per lane.
template <class Engine>
__global__ void __launch_bounds__(Engine::kThreads, Engine::kMinWaves)
    compress_kernel(const DevConfig d, const uint32_t *__restrict__ consts, const uint64_t *__restrict__ in,
                    uint64_t *__restrict__ out, size_t n) {
    Engine e(d, consts);
    const size_t gid = (size_t)blockIdx.x * Engine::kThreads + threadIdx.x;
    const bool active = gid < n;
    const uint32_t *pair = reinterpret_cast<const uint32_t *>(in + (active ? gid : 0) * 8);
    e.zero();
    e.set(e.c.capacity, e.from_abi(abi_load(pair)));
    e.set(e.c.capacity + 1, e.from_abi(abi_load(pair + 8)));
    e.permute(e.c.capacity, e.c.capacity + 1, e.c.capacity >= 1);   // only the digest lane of the result is read; lane 0 (capacity) went in as zero
    const Abi digest = e.to_abi(e.get(e.c.capacity));
    if (active) abi_store(reinterpret_cast<uint32_t *>(out + gid * 4), digest);
}

// ------------------------------------------------------------------------------------------------
// QuadEngine: ONE state spread over a quad of lanes (pmx_permute.hpp, cooperative schedule; t = 3): lane q of a quad
// holds state element q, lane 3 is the spare that squares in the folded sparse rounds; the values a round exchanges
// travel by quad-broadcast DPP moves.  For latency-bound launches - the narrow levels of a tree, a handful of sponges -
// where what is paid is the length of one permutation's dependent chain: 32 k instructions here, 58-67 k with one lane
// per state.  LDS: the cooperative table, staged once per workgroup (256 threads = 64 states).
// ------------------------------------------------------------------------------------------------
template <int ALPHA>
struct QuadEngine {
    Rounds c;
    FieldRt f;
    Fe one;
    const uint32_t *coop;
    uint32_t q, role;
    Fe s;   // this lane's element of its quad's state

    static size_t lds_bytes(const DevConfig &d) { return (size_t)d.rounds.total_rounds * 3 * kCoopElems * kFeStride * 4; }

    __device__ __forceinline__ QuadEngine(const DevConfig &d, const uint32_t *consts) : c(d.rounds), f(d.field), one(d.one) {
        f.io = consts + d.io_offset;
        const uint32_t table_chunks = c.total_rounds * 3 * kCoopElems * kFeStride / 4;
        const uint4 *g4 = reinterpret_cast<const uint4 *>(consts + d.coop_offset);
        for (uint32_t k = threadIdx.x; k < table_chunks; k += 256) pmx_lds[k] = g4[k];
        __syncthreads();
        coop = reinterpret_cast<const uint32_t *>(pmx_lds);
        q = threadIdx.x & 3;
        role = q < 3 ? q : 2;   // lane 3 reads lane 2's entries; in the uniform rounds it shadows lane 2 (its result is never read)
        s = fe_zero();
    }

    __device__ __forceinline__ Fe from_abi(const Abi &x) const { return fe_from_abi_scaled(x); }      // the optimised schedule carries the ABI residue as its internal form (pmx_field.hpp: fe_from_abi_scaled)
    __device__ __forceinline__ Abi to_abi(const Fe &x) const { return fe_to_abi_scaled(x, f); }

    // element held by lane `lane` of this quad, in every lane
    template <int LANE>
    __device__ __forceinline__ static Fe quad(const Fe &v) {
        Fe r;
#pragma unroll
        // quad_perm [l, l, l, l]; every lane reads a live lane of its own quad, so there is no "old" value to keep: mov_dpp,
        // not update_dpp(0, ...), which costs a v_mov_b32 of the 0 before every move (27 per round on a lone wave's chain)
        for (int w = 0; w < kN; ++w) r.l[w] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[w], LANE * 0x55, 0xf, 0xf, false);
        return r;
    }

    // callers keep whole quads active or inactive together (the DPP moves read the other lanes of the quad)
    __device__ __forceinline__ void permute() {
        const uint32_t first_partial = c.half_full, last_partial = c.half_full + c.partial_rounds - 1;
        for (uint32_t r = 0; r < c.total_rounds; ++r) {
            const uint32_t *entry = coop + ((size_t)r * 3 + role) * kCoopElems * kFeStride;
            if (kCoopFolded<ALPHA> && r >= first_partial && r < last_partial) {   // sparse round, three multiplications deep
                const Fe x = quad<0>(fe_add_lazy(s, fe_const(entry)));
                const Fe res_a = coop_fold_a(q, x, entry, f);
                const Fe res_b = coop_fold_b(q, s, res_a, entry, f);
                Fe xpow = res_b;
#pragma unroll
                for (int k = 0; k < kCoopExtraSquarings<ALPHA>; ++k) xpow = mont_sqr(xpow, f);
                s = coop_fold_c(q, s, quad<3>(xpow), res_a, quad<1>(res_b), quad<2>(res_b), f);
                continue;
            }
            const Fe z = coop_pre<ALPHA>(s, entry, is_full_round(r, c) || q == 0, c, one, f);
            Fe zz[3];
            zz[0] = quad<0>(z);
            zz[1] = quad<1>(z);
            zz[2] = quad<2>(z);
            s = coop_layer_is_norm(r, c) ? coop_post_norm(zz, entry, f) : coop_post(zz, entry, f);
        }
        // lane 3 carries scratch values through the rounds; keep them bounded for the next call's lazy adds
        if (q == 3) s = fe_zero();
    }
};

// 2-to-1 compression on the quad engine (capacity 1, rate 2): state = [0, l, r], digest = element 1.
template <int ALPHA>
__global__ void __launch_bounds__(256, 2)
    compress_coop_kernel(const DevConfig d, const uint32_t *__restrict__ consts, const uint64_t *__restrict__ in,
                         uint64_t *__restrict__ out, size_t n) {
    QuadEngine<ALPHA> e(d, consts);
    const size_t g = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool active = g < n;
    if (active && (e.q == 1 || e.q == 2)) e.s = e.from_abi(abi_load(reinterpret_cast<const uint32_t *>(in + (g * 2 + (e.q - 1)) * 4)));
    e.permute();
    const Abi digest = e.to_abi(e.s);
    if (active && e.q == 1) abi_store(reinterpret_cast<uint32_t *>(out + g * 4), digest);   // state[capacity]
}

// The permutation itself on the quad engine (any rate / capacity split of width 3: lane q holds element q).
template <int ALPHA>
__global__ void __launch_bounds__(256, 2)
    permute_quad_kernel(const DevConfig d, const uint32_t *__restrict__ consts, uint64_t *__restrict__ states, size_t n) {
    QuadEngine<ALPHA> e(d, consts);
    const size_t g = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool active = g < n;
    uint32_t *mine = reinterpret_cast<uint32_t *>(states + ((active ? g : 0) * 3 + e.role) * 4);
    if (active && e.q < 3) e.s = e.from_abi(abi_load(mine));
    e.permute();
    const Abi v = e.to_abi(e.s);
    if (active && e.q < 3) abi_store(mine, v);
}

// The fixed-shape hash on the quad engine (capacity 1, rate 2): the same wave-uniform state machine as hash_kernel.
template <int ALPHA>
__global__ void __launch_bounds__(256, 2)
    hash_quad_kernel(const DevConfig d, const uint32_t *__restrict__ consts, const uint64_t *__restrict__ in, size_t in_len,
                     uint64_t *__restrict__ out, size_t out_len, size_t n) {
    QuadEngine<ALPHA> e(d, consts);
    const Rounds &c = e.c;
    const size_t g = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool active = g < n;
    const uint64_t *row_in = in + (active ? g : 0) * in_len * 4;
    uint64_t *row_out = out + (active ? g : 0) * out_len * 4;
    size_t k_in = 0, rem = out_len, pos = 0;
    uint32_t idx = 0;
    bool squeezing = false, need = false;
    for (;;) {
        if (need) {
            e.permute();
            need = false;
        }
        if (k_in < in_len) {                                   // absorb_internal, mod.rs:121-150
            if (idx == c.rate) {                               // rate full and more input remains
                need = true;
                idx = 0;
                continue;
            }
            Fe x = fe_zero();
            if (active) x = e.from_abi(abi_load(reinterpret_cast<const uint32_t *>(row_in + 4 * k_in)));
            if (e.q == c.capacity + idx) e.s = fe_normalize(fe_add_lazy(e.s, x));
            ++idx;
            ++k_in;
            continue;
        }
        if (!squeezing) {                                      // Absorbing -> permute, squeeze from 0 (mod.rs:324-328)
            squeezing = true;
            need = true;
            idx = 0;
            continue;
        }
        const bool last = idx + rem <= c.rate;                 // squeeze_internal, mod.rs:153-182
        const uint32_t take = last ? (uint32_t)rem : c.rate - idx;
        const Abi v = e.to_abi(e.s);
        for (uint32_t k = 0; k < take; ++k)
            if (active && e.q == c.capacity + idx + k) abi_store(reinterpret_cast<uint32_t *>(row_out + 4 * (pos + k)), v);
        if (last) break;
        need = rem != c.rate;                                  // mod.rs:175, tested before the slice is advanced
        rem -= take;
        pos += take;
        idx = 0;
    }
}

// The duplex-sponge driver on the quad engine (capacity 1, rate 2), same semantics as absorb_kernel / squeeze_kernel
// below: a handful of sponges - the single PoseidonSponge of the trait shims - is all latency.  Mode and index are
// per-sponge values, identical in the four lanes of a quad, so quads diverge as units.
template <int ALPHA>
__global__ void __launch_bounds__(256, 2)
    absorb_quad_kernel(const DevConfig d, const uint32_t *__restrict__ consts, uint64_t *__restrict__ states,
                       uint32_t *__restrict__ mode_tag, uint32_t *__restrict__ mode_index, const uint64_t *__restrict__ in,
                       size_t in_len, size_t n) {
    QuadEngine<ALPHA> e(d, consts);
    const size_t g = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool active = g < n;
    uint32_t *mine = reinterpret_cast<uint32_t *>(states + ((active ? g : 0) * 3 + e.role) * 4);
    if (active && e.q < 3) e.s = e.from_abi(abi_load(mine));
    uint32_t idx = 0;
    if (active) idx = (mode_tag[g] == PMX_MODE_ABSORBING) ? mode_index[g] : e.c.rate;
    if (idx > e.c.rate) idx = e.c.rate;                        // device-resident mode words are not validated by the host
    const uint64_t *row = in + (active ? g : 0) * in_len * 4;
    size_t k = active ? 0 : in_len;                            // per-sponge input cursor (see absorb_elements)
    while (__builtin_amdgcn_ballot_w64(k < in_len)) {
        for (uint32_t j = 0; j < e.c.rate; ++j) {
            if (k < in_len && idx < e.c.rate) {
                const Fe x = e.from_abi(abi_load(reinterpret_cast<const uint32_t *>(row + 4 * k)));
                if (e.q == e.c.capacity + idx) e.s = fe_normalize(fe_add_lazy(e.s, x));   // state[capacity + idx] += element (mod.rs:128,143)
                idx += 1;
                k += 1;
            }
        }
        const bool need = k < in_len && idx == e.c.rate;       // rate filled and more input remains (mod.rs:137-148, :241-252)
        if (__builtin_amdgcn_ballot_w64(need)) {
            if (need) {
                e.permute();
                idx = 0;
            }
        }
    }
    const Abi v = e.to_abi(e.s);
    if (active && e.q < 3) abi_store(mine, v);
    if (active && e.q == 0) {
        mode_tag[g] = PMX_MODE_ABSORBING;                      // mod.rs:130-132
        mode_index[g] = idx;
    }
}

template <int ALPHA>
__global__ void __launch_bounds__(256, 2)
    squeeze_quad_kernel(const DevConfig d, const uint32_t *__restrict__ consts, uint64_t *__restrict__ states,
                        uint32_t *__restrict__ mode_tag, uint32_t *__restrict__ mode_index, uint64_t *__restrict__ out,
                        size_t out_len, size_t n) {
    QuadEngine<ALPHA> e(d, consts);
    const size_t g = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool active = g < n;
    uint32_t *mine = reinterpret_cast<uint32_t *>(states + ((active ? g : 0) * 3 + e.role) * 4);
    if (active && e.q < 3) e.s = e.from_abi(abi_load(mine));
    uint32_t idx = 0;
    bool need = true;                                          // Absorbing -> permute, start at 0 (mod.rs:324-328)
    if (active && mode_tag[g] == PMX_MODE_SQUEEZING) {         // mod.rs:330-336
        idx = mode_index[g];
        if (idx > e.c.rate) idx = e.c.rate;
        need = idx == e.c.rate;
        if (need) idx = 0;
    }
    uint64_t *row = out + (active ? g : 0) * out_len * 4;
    size_t rem = out_len, pos = 0;
    bool done = !active;
    while (__builtin_amdgcn_ballot_w64(!done)) {               // squeeze_internal, mod.rs:153-182
        const bool do_perm = !done && need;
        if (__builtin_amdgcn_ballot_w64(do_perm)) {
            if (do_perm) e.permute();
        }
        if (!done) {
            const bool last = idx + rem <= e.c.rate;
            const uint32_t take = last ? (uint32_t)rem : e.c.rate - idx;
            const Abi v = e.to_abi(e.s);                 // every lane converts its own element; the matching one stores
            for (uint32_t k = 0; k < take; ++k)
                if (e.q == e.c.capacity + idx + k) abi_store(reinterpret_cast<uint32_t *>(row + 4 * (pos + k)), v);
            if (last) {
                idx += take;
                done = true;
            } else {
                need = rem != e.c.rate;                        // mod.rs:175, tested before the slice is advanced
                rem -= take;
                pos += take;
                idx = 0;
            }
        }
    }
    const Abi v = e.to_abi(e.s);
    if (active && e.q < 3) abi_store(mine, v);
    if (active && e.q == 0) {
        mode_tag[g] = PMX_MODE_SQUEEZING;                      // mod.rs:162-164
        mode_index[g] = idx;
    }
}

template <class Engine>
__global__ void __launch_bounds__(Engine::kThreads, Engine::kMinWavesDriver)
    absorb_kernel(const DevConfig d, const uint32_t *__restrict__ consts, uint64_t *__restrict__ states,
                  uint32_t *__restrict__ mode_tag, uint32_t *__restrict__ mode_index, const uint64_t *__restrict__ in,
                  size_t in_len, size_t n) {
    static_assert(!Engine::kWaveUniformOnly, "this engine's permutation cannot run under the per-lane EXEC masks of absorb_elements");
    Engine e(d, consts);
    const size_t gid = (size_t)blockIdx.x * Engine::kThreads + threadIdx.x;
    const bool active = gid < n;
    e.load_states(states, n);
    uint32_t idx = 0;
    if (active) idx = (mode_tag[gid] == PMX_MODE_ABSORBING) ? mode_index[gid] : e.c.rate;
    if (idx > e.c.rate) idx = e.c.rate;                        // device-resident mode words are not validated by the host
    idx = absorb_elements(e, in + (active ? gid : 0) * in_len * 4, in_len, idx, active);
    e.store_states(states, n);
    if (active) {
        mode_tag[gid] = PMX_MODE_ABSORBING;                    // mod.rs:130-132
        mode_index[gid] = idx;
    }
}

template <class Engine>
__global__ void __launch_bounds__(Engine::kThreads, Engine::kMinWavesDriver)
    squeeze_kernel(const DevConfig d, const uint32_t *__restrict__ consts, uint64_t *__restrict__ states,
                   uint32_t *__restrict__ mode_tag, uint32_t *__restrict__ mode_index, uint64_t *__restrict__ out,
                   size_t out_len, size_t n) {
    static_assert(!Engine::kWaveUniformOnly, "this engine's permutation cannot run under the per-lane EXEC masks of squeeze_elements");
    Engine e(d, consts);
    const size_t gid = (size_t)blockIdx.x * Engine::kThreads + threadIdx.x;
    const bool active = gid < n;
    e.load_states(states, n);
    uint32_t idx = 0;
    bool need = true;                                          // Absorbing -> permute, start at 0 (mod.rs:324-328)
    if (active && mode_tag[gid] == PMX_MODE_SQUEEZING) {       // mod.rs:330-336
        idx = mode_index[gid];
        if (idx > e.c.rate) idx = e.c.rate;                    // see absorb_kernel
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

// The absorb / squeeze driver of the wide states as PASSES (pmx_sponge_plan.hpp; mod.rs:121-182, 232-254, 321-341).
// A sponge's call alternates data movement and permutations; pass p is "the sponge's own p-th permutation, then whatever
// follows it up to the next one".  Data movement = input elements added into the rate portion (field addition on the ABI
// residues, mod.rs:128,143) or rate elements copied out (mod.rs:159-170).  It never indexes registers dynamically:
//   - a chunk IN FRONT of a permutation is added as the state comes into the permuting kernel's registers (AbsorbAdjust);
//   - a chunk right BEHIND a permutation is copied out of the LDS staging of that kernel's store (CopyOut);
//   - a chunk no permutation follows is moved in global memory by the lane that owns the sponge (sponge_walk), which also
//     rewrites the mode words when the call is over for the sponge (mod.rs:130-132, 162-164).
// Until then every step re-derives its share from the sponge's ORIGINAL mode words.
//   sponge_first_kernel    pass 0 over the whole batch in place: walk, then the workgroup permutes if any of its sponges is
//                          due and only those take the result; walk on; sponges due again go onto list 1 (one atomic per wave).
//   permute_listed_kernel  pass p >= 1: the permutation of the sponges on list p, then walk on -> list p + 1 or finished.
// Both run the permutation wave-uniform on the permutation engine of the width (the matrix-core one at t = 7..9) with
// nothing per-sponge live across it but one ballot; and from the second permutation on a batch in mixed modes costs the
// permutations the reference would execute, not max-over-a-wave of them.
template <bool SQUEEZE>
__device__ __forceinline__ bool sponge_walk(const Rounds &c, const uint32_t *__restrict__ p32, uint64_t *__restrict__ states, uint32_t *__restrict__ mode_tag,
                                            uint32_t *__restrict__ mode_index, uint64_t *__restrict__ io, size_t len, size_t sponge, bool active,
                                            uint32_t pass, uint32_t last_pass, bool first_move_done = false) {
    if (!active) return false;
    const uint32_t tag = mode_tag[sponge], index = mode_index[sponge], t_all = c.rate + c.capacity;
    for (uint32_t q = pass;; ++q) {
        const SpongePass sp = SQUEEZE ? squeeze_pass(tag, index, (uint32_t)len, c.rate, c.capacity, q) : absorb_pass(tag, index, (uint32_t)len, c.rate, c.capacity, q);
        // absorb: the chunk in front of a permutation is added by the kernel that permutes, as the state comes into its
        // registers (AbsorbAdjust) - only a chunk no permutation follows is added here, in memory
        if (!SQUEEZE && sp.permute) return true;
        uint32_t *st = reinterpret_cast<uint32_t *>(states + (sponge * t_all + sp.state_pos) * 4);
        uint32_t *row = reinterpret_cast<uint32_t *>(io + (sponge * len + sp.first) * 4);
        // squeeze: the chunk right behind a permutation was copied out of the LDS staging by the kernel that permuted (CopyOut)
        const uint32_t todo = (SQUEEZE && first_move_done && q == pass) ? 0 : sp.count;
        for (uint32_t j = 0; j < todo; ++j) {
            if constexpr (SQUEEZE) {
                abi_store(row + 8 * j, abi_load(st + 8 * j));
            } else {
                // state[capacity + i] += element: both fully reduced residues, the sum reduced exactly (no multiplication)
                abi_store(st + 8 * j, abi_add_mod(abi_load(st + 8 * j), abi_load(row + 8 * j), p32));
            }
        }
        if (sp.permute) return true;   // its permutation q follows: only q == pass can get here (permutations are numbered consecutively)
        if (q >= last_pass) {          // the call is over for this sponge
            mode_tag[sponge] = SQUEEZE ? PMX_MODE_SQUEEZING : PMX_MODE_ABSORBING;
            mode_index[sponge] = sp.end_index;
            return false;
        }
    }
}

// the sponges of this wave whose next permutation is due go onto the next pass's list, in lane order (one atomic per wave)
__device__ __forceinline__ void sponge_queue(bool queued, size_t sponge, uint32_t *__restrict__ list_next, uint32_t *__restrict__ count_next) {
    const uint64_t want = __builtin_amdgcn_ballot_w64(queued);
    if (want) {
        const uint32_t lane = threadIdx.x & 63, leader = (uint32_t)__builtin_ctzll(want);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(count_next, (uint32_t)__builtin_popcountll(want));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (queued) list_next[base + (uint32_t)__builtin_popcountll(want & ((1ull << lane) - 1))] = (uint32_t)sponge;
    }
}

// Pass 0 over the whole batch, in place: most calls send (nearly) every sponge through a first permutation - any absorb that
// overflows the rate, any squeeze of an absorbing sponge - so it is not worth a list; a workgroup none of whose sponges
// permutes leaves early, and the moves in front of the permutation hide under the other workgroups' arithmetic.
template <class Engine, bool SQUEEZE>
__global__ void __launch_bounds__(Engine::kThreads, Engine::kMinWaves)
    sponge_first_kernel(const DevConfig d, const uint32_t *__restrict__ consts, uint64_t *__restrict__ states,
                        uint32_t *__restrict__ mode_tag, uint32_t *__restrict__ mode_index, uint64_t *__restrict__ io, size_t len, size_t n,
                        uint32_t last_pass, uint32_t *__restrict__ list1, uint32_t *__restrict__ count1) {
    const size_t gid = (size_t)blockIdx.x * Engine::kThreads + threadIdx.x;
    const uint32_t *p32 = consts + d.io_offset + kIoP32;   // the modulus as 8 x 32-bit limbs (wave-uniform: scalar loads)
    const bool due = sponge_walk<SQUEEZE>(d.rounds, p32, states, mode_tag, mode_index, io, len, gid, gid < n, 0, last_pass);
    const uint64_t due_mask = __builtin_amdgcn_ballot_w64(due);       // wave-uniform: the only thing live across the permutation
    {   // workgroup vote through the first word of the DYNAMIC LDS (free until the engine is built): __syncthreads_or keeps a static
        // word of its own, which would come on top of the engine's dynamic LDS (72 KiB at t = 9: two workgroups per CU)
        uint32_t *flag = reinterpret_cast<uint32_t *>(pmx_lds);
        if (threadIdx.x == 0) *flag = 0;
        __syncthreads();
        if (due_mask != 0 && (threadIdx.x & 63) == 0) *flag = 1;   // (every writer writes the same value)
        __syncthreads();
        const bool any = *flag != 0;
        __syncthreads();
        if (!any) return;
    }
    {
        Engine e(d, consts);
        typename Engine::AbsorbAdjust add{nullptr, p32, 0, 0};
        if (!SQUEEZE && due) {   // the chunk in front of this sponge's permutation 0 (none if its mode asks for the permutation up front)
            const SpongePass sp = absorb_pass(mode_tag[gid], mode_index[gid], (uint32_t)len, d.rounds.rate, d.rounds.capacity, 0);
            add.row = reinterpret_cast<const uint32_t *>(io + (gid * len + sp.first) * 4);
            add.pos = sp.state_pos;
            add.count = sp.count;
        }
        e.load_states(states, n, add);
        e.permute(0, e.c.rate + e.c.capacity);   // (run-time width: see permute_kernel)
        typename Engine::CopyOut out{nullptr, 0, 0};
        if (SQUEEZE && ((due_mask >> (threadIdx.x & 63)) & 1)) {   // the chunk right behind permutation 0
            const SpongePass sp = squeeze_pass(mode_tag[gid], mode_index[gid], (uint32_t)len, d.rounds.rate, d.rounds.capacity, 1);
            out.row = reinterpret_cast<uint32_t *>(io + (gid * len + sp.first) * 4);
            out.pos = sp.state_pos;
            out.count = sp.count;
        }
        e.store_states_where(states, n, due_mask, out);
    }
    const bool mine_due = (due_mask >> (threadIdx.x & 63)) & 1;
    // (the wave's span was written by other lanes of the SAME wave after a workgroup barrier: make it visible to this lane's loads)
    __threadfence_block();
    const bool again = sponge_walk<SQUEEZE>(d.rounds, p32, states, mode_tag, mode_index, io, len, gid, mine_due, 1, last_pass, true);
    sponge_queue(again, gid, list1, count1);
}

template <class Engine, bool SQUEEZE>
__global__ void __launch_bounds__(Engine::kThreads, Engine::kMinWaves)
    permute_listed_kernel(const DevConfig d, const uint32_t *__restrict__ consts, uint64_t *__restrict__ states,
                          uint32_t *__restrict__ mode_tag, uint32_t *__restrict__ mode_index, uint64_t *__restrict__ io, size_t len,
                          uint32_t pass, uint32_t last_pass, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_count,
                          uint32_t *__restrict__ list_next, uint32_t *__restrict__ count_next) {
    // The count and the list were produced by the atomics and stores of the PREVIOUS launch, at addresses an earlier launch of
    // this very call has already read (the two lists alternate, the counters share a cache line): they are read with
    // device-scope atomic loads, past the scalar cache and the vector L1, which a back-to-back launch does not always
    // find invalidated (tools/soak.py: calls with two or more listed passes lost sponges to a stale zero count).
    const uint32_t count = __hip_atomic_load(const_cast<uint32_t *>(list_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((size_t)blockIdx.x * Engine::kThreads >= count) return;      // (the grid is sized for the whole batch)
    const size_t slot = (size_t)blockIdx.x * Engine::kThreads + threadIdx.x;
    const bool active = slot < count;
    const size_t sponge = __hip_atomic_load(const_cast<uint32_t *>(list) + (active ? slot : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        Engine e(d, consts);
        uint64_t *mine = states + sponge * (size_t)(e.c.rate + e.c.capacity) * 4;
        const uint32_t *p32 = consts + d.io_offset + kIoP32;
        typename Engine::AbsorbAdjust add{nullptr, p32, 0, 0};
        if (!SQUEEZE && active) {    // the chunk in front of this sponge's permutation `pass`
            const SpongePass sp = absorb_pass(mode_tag[sponge], mode_index[sponge], (uint32_t)len, d.rounds.rate, d.rounds.capacity, pass);
            add.row = reinterpret_cast<const uint32_t *>(io + (sponge * len + sp.first) * 4);
            add.pos = sp.state_pos;
            add.count = sp.count;
        }
        e.load_state_at(mine, add);
        e.permute(0, e.c.rate + e.c.capacity);   // (run-time width: see permute_kernel)
        typename Engine::CopyOut out{nullptr, 0, 0};
        if (SQUEEZE && active) {     // the chunk right behind it
            const SpongePass sp = squeeze_pass(mode_tag[sponge], mode_index[sponge], (uint32_t)len, d.rounds.rate, d.rounds.capacity, pass + 1);
            out.row = reinterpret_cast<uint32_t *>(io + (sponge * len + sp.first) * 4);
            out.pos = sp.state_pos;
            out.count = sp.count;
        }
        // a batch in ONE mode lists whole waves of consecutive sponges (the start kernel appends a wave's sponges in lane order)
        const uint32_t lane = threadIdx.x & 63;
        const size_t lead = (size_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sponge) | ((size_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sponge >> 32)) << 32);
        const uint64_t live = __builtin_amdgcn_ballot_w64(active);
        const bool together = __builtin_amdgcn_ballot_w64(!active || sponge == lead + lane) == ~0ull;
        e.store_state_at(mine, active, together, states + lead * (size_t)(e.c.rate + e.c.capacity) * 4, (uint32_t)__builtin_popcountll(live), out);
    }
    // (written through the wave's LDS region by the lanes of the SAME wave: make it visible to this lane's loads)
    __threadfence_block();
    const bool again = sponge_walk<SQUEEZE>(d.rounds, consts + d.io_offset + kIoP32, states, mode_tag, mode_index, io, len, sponge, active, pass + 1, last_pass, true);
    sponge_queue(again, sponge, list_next, count_next);
}

// ------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------
template <class Engine>
struct Launch {
    static int grid(size_t n) { return (int)((n + Engine::kThreads - 1) / Engine::kThreads); }
    static size_t lds(const DevConfig &c, uint32_t t) { return Engine::lds_bytes(c, t); }
    // more than 64 KiB of dynamic LDS has to be asked for per kernel (and device)
    template <class K>
    static void allow_lds(K kernel, size_t bytes) {
        if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    }

    static hipError_t permute(const DevConfig &c, uint32_t t, uint64_t *states, size_t n, hipStream_t st) {
        allow_lds(permute_kernel<Engine>, Engine::lds_bytes(c, t));
        hipLaunchKernelGGL(permute_kernel<Engine>, dim3(grid(n)), dim3(Engine::kThreads), Engine::lds_bytes(c, t), st, c,
                           c.consts, states, n);
        return hipGetLastError();
    }
    static hipError_t hash(const DevConfig &c, uint32_t t, const uint64_t *in, size_t in_len, uint64_t *out,
                           size_t out_len, size_t n, hipStream_t st) {
        allow_lds(hash_kernel<Engine>, Engine::lds_bytes(c, t));
        hipLaunchKernelGGL(hash_kernel<Engine>, dim3(grid(n)), dim3(Engine::kThreads), Engine::lds_bytes(c, t), st, c,
                           c.consts, in, in_len, out, out_len, n);
        return hipGetLastError();
    }
    static hipError_t compress(const DevConfig &c, uint32_t t, const uint64_t *in, uint64_t *out, size_t n, hipStream_t st) {
        allow_lds(compress_kernel<Engine>, Engine::lds_bytes(c, t));
        hipLaunchKernelGGL(compress_kernel<Engine>, dim3(grid(n)), dim3(Engine::kThreads), Engine::lds_bytes(c, t), st, c,
                           c.consts, in, out, n);
        return hipGetLastError();
    }
    static hipError_t absorb(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                             const uint64_t *in, size_t in_len, size_t n, hipStream_t st, const PassScratch & = PassScratch{}) {
        hipLaunchKernelGGL(absorb_kernel<Engine>, dim3(grid(n)), dim3(Engine::kThreads), Engine::lds_bytes(c, t), st, c,
                           c.consts, states, tag, index, in, in_len, n);
        return hipGetLastError();
    }
    static hipError_t squeeze(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                              uint64_t *out, size_t out_len, size_t n, hipStream_t st, const PassScratch & = PassScratch{}) {
        hipLaunchKernelGGL(squeeze_kernel<Engine>, dim3(grid(n)), dim3(Engine::kThreads), Engine::lds_bytes(c, t), st, c,
                           c.consts, states, tag, index, out, out_len, n);
        return hipGetLastError();
    }
    // the whole absorb / squeeze call as passes (pmx_sponge_plan.hpp): pass 0 over the batch, then one launch per further
    // permutation a sponge of the batch can need.  The two lists and the per-pass counters live in scratch the caller's
    // context keeps per stream (PassScratch).
    template <bool SQUEEZE>
    static hipError_t sponge_passes(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index, uint64_t *io,
                                    size_t len, size_t n, hipStream_t st, const PassScratch &provider) {
        const size_t passes = SQUEEZE ? squeeze_passes(len, c.rounds.rate) : absorb_passes(len, c.rounds.rate);
        if (passes == 0 || n == 0) return hipSuccess;
        if (n > 0xffffffffull || len > kSpongeMaxLen) return hipErrorInvalidValue;
        allow_lds(permute_listed_kernel<Engine, SQUEEZE>, Engine::lds_bytes(c, t));
        allow_lds(sponge_first_kernel<Engine, SQUEEZE>, Engine::lds_bytes(c, t));
        const uint32_t last = (uint32_t)(passes - 1);    // (no sponge permutes in the last pass)
        uint32_t *scratch = nullptr;                     // [counters, padded to 64 words | list A: n | list B: n]
        const size_t head = (passes + 63) / 64 * 64;
        hipError_t e = hipSuccess;
        uint32_t *lists[2] = {nullptr, nullptr};
        if (last > 1) {                                  // a second permutation is possible: its sponges travel on a list
            // (NOT hipMallocAsync / hipFreeAsync: with several contexts - several streams - alive, ROCm 7.0's stream-ordered pool
            // handed out blocks whose earlier use was still in flight; tools/soak.py lost sponges and took a memory fault that way)
            e = provider.get(provider.owner, st, (head + 2 * n) * 4, &scratch);
            if (e != hipSuccess) return e;
            // (from here on every way out passes provider.done: the block is released by the event recorded there)
            lists[0] = scratch + head;
            lists[1] = scratch + head + n;
            e = hipMemsetAsync(scratch, 0, head * 4, st);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL((sponge_first_kernel<Engine, SQUEEZE>), dim3(grid(n)), dim3(Engine::kThreads), Engine::lds_bytes(c, t), st, c, c.consts,
                               states, tag, index, io, len, n, last, lists[1], scratch ? scratch + 1 : nullptr);
            e = hipGetLastError();
        }
        for (uint32_t p = 1; p < last && e == hipSuccess; ++p) {
            hipLaunchKernelGGL((permute_listed_kernel<Engine, SQUEEZE>), dim3(grid(n)), dim3(Engine::kThreads), Engine::lds_bytes(c, t), st, c,
                               c.consts, states, tag, index, io, len, p, last, lists[p & 1], scratch + p, lists[(p + 1) & 1], scratch + p + 1);
            e = hipGetLastError();
        }
        if (scratch && provider.done) provider.done(provider.owner, st, scratch);   // behind the last launch that reads the lists
        return e;
    }
    // what a launch of `op` would run on (pmx_ctx_engine_info): filled by the engine, completed per kernel family here
    static hipError_t describe(const DevConfig &c, uint32_t t, int op, size_t len, EngineInfo *o) {
        Engine::describe(*o);
        const bool driver = op == PMX_OP_ABSORB || op == PMX_OP_SQUEEZE;
        o->waves_per_simd = driver ? Engine::kMinWavesDriver : Engine::kMinWaves;
        o->lds_bytes = (uint32_t)Engine::lds_bytes(c, t);
        o->launches = 1;
        return hipSuccess;
    }
};

#if PMX_TU == 99
// ---- tuning aid (Makefile target asm1): ONE kernel of one window engine, for reading its ISA and register report in seconds ----------
#if !defined(PMX_ONE_T) || !defined(PMX_ONE_ALPHA)
#error "PMX_TU = 99 (make asm1) names its engine with -DPMX_ONE_T=<width> -DPMX_ONE_ALPHA=<5 | 0>"
#endif
template __global__ void permute_kernel<HybridEngine<PMX_ONE_T, PMX_ONE_ALPHA>>(const DevConfig, const uint32_t *__restrict__, uint64_t *__restrict__, size_t);
#elif PMX_TU != 0
// ---- window engines of this translation unit ------------------------------------------------------------------------
// four translation units (they dominate the build time, so they compile in parallel): the exponent (1, 3: alpha = 5; 2, 4: any other) x the
// widths (1, 2: t = 3 .. 6; 3, 4: t = 7 .. 9).  The public launchers of TU 0 pick the half by t, after they have checked that the config
// has the engine's tables (DevConfig::mfma_dense) and that its LDS fits the device.
#if PMX_TU == 1 || PMX_TU == 3
#define PMX_HYB_ALPHA 5
#else
#define PMX_HYB_ALPHA 0
#endif
#if PMX_TU == 1
#define PMX_HYB_NAME(op) hybrid5n_##op
#elif PMX_TU == 3
#define PMX_HYB_NAME(op) hybrid5w_##op
#elif PMX_TU == 2
#define PMX_HYB_NAME(op) hybridgn_##op
#else
#define PMX_HYB_NAME(op) hybridgw_##op
#endif
#if PMX_TU == 1 || PMX_TU == 2
#define PMX_HYB_DISPATCH(CALL)                                             \
    switch (t) {                                                           \
        case 3: return Launch<HybridEngine<3, PMX_HYB_ALPHA>>::CALL;       \
        case 4: return Launch<HybridEngine<4, PMX_HYB_ALPHA>>::CALL;       \
        case 5: return Launch<HybridEngine<5, PMX_HYB_ALPHA>>::CALL;       \
        case 6: return Launch<HybridEngine<6, PMX_HYB_ALPHA>>::CALL;       \
        default: return hipErrorInvalidValue;                              \
    }
#else
#define PMX_HYB_DISPATCH(CALL)                                             \
    switch (t) {                                                           \
        case 7: return Launch<HybridEngine<7, PMX_HYB_ALPHA>>::CALL;       \
        case 8: return Launch<HybridEngine<8, PMX_HYB_ALPHA>>::CALL;       \
        case 9: return Launch<HybridEngine<9, PMX_HYB_ALPHA>>::CALL;       \
        default: return hipErrorInvalidValue;                              \
    }
#endif
hipError_t PMX_HYB_NAME(permute)(const DevConfig &c, uint32_t t, uint64_t *states, size_t n, hipStream_t st) { PMX_HYB_DISPATCH(permute(c, t, states, n, st)); }
hipError_t PMX_HYB_NAME(hash)(const DevConfig &c, uint32_t t, const uint64_t *in, size_t in_len, uint64_t *out, size_t out_len,
                              size_t n, hipStream_t st) {
    PMX_HYB_DISPATCH(hash(c, t, in, in_len, out, out_len, n, st));
}
hipError_t PMX_HYB_NAME(compress)(const DevConfig &c, uint32_t t, const uint64_t *in, uint64_t *out, size_t n, hipStream_t st) {
    PMX_HYB_DISPATCH(compress(c, t, in, out, n, st));
}
// absorb / squeeze run as passes on the permutation engine of the width (pmx_sponge_plan.hpp)
hipError_t PMX_HYB_NAME(absorb)(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                                const uint64_t *in, size_t len, size_t n, hipStream_t st, const PassScratch &scratch) {
    uint64_t *io = const_cast<uint64_t *>(in);   // (the absorb form of the pass kernel only reads `io`)
    PMX_HYB_DISPATCH(template sponge_passes<false>(c, t, states, tag, index, io, len, n, st, scratch));
}
hipError_t PMX_HYB_NAME(squeeze)(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                                 uint64_t *out, size_t len, size_t n, hipStream_t st, const PassScratch &scratch) {
    PMX_HYB_DISPATCH(template sponge_passes<true>(c, t, states, tag, index, out, len, n, st, scratch));
}
// LDS one workgroup of the width's engine asks for (the launchers of TU 0 compare it with the device's limit)
size_t PMX_HYB_NAME(lds_bytes)(const DevConfig &c, uint32_t t) { PMX_HYB_DISPATCH(lds(c, t)); }
// pmx_ctx_engine_info for the window engines
hipError_t PMX_HYB_NAME(describe)(const DevConfig &c, uint32_t t, int op, size_t len, EngineInfo *o) {
    const bool passes = op == PMX_OP_ABSORB || op == PMX_OP_SQUEEZE;
    const int op_engine = passes ? PMX_OP_PERMUTE : op;        // a pass is the permutation engine's launch
    auto plain = [&]() -> hipError_t { PMX_HYB_DISPATCH(describe(c, t, op_engine, len, o)); };
    const hipError_t e = plain();
    if (e == hipSuccess && passes) {
        const size_t passes_z = op == PMX_OP_SQUEEZE ? squeeze_passes(len, c.rounds.rate) : absorb_passes(len, c.rounds.rate);
        const int n_passes = passes_z > 0x7fffffff ? 0x7fffffff : (int)passes_z;
        o->launches = n_passes > 1 ? n_passes - 1 : n_passes;  // one launch per permutation a sponge of the batch can need
        std::snprintf(o->engine + std::strlen(o->engine), sizeof o->engine - std::strlen(o->engine), " x passes");
    }
    return e;
}

#else  // PMX_TU == 0
// ---- public launchers -------------------------------------------------------------------------------------------------
#define PMX_HYB_DECL(P)                                                                                                          \
    hipError_t P##permute(const DevConfig &, uint32_t, uint64_t *, size_t, hipStream_t);                                         \
    hipError_t P##hash(const DevConfig &, uint32_t, const uint64_t *, size_t, uint64_t *, size_t, size_t, hipStream_t);          \
    hipError_t P##compress(const DevConfig &, uint32_t, const uint64_t *, uint64_t *, size_t, hipStream_t);                      \
    hipError_t P##absorb(const DevConfig &, uint32_t, uint64_t *, uint32_t *, uint32_t *, const uint64_t *, size_t, size_t, hipStream_t, const PassScratch &); \
    hipError_t P##squeeze(const DevConfig &, uint32_t, uint64_t *, uint32_t *, uint32_t *, uint64_t *, size_t, size_t, hipStream_t, const PassScratch &); \
    size_t P##lds_bytes(const DevConfig &, uint32_t);                                                                            \
    hipError_t P##describe(const DevConfig &, uint32_t, int, size_t, EngineInfo *);
PMX_HYB_DECL(hybrid5n_)
PMX_HYB_DECL(hybrid5w_)
PMX_HYB_DECL(hybridgn_)
PMX_HYB_DECL(hybridgw_)

// Engine choice (round 6: three engines, one per regime - the register engine of t = 3 and the VALU-row hybrids of rounds 1-4 are gone):
//   QuadEngine     t = 3, at most 32768 units: one state per quad of lanes - the call is one permutation's latency
//   HybridEngine   t = 3 .. 9, configs that have the window tables (DevConfig::mfma_dense: the optimised schedule exists, at least two
//                  full rounds, the window algebra meets no zero - every config of the reference's tables, any modulus): alpha = 5
//                  specialised, any other exponent on the generic S-box; absorb / squeeze as passes
//   LdsEngine      everything else (t = 2, t >= 10, no partial section, a zero in the algebra): run-time width, the reference's dense schedule
// alpha 5 and 17 have dedicated addition chains in the quad and run-time-width engines, other exponents share the generic S-box.
static bool window_engine(const DevConfig &c, uint32_t t) {
    if (!c.has_opt || !c.mfma_dense || t < (uint32_t)PMX_MFMA_MIN_T || t > (uint32_t)PMX_MFMA_MAX_T) return false;
    const size_t lds = c.rounds.alpha == 5 ? (t <= 6 ? hybrid5n_lds_bytes(c, t) : hybrid5w_lds_bytes(c, t)) : (t <= 6 ? hybridgn_lds_bytes(c, t) : hybridgw_lds_bytes(c, t));
    return lds <= (size_t)c.max_lds_bytes;
}
// the window engine of a width: the exponent's half of the family, then the width's
#define PMX_WINDOW(CALL) (c.rounds.alpha == 5 ? (t <= 6 ? hybrid5n_##CALL : hybrid5w_##CALL) : (t <= 6 ? hybridgn_##CALL : hybridgw_##CALL))
#define PMX_LDS_ENGINE(CALL)                                                 \
    do {                                                                     \
        if (c.rounds.alpha == 5) return Launch<LdsEngine<5>>::CALL;          \
        if (c.rounds.alpha == 17) return Launch<LdsEngine<17>>::CALL;        \
        return Launch<LdsEngine<0>>::CALL;                                   \
    } while (0)

// the quad engine's table exists (t = 3, optimised schedule) and fits LDS; the lane of each element is fixed by the
// split only where elements are addressed through capacity / rate (quad_shape below)
static bool quad_table(const DevConfig &c, uint32_t t) {
    return t == 3 && c.has_opt && (size_t)c.rounds.total_rounds * 3 * kCoopElems * kFeStride * 4 <= (size_t)c.max_lds_bytes;
}
static bool quad_shape(const DevConfig &c, uint32_t t) { return quad_table(c, t) && c.rounds.capacity == 1 && c.rounds.rate == 2; }

// Small batches are latency: the quad engine up to this many states / rows / sponges / compressions (one dependent chain of 32 k
// instead of 55 k instructions: 0.066 ms up to 4096 states, 0.078 at 2^14, 0.132 at 2^15; above, the one-lane-per-state engine fills the
// chip better - profiles/r05/v_ab_t3_engine_threshold_32769.txt, w_ab_quad_kernels_up_to_16384_only_not_kept.txt).
static constexpr size_t kQuadMaxUnits = 32768;
#define PMX_QUAD_LAUNCH(KERNEL, ...)                                                                                        \
    do {                                                                                                                    \
        const dim3 grid_((unsigned)((n + 63) / 64));                                                                        \
        if (c.rounds.alpha == 5) hipLaunchKernelGGL(KERNEL<5>, grid_, dim3(256), QuadEngine<5>::lds_bytes(c), st, c, c.consts, __VA_ARGS__);        \
        else if (c.rounds.alpha == 17) hipLaunchKernelGGL(KERNEL<17>, grid_, dim3(256), QuadEngine<17>::lds_bytes(c), st, c, c.consts, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<0>, grid_, dim3(256), QuadEngine<0>::lds_bytes(c), st, c, c.consts, __VA_ARGS__);    \
        return hipGetLastError();                                                                                           \
    } while (0)

hipError_t launch_permute(const DevConfig &c, uint32_t t, uint64_t *states, size_t n, hipStream_t st) {
    if (quad_table(c, t) && n <= kQuadMaxUnits) PMX_QUAD_LAUNCH(permute_quad_kernel, states, n);
    if (window_engine(c, t)) return PMX_WINDOW(permute(c, t, states, n, st));
    PMX_LDS_ENGINE(permute(c, t, states, n, st));
}
hipError_t launch_hash(const DevConfig &c, uint32_t t, const uint64_t *in, size_t in_len, uint64_t *out, size_t out_len,
                       size_t n, hipStream_t st) {
    if (quad_shape(c, t) && n <= kQuadMaxUnits) PMX_QUAD_LAUNCH(hash_quad_kernel, in, in_len, out, out_len, n);
    if (window_engine(c, t)) return PMX_WINDOW(hash(c, t, in, in_len, out, out_len, n, st));
    PMX_LDS_ENGINE(hash(c, t, in, in_len, out, out_len, n, st));
}
hipError_t launch_compress(const DevConfig &c, uint32_t t, const uint64_t *in, uint64_t *out, size_t n, hipStream_t st) {
    // (the split (rate 3, capacity 0) of the same width takes the one-lane-per-state engine at every level)
    if (quad_shape(c, t) && n <= kQuadMaxUnits) PMX_QUAD_LAUNCH(compress_coop_kernel, in, out, n);
    if (window_engine(c, t)) return PMX_WINDOW(compress(c, t, in, out, n, st));
    PMX_LDS_ENGINE(compress(c, t, in, out, n, st));
}
hipError_t launch_absorb(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                         const uint64_t *in, size_t in_len, size_t n, hipStream_t st, const PassScratch &scratch) {
    if (quad_shape(c, t) && n <= kQuadMaxUnits) PMX_QUAD_LAUNCH(absorb_quad_kernel, states, tag, index, in, in_len, n);
    if (window_engine(c, t)) return PMX_WINDOW(absorb(c, t, states, tag, index, in, in_len, n, st, scratch));
    PMX_LDS_ENGINE(absorb(c, t, states, tag, index, in, in_len, n, st, scratch));
}
hipError_t launch_squeeze(const DevConfig &c, uint32_t t, uint64_t *states, uint32_t *tag, uint32_t *index,
                          uint64_t *out, size_t out_len, size_t n, hipStream_t st, const PassScratch &scratch) {
    if (quad_shape(c, t) && n <= kQuadMaxUnits) PMX_QUAD_LAUNCH(squeeze_quad_kernel, states, tag, index, out, out_len, n);
    if (window_engine(c, t)) return PMX_WINDOW(squeeze(c, t, states, tag, index, out, out_len, n, st, scratch));
    PMX_LDS_ENGINE(squeeze(c, t, states, tag, index, out, out_len, n, st, scratch));
}

// ---- pmx_ctx_engine_info: the same conditions, describing instead of launching ------------------------------------------
static hipError_t describe_quad(const DevConfig &c, EngineInfo *o) {
    std::snprintf(o->engine, sizeof o->engine, "QuadEngine<%d>", c.rounds.alpha == 5 ? 5 : c.rounds.alpha == 17 ? 17 : 0);
    o->threads = 256;            // 64 states, one per quad of lanes
    o->waves_per_simd = 2;       // __launch_bounds__(256, 2)
    o->lds_bytes = (uint32_t)QuadEngine<0>::lds_bytes(c);
    o->optimised = 1;            // element form, sparse rounds folded three multiplications deep
    o->launches = 1;
    return hipSuccess;
}
hipError_t describe_launch(const DevConfig &c, uint32_t t, int op, size_t n, size_t len, EngineInfo *o) {
    std::memset(o, 0, sizeof *o);
    o->width = (int)t;
    switch (op) {
        case PMX_OP_PERMUTE:
            if (quad_table(c, t) && n <= kQuadMaxUnits) return describe_quad(c, o);
            break;
        case PMX_OP_HASH:
        case PMX_OP_ABSORB:
        case PMX_OP_SQUEEZE:
        case PMX_OP_COMPRESS:
            if (quad_shape(c, t) && n <= kQuadMaxUnits) return describe_quad(c, o);
            break;
        default:
            return hipErrorInvalidValue;
    }
    if (window_engine(c, t)) return PMX_WINDOW(describe(c, t, op, len, o));
    PMX_LDS_ENGINE(describe(c, t, op, len, o));
}

// ---- authentication paths (pmx_merkle_verify_paths_dev) --------------------------------------------------------------
// Pure data movement: four lanes per path, one 16-byte quarter of the 64-byte pair each.
__global__ void __launch_bounds__(256) path_pairs_kernel(const uint4 *__restrict__ cur, const uint4 *__restrict__ paths,
                                                         const uint64_t *__restrict__ indices, size_t depth, size_t level,
                                                         uint4 *__restrict__ pairs, size_t k) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, i = gid >> 2;
    if (i >= k) return;
    const uint32_t quarter = gid & 3, half = quarter & 1;
    const bool right = (indices[i] >> level) & 1;          // the running node is the right child at this level
    const bool from_cur = (quarter >> 1) == (right ? 1u : 0u);
    pairs[gid] = from_cur ? cur[i * 2 + half] : paths[(i * depth + level) * 2 + half];
}

__global__ void __launch_bounds__(256) path_check_kernel(const uint4 *__restrict__ cur, const uint4 *__restrict__ root,
                                                         const uint64_t *__restrict__ indices, size_t depth,
                                                         uint8_t *__restrict__ ok, size_t k) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= k) return;
    const uint4 a = cur[i * 2], b = cur[i * 2 + 1], ra = root[0], rb = root[1];
    const bool same = a.x == ra.x && a.y == ra.y && a.z == ra.z && a.w == ra.w && b.x == rb.x && b.y == rb.y && b.z == rb.z && b.w == rb.w;
    ok[i] = (same && (indices[i] >> depth) == 0) ? 1 : 0;   // depth < 64 (checked by the caller)
}

hipError_t launch_path_pairs(const uint64_t *cur, const uint64_t *paths, const uint64_t *indices, size_t depth, size_t level,
                             uint64_t *pairs, size_t k, hipStream_t st) {
    hipLaunchKernelGGL(path_pairs_kernel, dim3((unsigned)((k * 4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint4 *>(cur),
                       reinterpret_cast<const uint4 *>(paths), indices, depth, level, reinterpret_cast<uint4 *>(pairs), k);
    return hipGetLastError();
}
hipError_t launch_path_check(const uint64_t *cur, const uint64_t *root, const uint64_t *indices, size_t depth, uint8_t *ok,
                             size_t k, hipStream_t st) {
    hipLaunchKernelGGL(path_check_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint4 *>(cur),
                       reinterpret_cast<const uint4 *>(root), indices, depth, ok, k);
    return hipGetLastError();
}
#endif  // PMX_TU

}  // namespace pmx

