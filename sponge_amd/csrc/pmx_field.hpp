// Device-side prime-field arithmetic for gfx950: 256-bit Montgomery residues as 8 x 32-bit limbs
// (the same bytes as the ABI's 4 little-endian u64 limbs).  CDNA4 has no 64x64 multiplier; the unit of
// work is v_mad_u64_u32 (32x32+64 -> 64), so the limb loops are written at 32-bit granularity.
//
// Replaces the ark-ff Fp<MontBackend<_,4>,4> operations the reference's hot path calls:
//   add_assign / +=   reference src/poseidon/mod.rs:78,88,128,143
//   mul               reference src/poseidon/mod.rs:87
//   pow(&[alpha])     reference src/poseidon/mod.rs:67,72
// Every function returns a fully reduced residue in [0, p), so results are limb-identical to ark-ff's.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "pmx_internal.hpp"

namespace pmx {

struct Fe {
    uint32_t l[8];
};

// Modulus view used by the arithmetic: runtime values that live in SGPRs (kernel arguments).
struct FieldRt {
    uint32_t p[8];       // modulus limbs
    uint32_t inv32;      // -p^-1 mod 2^32
};

__device__ __forceinline__ Fe fe_zero() {
    Fe z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.l[i] = 0;
    return z;
}

// t (8 limbs + carry word `hi`) -> t - p if t >= p
__device__ __forceinline__ Fe fe_cond_sub(const uint32_t t[8], uint32_t hi, const FieldRt &f) {
    uint32_t d[8];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t v = (uint64_t)t[i] - f.p[i] - borrow;
        d[i] = (uint32_t)v;
        borrow = (uint32_t)(v >> 32) & 1u;
    }
    const bool take = (hi != 0) | (borrow == 0);
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = take ? d[i] : t[i];
    return r;
}

__device__ __forceinline__ Fe fe_add(const Fe &a, const Fe &b, const FieldRt &f) {
    uint32_t s[8];
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t v = (uint64_t)a.l[i] + b.l[i] + carry;
        s[i] = (uint32_t)v;
        carry = (uint32_t)(v >> 32);
    }
    return fe_cond_sub(s, carry, f);
}

// Montgomery product a*b*2^-256 mod p.  Word-serial: one row of a*b[i], then one reduction row.
__device__ __forceinline__ Fe fe_mul(const Fe &a, const Fe &b, const FieldRt &f) {
    uint32_t t[8];
    uint32_t t8 = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t v = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint32_t)v;
            c = (uint32_t)(v >> 32);
        }
        uint64_t v = (uint64_t)t8 + c;
        t8 = (uint32_t)v;
        uint32_t t9 = (uint32_t)(v >> 32);
        const uint32_t m = t[0] * f.inv32;
        v = (uint64_t)m * f.p[0] + t[0];
        c = (uint32_t)(v >> 32);
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            v = (uint64_t)m * f.p[j] + t[j] + c;
            t[j - 1] = (uint32_t)v;
            c = (uint32_t)(v >> 32);
        }
        v = (uint64_t)t8 + c;
        t[7] = (uint32_t)v;
        t8 = t9 + (uint32_t)(v >> 32);
    }
    return fe_cond_sub(t, t8, f);
}

__device__ __forceinline__ Fe fe_sqr(const Fe &a, const FieldRt &f) { return fe_mul(a, a, f); }

// x^alpha.  5 and 17 use the shortest chains; anything else is MSB-first square-and-multiply seeded
// with x (alpha is wave-uniform, so the branches are scalar).  alpha == 0 -> one, alpha == 1 -> x.
template <int ALPHA>
__device__ __forceinline__ Fe fe_sbox(const Fe &x, uint64_t alpha, const Fe &one, const FieldRt &f) {
    if constexpr (ALPHA == 5) {
        const Fe x2 = fe_sqr(x, f);
        const Fe x4 = fe_sqr(x2, f);
        return fe_mul(x4, x, f);
    } else if constexpr (ALPHA == 17) {
        Fe y = fe_sqr(x, f);
        y = fe_sqr(y, f);
        y = fe_sqr(y, f);
        y = fe_sqr(y, f);
        return fe_mul(y, x, f);
    } else {
        if (alpha == 0) return one;
        int top = 63 - __builtin_clzll(alpha);
        Fe acc = x;
        for (int bit = top - 1; bit >= 0; --bit) {
            acc = fe_sqr(acc, f);
            if ((alpha >> bit) & 1) acc = fe_mul(acc, x, f);
        }
        return acc;
    }
}

// 32-byte element <-> two 16-byte vectors
__device__ __forceinline__ Fe fe_from_u4(const uint4 &lo, const uint4 &hi) {
    Fe r;
    r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
    r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
    return r;
}
__device__ __forceinline__ uint4 fe_lo(const Fe &a) { return make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]); }
__device__ __forceinline__ uint4 fe_hi(const Fe &a) { return make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]); }

__device__ __forceinline__ Fe fe_load(const uint32_t *ptr) {  // 16-byte aligned (global or LDS)
    const uint4 *q = reinterpret_cast<const uint4 *>(ptr);
    return fe_from_u4(q[0], q[1]);
}
__device__ __forceinline__ void fe_store(uint32_t *ptr, const Fe &a) {
    uint4 *q = reinterpret_cast<uint4 *>(ptr);
    q[0] = fe_lo(a);
    q[1] = fe_hi(a);
}

}  // namespace pmx
