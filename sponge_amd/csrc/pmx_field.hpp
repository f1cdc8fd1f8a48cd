// Prime-field arithmetic for gfx950: 256-bit Montgomery residues as 8 x 32-bit limbs (the same bytes as
// the ABI's 4 little-endian u64 limbs).  CDNA4 has no 64x64 multiplier; the unit of multiply work is
// v_mad_u64_u32 (32x32+64 -> 64 with carry-out), so everything is written at 32-bit granularity.
//
// Structure (product scanning / Comba): a column accumulator of 96 bits (64-bit `acc` + 32-bit `ovf`)
// receives one v_mad_u64_u32 + one v_addc_co_u32 per limb product.  Wide (512-bit) results are only
// reduced when needed:
//     mul_wide / sqr_wide / dot_wide   ->  17-limb unreduced sums of products
//     redc                             ->  Montgomery reduction of such a sum, fully reduced to [0, p)
// so an MDS row (a t-term dot product) pays ONE reduction, and a squaring does 36 instead of 64 products.
//
// Replaces the ark-ff Fp<MontBackend<_,4>,4> operations the reference's hot path calls:
//   add_assign / +=   reference src/poseidon/mod.rs:78,88,128,143
//   mul               reference src/poseidon/mod.rs:87
//   pow(&[alpha])     reference src/poseidon/mod.rs:67,72
// Every public result is a fully reduced residue in [0, p): limb-identical to ark-ff's.
//
// The same source compiles for the host (portable C++ in place of the two asm statements) so that the
// algorithms are unit-tested on CPU against the oracle (tools/host_field_check.cpp).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PMX_FN __host__ __device__ __forceinline__
#else
#define PMX_FN inline
#endif

namespace pmx {

struct Fe {
    uint32_t l[8];
};

// Modulus view: run-time values, wave-uniform (SGPRs on the device).
struct FieldRt {
    uint32_t p[8];   // modulus limbs
    uint32_t inv32;  // -p^-1 mod 2^32
};

// An unreduced sum of at most `lazy_terms` products of reduced residues: value < 2^512 + small, 17 limbs.
struct Wide {
    uint32_t w[17];
};

PMX_FN Fe fe_zero() {
    Fe z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.l[i] = 0;
    return z;
}

// ---- column accumulator primitives -------------------------------------------------------------------
// (acc, ovf) += a * b
PMX_FN void mac(uint64_t &acc, uint32_t &ovf, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry) : "v"(a), "v"(b));
    asm("v_addc_co_u32 %0, %1, 0, %0, %1" : "+v"(ovf), "+s"(carry));
#else
    const unsigned __int128 t = (unsigned __int128)acc + (uint64_t)a * b;
    acc = (uint64_t)t;
    ovf += (uint32_t)(t >> 64);
#endif
}

// same with b wave-uniform (an SGPR operand: modulus limbs, scalar-loaded constants)
PMX_FN void mac_s(uint64_t &acc, uint32_t &ovf, uint32_t a, uint32_t b_uniform) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry) : "v"(a), "s"(b_uniform));
    asm("v_addc_co_u32 %0, %1, 0, %0, %1" : "+v"(ovf), "+s"(carry));
#else
    mac(acc, ovf, a, b_uniform);
#endif
}

// (acc, ovf) += x  (32-bit)
PMX_FN void acc_add(uint64_t &acc, uint32_t &ovf, uint32_t x) {
    const uint64_t prev = acc;
    acc += x;
    ovf += (acc < prev) ? 1u : 0u;
}

// (acc, ovf) = 2 * (acc, ovf)
PMX_FN void acc_double(uint64_t &acc, uint32_t &ovf) {
    ovf = (ovf << 1) | (uint32_t)(acc >> 63);
    acc <<= 1;
}

// emit the low limb and shift the accumulator down one limb
PMX_FN uint32_t acc_shift(uint64_t &acc, uint32_t &ovf) {
    const uint32_t out = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)ovf << 32);
    ovf = 0;
    return out;
}

// ---- conditional subtraction / modular add ---------------------------------------------------------------
// t (8 limbs + carry word `hi`) -> t - p if t >= p
PMX_FN Fe fe_cond_sub(const uint32_t t[8], uint32_t hi, const FieldRt &f, uint32_t *hi_out = nullptr) {
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
    if (hi_out) *hi_out = take ? hi - borrow : hi;
    return r;
}

PMX_FN Fe fe_add(const Fe &a, const Fe &b, const FieldRt &f) {
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

// ---- wide products -----------------------------------------------------------------------------------------
// Column k of the schoolbook product a*b, accumulated into (acc, ovf).  BS: b is wave-uniform.
template <bool BS>
PMX_FN void mul_column(uint64_t &acc, uint32_t &ovf, const Fe &a, const Fe &b, int k) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = k - i;
        if (j >= 0 && j < 8) {
            if constexpr (BS) mac_s(acc, ovf, a.l[i], b.l[j]);
            else mac(acc, ovf, a.l[i], b.l[j]);
        }
    }
}

// r = sum_{j<T} a[j] * b[j]   (unreduced; the caller guarantees the sum fits the lazy bound)
template <int T, bool BS>
PMX_FN Wide dot_wide(const Fe *a, const Fe *b) {
    Wide r;
    uint64_t acc = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
#pragma unroll
        for (int j = 0; j < T; ++j) mul_column<BS>(acc, ovf, a[j], b[j], k);
        r.w[k] = acc_shift(acc, ovf);
    }
    r.w[15] = (uint32_t)acc;
    r.w[16] = (uint32_t)(acc >> 32);
    return r;
}

PMX_FN Wide mul_wide(const Fe &a, const Fe &b) { return dot_wide<1, false>(&a, &b); }

// a^2: cross products once, doubled, plus the squares (36 limb products instead of 64)
PMX_FN Wide sqr_wide(const Fe &a) {
    Wide r;
    uint64_t acc = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        // carry from the previous column is already in (acc, ovf); cross terms go to a fresh accumulator
        uint64_t cacc = 0;
        uint32_t covf = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = k - i;
            if (j > i && j < 8) mac(cacc, covf, a.l[i], a.l[j]);
        }
        acc_double(cacc, covf);
        if ((k & 1) == 0) mac(cacc, covf, a.l[k / 2], a.l[k / 2]);
        // (acc, ovf) += (cacc, covf)
        const uint64_t prev = acc;
        acc += cacc;
        ovf += covf + ((acc < prev) ? 1u : 0u);
        r.w[k] = acc_shift(acc, ovf);
    }
    r.w[15] = (uint32_t)acc;
    r.w[16] = (uint32_t)(acc >> 32);
    return r;
}

// r += x * 2^256  (adds a reduced residue into the upper half: used for  w*z0 + z_i  before one reduction)
PMX_FN void wide_add_hi(Wide &r, const Fe &x) {
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t v = (uint64_t)r.w[8 + i] + x.l[i] + carry;
        r.w[8 + i] = (uint32_t)v;
        carry = (uint32_t)(v >> 32);
    }
    r.w[16] += carry;
}

// Montgomery reduction: (T + m*p) / 2^256 fully reduced to [0, p).
// Requires T < 2^256 * 2p  (so the quotient is < 3p): the host checks the lazy bounds per modulus.
PMX_FN Fe redc(const Wide &t, const FieldRt &f) {
    uint32_t m[8];
    uint32_t out[8];
    uint64_t acc = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        acc_add(acc, ovf, t.w[k]);
#pragma unroll
        for (int j = 0; j < k; ++j) mac_s(acc, ovf, m[j], f.p[k - j]);
        m[k] = (uint32_t)acc * f.inv32;
        mac_s(acc, ovf, m[k], f.p[0]);
        (void)acc_shift(acc, ovf);  // low limb is zero by construction
    }
#pragma unroll
    for (int k = 8; k < 16; ++k) {
        acc_add(acc, ovf, t.w[k]);
#pragma unroll
        for (int j = k - 7; j < 8; ++j) mac_s(acc, ovf, m[j], f.p[k - j]);
        out[k - 8] = acc_shift(acc, ovf);
    }
    uint32_t hi = (uint32_t)acc + t.w[16];
    // quotient < 3p: at most two subtractions; after the first it is < 2p <= 2^256 (p < 2^255 is enforced)
    uint32_t hi2;
    const Fe once = fe_cond_sub(out, hi, f, &hi2);
    return fe_cond_sub(once.l, hi2, f);
}

PMX_FN Fe fe_mul(const Fe &a, const Fe &b, const FieldRt &f) { return redc(mul_wide(a, b), f); }
PMX_FN Fe fe_sqr(const Fe &a, const FieldRt &f) { return redc(sqr_wide(a), f); }

// x^alpha.  5 and 17 use the shortest chains; anything else is MSB-first square-and-multiply seeded
// with x (alpha is wave-uniform, so the branches are scalar).  alpha == 0 -> one, alpha == 1 -> x.
template <int ALPHA>
PMX_FN Fe fe_sbox(const Fe &x, uint64_t alpha, const Fe &one, const FieldRt &f) {
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
        const int top = 63 - __builtin_clzll(alpha);
        Fe acc = x;
        for (int bit = top - 1; bit >= 0; --bit) {
            acc = fe_sqr(acc, f);
            if ((alpha >> bit) & 1) acc = fe_mul(acc, x, f);
        }
        return acc;
    }
}

#if defined(__HIPCC__)
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
#endif

}  // namespace pmx
