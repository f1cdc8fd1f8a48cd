// Prime-field arithmetic of the gfx950 Poseidon kernels.
//
// Measured on MI355X (profiles/r01/valu_microbench.txt): v_mad_u64_u32 (32x32+64 -> 64) issues at HALF
// the v_add_u32 rate, and so does every carry instruction (v_add_co / v_addc_co) and every 64-bit add.
// A saturated 8 x 32-bit representation pays one carry instruction per limb product, i.e. doubles the
// multiply cost.  The kernels therefore compute in an UNSATURATED radix:
//
//     9 limbs of 29 bits (261 bits) held in uint32_t, Montgomery radix R' = 2^261.
//
//   * a limb product is < 2^58 (2^59/2^60 with lazily added operands), so up to 27 of them plus the 9
//     reduction products are summed in ONE 64-bit column accumulator by bare v_mad_u64_u32 - no carries;
//   * p < 2^255 leaves >= 6 spare bits: Montgomery outputs are < 1.3 p without any conditional
//     subtraction, additions are plain limb-wise adds, and nothing is compared against p until the final
//     conversion back to the ABI form;
//   * multiplication and reduction are interleaved column by column (one live accumulator), the chain of a
//     column seeded with the carry of the previous one (the build disables LLVM's Reassociate pass, which would
//     undo that - csrc/Makefile);
//   * products by CONSTANTS (every multiplication of the permutation except the S-box) are layers on the matrix cores (pmx_mfma.hpp);
//     the one such product left on the VALU - the history term of a t = 3 window - takes a shifted table: nine precomputed residues
//     of the constant, 81 + 18 multiplies instead of 81 + 81 (tab_lanes_stream below).
//
// ABI form (4 x u64 = 8 x u32 limbs, x * 2^256 mod p, fully reduced - ark-ff's Fp<MontBackend<_,4>,4>)
// is converted on load (x2^256 -> x2^261: one Montgomery product with 2^266 mod p) and on store (one
// product with 2^256 mod p, then the only exact reduction to [0, p)).  Results are therefore
// limb-identical to ark-ff's for:
//   add_assign / +=   reference src/poseidon/mod.rs:78,88,128,143
//   mul               reference src/poseidon/mod.rs:87
//   pow(&[alpha])     reference src/poseidon/mod.rs:67,72
//
// Bounds (B = value / p; "norm" = every limb < 2^29, "lazy" = every limb < 2^30):
//   redc output      norm, B < T / (p * 2^261) + 1            (T = the reduced integer)
//   fe_add_lazy      inputs norm -> output lazy, B = Ba + Bb
//   column sums      <= 27 products (one side lazy) + 9 reduction products + carry  < 2^64;  table products: <= 6
//                    normalised terms of 9 products + 2 reduction products + carry      (tests/test_hostcheck.py)
//   table product    norm, B < Bs + 1 + 2^-20 (s: the addend)
// p < 2^255  =>  p / 2^261 < 2^-6: with every operand B <= 4, T <= 3 * 16 p^2 gives outputs B < 1.75.
//
// The same source compiles for the host so the algorithms are unit-tested on CPU against the oracle
// (tests/hostcheck/); on the device `acc += (uint64_t)a * b` is exactly one v_mad_u64_u32.
#pragma once
#include <cstdint>
#include <type_traits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PMX_FN __host__ __device__ __forceinline__
#else
#define PMX_FN inline
#endif

// Host-only instrumentation for tests/hostcheck (never defined in the product build): PMX_TRACK(tag, x, f) reports
// an intermediate element so the test can record its largest limb and its magnitude value / p.
#if defined(PMX_HOSTCHECK) && !defined(__HIPCC__)
#define PMX_TRACK(tag, x, f) ::pmx::hostcheck_track(tag, x, f)
#else
#define PMX_TRACK(tag, x, f) ((void)0)
#endif

namespace pmx {

constexpr int kW = 29;                       // bits per limb
constexpr int kN = 9;                        // limbs
constexpr uint32_t kMask = (1u << kW) - 1;
constexpr int kFeStride = 12;                // u32 words per stored constant (9 used; 48-byte aligned rows)

// Scheduling fence (device only): nothing is moved across it by the machine scheduler.  Keeps the independent lane
// updates of a wide state from being interleaved (t = 6, 8 spilled kilobytes to scratch without it) and bounds how
// much of a constant stream is hoisted into SGPRs (tab_lanes_stream).
#if defined(__HIP_DEVICE_COMPILE__)
#define PMX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define PMX_SCHED_FENCE() ((void)0)
#endif

// Fence for a software-pipelined constant stream (tab_lanes_stream): sched_barrier alone only orders instructions that
// have side effects - instruction selection is free to place the multiplies of every chunk after the last barrier,
// which it does.  Passing the accumulator through an empty volatile asm with a memory clobber pins the multiplies
// that produce it before the fence and the constant loads that follow it after.
#if defined(__HIP_DEVICE_COMPILE__)
#define PMX_STREAM_FENCE(acc)                          \
    do {                                               \
        asm volatile("" : "+v"(acc) : : "memory");     \
        __builtin_amdgcn_sched_barrier(0);             \
    } while (0)
#else
#define PMX_STREAM_FENCE(acc) ((void)0)
#endif

// Guaranteed compile-time unrolling of the element loops (#pragma unroll gives up on bodies this large, and a
// rolled loop would index the register-resident state dynamically, i.e. push it to scratch memory).
template <int I, int N, class F>
PMX_FN void static_for(F &&fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        static_for<I + 1, N>(fn);
    }
}


struct Fe {
    uint32_t l[kN];
};

// ABI element: 8 x 32-bit limbs of x * 2^256 mod p
struct Abi {
    uint32_t w[8];
};

// Modulus view: wave-uniform run-time values (SGPRs on the device).  Only what the arithmetic of every round needs
// travels by value; the constants of the ABI conversions sit in the constant table behind `io` and are fetched
// where a conversion happens, so they do not occupy SGPRs for the length of a permutation.
struct FieldRt {
    uint32_t p[kN];       // modulus, 29-bit limbs
    uint32_t pinv;        // -p^-1 mod 2^29
    uint32_t unit;        // 1, as a run-time value: x * unit + acc is a single v_mad_u64_u32 (mont_mul_add, tab_col_end)
    const uint32_t *io;   // kIoWords words: [p as 8 x 32-bit limbs | 2^266 mod p | 2^256 mod p (9 x 29-bit limbs each)]
};
constexpr int kIoP32 = 0, kIoToInt = 8, kIoToAbi = 8 + kN, kIoWords = 28;

#if defined(PMX_HOSTCHECK) && !defined(__HIPCC__)
void hostcheck_track(int tag, const Fe &x, const FieldRt &f);   // defined in tests/hostcheck/pmx_hostcheck.cpp
void hostcheck_below_2_256(const Fe &x);                        // counts the elements that are not (pmx_mfma.hpp: inputs of a matrix-core layer)
#endif

// a stored constant: kFeStride words, 9 used (wave-uniform address -> scalar loads / LDS broadcast)
PMX_FN Fe fe_const(const uint32_t *ptr) {
    Fe r;
#pragma unroll
    for (int i = 0; i < kN; ++i) r.l[i] = ptr[i];
    return r;
}

PMX_FN Fe fe_zero() {
    Fe z;
#pragma unroll
    for (int i = 0; i < kN; ++i) z.l[i] = 0;
    return z;
}

// One Montgomery step on a column accumulator whose lower columns are already cleared: picks m_k so that
// acc + m_k p_0 = 0 mod 2^29 and leaves acc = (acc + m_k p_0) >> 29, the carry into the next column.
// (A variant for p = 1 mod 2^29 - BLS12-381 Fr: m_k = -acc mod 2^29, carry = (acc + 2^29 - 1) >> 29, i.e. a subtract
// and a 64-bit add in place of v_mul_lo_u32 and the multiply - measured +1.3 % on C2, +0.4 % on the hash driver and
// -1 % on the Merkle tree; not kept.)
PMX_FN uint32_t mont_step(uint64_t &acc, const FieldRt &f) {
    const uint32_t m = ((uint32_t)acc * f.pinv) & kMask;
    acc += (uint64_t)m * f.p[0];   // low 29 bits are now zero
    acc >>= kW;
    return m;
}

// limb-wise add, no carry propagation: inputs norm -> output lazy
PMX_FN Fe fe_add_lazy(const Fe &a, const Fe &b) {
    Fe r;
#pragma unroll
    for (int i = 0; i < kN; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// carry propagation: any limbs < 2^32 (value < 2^261) -> norm
PMX_FN Fe fe_normalize(const Fe &a) {
    Fe r;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < kN - 1; ++i) {
        const uint32_t v = a.l[i] + carry;   // < 2^32 as long as a.l[i] < 2^32 - 8
        r.l[i] = v & kMask;
        carry = v >> kW;
    }
    r.l[kN - 1] = a.l[kN - 1] + carry;
    return r;
}

// ---- interleaved (column-wise) Montgomery dot product -------------------------------------------------------
// returns  (sum_{j<T} a[j]*b[j]) * 2^-261  mod p  as a norm element, B < sum / (p 2^261) + 1.
// Operands: a[] lazy (limbs < 2^30), b[] norm.  Column k of the products and of the running m*p are summed
// in 64-bit accumulators; one accumulator takes at most 3 terms (27 products * 2^59 + 9 * 2^58 + carry
// < 2^64, checked in tests/test_hostcheck.py), so wider dots use ceil(T/3) accumulators whose low 29 bits
// and carries are combined once per column.
// ADD: + s, the addend entering the upper columns (as s * 2^261 before the reduction) like in mont_mul_add below: the sum
// comes out with normalised limbs, B < sum / (p 2^261) + Bs + 1.
template <int T, bool ADD>
PMX_FN Fe mont_dot_impl(const Fe *a, const Fe *b, const Fe &s, const FieldRt &f) {
    constexpr int G = (T + 2) / 3;
    uint32_t m[kN];
    Fe out;
    uint64_t acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0;
#pragma unroll
    for (int k = 0; k < 2 * kN - 1; ++k) {
        const int lo_i = k < kN ? 0 : k - (kN - 1);
        const int hi_i = k < kN ? k : kN - 1;
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int i = lo_i; i <= hi_i; ++i) acc[t / 3] += (uint64_t)a[t].l[i] * b[t].l[k - i];
        }
#pragma unroll
        for (int j = lo_i; j <= hi_i; ++j) {
            if (j < k || k >= kN) acc[0] += (uint64_t)m[j] * f.p[k - j];   // m_k itself is added below
        }
        if constexpr (G == 1) {
            if (k < kN) {
                m[k] = mont_step(acc[0], f);
            } else {
                if constexpr (ADD) acc[0] += (uint64_t)s.l[k - kN] * f.unit;
                out.l[k - kN] = (uint32_t)acc[0] & kMask;
                acc[0] >>= kW;
            }
        } else {
            uint32_t low = (uint32_t)acc[0] & kMask;
            if constexpr (ADD) {
                if (k >= kN) low += s.l[k - kN];
            }
            uint64_t carry = acc[0] >> kW;
#pragma unroll
            for (int g = 1; g < G; ++g) {
                low += (uint32_t)acc[g] & kMask;
                carry += acc[g] >> kW;
                acc[g] = 0;
            }
            if (k < kN) {
                m[k] = (low * f.pinv) & kMask;
                const uint64_t v = (uint64_t)m[k] * f.p[0] + low;   // low 29 bits are zero
                acc[0] = carry + (v >> kW);
            } else {
                out.l[k - kN] = low & kMask;
                acc[0] = carry + (low >> kW);
            }
        }
    }
    out.l[kN - 1] = (uint32_t)acc[0];
    if constexpr (ADD) out.l[kN - 1] += s.l[kN - 1];
    return out;
}

template <int T>
PMX_FN Fe mont_dot(const Fe *a, const Fe *b, const FieldRt &f) { return mont_dot_impl<T, false>(a, b, a[0], f); }
template <int T>
PMX_FN Fe mont_dot_add(const Fe *a, const Fe *b, const Fe &s, const FieldRt &f) { return mont_dot_impl<T, true>(a, b, s, f); }

PMX_FN Fe mont_mul(const Fe &a, const Fe &b, const FieldRt &f) { return mont_dot<1>(&a, &b, f); }

// a * b * 2^-261 + s  in one pass: the addend enters the upper columns of the product (as s * 2^261 before the
// reduction), so the sum comes out with normalised limbs and no separate addition / magnitude-cap pass.
// a lazy, b norm, s norm with value < 2^261 - 2p.  Result norm, B < Ba * Bb * p / 2^261 + Bs + 1: the accumulator
// this is used for (identity lanes of the sparse partial rounds) grows by about one p per round; the number of
// rounds is bounded by the host (pmx_prepare.hpp: opt_schedule_lane_headroom).
PMX_FN Fe mont_mul_add(const Fe &a, const Fe &b, const Fe &s, const FieldRt &f) {
    uint32_t m[kN];
    Fe out;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * kN - 1; ++k) {
        const int lo_i = k < kN ? 0 : k - (kN - 1);
        const int hi_i = k < kN ? k : kN - 1;
#pragma unroll
        for (int i = lo_i; i <= hi_i; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int j = lo_i; j <= hi_i; ++j) {
            if (j < k || k >= kN) acc += (uint64_t)m[j] * f.p[k - j];
        }
        if (k < kN) {
            m[k] = mont_step(acc, f);
        } else {
            acc += (uint64_t)s.l[k - kN] * f.unit;   // one mad; a plain 64-bit add would first widen s.l[] into a register pair
            out.l[k - kN] = (uint32_t)acc & kMask;
            acc >>= kW;
        }
    }
    out.l[kN - 1] = (uint32_t)acc + s.l[kN - 1];
    return out;
}

// a^2 * 2^-261: cross products once against the doubled operand (45 limb products instead of 81)
PMX_FN Fe mont_sqr(const Fe &a, const FieldRt &f) {
    uint32_t d[kN];  // 2 * a_j
#pragma unroll
    for (int i = 0; i < kN; ++i) d[i] = a.l[i] << 1;
    uint32_t m[kN];
    Fe out;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * kN - 1; ++k) {
#pragma unroll
        for (int i = 0; i < kN; ++i) {
            const int j = k - i;
            if (j > i && j < kN) acc += (uint64_t)a.l[i] * d[j];
        }
        if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
        if (k < kN) {
#pragma unroll
            for (int j = 0; j < k; ++j) acc += (uint64_t)m[j] * f.p[k - j];
            m[k] = mont_step(acc, f);
        } else {
#pragma unroll
            for (int j = k - (kN - 1); j < kN; ++j) acc += (uint64_t)m[j] * f.p[k - j];
            out.l[k - kN] = (uint32_t)acc & kMask;
            acc >>= kW;
        }
    }
    out.l[kN - 1] = (uint32_t)acc;
    return out;
}

// ---- products by CONSTANTS: shifted tables -----------------------------------------------------------------------
// For a constant C the host stores the nine residues  T_j = C * 2^(29 j + 58) mod p  (canonical, 9 limbs each), a row
// of N constants in consumption order (layout below).  Then for any element z (internal form, limbs z_j)
//     U = sum_j z_j * T_j  =  z * C * 2^58   (mod p),        U < 9 * 2^29 * p,
// is NINE columns of nine limb products, and two Montgomery steps (18 more multiplies) turn it into
// z * C * 2^-261 * 2^261 - the same value mont_mul(z, C~) returns - with the result below (1 + 2^-20) p.  That is 99
// multiplies for a product by a constant instead of 162, and 81 N + 18 for an N-term dot product instead of
// 81 N + 81; the price is 10x the table size (pmx_prepare.hpp), streamed through the scalar cache.
// Operands z must be norm (limbs < 2^29): a column then holds 9 products per term, at most 6 terms (+ 2 reduction
// products + carry < 2^64) per accumulator; wider dots use two accumulators combined once per column.
// ADD: returns sum_i z_i * C_i + s, the addend entering two columns up (s * 2^58 before the two steps), as in
// mont_mul_add; s norm, result norm with B < Bs + 1 + 2^-20.
constexpr int kTabSteps = 2;
// Table layout (pmx_prepare.hpp: put_shifted_row).  The words are grouped into CHUNKS of 27 padded to 32, in the
// order the products consume them, so that the scalar loads of one chunk never straddle the next (the compiler merges
// contiguous scalar loads up to 16 dwords, and a load that covers two chunks keeps the second one's SGPRs alive):
//   row of N >= 2 constants:  chunk (k, g) = column k of terms 3g .. 3g+2   at word (k * ceil(N/3) + g) * 32,
//                             inside it limb k of T_j(C_i) at (i - 3g) * 9 + j
//   single constant:          chunk h = columns 3h .. 3h+2                   at word h * 32,  inside it (k - 3h) * 9 + j
constexpr int kTabChunk = 3;
constexpr int kTabChunkWords = 32;
constexpr int kTabOneWords = (kN / kTabChunk) * kTabChunkWords;   // 96
PMX_FN constexpr int tab_row_words(int n) { return n == 1 ? kTabOneWords : kN * ((n + kTabChunk - 1) / kTabChunk) * kTabChunkWords; }
template <int N>
PMX_FN constexpr int tab_index(int i, int k, int j) {
    return N == 1 ? (k / kTabChunk) * kTabChunkWords + (k % kTabChunk) * kN + j
                  : (k * ((N + kTabChunk - 1) / kTabChunk) + i / kTabChunk) * kTabChunkWords + (i % kTabChunk) * kN + j;
}

// end of column k of a table product: the two reduction chains reach up to here, then either a Montgomery step
// (k < 2) or a result limb (with the addend entering first): s norm, result norm with B < Bs + 1 + 2^-20
PMX_FN void tab_col_end(int k, uint64_t &acc, uint32_t (&m)[kTabSteps], Fe &out, const Fe &s, const FieldRt &f) {
#pragma unroll
    for (int q = 0; q < kTabSteps; ++q) {
        if (q < k && k - q < kN) acc += (uint64_t)m[q] * f.p[k - q];
    }
    if (k < kTabSteps) {
        m[k] = mont_step(acc, f);
    } else {
        acc += (uint64_t)s.l[k - kTabSteps] * f.unit;
        if (k < kN + kTabSteps - 1) {
            out.l[k - kTabSteps] = (uint32_t)acc & kMask;
            acc >>= kW;
        } else {
            out.l[k - kTabSteps] = (uint32_t)acc;
        }
    }
}

// s[l] <- s[l] + z0 * w_l for L single constants whose tables follow each other (the identity lanes of one sparse
// round), as ONE stream: three columns per chunk, the next lane's first chunk in flight during the last of this one
template <int L>
PMX_FN void tab_lanes_stream(const Fe &z0, const uint32_t *tab, Fe *s, const FieldRt &f) {
    constexpr int kParts = kN / kTabChunk;   // chunks per lane
    constexpr int kChunks = L * kParts;
    uint32_t m[kTabSteps];
    Fe out;
    uint64_t acc = 0;
    uint32_t buf[2][kTabChunk * kN];
    auto load = [&](auto cc, uint32_t *b) {
        constexpr int c = decltype(cc)::value;
#pragma unroll
        for (int w = 0; w < kTabChunk * kN; ++w) b[w] = tab[(c / kParts) * kTabOneWords + (c % kParts) * kTabChunkWords + w];
    };
    load(std::integral_constant<int, 0>{}, buf[0]);
    static_for<0, kChunks>([&](auto cc) {
        constexpr int c = decltype(cc)::value, l = c / kParts, h = c % kParts;
        if constexpr (c + 1 < kChunks) load(std::integral_constant<int, c + 1>{}, buf[(c + 1) & 1]);
#pragma unroll
        for (int kk = 0; kk < kTabChunk; ++kk) {
#pragma unroll
            for (int j = 0; j < kN; ++j) acc += (uint64_t)z0.l[j] * buf[c & 1][kk * kN + j];
            tab_col_end(kTabChunk * h + kk, acc, m, out, s[l], f);
        }
        if constexpr (h == kParts - 1) {
#pragma unroll
            for (int k = kN; k < kN + kTabSteps; ++k) tab_col_end(k, acc, m, out, s[l], f);
            s[l] = out;
            acc = 0;
        }
        PMX_STREAM_FENCE(acc);
    });
}

// ---- run-time-width dot products: explicit column array ---------------------------------------------------------
// 18 64-bit columns; terms are added with cols_mul_acc and the columns re-compressed at least every 3 terms.
struct Cols {
    uint64_t c[2 * kN];
};

PMX_FN void cols_zero(Cols &t) {
#pragma unroll
    for (int k = 0; k < 2 * kN; ++k) t.c[k] = 0;
}

PMX_FN void cols_mul_acc(Cols &t, const Fe &a, const Fe &b) {
#pragma unroll
    for (int i = 0; i < kN; ++i) {
#pragma unroll
        for (int j = 0; j < kN; ++j) t.c[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
}

// push every column's bits above 29 into the next column
PMX_FN void cols_compress(Cols &t) {
#pragma unroll
    for (int k = 0; k < 2 * kN - 1; ++k) {
        t.c[k + 1] += t.c[k] >> kW;
        t.c[k] &= kMask;
    }
}

// Montgomery reduction of 18 columns, compressed or not: column k is consumed as (carry from below + its own sum), so
// the only requirement is that no column overflows 64 bits once its up to nine m_j p_i products and the carry (< 2^35)
// are added - callers bound their column sums accordingly (permute_dense_rt; tests/test_hostcheck.py replays the worst case).
// Norm result, B < value / (p 2^261) + 1.
// ADD: + s, entering the upper columns (s * 2^261) on their way out - one multiply-by-one v_mad per limb, nothing live
// before the reduction; s norm, B grows by Bs.
template <bool ADD = false>
PMX_FN Fe cols_redc(Cols &t, const FieldRt &f, const Fe *s = nullptr) {
#pragma unroll
    for (int k = 0; k < kN; ++k) {
        const uint32_t m = ((uint32_t)t.c[k] * f.pinv) & kMask;
#pragma unroll
        for (int j = 0; j < kN; ++j) t.c[k + j] += (uint64_t)m * f.p[j];
        t.c[k + 1] += t.c[k] >> kW;
    }
    Fe out;
#pragma unroll
    for (int k = kN; k < 2 * kN - 1; ++k) {
        if constexpr (ADD) t.c[k] += (uint64_t)s->l[k - kN] * f.unit;
        out.l[k - kN] = (uint32_t)t.c[k] & kMask;
        t.c[k + 1] += t.c[k] >> kW;
    }
    out.l[kN - 1] = (uint32_t)t.c[2 * kN - 1];
    if constexpr (ADD) out.l[kN - 1] += s->l[kN - 1];
    return out;
}

// x^alpha on the internal form.  5 and 17 use the shortest chains; anything else is MSB-first
// square-and-multiply seeded with x (alpha is wave-uniform, so the branches are scalar).
// x may be lazy with B <= 4, or norm with B < 7.6 (a window S-box input with a small-integer history term, pmx_permute.hpp: B^2 < 2^261 / p);
// the result is norm with B < 1.3 (alpha >= 4; pmx_prepare.hpp: opt_schedule_lane_headroom has the smaller exponents).  `one` = 2^261 mod p.
template <int ALPHA>
PMX_FN Fe fe_sbox(const Fe &x, uint64_t alpha, const Fe &one, const FieldRt &f) {
    if constexpr (ALPHA == 5) {
        const Fe x2 = mont_sqr(x, f);
        const Fe x4 = mont_sqr(x2, f);
        return mont_mul(x4, x, f);
    } else if constexpr (ALPHA == 17) {
        Fe y = mont_sqr(x, f);
        y = mont_sqr(y, f);
        y = mont_sqr(y, f);
        y = mont_sqr(y, f);
        return mont_mul(y, x, f);
    } else {
        if (alpha == 0) return one;
        // alpha = 1: x itself, but as a Montgomery PRODUCT (x * 2^261 * 2^-261): the optimised schedules add an S-box output into rows
        // unreduced and cut it into 32 bytes for the matrix cores, both on the bound of a product (B < 1.3, below 2^256), not of a lazy sum
        if (alpha == 1) return mont_mul(x, one, f);
        const int top = 63 - __builtin_clzll(alpha);
        Fe acc = mont_sqr(x, f);
        if ((alpha >> (top - 1)) & 1) acc = mont_mul(acc, x, f);
        for (int bit = top - 2; bit >= 0; --bit) {
            acc = mont_sqr(acc, f);
            if ((alpha >> bit) & 1) acc = mont_mul(acc, x, f);
        }
        return acc;
    }
}

// ---- ABI <-> internal ------------------------------------------------------------------------------------------
// bit-slice 8 x 32 -> 9 x 29 (value < 2^256)
PMX_FN Fe limbs_32_to_29(const Abi &x) {
    Fe r;
#pragma unroll
    for (int i = 0; i < kN; ++i) {
        const int bit = kW * i, wi = bit / 32, sh = bit % 32;
        uint64_t pair = x.w[wi];
        if (wi + 1 < 8) pair |= (uint64_t)x.w[wi + 1] << 32;
        r.l[i] = (uint32_t)(pair >> sh) & kMask;
    }
    return r;
}

// 9 x 29 (norm, value < 2^256) -> 8 x 32
PMX_FN Abi limbs_29_to_32(const Fe &x) {
    Abi r;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int bit = 32 * k, li = bit / kW, sh = bit % kW;   // word k starts inside limb li
        uint64_t v = (uint64_t)x.l[li] >> sh;
        if (li + 1 < kN) v |= (uint64_t)x.l[li + 1] << (kW - sh);
        if (li + 2 < kN && 2 * kW - sh < 32) v |= (uint64_t)x.l[li + 2] << (2 * kW - sh);
        r.w[k] = (uint32_t)v;
    }
    return r;
}

// x*2^256 (reduced) -> x*2^261 (norm, B < 1.02)
PMX_FN Fe fe_from_abi(const Abi &x, const FieldRt &f) { return mont_mul(limbs_32_to_29(x), fe_const(f.io + kIoToInt), f); }

// x*2^261 (norm or lazy, B <= 4) -> x*2^256 fully reduced to [0, p)
PMX_FN Abi fe_to_abi(const Fe &x, const FieldRt &f) {
    const Abi t = limbs_29_to_32(mont_mul(x, fe_const(f.io + kIoToAbi), f));   // B < 1.1: at most one subtraction
    uint32_t d[8];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t v = (uint64_t)t.w[i] - f.io[kIoP32 + i] - borrow;
        d[i] = (uint32_t)v;
        borrow = (uint32_t)(v >> 32) & 1u;
    }
    Abi r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.w[i] = borrow ? t.w[i] : d[i];
    return r;
}

// ---- ABI <-> internal without a multiplication (optimised schedule) ---------------------------------------------------------
// The optimised schedule carries its state scaled lane by lane anyway (pmx_prepare.hpp: derive_opt_tables); with the scaling
// 2^-5 at both ends of the permutation the internal form of a state element x is (x / 32) * 2^261 = x * 2^256 - the ABI
// residue itself.  Entering is bit-slicing 8 x 32 -> 9 x 29, leaving is the exact reduction to [0, p) and slicing back; the
// elements a sponge absorbs (added to the state in the same scaled coordinates) and squeezes are converted the same way.
PMX_FN Fe fe_from_abi_scaled(const Abi &x) { return limbs_32_to_29(x); }

// x norm with value below 3 p (a state lane after a permutation is below 1.3 p, after absorbing one more element 2.3 p)
PMX_FN Abi fe_to_abi_scaled(const Fe &x, const FieldRt &f) {
    Fe t = x;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        Fe d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < kN; ++i) {
            const uint32_t v = t.l[i] - f.p[i] - borrow;   // limbs below 2^29: bit 31 of the wrapped difference is the borrow
            d.l[i] = v & kMask;
            borrow = v >> 31;
        }
#pragma unroll
        for (int i = 0; i < kN; ++i) t.l[i] = borrow ? t.l[i] : d.l[i];
    }
    return limbs_29_to_32(t);
}

// a + b mod p on two fully reduced ABI residues (p32: the modulus as 8 x 32-bit limbs, FieldRt::io + kIoP32): one carry chain
// up, one borrow chain down, a select - the `state[capacity + i] += element` of the sponge drivers (mod.rs:128,143), which is
// a field addition and needs no Montgomery arithmetic.  a, b < p < 2^255: the sum fits 256 bits and one subtraction reduces it.
// (Unreduced device-resident data - the host entry points reject it - is still processed modulo p here as in the per-lane
// kernels: the carry out of the 256-bit sum takes part in the select, so a sum of 2^256 or more is reduced once as well.)
PMX_FN Abi abi_add_mod(const Abi &a, const Abi &b, const uint32_t *p32) {
    uint32_t sum[8], dif[8];
    uint32_t carry = 0, borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t v = (uint64_t)a.w[i] + b.w[i] + carry;
        sum[i] = (uint32_t)v;
        carry = (uint32_t)(v >> 32);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t v = (uint64_t)sum[i] - p32[i] - borrow;
        dif[i] = (uint32_t)v;
        borrow = (uint32_t)(v >> 32) & 1u;
    }
    Abi r;
    const bool keep_sum = borrow && !carry;   // sum < p
#pragma unroll
    for (int i = 0; i < 8; ++i) r.w[i] = keep_sum ? sum[i] : dif[i];
    return r;
}

#if defined(__HIPCC__)
// 32-byte ABI element <-> two 16-byte vectors
__device__ __forceinline__ Abi abi_from_u4(const uint4 &lo, const uint4 &hi) {
    Abi r;
    r.w[0] = lo.x; r.w[1] = lo.y; r.w[2] = lo.z; r.w[3] = lo.w;
    r.w[4] = hi.x; r.w[5] = hi.y; r.w[6] = hi.z; r.w[7] = hi.w;
    return r;
}
__device__ __forceinline__ uint4 abi_lo(const Abi &a) { return make_uint4(a.w[0], a.w[1], a.w[2], a.w[3]); }
__device__ __forceinline__ uint4 abi_hi(const Abi &a) { return make_uint4(a.w[4], a.w[5], a.w[6], a.w[7]); }

__device__ __forceinline__ Abi abi_load(const uint32_t *ptr) {  // 16-byte aligned global or LDS address
    const uint4 *q = reinterpret_cast<const uint4 *>(ptr);
    return abi_from_u4(q[0], q[1]);
}
__device__ __forceinline__ void abi_store(uint32_t *ptr, const Abi &a) {
    uint4 *q = reinterpret_cast<uint4 *>(ptr);
    q[0] = abi_lo(a);
    q[1] = abi_hi(a);
}
#endif

}  // namespace pmx
