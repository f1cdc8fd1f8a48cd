// Host-side 256-bit prime-field helpers (4 x u64 limbs, Montgomery R = 2^256).
// Used only off the data path: parameter generation (Grain LFSR, Cauchy MDS), Montgomery constants,
// config validation and canonical<->Montgomery conversion.  The batch data path is HIP-only.
#pragma once
#include <cstdint>
#include <cstring>

namespace pmx {

typedef unsigned __int128 u128;

struct U256 {
    uint64_t l[4];
};

inline bool u256_geq(const U256 &a, const U256 &b) {
    for (int i = 3; i >= 0; --i)
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    return true;
}

inline bool u256_is_zero(const U256 &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }

// r = a - b, returns borrow
inline uint64_t u256_sub(U256 &r, const U256 &a, const U256 &b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}

// r = a + b, returns carry
inline uint64_t u256_add(U256 &r, const U256 &a, const U256 &b) {
    uint64_t carry = 0;
    for (int i = 0; i < 4; ++i) {
        u128 s = (u128)a.l[i] + b.l[i] + carry;
        r.l[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
    }
    return carry;
}

struct HostField {
    U256 p;
    uint64_t inv;  // -p^-1 mod 2^64
    U256 r;        // 2^256 mod p  (Montgomery one)
    U256 r2;       // 2^512 mod p

    // p must be odd and > 1.
    bool init(const uint64_t modulus[4]) {
        std::memcpy(p.l, modulus, sizeof p.l);
        if ((p.l[0] & 1) == 0) return false;
        if (p.l[1] == 0 && p.l[2] == 0 && p.l[3] == 0 && p.l[0] < 3) return false;
        // Newton iteration for p^-1 mod 2^64 (doubles the correct bits each step)
        uint64_t x = p.l[0];
        for (int i = 0; i < 6; ++i) x *= 2 - p.l[0] * x;
        inv = (uint64_t)0 - x;
        // r = 2^256 mod p by 256 modular doublings of 1; r2 by 256 more
        U256 v = {{1, 0, 0, 0}};
        if (!u256_geq(p, v)) return false;
        for (int i = 0; i < 256; ++i) v = add(v, v);
        r = v;
        for (int i = 0; i < 256; ++i) v = add(v, v);
        r2 = v;
        return true;
    }

    unsigned bits() const {
        for (int i = 3; i >= 0; --i)
            if (p.l[i]) return 64u * (unsigned)i + (64u - (unsigned)__builtin_clzll(p.l[i]));
        return 0;
    }

    U256 add(const U256 &a, const U256 &b) const {
        U256 s;
        uint64_t c = u256_add(s, a, b);
        if (c || u256_geq(s, p)) u256_sub(s, s, p);
        return s;
    }

    // Montgomery product a*b*2^-256 mod p (word-serial, reduction interleaved)
    U256 mul(const U256 &a, const U256 &b) const {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            uint64_t c = 0;
            for (int j = 0; j < 4; ++j) {
                u128 v = (u128)a.l[j] * b.l[i] + t[j] + c;
                t[j] = (uint64_t)v;
                c = (uint64_t)(v >> 64);
            }
            u128 v = (u128)t[4] + c;
            t[4] = (uint64_t)v;
            t[5] = (uint64_t)(v >> 64);
            const uint64_t m = t[0] * inv;
            v = (u128)m * p.l[0] + t[0];
            c = (uint64_t)(v >> 64);
            for (int j = 1; j < 4; ++j) {
                v = (u128)m * p.l[j] + t[j] + c;
                t[j - 1] = (uint64_t)v;
                c = (uint64_t)(v >> 64);
            }
            v = (u128)t[4] + c;
            t[3] = (uint64_t)v;
            t[4] = t[5] + (uint64_t)(v >> 64);
        }
        U256 out = {{t[0], t[1], t[2], t[3]}};
        if (t[4] || u256_geq(out, p)) u256_sub(out, out, p);
        return out;
    }

    U256 neg(const U256 &a) const {
        if (u256_is_zero(a)) return a;
        U256 d;
        u256_sub(d, p, a);
        return d;
    }
    U256 sub(const U256 &a, const U256 &b) const { return add(a, neg(b)); }

    U256 to_mont(const U256 &x) const { return mul(x, r2); }
    U256 from_mont(const U256 &x) const {
        U256 one = {{1, 0, 0, 0}};
        return mul(x, one);
    }

    // a^(p-2) in the Montgomery domain (Fermat inverse; a != 0)
    U256 inverse(const U256 &a) const {
        U256 e = p, two = {{2, 0, 0, 0}};
        u256_sub(e, e, two);
        U256 acc = r;
        for (int bit = 255; bit >= 0; --bit) {
            acc = mul(acc, acc);
            if ((e.l[bit / 64] >> (bit % 64)) & 1) acc = mul(acc, a);
        }
        return acc;
    }
};

}  // namespace pmx
