// Input encodings of the reference's Absorb / AbsorbWithLength traits (src/absorb.rs) for the C++ host mirror, plus
// CryptographicSponge::fork (src/lib.rs:149-157) and the absorb! / collect_sponge_* macros (src/absorb.rs:319-355).
// Overload set:  to_sponge_bytes(x, dest)  /  to_sponge_field_elements(field, x, dest)  for
//   uint8_t..uint64_t, unsigned __int128, int8_t..int64_t, bool, Fp (native), std::vector<A>, std::optional<A>,
//   WithLength<std::vector<A>>, TEAffine / SWAffine (curve points as their base-field coordinates, src/absorb.rs:232-254).
// Two encodings rest on ark-ff / ark-serialize behaviour outside the reference tree: byte slices are packed
// (MODULUS_BIT_SIZE-1)/8 bytes per element after a u64-LE length prefix (src/absorb.rs:135-139), and an Fp
// serialises as ceil(MODULUS_BIT_SIZE/8) little-endian bytes of its canonical value (:153-155).
#pragma once
#include <optional>
#include <type_traits>

#include "poseidon_sponge.hpp"

namespace pmx_host {

template <class V>
struct WithLength {   // AbsorbWithLength::to_sponge_*_with_length (src/absorb.rs:84-101)
    const V &seq;
};
template <class V>
WithLength<V> with_length(const V &v) { return WithLength<V>{v}; }

// ---- bytes ----------------------------------------------------------------------------------------------------------
template <class U, std::enable_if_t<std::is_integral_v<U> && !std::is_same_v<U, bool>, int> = 0>
inline void to_sponge_bytes(U x, std::vector<uint8_t> &dest) {   // to_le_bytes (two's complement for signed)
    using W = std::make_unsigned_t<U>;
    W w = (W)x;
    for (size_t i = 0; i < sizeof(U); ++i) dest.push_back((uint8_t)(w >> (8 * i)));
}
inline void to_sponge_bytes(unsigned __int128 x, std::vector<uint8_t> &dest) {
    for (int i = 0; i < 16; ++i) dest.push_back((uint8_t)(x >> (8 * i)));
}
inline void to_sponge_bytes(bool x, std::vector<uint8_t> &dest) { dest.push_back(x ? 1 : 0); }
struct FpOf {   // an element together with its field (serialisation needs the modulus size)
    const Field &f;
    Fp x;
};
inline void to_sponge_bytes(const FpOf &e, std::vector<uint8_t> &dest) {   // serialize_compressed
    const auto c = fp_into_bigint(e.f, e.x);
    const uint8_t *b = reinterpret_cast<const uint8_t *>(c.data());
    dest.insert(dest.end(), b, b + (e.f.modulus_bit_size() + 7) / 8);
}
// Curve points (src/absorb.rs:232-254): ark-ec's ToConstraintField gives [x, y] for a twisted-Edwards affine point and
// [x, y, infinity] for a short-Weierstrass one, over the curve's base field `f`.
struct TEAffine {
    const Field &f;
    Fp x, y;
    std::vector<FpOf> to_field_elements() const { return {FpOf{f, x}, FpOf{f, y}}; }
};
struct SWAffine {
    const Field &f;
    Fp x, y;
    bool infinity;
    std::vector<FpOf> to_field_elements() const {
        return {FpOf{f, x}, FpOf{f, y}, FpOf{f, fp_from_bigint(f, {infinity ? 1ull : 0ull, 0, 0, 0})}};
    }
};
template <class P, std::enable_if_t<std::is_same_v<P, TEAffine> || std::is_same_v<P, SWAffine>, int> = 0>
inline void to_sponge_bytes(const P &pt, std::vector<uint8_t> &dest) {   // Vec<BaseField>::serialize_compressed
    const auto elems = pt.to_field_elements();
    to_sponge_bytes((uint64_t)elems.size(), dest);
    for (const FpOf &e : elems) to_sponge_bytes(e, dest);
}
template <class A>
inline void to_sponge_bytes(const std::vector<A> &v, std::vector<uint8_t> &dest) {
    for (const A &a : v) to_sponge_bytes(a, dest);   // batch_to_sponge_bytes (u8: extend_from_slice - same bytes)
}
template <class A>
inline void to_sponge_bytes(const std::optional<A> &o, std::vector<uint8_t> &dest) {
    to_sponge_bytes(o.has_value(), dest);
    if (o) to_sponge_bytes(*o, dest);
}
template <class V>
inline void to_sponge_bytes(const WithLength<V> &w, std::vector<uint8_t> &dest) {
    to_sponge_bytes((uint64_t)w.seq.size(), dest);   // usize as u64
    to_sponge_bytes(w.seq, dest);
}

// ---- field elements ---------------------------------------------------------------------------------------------------
inline Fp fp_from_u128(const Field &f, unsigned __int128 v) { return fp_from_bigint(f, {(uint64_t)v, (uint64_t)(v >> 64), 0, 0}); }
inline Fp fp_neg_of_u128(const Field &f, unsigned __int128 mag) {   // -F::from(mag), mag != 0, mag < p
    std::array<uint64_t, 4> m{(uint64_t)mag, (uint64_t)(mag >> 64), 0, 0}, d{};
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned __int128 t = (unsigned __int128)f.modulus[i] - m[i] - borrow;
        d[i] = (uint64_t)t;
        borrow = (t >> 64) & 1;
    }
    return fp_from_bigint(f, d);
}
template <class U, std::enable_if_t<std::is_integral_v<U> && std::is_unsigned_v<U> && !std::is_same_v<U, bool>, int> = 0>
inline void to_sponge_field_elements(const Field &f, U x, std::vector<Fp> &dest) { dest.push_back(fp_from_u128(f, x)); }
inline void to_sponge_field_elements(const Field &f, unsigned __int128 x, std::vector<Fp> &dest) { dest.push_back(fp_from_u128(f, x)); }
template <class I, std::enable_if_t<std::is_integral_v<I> && std::is_signed_v<I>, int> = 0>
inline void to_sponge_field_elements(const Field &f, I x, std::vector<Fp> &dest) {   // src/absorb.rs:190-196
    if (x >= 0) dest.push_back(fp_from_u128(f, (unsigned __int128)x));
    else dest.push_back(fp_neg_of_u128(f, (unsigned __int128)(-(__int128)x)));
}
inline void to_sponge_field_elements(const Field &f, bool x, std::vector<Fp> &dest) { dest.push_back(fp_from_u128(f, x ? 1 : 0)); }
inline void to_sponge_field_elements(const Field &f, const FpOf &e, std::vector<Fp> &dest) {
    if (e.f == f) dest.push_back(e.x);   // field_cast; non-native single elements are dropped (`let _ =`, :157)
}
template <class P, std::enable_if_t<std::is_same_v<P, TEAffine> || std::is_same_v<P, SWAffine>, int> = 0>
inline void to_sponge_field_elements(const Field &f, const P &pt, std::vector<Fp> &dest) {
    for (const FpOf &e : pt.to_field_elements()) {   // field_cast::<BaseField, F>(..).unwrap()
        if (!(e.f == f)) throw Error(PMX_ERR_ARG, "Trying to absorb non-native field elements");
        dest.push_back(e.x);
    }
}
// &[u8]: u64-LE length, then the bytes, packed (src/absorb.rs:135-139 + ark-ff ToConstraintField for [u8])
inline void to_sponge_field_elements(const Field &f, const std::vector<uint8_t> &bytes, std::vector<Fp> &dest) {
    std::vector<uint8_t> all;
    to_sponge_bytes((uint64_t)bytes.size(), all);
    all.insert(all.end(), bytes.begin(), bytes.end());
    const size_t step = (f.modulus_bit_size() - 1) / 8;
    for (size_t off = 0; off < all.size(); off += step) {
        std::array<uint64_t, 4> c{0, 0, 0, 0};
        const size_t n = std::min(step, all.size() - off);
        std::memcpy(c.data(), all.data() + off, n);   // little-endian host
        dest.push_back(fp_from_bigint(f, c));
    }
}
template <class A, std::enable_if_t<!std::is_same_v<A, uint8_t>, int> = 0>
inline void to_sponge_field_elements(const Field &f, const std::vector<A> &v, std::vector<Fp> &dest) {
    if constexpr (std::is_same_v<A, FpOf>) {
        for (const FpOf &e : v) {   // field_cast(batch).unwrap(), :159-164
            if (!(e.f == f)) throw Error(PMX_ERR_ARG, "Trying to absorb non-native field elements");
            dest.push_back(e.x);
        }
    } else {
        for (const A &a : v) to_sponge_field_elements(f, a, dest);
    }
}
template <class A>
inline void to_sponge_field_elements(const Field &f, const std::optional<A> &o, std::vector<Fp> &dest) {
    to_sponge_field_elements(f, o.has_value(), dest);
    if (o) to_sponge_field_elements(f, *o, dest);
}
template <class V>
inline void to_sponge_field_elements(const Field &f, const WithLength<V> &w, std::vector<Fp> &dest) {
    to_sponge_field_elements(f, (uint64_t)w.seq.size(), dest);
    to_sponge_field_elements(f, w.seq, dest);
}

// ---- sponge-side helpers ------------------------------------------------------------------------------------------------
// CryptographicSponge::absorb(&impl Absorb) and the absorb! macro (one absorb call per argument, in order)
template <class... A>
inline void absorb(PoseidonSponge &sponge, const A &...items) {
    auto one = [&](const auto &x) {
        std::vector<Fp> elems;
        to_sponge_field_elements(sponge.parameters.field, x, elems);
        sponge.absorb(elems);
    };
    (one(items), ...);
}
template <class... A>
inline std::vector<uint8_t> collect_sponge_bytes(const A &...items) {
    std::vector<uint8_t> out;
    (to_sponge_bytes(items, out), ...);
    return out;
}
template <class... A>
inline std::vector<Fp> collect_sponge_field_elements(const Field &f, const A &...items) {
    std::vector<Fp> out;
    (to_sponge_field_elements(f, items, out), ...);
    return out;
}
// CryptographicSponge::fork (src/lib.rs:149-157)
inline PoseidonSponge fork(const PoseidonSponge &sponge, const std::vector<uint8_t> &domain) {
    PoseidonSponge new_sponge = sponge;
    std::vector<uint8_t> input;
    to_sponge_bytes((uint64_t)domain.size(), input);
    input.insert(input.end(), domain.begin(), domain.end());
    absorb(new_sponge, input);
    return new_sponge;
}

}  // namespace pmx_host
