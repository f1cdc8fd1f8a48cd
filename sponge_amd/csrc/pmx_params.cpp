// Host-only parameter generation behind the C ABI:
//   pmx_find_poseidon_ark_and_mds   <- find_poseidon_ark_and_mds   (reference src/poseidon/traits.rs:105-146)
//                                      PoseidonGrainLFSR            (reference src/poseidon/grain_lfsr.rs:15-189)
//   pmx_mont_constants / pmx_to_mont / pmx_from_mont  <- ark-ff MontConfig constants, from_bigint / into_bigint
// The Grain register is held as one 80-bit shift register in a 128-bit integer (bit i = b_i, b_0 the
// oldest bit): one clock is state = (state >> 1) | (feedback << 79).
#include "../../include/poseidon_mi355x.h"
#include "pmx_host_field.hpp"
#include "pmx_internal.hpp"

#include <vector>

namespace pmx {

class GrainLfsr {
public:
    GrainLfsr(bool sbox_is_inverse, uint64_t prime_bits, uint64_t width, uint64_t rf, uint64_t rp)
        : nbits_(prime_bits), s_(0) {
        put_field(1, 1, 1);                         // b0 b1 = 01: prime field            grain_lfsr.rs:25
        put_field(5, 5, sbox_is_inverse ? 1 : 0);   // b2..b5: S-box kind                 grain_lfsr.rs:28-32
        put_field(6, 17, prime_bits);               // n, big-endian                      grain_lfsr.rs:35-41
        put_field(18, 29, width);                   // t                                  grain_lfsr.rs:44-50
        put_field(30, 39, rf);                      // R_F                                grain_lfsr.rs:53-59
        put_field(40, 49, rp);                      // R_P                                grain_lfsr.rs:62-68
        put_field(50, 79, 0x3FFFFFFFu);             // thirty ones                        grain_lfsr.rs:71-73
        for (int i = 0; i < 160; ++i) clock();      // discard 160 bits                   grain_lfsr.rs:176-188
    }

    // n-bit integer, first generated bit most significant (grain_lfsr.rs:119-123, 141-153); n <= 256
    U256 next_integer() {
        U256 v = {{0, 0, 0, 0}};
        for (uint64_t k = 0; k < nbits_; ++k) {
            const unsigned pos = (unsigned)(nbits_ - 1 - k);
            if (next_bit()) v.l[pos / 64] |= (uint64_t)1 << (pos % 64);
        }
        return v;
    }

private:
    uint64_t nbits_;
    u128 s_;

    void put_field(unsigned first, unsigned last, uint64_t value) {  // value big-endian into b_first..b_last
        for (unsigned pos = last + 1; pos-- > first;) {
            if (value & 1) s_ |= (u128)1 << pos;
            value >>= 1;
        }
    }

    unsigned clock() {  // taps 62,51,38,23,13,0 relative to the oldest bit (grain_lfsr.rs:163-168)
        const unsigned fb = (unsigned)(((s_ >> 62) ^ (s_ >> 51) ^ (s_ >> 38) ^ (s_ >> 23) ^ (s_ >> 13) ^ s_) & 1);
        s_ = (s_ >> 1) | ((u128)fb << 79);
        return fb;
    }

    unsigned next_bit() {  // self-shrinking output: keep the 2nd bit of a pair whose 1st bit is 1 (:89-103)
        for (;;) {
            const unsigned first = clock();
            const unsigned second = clock();
            if (first) return second;
        }
    }
};

}  // namespace pmx

using namespace pmx;

extern "C" int pmx_mont_constants(const uint64_t modulus[PMX_LIMBS], uint64_t *inv, uint64_t r[PMX_LIMBS],
                                  uint64_t r2[PMX_LIMBS]) {
    if (!modulus) return set_error(PMX_ERR_ARG, "pmx_mont_constants: null modulus");
    HostField f;
    if (!f.init(modulus)) return set_error(PMX_ERR_CONFIG, "modulus must be odd and > 2");
    if (inv) *inv = f.inv;
    if (r) std::memcpy(r, f.r.l, sizeof f.r.l);
    if (r2) std::memcpy(r2, f.r2.l, sizeof f.r2.l);
    return PMX_OK;
}

static int convert(const uint64_t modulus[PMX_LIMBS], uint64_t *elems, size_t n, bool to) {
    if (!modulus || (!elems && n)) return set_error(PMX_ERR_ARG, "pmx_to/from_mont: null pointer");
    HostField f;
    if (!f.init(modulus)) return set_error(PMX_ERR_CONFIG, "modulus must be odd and > 2");
    for (size_t i = 0; i < n; ++i) {
        U256 v;
        std::memcpy(v.l, elems + 4 * i, sizeof v.l);
        if (u256_geq(v, f.p)) return set_error(PMX_ERR_ARG, "element %zu is not reduced (>= modulus)", i);
        v = to ? f.to_mont(v) : f.from_mont(v);
        std::memcpy(elems + 4 * i, v.l, sizeof v.l);
    }
    return PMX_OK;
}

extern "C" int pmx_to_mont(const uint64_t modulus[PMX_LIMBS], uint64_t *elems, size_t n) {
    return convert(modulus, elems, n, true);
}

extern "C" int pmx_from_mont(const uint64_t modulus[PMX_LIMBS], uint64_t *elems, size_t n) {
    return convert(modulus, elems, n, false);
}

extern "C" int pmx_find_poseidon_ark_and_mds(const uint64_t modulus[PMX_LIMBS], uint64_t prime_bits, uint32_t rate,
                                             uint32_t full_rounds, uint32_t partial_rounds, uint32_t skip_matrices,
                                             uint64_t *ark_out, uint64_t *mds_out) {
    if (!modulus || !ark_out || !mds_out) return set_error(PMX_ERR_ARG, "pmx_find_poseidon_ark_and_mds: null pointer");
    HostField f;
    if (!f.init(modulus)) return set_error(PMX_ERR_CONFIG, "modulus must be odd and > 2");
    // assert_eq!(F::MODULUS_BIT_SIZE, prime_num_bits)  grain_lfsr.rs:112,136
    if (prime_bits != f.bits()) return set_error(PMX_ERR_CONFIG, "prime_bits %llu != bit size of modulus %u",
                                                 (unsigned long long)prime_bits, f.bits());
    // the seed fields are 12/10 bits wide (grain_lfsr.rs:34-68)
    if (rate == 0 || rate + 1 >= (1u << 12) || full_rounds >= (1u << 10) || partial_rounds >= (1u << 10))
        return set_error(PMX_ERR_CONFIG, "rate/rounds do not fit the Grain seed fields");
    const uint32_t t = rate + 1;
    GrainLfsr lfsr(false, prime_bits, t, full_rounds, partial_rounds);

    // round constants: rejection sampling of n-bit integers (grain_lfsr.rs:108-133, traits.rs:120-123)
    const size_t n_ark = (size_t)(full_rounds + partial_rounds) * t;
    for (size_t k = 0; k < n_ark; ++k) {
        U256 v;
        do {
            v = lfsr.next_integer();
        } while (u256_geq(v, f.p));
        v = f.to_mont(v);
        std::memcpy(ark_out + 4 * k, v.l, sizeof v.l);
    }

    // n-bit integers reduced mod p (grain_lfsr.rs:135-159): v < 2^n <= 2p, one conditional subtraction
    auto next_mod_p = [&]() {
        U256 v = lfsr.next_integer();
        if (u256_geq(v, f.p)) u256_sub(v, v, f.p);
        return f.to_mont(v);
    };
    for (uint32_t s = 0; s < skip_matrices; ++s)              // traits.rs:127-129
        for (uint32_t k = 0; k < 2 * t; ++k) (void)next_mod_p();
    std::vector<U256> xs(t), ys(t);
    for (uint32_t i = 0; i < t; ++i) xs[i] = next_mod_p();    // traits.rs:136
    for (uint32_t i = 0; i < t; ++i) ys[i] = next_mod_p();    // traits.rs:137
    for (uint32_t i = 0; i < t; ++i) {
        for (uint32_t j = 0; j < t; ++j) {                    // Cauchy matrix, traits.rs:139-143
            const U256 sum = f.add(xs[i], ys[j]);
            if (u256_is_zero(sum))  // the reference's `.inverse().unwrap()` panics here
                return set_error(PMX_ERR_CONFIG, "xs[%u] + ys[%u] == 0: matrix not invertible, raise skip_matrices", i, j);
            const U256 m = f.inverse(sum);
            std::memcpy(mds_out + 4 * ((size_t)i * t + j), m.l, sizeof m.l);
        }
    }
    return PMX_OK;
}
