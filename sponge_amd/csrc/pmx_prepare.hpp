// Host-side preparation of a validated config for the kernels: modulus view, conversion constants and the
// round constants / MDS matrix rewritten into the internal field form (29-bit limbs, x * 2^261 mod p).
// Shared by pmx_api.cpp (which uploads the table) and the CPU-side algorithm check in tests/hostcheck.
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "../../include/poseidon_mi355x.h"
#include "pmx_field.hpp"
#include "pmx_host_field.hpp"
#include "pmx_permute.hpp"

namespace pmx {

inline void to_limbs29(const U256 &x, uint32_t out[kN]) {
    for (int i = 0; i < kN; ++i) {
        const int bit = kW * i, wi = bit / 64, sh = bit % 64;
        unsigned __int128 pair = x.l[wi];
        if (wi + 1 < 4) pair |= (unsigned __int128)x.l[wi + 1] << 64;
        out[i] = (uint32_t)(pair >> sh) & kMask;
    }
}

inline U256 times_pow2(const HostField &f, U256 v, int k) {  // v * 2^k mod p
    for (int i = 0; i < k; ++i) v = f.add(v, v);
    return v;
}

struct Prepared {
    HostField hf;
    FieldRt f;
    Fe one;                         // 2^261 mod p
    Rounds c;
    uint32_t t;
    std::vector<uint32_t> consts;   // ark [rounds][t][kFeStride] | mds [t][t][kFeStride]
    size_t mds_offset;              // in words
};

// Returns PMX_OK or an error code with a message in `err`.
inline int prepare(const pmx_config *cfg, Prepared &out, std::string &err) {
    const uint64_t t64 = (uint64_t)cfg->rate + cfg->capacity;
    if (cfg->rate == 0) { err = "rate must be >= 1"; return PMX_ERR_CONFIG; }
    if (t64 > PMX_MAX_WIDTH) { err = "width exceeds PMX_MAX_WIDTH"; return PMX_ERR_UNSUPPORTED; }
    if (cfg->full_rounds % 2) { err = "full_rounds must be even (RF/2 rounds on each side, mod.rs:96)"; return PMX_ERR_CONFIG; }
    const uint64_t rounds = (uint64_t)cfg->full_rounds + cfg->partial_rounds;
    if (rounds == 0 || rounds > 4096) { err = "round count out of range"; return PMX_ERR_CONFIG; }
    HostField &hf = out.hf;
    if (!hf.init(cfg->modulus)) { err = "modulus must be odd and > 2"; return PMX_ERR_CONFIG; }
    // the unsaturated 9 x 29-bit arithmetic needs 6 spare bits below 2^261 (pmx_field.hpp)
    if (hf.bits() > 255) { err = "modulus must be < 2^255 (BLS12-381 Fr and BN254 Fr are 255 and 254 bits)"; return PMX_ERR_UNSUPPORTED; }
    if (hf.bits() < 225) { err = "modulus must be at least 225 bits"; return PMX_ERR_UNSUPPORTED; }
    const uint32_t t = (uint32_t)t64;
    const size_t n_ark = (size_t)rounds * t, n_mds = (size_t)t * t;
    out.t = t;
    out.consts.assign((n_ark + n_mds) * kFeStride, 0u);
    out.mds_offset = n_ark * kFeStride;
    for (size_t k = 0; k < n_ark + n_mds; ++k) {
        const uint64_t *src = k < n_ark ? cfg->ark + 4 * k : cfg->mds + 4 * (k - n_ark);
        U256 v;
        std::memcpy(v.l, src, sizeof v.l);
        if (u256_geq(v, hf.p)) {  // every constant must be a reduced residue (ark-ff's invariant)
            err = std::string(k < n_ark ? "ark" : "mds") + " constant " + std::to_string(k < n_ark ? k : k - n_ark) + " is not reduced";
            return PMX_ERR_CONFIG;
        }
        to_limbs29(times_pow2(hf, v, 5), &out.consts[k * kFeStride]);   // x*2^256 -> x*2^261
    }
    FieldRt &f = out.f;
    to_limbs29(hf.p, f.p);
    f.pinv = (uint32_t)hf.inv & kMask;
    std::memcpy(f.p32, hf.p.l, 32);
    to_limbs29(times_pow2(hf, hf.r, 10), f.to_int.l);   // 2^266 mod p
    to_limbs29(hf.r, f.to_abi.l);                       // 2^256 mod p
    to_limbs29(times_pow2(hf, hf.r, 5), out.one.l);     // 2^261 mod p
    out.c.rate = cfg->rate;
    out.c.capacity = cfg->capacity;
    out.c.half_full = cfg->full_rounds / 2;
    out.c.partial_rounds = cfg->partial_rounds;
    out.c.total_rounds = (uint32_t)rounds;
    out.c.alpha = cfg->alpha;
    return PMX_OK;
}

}  // namespace pmx
