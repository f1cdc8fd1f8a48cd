// Internal declarations shared by the translation units of libposeidon_mi355x.so.
#pragma once
#include <cstddef>
#include <cstdint>

namespace pmx {

// Records a printf-style message for pmx_last_error() (thread-local) and returns `code`.
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Kernel-argument view of a validated config.  Constants live in device memory as 32-bit-limb
// Montgomery residues: ark [rounds][t][8] u32, then mds [t][t][8] u32 (same bytes as the 4 x u64 ABI form).
struct DevConfig {
    const uint32_t *consts;   // device: ark followed by mds
    uint32_t n_const_words;   // u32 words in consts (staged into LDS by each workgroup)
    uint32_t rate;
    uint32_t capacity;
    uint32_t half_full;       // full_rounds / 2
    uint32_t partial_rounds;
    uint32_t total_rounds;
    uint32_t alpha_lo, alpha_hi;
    uint32_t p[8];            // modulus, 32-bit limbs
    uint32_t inv32;           // -p^-1 mod 2^32
    uint32_t one[8];          // 2^256 mod p
};

}  // namespace pmx
