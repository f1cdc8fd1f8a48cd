// Internal declarations shared by the translation units of libposeidon_mi355x.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include <exception>
#include <new>

#include "pmx_permute.hpp"

namespace pmx {

// Records a printf-style message for pmx_last_error() (thread-local) and returns `code`.
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Nothing may unwind through the C ABI (include/poseidon_mi355x.h: "nothing throws or aborts across the boundary"; the
// caller is Rust, where a foreign exception is undefined behaviour).  Every extern "C" body that can allocate, take a
// lock or start a thread is written between PMX_ABI_BEGIN / PMX_ABI_END: the body becomes a lambda run under a
// catch-all, so `return code;` inside it keeps its meaning.
template <class F>
int abi_guard(const char *who, F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return set_error(PMX_ERR_HOST, "%s: out of host memory", who);
    } catch (const std::exception &e) {
        return set_error(PMX_ERR_HOST, "%s: %s", who, e.what());
    } catch (...) {
        return set_error(PMX_ERR_HOST, "%s: unknown exception", who);
    }
}
#define PMX_ABI_BEGIN(who) return ::pmx::abi_guard(who, [&]() -> int {
#define PMX_ABI_END });

// Kernel-argument block of a validated config (see pmx_prepare.hpp).  The constant table lives in device
// memory in the internal field form: ark [rounds][t][kFeStride] words, then mds [t][t][kFeStride] words.
struct DevConfig {
    const uint32_t *consts;   // device; layout: pmx_prepare.hpp Prepared::consts
    uint32_t n_const_words;   // words in consts
    uint32_t mds_offset;      // word offsets inside consts
    uint32_t opt_offset, coop_offset;   // ark' of the optimised schedule; the quad engine's table (t = 3)
    uint32_t mfma_offset;     // int8 tables of the dense layers (pmx_mfma.hpp); valid when mfma_dense
    uint32_t mfma_dense;      // 1: those tables exist (t and modulus qualify: pmx_prepare.hpp)
    uint32_t win_offset;      // window tables of the partial section (pmx_mfma.hpp); valid when mfma_dense and the width takes windows
    uint32_t io_offset;       // FieldRt::io block; `field.io` itself holds a HOST address and is re-pointed by the engines
    uint32_t has_opt;         // optimised schedule tables present (and, for t = 3, the cooperative table)
    uint32_t max_lds_bytes;   // LDS one workgroup may ask for on this device (launcher-side engine choice only)
    Rounds rounds;
    FieldRt field;
    Fe one;                   // 2^261 mod p
};

}  // namespace pmx
