// C++ host-side mirror of the reference's Poseidon sponge interface over the C ABI (include/poseidon_mi355x.h).
// The reference is Rust (no toolchain in this image), so the host layer above the ABI is C++; names, argument
// meaning and failure behaviour follow the reference (file:line in arkworks-rs/sponge):
//   DuplexSpongeMode                        src/lib.rs:198-210
//   PoseidonConfig / PoseidonConfig::new    src/poseidon/mod.rs:23-42, 185-214      (asserts -> pmx_host::Error)
//   PoseidonSponge {parameters,state,mode}  src/poseidon/mod.rs:51-60
//     new / absorb / squeeze_bytes / squeeze_bits / squeeze_field_elements      :216-318
//     squeeze_native_field_elements                                            :320-342
//     into_state / from_state (SpongeExt)                                      :344-367, src/lib.rs:188-195
//   find_poseidon_ark_and_mds               src/poseidon/traits.rs:105-146
//   get_default_poseidon_parameters         src/poseidon/traits.rs:59-102, table src/test.rs:13-32
// Every permutation runs on the GPU; a single PoseidonSponge is the n = 1 case of the batch driver.
// Where the reference panics (assert_eq!, unwrap) this layer throws pmx_host::Error.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/poseidon_mi355x.h"

namespace pmx_host {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error("pmx error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) {
    if (rc != PMX_OK) throw Error(rc, pmx_last_error());
}

// A prime field with 4-limb elements (ark-ff Fp256).  Elements are Montgomery residues, R = 2^256.
struct Field {
    std::array<uint64_t, 4> modulus;
    unsigned modulus_bit_size() const {   // PrimeField::MODULUS_BIT_SIZE
        for (int i = 3; i >= 0; --i)
            if (modulus[i]) return 64u * i + (64u - (unsigned)__builtin_clzll(modulus[i]));
        return 0;
    }
    bool operator==(const Field &o) const { return modulus == o.modulus; }
    static Field bls12_381_fr() {   // src/test.rs:6
        return {{0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}};
    }
    static Field bn254_fr() {
        return {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
    }
};

struct Fp {
    std::array<uint64_t, 4> l{};   // x * 2^256 mod p, little-endian limbs (ark-ff in-memory form)
    bool operator==(const Fp &o) const { return l == o.l; }
    bool operator!=(const Fp &o) const { return !(*this == o); }
};
static_assert(sizeof(Fp) == 32, "Fp must be 4 x u64 like ark-ff's Fp256");

inline Fp fp_from_bigint(const Field &f, const std::array<uint64_t, 4> &canonical) {   // F::from_bigint
    Fp x;
    x.l = canonical;
    check(pmx_to_mont(f.modulus.data(), x.l.data(), 1));
    return x;
}
inline Fp fp_from_u64(const Field &f, uint64_t v) { return fp_from_bigint(f, {v, 0, 0, 0}); }   // Fr::from(v)
inline std::array<uint64_t, 4> fp_into_bigint(const Field &f, const Fp &x) {                    // into_bigint
    std::array<uint64_t, 4> c = x.l;
    check(pmx_from_mont(f.modulus.data(), c.data(), 1));
    return c;
}
// decimal string (the argument of MontFp!) -> element; digits only
inline Fp fp_from_decimal(const Field &f, const std::string &dec) {
    std::array<uint64_t, 4> acc{0, 0, 0, 0};
    for (char ch : dec) {
        if (ch < '0' || ch > '9') throw Error(PMX_ERR_ARG, "not a decimal digit");
        unsigned __int128 carry = (unsigned)(ch - '0');
        for (int i = 0; i < 4; ++i) {
            unsigned __int128 v = (unsigned __int128)acc[i] * 10 + carry;
            acc[i] = (uint64_t)v;
            carry = v >> 64;
        }
        if (carry) throw Error(PMX_ERR_ARG, "decimal literal exceeds 256 bits");
    }
    return fp_from_bigint(f, acc);
}

// std::allocator-compatible allocator for page-locked host memory (pmx_host_alloc): with
// std::vector<Fp, PinnedAllocator<Fp>> the host entry points pipeline the PCIe copies with the kernel.
template <class T>
struct PinnedAllocator {
    using value_type = T;
    PinnedAllocator() = default;
    template <class U>
    PinnedAllocator(const PinnedAllocator<U> &) {}
    T *allocate(size_t n) {
        void *p = nullptr;
        check(pmx_host_alloc(&p, n * sizeof(T)));
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t) noexcept { pmx_host_free(p); }
    template <class U>
    bool operator==(const PinnedAllocator<U> &) const { return true; }
    template <class U>
    bool operator!=(const PinnedAllocator<U> &) const { return false; }
};

struct DuplexSpongeMode {
    enum Tag : uint32_t { kAbsorbing = PMX_MODE_ABSORBING, kSqueezing = PMX_MODE_SQUEEZING } tag;
    size_t index;   // next_absorb_index / next_squeeze_index
    static DuplexSpongeMode Absorbing(size_t next_absorb_index) { return {kAbsorbing, next_absorb_index}; }
    static DuplexSpongeMode Squeezing(size_t next_squeeze_index) { return {kSqueezing, next_squeeze_index}; }
    bool operator==(const DuplexSpongeMode &o) const { return tag == o.tag && index == o.index; }
};

// pmx_ctx holder: one validated config resident on one GPU, taken from the library's process-wide cache
// (pmx_ctx_acquire): sponges made from equal configs - each PoseidonSponge::make copies its PoseidonConfig, like
// CryptographicSponge::new clones the parameters (src/poseidon/mod.rs:219-230) - share one set of device tables.
class Context {
public:
    Context(const pmx_config &c, int device) { check(pmx_ctx_acquire(&c, device, &h_)); }
    ~Context() { pmx_ctx_release(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    pmx_ctx *get() const { return h_; }

private:
    pmx_ctx *h_ = nullptr;
};

struct PoseidonConfig {
    Field field;
    size_t full_rounds, partial_rounds;
    uint64_t alpha;
    std::vector<std::vector<Fp>> ark;   // ark[round][i]
    std::vector<std::vector<Fp>> mds;   // mds[i][j]
    size_t rate, capacity;

    // PoseidonConfig::new (argument order of src/poseidon/mod.rs:187-195) with its asserts (:196-203)
    static PoseidonConfig make(const Field &field, size_t full_rounds, size_t partial_rounds, uint64_t alpha,
                               std::vector<std::vector<Fp>> mds, std::vector<std::vector<Fp>> ark, size_t rate,
                               size_t capacity) {
        if (ark.size() != full_rounds + partial_rounds) throw Error(PMX_ERR_CONFIG, "ark.len() != full_rounds + partial_rounds");
        for (auto &row : ark)
            if (row.size() != rate + capacity) throw Error(PMX_ERR_CONFIG, "ark row length != rate + capacity");
        if (mds.size() != rate + capacity) throw Error(PMX_ERR_CONFIG, "mds.len() != rate + capacity");
        for (auto &row : mds)
            if (row.size() != rate + capacity) throw Error(PMX_ERR_CONFIG, "mds row length != rate + capacity");
        return PoseidonConfig{field, full_rounds, partial_rounds, alpha, std::move(ark), std::move(mds), rate, capacity, {}};
    }

    size_t width() const { return rate + capacity; }

    // the device context of this config (created on first use)
    std::shared_ptr<Context> context(int device = 0) const {
        if (!ctx_ || ctx_device_ != device) {
            std::vector<uint64_t> a, m;
            for (auto &row : ark) for (auto &x : row) a.insert(a.end(), x.l.begin(), x.l.end());
            for (auto &row : mds) for (auto &x : row) m.insert(m.end(), x.l.begin(), x.l.end());
            pmx_config c{};
            c.full_rounds = (uint32_t)full_rounds;
            c.partial_rounds = (uint32_t)partial_rounds;
            c.alpha = alpha;
            c.rate = (uint32_t)rate;
            c.capacity = (uint32_t)capacity;
            std::memcpy(c.modulus, field.modulus.data(), 32);
            c.ark = a.data();
            c.mds = m.data();
            ctx_ = std::make_shared<Context>(c, device);
            ctx_device_ = device;
        }
        return ctx_;
    }

    mutable std::shared_ptr<Context> ctx_;
    mutable int ctx_device_ = -1;
};

// find_poseidon_ark_and_mds (traits.rs:105-146): {ark, mds}
inline std::pair<std::vector<std::vector<Fp>>, std::vector<std::vector<Fp>>> find_poseidon_ark_and_mds(
    const Field &f, uint64_t prime_bits, size_t rate, uint64_t full_rounds, uint64_t partial_rounds, uint64_t skip_matrices) {
    const size_t t = rate + 1, rounds = full_rounds + partial_rounds;
    std::vector<uint64_t> a(rounds * t * 4), m(t * t * 4);
    check(pmx_find_poseidon_ark_and_mds(f.modulus.data(), prime_bits, (uint32_t)rate, (uint32_t)full_rounds,
                                        (uint32_t)partial_rounds, (uint32_t)skip_matrices, a.data(), m.data()));
    std::vector<std::vector<Fp>> ark(rounds, std::vector<Fp>(t)), mds(t, std::vector<Fp>(t));
    for (size_t r = 0; r < rounds; ++r) for (size_t i = 0; i < t; ++i) std::memcpy(ark[r][i].l.data(), &a[(r * t + i) * 4], 32);
    for (size_t i = 0; i < t; ++i) for (size_t j = 0; j < t; ++j) std::memcpy(mds[i][j].l.data(), &m[(i * t + j) * 4], 32);
    return {ark, mds};
}

// PoseidonDefaultConfigEntry table of the reference's test field (src/test.rs:13-32): rate, alpha, RF, RP, skip
struct PoseidonDefaultConfigEntry { size_t rate; uint64_t alpha; size_t full_rounds, partial_rounds, skip_matrices; };
inline const PoseidonDefaultConfigEntry *bls12_381_default_table(bool optimized_for_weights) {
    static const PoseidonDefaultConfigEntry constraints[7] = {{2, 17, 8, 31, 0}, {3, 5, 8, 56, 0}, {4, 5, 8, 56, 0}, {5, 5, 8, 57, 0},
                                                             {6, 5, 8, 57, 0}, {7, 5, 8, 57, 0}, {8, 5, 8, 57, 0}};
    static const PoseidonDefaultConfigEntry weights[7] = {{2, 257, 8, 13, 0}, {3, 257, 8, 13, 0}, {4, 257, 8, 13, 0}, {5, 257, 8, 13, 0},
                                                         {6, 257, 8, 13, 0}, {7, 257, 8, 13, 0}, {8, 257, 8, 13, 0}};
    return optimized_for_weights ? weights : constraints;
}

// PoseidonDefaultConfigField::get_default_poseidon_parameters (traits.rs:69-102); only BLS12-381 Fr has a table
inline std::optional<PoseidonConfig> get_default_poseidon_parameters(const Field &f, size_t rate, bool optimized_for_weights) {
    if (!(f == Field::bls12_381_fr())) return std::nullopt;
    const PoseidonDefaultConfigEntry *tab = bls12_381_default_table(optimized_for_weights);
    for (int k = 0; k < 7; ++k) {
        if (tab[k].rate != rate) continue;
        auto am = find_poseidon_ark_and_mds(f, f.modulus_bit_size(), rate, tab[k].full_rounds, tab[k].partial_rounds, tab[k].skip_matrices);
        return PoseidonConfig::make(f, tab[k].full_rounds, tab[k].partial_rounds, tab[k].alpha, am.second, am.first, rate, 1);
    }
    return std::nullopt;
}

// FieldElementSize (src/lib.rs:34-58): Full, or Truncated(num_bits)
struct FieldElementSize {
    bool truncated = false;
    size_t bits = 0;
    static FieldElementSize Full() { return {}; }
    static FieldElementSize Truncated(size_t num_bits) { return {true, num_bits}; }
    // num_bits::<F>() (src/lib.rs:45-52): panics when the request exceeds the field, and otherwise ALWAYS answers
    // MODULUS_BIT_SIZE - 1 - the Truncated value does not shorten the element in the reference
    size_t num_bits(const Field &f) const {
        if (truncated && bits > f.modulus_bit_size()) throw Error(PMX_ERR_ARG, "num_bits is greater than the capacity of the field.");
        return f.modulus_bit_size() - 1;
    }
    bool operator==(const FieldElementSize &o) const { return truncated == o.truncated && (!truncated || bits == o.bits); }
};

struct PoseidonSpongeState {   // src/poseidon/mod.rs:346-349
    std::vector<Fp> state;
    DuplexSpongeMode mode;
};

// n independent sponges advanced together (the batched form of the trait surface)
class BatchPoseidonSponge {
public:
    PoseidonConfig parameters;
    std::vector<Fp> state;             // [n][t]
    std::vector<uint32_t> mode_tag;    // [n]
    std::vector<uint32_t> mode_index;  // [n]

    static BatchPoseidonSponge make(const PoseidonConfig &params, size_t n, int device = 0) {   // n x CryptographicSponge::new
        BatchPoseidonSponge s{params, std::vector<Fp>(n * params.width()), std::vector<uint32_t>(n, PMX_MODE_ABSORBING),
                              std::vector<uint32_t>(n, 0), n, device};
        return s;
    }
    size_t size() const { return n_; }

    // every sponge absorbs its own L elements: input [n][L]
    void absorb(const std::vector<Fp> &input) {
        if (n_ == 0 || input.empty()) return;   // mod.rs:234-236
        const size_t L = input.size() / n_;
        check(pmx_sponge_absorb_batch(parameters.context(device_)->get(), state[0].l.data(), mode_tag.data(), mode_index.data(),
                                      input[0].l.data(), L, n_));
    }
    std::vector<Fp> squeeze_native_field_elements(size_t num_elements) {   // [n][num_elements]
        std::vector<Fp> out(n_ * num_elements);
        if (n_ == 0) return out;
        Fp dummy;
        check(pmx_sponge_squeeze_batch(parameters.context(device_)->get(), state[0].l.data(), mode_tag.data(), mode_index.data(),
                                       out.empty() ? dummy.l.data() : out[0].l.data(), num_elements, n_));
        return out;
    }

private:
    BatchPoseidonSponge(PoseidonConfig p, std::vector<Fp> st, std::vector<uint32_t> tag, std::vector<uint32_t> idx, size_t n, int dev)
        : parameters(std::move(p)), state(std::move(st)), mode_tag(std::move(tag)), mode_index(std::move(idx)), n_(n), device_(dev) {}
    size_t n_;
    int device_;
};

class PoseidonSponge {
public:
    PoseidonConfig parameters;
    std::vector<Fp> state;
    DuplexSpongeMode mode;

    static PoseidonSponge make(const PoseidonConfig &params, int device = 0) {   // CryptographicSponge::new, mod.rs:219-230
        return PoseidonSponge{params, std::vector<Fp>(params.width()), DuplexSpongeMode::Absorbing(0), device};
    }

    void absorb(const std::vector<Fp> &input) {   // native field elements (mod.rs:232-254)
        if (input.empty()) return;
        uint32_t tag = mode.tag, idx = (uint32_t)mode.index;
        check(pmx_sponge_absorb_batch(parameters.context(device_)->get(), state[0].l.data(), &tag, &idx, input[0].l.data(), input.size(), 1));
        mode = {(DuplexSpongeMode::Tag)tag, idx};
    }
    std::vector<Fp> squeeze_native_field_elements(size_t num_elements) {   // mod.rs:321-341
        std::vector<Fp> out(num_elements);
        uint32_t tag = mode.tag, idx = (uint32_t)mode.index;
        Fp dummy;
        check(pmx_sponge_squeeze_batch(parameters.context(device_)->get(), state[0].l.data(), &tag, &idx,
                                       out.empty() ? dummy.l.data() : out[0].l.data(), num_elements, 1));
        mode = {(DuplexSpongeMode::Tag)tag, idx};
        return out;
    }
    // native-field case of squeeze_field_elements::<F> (mod.rs:306-311)
    std::vector<Fp> squeeze_field_elements(size_t num_elements) { return squeeze_native_field_elements(num_elements); }

    std::vector<uint8_t> squeeze_bytes(size_t num_bytes) {   // mod.rs:256-270
        const size_t usable = (parameters.field.modulus_bit_size() - 1) / 8;
        const size_t n = (num_bytes + usable - 1) / usable;
        std::vector<uint8_t> bytes;
        for (const Fp &e : squeeze_native_field_elements(n)) {
            const auto c = fp_into_bigint(parameters.field, e);
            const uint8_t *b = reinterpret_cast<const uint8_t *>(c.data());   // little-endian host
            bytes.insert(bytes.end(), b, b + usable);
        }
        bytes.resize(num_bytes);
        return bytes;
    }
    std::vector<bool> squeeze_bits(size_t num_bits) {   // mod.rs:272-286
        const size_t usable = parameters.field.modulus_bit_size() - 1;
        const size_t n = (num_bits + usable - 1) / usable;
        std::vector<bool> bits;
        for (const Fp &e : squeeze_native_field_elements(n)) {
            const auto c = fp_into_bigint(parameters.field, e);
            for (size_t k = 0; k < usable; ++k) bits.push_back((c[k / 64] >> (k % 64)) & 1);
        }
        bits.resize(num_bits);
        return bits;
    }

    // squeeze_field_elements_with_sizes::<F2> (mod.rs:288-304).  Same characteristic: the native path below.
    // Otherwise the default of src/lib.rs:61-100: one squeeze_bits call for all elements, num_bits::<F2>() bits each,
    // little-endian, through from_le_bytes_mod_order (the value is < 2^(bits(p2)-1) <= p2, so nothing is reduced).
    // Elements come back as residues of `f2`.
    std::vector<Fp> squeeze_field_elements_with_sizes(const std::vector<FieldElementSize> &sizes, const Field &f2) {
        if (f2 == parameters.field) return squeeze_native_field_elements_with_sizes(sizes);
        return squeeze_with_sizes_default(sizes, f2);
    }
    // squeeze_field_elements::<F2> (mod.rs:306-317)
    std::vector<Fp> squeeze_field_elements(size_t num_elements, const Field &f2) {
        if (f2 == parameters.field) return squeeze_native_field_elements(num_elements);
        return squeeze_field_elements_with_sizes(std::vector<FieldElementSize>(num_elements), f2);
    }
    // FieldBasedCryptographicSponge::squeeze_native_field_elements_with_sizes (src/lib.rs:166-182)
    std::vector<Fp> squeeze_native_field_elements_with_sizes(const std::vector<FieldElementSize> &sizes) {
        bool all_full = true;
        for (const auto &sz : sizes) all_full = all_full && !sz.truncated;
        if (all_full) return squeeze_native_field_elements(sizes.size());
        return squeeze_with_sizes_default(sizes, parameters.field);
    }

    // SpongeExt (src/lib.rs:188-195, mod.rs:351-367)
    PoseidonSpongeState into_state() && { return {std::move(state), mode}; }
    static PoseidonSponge from_state(PoseidonSpongeState st, const PoseidonConfig &params, int device = 0) {
        PoseidonSponge s = make(params, device);
        s.mode = st.mode;
        s.state = std::move(st.state);
        return s;
    }

private:
    std::vector<Fp> squeeze_with_sizes_default(const std::vector<FieldElementSize> &sizes, const Field &f2) {
        std::vector<Fp> out;
        if (sizes.empty()) return out;   // src/lib.rs:65-67: no squeeze at all
        size_t total = 0;
        for (const auto &sz : sizes) total += sz.num_bits(f2);
        const std::vector<bool> bits = squeeze_bits(total);
        size_t at = 0;
        for (const auto &sz : sizes) {
            const size_t nb = sz.num_bits(f2);
            std::array<uint64_t, 4> v{0, 0, 0, 0};
            for (size_t k = 0; k < nb; ++k)
                if (bits[at + k]) v[k / 64] |= 1ull << (k % 64);
            at += nb;
            out.push_back(fp_from_bigint(f2, v));
        }
        return out;
    }
    PoseidonSponge(PoseidonConfig p, std::vector<Fp> st, DuplexSpongeMode m, int dev)
        : parameters(std::move(p)), state(std::move(st)), mode(m), device_(dev) {}
    int device_;
};

}  // namespace pmx_host
