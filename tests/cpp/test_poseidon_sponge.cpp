// The reference's own tests for this path, re-expressed against the C++ host mirror
// (sponge_amd/host/poseidon_sponge.hpp).  `--host-only` runs the parts that need no GPU.
//   test_grain_lfsr_consistency                      src/poseidon/grain_lfsr.rs:197-213
//   bls12_381_fr_poseidon_default_parameters_test    src/poseidon/traits.rs:163-358 (rates 2, 3, 8)
//   test_poseidon_sponge_consistency                 src/poseidon/mod.rs:376-399        (GPU)
//   test_squeeze_cast_native                         src/poseidon/tests.rs:71-85         (GPU)
#include <cstdio>
#include <cstring>
#include <string>

#include "../../sponge_amd/host/poseidon_sponge.hpp"

using namespace pmx_host;

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static void test_grain_lfsr_and_default_parameters() {
    const Field Fr = Field::bls12_381_fr();
    auto c2 = get_default_poseidon_parameters(Fr, 2, false).value();
    EXPECT(c2.ark[0][0] == fp_from_decimal(Fr, "27117311055620256798560880810000042840428971800021819916023577129547249660720"));
    EXPECT(c2.ark[0][1] == fp_from_decimal(Fr, "51641662388546346858987925410984003801092143452466182801674685248597955169158"));
    EXPECT(c2.mds[0][0] == fp_from_decimal(Fr, "26017457457808754696901916760153646963713419596921330311675236858336250747575"));
    EXPECT(c2.alpha == 17 && c2.full_rounds == 8 && c2.partial_rounds == 31 && c2.capacity == 1);
    auto c3 = get_default_poseidon_parameters(Fr, 3, false).value();
    EXPECT(c3.ark[0][0] == fp_from_decimal(Fr, "11865901593870436687704696210307853465124332568266803587887584059192277437537"));
    EXPECT(c3.mds[0][0] == fp_from_decimal(Fr, "18791275321793747281053101601584820964683215017313972132092847596434094368732"));
    auto w8 = get_default_poseidon_parameters(Fr, 8, true).value();
    EXPECT(w8.ark[0][0] == fp_from_decimal(Fr, "16478680729975035007348178961232525927424769683353433314299437589237598655079"));
    EXPECT(w8.mds[0][0] == fp_from_decimal(Fr, "39160448583049384229582837387246752222769278402304070376350288593586064961857"));
    EXPECT(!get_default_poseidon_parameters(Fr, 9, false).has_value());
    // PoseidonConfig::new asserts (mod.rs:196-203)
    bool threw = false;
    try {
        auto bad_ark = c2.ark;
        bad_ark.pop_back();
        (void)PoseidonConfig::make(Fr, 8, 31, 17, c2.mds, bad_ark, 2, 1);
    } catch (const Error &e) { threw = e.code == PMX_ERR_CONFIG; }
    EXPECT(threw);
}

static void test_poseidon_sponge_consistency() {
    const Field Fr = Field::bls12_381_fr();
    auto sponge_param = get_default_poseidon_parameters(Fr, 2, false).value();
    auto sponge = PoseidonSponge::make(sponge_param);
    sponge.absorb({fp_from_u64(Fr, 0), fp_from_u64(Fr, 1), fp_from_u64(Fr, 2)});
    auto res = sponge.squeeze_native_field_elements(3);
    EXPECT(res[0] == fp_from_decimal(Fr, "40442793463571304028337753002242186710310163897048962278675457993207843616876"));
    EXPECT(res[1] == fp_from_decimal(Fr, "2664374461699898000291153145224099287711224021716202960480903840045233645301"));
    EXPECT(res[2] == fp_from_decimal(Fr, "50191078828066923662070228256530692951801504043422844038937334196346054068797"));
    EXPECT(sponge.mode == DuplexSpongeMode::Squeezing(1));
}

static void test_squeeze_cast_native_and_state_roundtrip() {
    const Field Fr = Field::bls12_381_fr();
    auto sponge_param = get_default_poseidon_parameters(Fr, 2, false).value();
    auto sponge1 = PoseidonSponge::make(sponge_param);
    sponge1.absorb({fp_from_u64(Fr, 114514)});
    auto sponge2 = sponge1;   // Clone
    EXPECT(sponge1.squeeze_native_field_elements(5) == sponge2.squeeze_field_elements(5));
    auto sponge3 = PoseidonSponge::from_state(std::move(sponge2).into_state(), sponge_param);
    auto bytes = sponge3.squeeze_bytes(40);
    auto elems = sponge1.squeeze_native_field_elements(2);
    const auto c0 = fp_into_bigint(Fr, elems[0]);
    EXPECT(bytes.size() == 40 && std::memcmp(bytes.data(), c0.data(), 31) == 0);
    // batch of 3 equals three single sponges
    auto batch = BatchPoseidonSponge::make(sponge_param, 3);
    std::vector<Fp> in;
    for (uint64_t k = 0; k < 3; ++k) { in.push_back(fp_from_u64(Fr, 10 * k)); in.push_back(fp_from_u64(Fr, 10 * k + 1)); in.push_back(fp_from_u64(Fr, 10 * k + 2)); }
    batch.absorb(in);
    auto out = batch.squeeze_native_field_elements(2);
    for (uint64_t k = 0; k < 3; ++k) {
        auto s = PoseidonSponge::make(sponge_param);
        s.absorb({in[3 * k], in[3 * k + 1], in[3 * k + 2]});
        auto o = s.squeeze_native_field_elements(2);
        EXPECT(o[0] == out[2 * k] && o[1] == out[2 * k + 1]);
    }
}

int main(int argc, char **argv) {
    const bool host_only = argc > 1 && std::string(argv[1]) == "--host-only";
    try {
        test_grain_lfsr_and_default_parameters();
        if (!host_only) {
            test_poseidon_sponge_consistency();
            test_squeeze_cast_native_and_state_roundtrip();
        } else {
            // without a device the data path must fail loudly, never fall back
            bool threw = false;
            if (pmx_device_count() == 0) {
                try {
                    auto p = get_default_poseidon_parameters(Field::bls12_381_fr(), 2, false).value();
                    auto s = PoseidonSponge::make(p);
                    s.absorb({fp_from_u64(p.field, 1)});
                } catch (const Error &e) { threw = e.code == PMX_ERR_HIP; }
                EXPECT(threw);
            }
        }
    } catch (const std::exception &e) {
        std::printf("EXCEPTION: %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "FAILED (%d)\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
