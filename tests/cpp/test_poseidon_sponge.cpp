// The reference's own tests for this path, re-expressed against the C++ host mirror
// (sponge_amd/host/poseidon_sponge.hpp).  `--host-only` runs the parts that need no GPU.
//   test_grain_lfsr_consistency                      src/poseidon/grain_lfsr.rs:197-213
//   bls12_381_fr_poseidon_default_parameters_test    src/poseidon/traits.rs:163-358 (rates 2, 3, 8)
//   test_poseidon_sponge_consistency                 src/poseidon/mod.rs:376-399        (GPU)
//   test_squeeze_cast_native                         src/poseidon/tests.rs:71-85         (GPU)
#include <cstdio>
#include <cstring>
#include <string>

#include "../../sponge_amd/host/absorb.hpp"

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

static void test_absorb_encodings_host() {
    const Field Fr = Field::bls12_381_fr();
    // "abc" -> u64-LE length 3, then the bytes, in one 31-byte chunk: 3 + 0x636261 * 2^64
    auto e = collect_sponge_field_elements(Fr, std::vector<uint8_t>{'a', 'b', 'c'});
    EXPECT(e.size() == 1 && e[0] == fp_from_bigint(Fr, {3, 0x636261, 0, 0}));
    // 8 + 24 = 32 bytes -> two elements
    EXPECT(collect_sponge_field_elements(Fr, std::vector<uint8_t>(24, 7)).size() == 2);
    // signed: -2 -> p - 2; bool; option; with_length
    auto m2 = collect_sponge_field_elements(Fr, (int16_t)-2);
    EXPECT(fp_into_bigint(Fr, m2[0])[0] == Fr.modulus[0] - 2);
    EXPECT(collect_sponge_bytes((int16_t)-2) == (std::vector<uint8_t>{0xfe, 0xff}));
    EXPECT(collect_sponge_bytes(std::optional<uint8_t>(3)) == (std::vector<uint8_t>{1, 3}));
    EXPECT(collect_sponge_bytes(std::optional<uint8_t>()) == (std::vector<uint8_t>{0}));
    auto wl = collect_sponge_field_elements(Fr, with_length(std::vector<uint64_t>{5, 6}));
    EXPECT(wl.size() == 3 && wl[0] == fp_from_u64(Fr, 2) && wl[2] == fp_from_u64(Fr, 6));
    // test_macros (src/poseidon/tests.rs:101-116): collect_* equals the long-hand calls
    std::vector<uint8_t> expected;
    to_sponge_bytes(std::vector<int32_t>{6, 5, 4, 3, 2, 1}, expected);
    to_sponge_bytes(FpOf{Fr, fp_from_u64(Fr, 42)}, expected);
    EXPECT(expected.size() == 24 + 32);
    EXPECT(collect_sponge_bytes(std::vector<int32_t>{6, 5, 4, 3, 2, 1}, FpOf{Fr, fp_from_u64(Fr, 42)}) == expected);
    // list_with_nonconstant_size_element (tests.rs:57-69): the per-list lengths separate the two encodings
    using B = std::vector<uint8_t>;
    const B a1{1, 2, 3, 4}, a2{5, 6}, b1{1, 2}, b2{3, 4, 5, 6};
    EXPECT(collect_sponge_bytes(with_length(a1), with_length(a2)) != collect_sponge_bytes(with_length(b1), with_length(b2)));
    EXPECT(collect_sponge_bytes(a1, a2) == collect_sponge_bytes(b1, b2));
    // curve points (src/absorb.rs:232-254): base-field coordinates, [x, y] / [x, y, infinity]; bytes = u64 count + elements
    const TEAffine te{Fr, fp_from_u64(Fr, 7), fp_from_u64(Fr, 9)};
    const SWAffine sw{Fr, fp_from_u64(Fr, 7), fp_from_u64(Fr, 9), true};
    auto tf = collect_sponge_field_elements(Fr, te), sf = collect_sponge_field_elements(Fr, sw);
    EXPECT(tf.size() == 2 && tf[0] == fp_from_u64(Fr, 7) && tf[1] == fp_from_u64(Fr, 9));
    EXPECT(sf.size() == 3 && sf[2] == fp_from_u64(Fr, 1));
    auto tb = collect_sponge_bytes(te);
    EXPECT(tb.size() == 8 + 64 && tb[0] == 2 && tb[8] == 7 && tb[40] == 9);
    EXPECT(collect_sponge_bytes(sw).size() == 8 + 96);
    const Field Bn = Field::bn254_fr();
    bool refused = false;
    try {
        (void)collect_sponge_field_elements(Fr, TEAffine{Bn, fp_from_u64(Bn, 1), fp_from_u64(Bn, 2)});
    } catch (const Error &) { refused = true; }
    EXPECT(refused);
}

static void test_macros_and_fork_on_gpu() {
    // src/poseidon/tests.rs:87-99: absorb!(s, vec![1..6], Fr::from(114514)) == two absorb calls; fork changes the output
    const Field Fr = Field::bls12_381_fr();
    auto sponge_param = get_default_poseidon_parameters(Fr, 2, false).value();
    auto sponge1 = PoseidonSponge::make(sponge_param);
    std::vector<Fp> e1, e2;
    to_sponge_field_elements(Fr, std::vector<int32_t>{1, 2, 3, 4, 5, 6}, e1);
    to_sponge_field_elements(Fr, FpOf{Fr, fp_from_u64(Fr, 114514)}, e2);
    sponge1.absorb(e1);
    sponge1.absorb(e2);
    auto sponge2 = PoseidonSponge::make(sponge_param);
    absorb(sponge2, std::vector<int32_t>{1, 2, 3, 4, 5, 6}, FpOf{Fr, fp_from_u64(Fr, 114514)});
    auto forked = fork(sponge2, {'d', 'o', 'm'});
    auto expected = sponge1.squeeze_native_field_elements(3);
    EXPECT(sponge2.squeeze_native_field_elements(3) == expected);
    EXPECT(forked.squeeze_native_field_elements(3) != expected);
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

static void test_squeeze_with_sizes() {
    // src/lib.rs:45-100, 166-182 and mod.rs:288-317
    const Field Fr = Field::bls12_381_fr(), Fq = Field::bn254_fr();
    auto sponge_param = get_default_poseidon_parameters(Fr, 2, false).value();
    auto base = PoseidonSponge::make(sponge_param);
    base.absorb({fp_from_u64(Fr, 7), fp_from_u64(Fr, 8)});
    // all-Full native sizes == plain native squeeze; the F2 == F case casts
    auto a = base, b = base, c = base;
    auto plain = a.squeeze_native_field_elements(3);
    EXPECT(b.squeeze_native_field_elements_with_sizes(std::vector<FieldElementSize>(3)) == plain);
    EXPECT(c.squeeze_field_elements(3, Fr) == plain);
    // a Truncated size takes the bit path: 254 bits per element out of one squeeze_bits stream
    auto d = base, e = base;
    auto trunc = d.squeeze_native_field_elements_with_sizes({FieldElementSize::Truncated(128), FieldElementSize::Full()});
    auto bits = e.squeeze_bits(2 * 254);
    for (int k = 0; k < 2; ++k) {
        std::array<uint64_t, 4> v{0, 0, 0, 0};
        for (size_t i = 0; i < 254; ++i) if (bits[254 * k + i]) v[i / 64] |= 1ull << (i % 64);
        EXPECT(trunc[k] == fp_from_bigint(Fr, v));
    }
    EXPECT(d.mode == e.mode && d.state == e.state);
    // non-native: BN254 Fr elements take 253 bits each
    auto f = base, g = base;
    auto nn = f.squeeze_field_elements(3, Fq);
    auto nbits = g.squeeze_bits(3 * 253);
    for (int k = 0; k < 3; ++k) {
        std::array<uint64_t, 4> v{0, 0, 0, 0};
        for (size_t i = 0; i < 253; ++i) if (nbits[253 * k + i]) v[i / 64] |= 1ull << (i % 64);
        EXPECT(fp_into_bigint(Fq, nn[k]) == v);
    }
    // empty request: no squeeze, the mode stays Absorbing
    auto h = base;
    EXPECT(h.squeeze_field_elements_with_sizes({}, Fq).empty() && h.mode == base.mode);
    // oversize Truncated panics in the reference (src/lib.rs:48)
    bool threw = false;
    try { (void)h.squeeze_field_elements_with_sizes({FieldElementSize::Truncated(255)}, Fq); } catch (const Error &) { threw = true; }
    EXPECT(threw);
}

// Device groups straight through the C ABI (what BatchPoseidon::new_multi binds): the batch sharded over every visible GPU
// equals the same batch on one device; the sharded tree's root equals the single-device tree's; the communicator RCCL
// built has as many ranks as the group asked for; shard arithmetic on the host.
static void test_device_group_on_gpu() {
    const Field Fr = Field::bls12_381_fr();
    auto p = get_default_poseidon_parameters(Fr, 2, false).value();
    std::vector<uint64_t> ark, mds;
    for (auto &row : p.ark) for (auto &x : row) ark.insert(ark.end(), x.l.begin(), x.l.end());
    for (auto &row : p.mds) for (auto &x : row) mds.insert(mds.end(), x.l.begin(), x.l.end());
    pmx_config c{};
    c.full_rounds = (uint32_t)p.full_rounds; c.partial_rounds = (uint32_t)p.partial_rounds; c.alpha = p.alpha;
    c.rate = (uint32_t)p.rate; c.capacity = (uint32_t)p.capacity;
    std::memcpy(c.modulus, Fr.modulus.data(), 32);
    c.ark = ark.data(); c.mds = mds.data();
    const int ndev = pmx_device_count();
    pmx_mgpu *g = nullptr;
    check(pmx_mgpu_create(&c, ndev, nullptr, &g));
    pmx_mgpu_info info{};
    check(pmx_mgpu_get_info(g, &info));
    EXPECT(info.world == ndev && info.n_local == ndev && info.comm_ranks == ndev && info.width == 3);
    size_t covered = 0;
    for (int r = 0; r < ndev; ++r) {
        size_t start = 0, count = 0;
        check(pmx_shard_bounds(10007, ndev, r, &start, &count));
        EXPECT(start == covered);
        covered += count;
    }
    EXPECT(covered == 10007);
    // 10007 states: group vs one device
    const size_t n = 10007;
    std::vector<uint64_t> a(n * 12), b;
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto &w : a) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = x; }
    for (size_t i = 0; i < n * 3; ++i) a[i * 4 + 3] &= 0x0fffffffffffffffull;   // < 2^252 < p: a reduced residue
    b = a;
    check(pmx_mgpu_permute_batch(g, a.data(), n));
    check(pmx_permute_batch(pmx_mgpu_ctx(g, 0), b.data(), n));
    EXPECT(a == b);
    // 2^12 leaves: sharded root vs single-device root (needs a power-of-two number of devices)
    if ((ndev & (ndev - 1)) == 0) {
        const size_t m = 4096;
        std::vector<uint64_t> leaves(b.begin(), b.begin() + m * 4);
        uint64_t r1[4], r2[4];
        check(pmx_mgpu_merkle_2to1(g, leaves.data(), m, r1));
        check(pmx_merkle_2to1(pmx_mgpu_ctx(g, 0), leaves.data(), m, nullptr, r2));
        EXPECT(std::memcmp(r1, r2, 32) == 0);
    }
    // device-resident flow without a HIP header: allocate, upload, permute twice on the group's stream, gather, download
    {
        const size_t m = 3000;
        size_t start = 0, count = 0;
        check(pmx_shard_bounds(m, ndev, 0, &start, &count));
        std::vector<uint64_t> host(b.begin(), b.begin() + m * 12), twice = host, back(m * 12);
        check(pmx_permute_batch(pmx_mgpu_ctx(g, 0), twice.data(), m));
        check(pmx_permute_batch(pmx_mgpu_ctx(g, 0), twice.data(), m));
        std::vector<uint64_t *> shard(ndev, nullptr), all(ndev, nullptr);
        for (int l = 0; l < ndev; ++l) {
            size_t s0 = 0, c0 = 0;
            check(pmx_shard_bounds(m, ndev, l, &s0, &c0));
            check(pmx_device_alloc(info.devices[l], (void **)&shard[l], c0 * 96));
            check(pmx_device_alloc(info.devices[l], (void **)&all[l], m * 96));
            check(pmx_device_upload(info.devices[l], shard[l], host.data() + s0 * 12, c0 * 96, pmx_mgpu_stream(g, l)));
        }
        check(pmx_mgpu_permute_shards_dev(g, shard.data(), m));
        check(pmx_mgpu_permute_shards_dev(g, shard.data(), m));
        check(pmx_mgpu_all_gather_dev(g, (const uint64_t *const *)shard.data(), all.data(), m, 3));
        check(pmx_device_download(info.devices[0], back.data(), all[0], m * 96, pmx_mgpu_stream(g, 0)));
        check(pmx_stream_synchronize(info.devices[0], pmx_mgpu_stream(g, 0)));
        EXPECT(back == twice);
        for (int l = 0; l < ndev; ++l) {
            check(pmx_device_free(info.devices[l], shard[l]));
            check(pmx_device_free(info.devices[l], all[l]));
        }
    }
    check(pmx_mgpu_destroy(g));
}

int main(int argc, char **argv) {
    const bool host_only = argc > 1 && std::string(argv[1]) == "--host-only";
    try {
        test_grain_lfsr_and_default_parameters();
        test_absorb_encodings_host();
        if (!host_only) {
            test_macros_and_fork_on_gpu();
            test_poseidon_sponge_consistency();
            test_squeeze_cast_native_and_state_roundtrip();
            test_squeeze_with_sizes();
            test_device_group_on_gpu();
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
