// AddressSanitizer + UBSan run of the host-side code paths (GPU sanitizers are unavailable on the pool): the
// kernel algorithm templates compiled for the host, the table preparation (incl. the optimised-schedule linear
// algebra) and the Grain-LFSR parameter generator.  Built and run by tests/test_sanitizers.py.
#include <cstdio>
#include <cstring>
#include <vector>

#include "pmx_hostcheck.cpp"
#include "../../sponge_amd/csrc/pmx_params.cpp"

namespace pmx {
int set_error(int code, const char *, ...) { return code; }
}

int main() {
    const uint64_t bls[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    const uint64_t bn[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    struct Case { const uint64_t *p; uint64_t bits; uint32_t rate, rf, rp; uint64_t alpha; };
    const Case cases[] = {{bls, 255, 2, 8, 31, 5}, {bls, 255, 2, 8, 31, 17}, {bls, 255, 3, 8, 56, 5}, {bn, 254, 8, 8, 57, 5}, {bls, 255, 2, 8, 13, 257}};
    for (const Case &c : cases) {
        const uint32_t t = c.rate + 1;
        std::vector<uint64_t> ark((size_t)(c.rf + c.rp) * t * 4), mds((size_t)t * t * 4);
        if (pmx_find_poseidon_ark_and_mds(c.p, c.bits, c.rate, c.rf, c.rp, 0, ark.data(), mds.data())) return 1;
        pmx_config cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.full_rounds = c.rf; cfg.partial_rounds = c.rp; cfg.alpha = c.alpha; cfg.rate = c.rate; cfg.capacity = 1;
        std::memcpy(cfg.modulus, c.p, 32);
        cfg.ark = ark.data(); cfg.mds = mds.data();
        const size_t n = 5;
        std::vector<uint64_t> base(n * t * 4);
        for (size_t i = 0; i < base.size(); ++i) base[i] = (i % 4 == 3) ? (0x0123456789abcdefull >> 3) % (c.p[3]) : 0x9e3779b97f4a7c15ull * (i + 1);
        std::vector<uint64_t> a = base, b = base, d = base, e = base;
        if (hc_permute(&cfg, a.data(), n)) return 2;
        if (hc_permute_rt(&cfg, b.data(), n)) return 3;
        if (hc_permute_hybrid_mfma(&cfg, d.data(), n)) return 4;      // the window engines' schedule (int8 tables, row finish)
        e = d;
        if (a != b || a != d || a != e) { std::printf("schedules disagree\n"); return 6; }
        if (t == 3) {
            std::vector<uint64_t> g = base;
            if (hc_permute_coop(&cfg, g.data(), n) || g != a) return 7;
        }
    }
    // the absorb / squeeze pass plan (pmx_sponge_plan.hpp) over every mode, index and length at several rates, and at the
    // extremes of its 32-bit arithmetic: every element moved exactly once, in order, and no permutation in the last pass
    for (int squeeze = 0; squeeze < 2; ++squeeze) {
        for (uint32_t rate : {1u, 2u, 3u, 8u, 15u}) {
            for (uint32_t tag = 0; tag < 2; ++tag) {
                for (uint32_t index = 0; index <= rate + 1; ++index) {
                    for (size_t len : {(size_t)0, (size_t)1, (size_t)rate - 0, (size_t)rate + 1, (size_t)3 * rate + 2, (size_t)1000003, (size_t)0x7fffffff}) {
                        const size_t passes = hc_sponge_passes(squeeze, len, rate);
                        if (!squeeze && len == 0) { if (passes != 0) return 8; continue; }
                        uint64_t moved = 0, out[5];
                        const size_t walk = len > 2000000 ? 4 : passes;    // (the longest lengths: the first passes and the last one)
                        for (size_t p = 0; p < passes; ++p) {
                            if (p >= walk && p + 1 < passes) continue;
                            hc_sponge_pass(squeeze, tag, index, len, rate, 1, p, out);
                            if (walk == passes) { if (out[3] && out[2] != moved) return 9; moved += out[3]; }
                            if (out[1] + out[3] > rate + 1 || out[4] > rate) return 10;
                            if (p + 1 == passes && out[0]) return 11;
                        }
                        if (walk == passes && moved != len) return 12;
                    }
                }
            }
        }
    }
    {   // a + b mod p on the ABI residues at the edges
        const uint64_t pm1[4] = {bls[0] - 1, bls[1], bls[2], bls[3]}, one[4] = {1, 0, 0, 0};
        uint64_t r[4];
        if (hc_field_op(bls, 4, pm1, one, r) || r[0] || r[1] || r[2] || r[3]) return 13;
        if (hc_field_op(bls, 4, pm1, pm1, r) || r[0] != bls[0] - 2 || r[3] != bls[3]) return 14;
    }
    std::printf("sanitized ok\n");
    return 0;
}
