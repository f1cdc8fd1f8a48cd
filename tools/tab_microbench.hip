// Timing probe for the shifted-constant tables (pmx_field.hpp: tab_dot): 31 partial rounds of the t = 3 schedule with
// products by constants as sum_j z_j * T_j + two Montgomery steps, against product + full reduction.  Random table contents: timing only, no values are checked.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -opt-disable=reassociate -I sponge_amd/csrc tools/tab_microbench.hip -o tools/tab_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "pmx_field.hpp"
using namespace pmx;

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t *o, const uint32_t *__restrict__ in, const uint32_t *__restrict__ tabs, FieldRt f, int rounds) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fe s[3];
    for (int e = 0; e < 3; ++e)
        for (int i = 0; i < 9; ++i) s[e].l[i] = in[(g * 3 + e) * 9 + i] & kMask;
    for (int r = 0; r < rounds; ++r) {
        Fe z[3];
        z[0] = fe_sbox<5>(s[0], 5, s[0], f);
        z[1] = s[1];
        z[2] = s[2];
        if (MODE == 0) {
            const uint32_t *sp = tabs + (size_t)r * 5 * kFeStride;
            Fe row[3];
            for (int j = 0; j < 3; ++j) row[j] = fe_const(sp + j * kFeStride);
            s[0] = mont_dot<3>(z, row, f);
            s[1] = mont_mul_add(z[0], fe_const(sp + 3 * kFeStride), s[1], f);
            s[2] = mont_mul_add(z[0], fe_const(sp + 4 * kFeStride), s[2], f);
        } else {
            const uint32_t *sp = tabs + (size_t)r * (tab_row_words(3) + 2 * kTabOneWords);
            s[0] = tab_dot<3, false>(z, sp, z[0], f);
            s[1] = tab_dot<1, true>(z, sp + tab_row_words(3), s[1], f);
            s[2] = tab_dot<1, true>(z, sp + tab_row_words(3) + kTabOneWords, s[2], f);
        }
    }
    for (int e = 0; e < 3; ++e)
        for (int i = 0; i < 9; ++i) o[(g * 3 + e) * 9 + i] = s[e].l[i];
}

int main() {
    const size_t n = 1 << 20;
    const int rounds = 31;
    std::vector<uint32_t> h(n * 27), t(rounds * 5 * 96);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 20); };
    for (auto &v : h) v = rnd();
    for (auto &v : t) v = rnd() & kMask;
    uint32_t *d_in, *d_out, *d_t;
    hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_out, h.size() * 4); hipMalloc(&d_t, t.size() * 4);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_t, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    FieldRt f{};
    f.io = nullptr;
    for (int i = 0; i < 9; ++i) f.p[i] = rnd() & kMask;
    f.p[0] |= 1; f.pinv = 0x12345677; f.unit = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 200; ++rep) { if (mode == 0) k<0><<<n / 256, 256>>>(d_out, d_in, d_t, f, rounds); else k<1><<<n / 256, 256>>>(d_out, d_in, d_t, f, rounds); }
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 20; ++rep) { if (mode == 0) k<0><<<n / 256, 256>>>(d_out, d_in, d_t, f, rounds); else k<1><<<n / 256, 256>>>(d_out, d_in, d_t, f, rounds); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.4f ms per launch of %d partial rounds on 2^20 states\n", mode, mode ? "shifted tables" : "product + full reduction", ms / 20, rounds);
    }
    return 0;
}
