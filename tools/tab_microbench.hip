// Feasibility probe for "shifted-constant tables": products by CONSTANTS computed as sum_j z_j * T_j with
// T_j = C * 2^(29 j + 58) mod p precomputed (81 words per constant), followed by two Montgomery steps, against the
// current product + full reduction.  Random table contents: timing only, no values are checked.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -opt-disable=reassociate -I sponge_amd/csrc tools/tab_microbench.hip -o tools/tab_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "pmx_field.hpp"
using namespace pmx;

// sum_i z[i] * C_i * 2^-261 from shifted tables tab_i[k * 9 + j] = limb k of T_j(C_i); optional addend s
template <int N, bool ADD>
__device__ __forceinline__ Fe tab_dot(const Fe (&z)[N], const uint32_t *const (&tab)[N], const Fe &s, const FieldRt &f) {
    constexpr int S = 2;
    uint32_t m[S];
    Fe out;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < kN + S; ++k) {
        if (k < kN) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
#pragma unroll
                for (int j = 0; j < kN; ++j) acc += (uint64_t)z[i].l[j] * tab[i][k * kN + j];
            }
        }
#pragma unroll
        for (int q = 0; q < S; ++q) {
            if (q < k && k - q < kN) acc += (uint64_t)m[q] * f.p[k - q];
        }
        if (k < S) {
            m[k] = mont_step(acc, f);
        } else {
            if (ADD) acc += (uint64_t)s.l[k - S] * f.unit;
            if (k < kN + S - 1) {
                out.l[k - S] = (uint32_t)acc & kMask;
                acc >>= kW;
            } else {
                out.l[k - S] = (uint32_t)acc;
            }
        }
    }
    return out;
}

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
            const uint32_t *sp = tabs + (size_t)r * 5 * 81;
            const uint32_t *const t3[3] = {sp, sp + 81, sp + 162};
            s[0] = tab_dot<3, false>(z, t3, z[0], f);
            const Fe z0[1] = {z[0]};
            const uint32_t *const t1[1] = {sp + 243};
            const uint32_t *const t2[1] = {sp + 324};
            s[1] = tab_dot<1, true>(z0, t1, s[1], f);
            s[2] = tab_dot<1, true>(z0, t2, s[2], f);
        }
    }
    for (int e = 0; e < 3; ++e)
        for (int i = 0; i < 9; ++i) o[(g * 3 + e) * 9 + i] = s[e].l[i];
}

int main() {
    const size_t n = 1 << 20;
    const int rounds = 31;
    std::vector<uint32_t> h(n * 27), t(rounds * 5 * 81);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 20); };
    for (auto &v : h) v = rnd();
    for (auto &v : t) v = rnd() & kMask;
    uint32_t *d_in, *d_out, *d_t;
    hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_out, h.size() * 4); hipMalloc(&d_t, t.size() * 4);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_t, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    FieldRt f{};
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
