// Do the matrix cores and the vector ALU of one gfx950 SIMD run side by side?  The window engines (pmx_mfma.hpp) put the products by
// constants on v_mfma_i32_32x32x32_i8 to take them off the VALU issue port; what that buys depends on whether the SIMD keeps issuing
// VALU instructions while a matrix-core instruction executes - from the same wave (independent instructions interleaved) or from the
// other wave on the SIMD.  Inline assembly, one statement per unrolled body, exactly k waves per SIMD (pinned by the LDS a block asks for).
//   unit of VALU work: 16 v_mad_u64_u32 on four independent accumulator chains
//   unit of matrix work: 2 v_mfma_i32_32x32x32_i8 on two accumulators (32 clocks of the matrix pipe each)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_valu_overlap_microbench.hip -o tools/mfma_valu_overlap_microbench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
extern __shared__ uint4 lds_pin[];

#define MAD4 "v_mad_u64_u32 %[a0], vcc, %[x0], %[y], %[a0]\n\tv_mad_u64_u32 %[a1], vcc, %[x1], %[y], %[a1]\n\tv_mad_u64_u32 %[a2], vcc, %[x2], %[y], %[a2]\n\tv_mad_u64_u32 %[a3], vcc, %[x3], %[y], %[a3]"
#define MAD8 MAD4 "\n\t" MAD4
#define MAD16 MAD8 "\n\t" MAD8
#define MF1 "v_mfma_i32_32x32x32_i8 %[d1], %[a], %[b], %[d1]"
#define MF2 "v_mfma_i32_32x32x32_i8 %[d2], %[a], %[b], %[d2]"
#define R4(x) x "\n\t" x "\n\t" x "\n\t" x

enum Kind { VALU_ONLY, MFMA_ONLY, INTERLEAVED, BURST, SPLIT_WAVES, HALF_VALU, N_KIND };
static const char *kNames[N_KIND] = {
    "VALU only: 16 mads per unit, every wave",
    "matrix only: 2 MFMA per unit, every wave",
    "one wave does both, interleaved: MFMA, 8 mads, MFMA, 8 mads",
    "one wave does both, in bursts: 2 MFMA, then 16 mads",
    "two kinds of waves on each SIMD: waves 0-3 of the block only mads, waves 4-7 only MFMA (two-wave launches only)",
    "interleaved with half the VALU work: MFMA, 4 mads, MFMA, 4 mads"};

template <int KIND>
__global__ void __launch_bounds__(512) bench(uint32_t *out, int trips, uint32_t seed) {
    uint64_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    uint32_t x0 = threadIdx.x | 1, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6, y = seed * 2654435761u | 1;
    v16i d1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d2 = d1;
    v4i a = {(int)x0, (int)x1, (int)x2, (int)x3}, b = {(int)y, (int)x1, (int)y, (int)x3};
    const bool matrix_wave = (threadIdx.x >> 8) & 1;   // waves 4-7 of a 512-thread block
    for (int i = 0; i < trips; ++i) {
#define OPS : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [d1] "+v"(d1), [d2] "+v"(d2) : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [y] "v"(y), [a] "v"(a), [b] "v"(b) : "vcc"
        if constexpr (KIND == VALU_ONLY) asm volatile(R4(MAD16) OPS);
        else if constexpr (KIND == MFMA_ONLY) asm volatile(R4(MF1 "\n\t" MF2) OPS);
        else if constexpr (KIND == INTERLEAVED) asm volatile(R4(MF1 "\n\t" MAD8 "\n\t" MF2 "\n\t" MAD8) OPS);
        else if constexpr (KIND == BURST) asm volatile(R4(MF1 "\n\t" MF2 "\n\t" MAD16) OPS);
        else if constexpr (KIND == HALF_VALU) asm volatile(R4(MF1 "\n\t" MAD4 "\n\t" MF2 "\n\t" MAD4) OPS);
        else {
            if (matrix_wave) asm volatile(R4(MF1 "\n\t" MF2) OPS);
            else asm volatile(R4(MAD16) OPS);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the matrix results have landed before anything reads them
    uint32_t r = (uint32_t)(a0 ^ a1 ^ a2 ^ a3) ^ (uint32_t)((a0 ^ a1 ^ a2 ^ a3) >> 32);
    for (int k = 0; k < 16; ++k) r ^= (uint32_t)d1[k] ^ (uint32_t)d2[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ lds_pin[0].x;
}

template <int KIND>
static double run(int waves_per_simd, int n_cu, size_t cu_lds, uint32_t *d_out) {
    const int threads = 256 * waves_per_simd;                 // one block per CU, `waves_per_simd` waves on each SIMD
    const size_t lds = std::min(cu_lds - 1024, (size_t)160 * 1024 - 1024);
    CHECK(hipFuncSetAttribute((const void *)bench<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int trips = 2048;                                   // x 4 units per trip
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<double> ms;
    for (int rep = 0; rep < 12; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<KIND>, dim3(n_cu), dim3(threads), lds, 0, d_out, trips, 1u + rep);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t = 0;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    std::sort(ms.begin() + 4, ms.end());                      // past the clock ramp
    const double med = ms[4 + (ms.size() - 4) / 2];
    return med * 1e6 / (trips * 4.0);                         // ns per unit of ONE wave's loop (all waves of a SIMD run concurrently)
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    int cu_lds = 0;
    if (hipDeviceGetAttribute(&cu_lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0) != hipSuccess || cu_lds <= 0) cu_lds = 160 * 1024;
    uint32_t *d_out = nullptr;
    CHECK(hipMalloc((void **)&d_out, (size_t)n_cu * 512 * 4));
    printf("%s, %d CUs.  ns per loop unit (a unit = 16 mads and / or 2 MFMA per wave; every wave of the launch runs the same number of units)\n", prop.name, n_cu);
    printf("(2.2 GHz: 16 mads at one per 4 clocks = 29 ns; 2 MFMA at 32 clocks each = 29 ns)\n");
    for (int w = 1; w <= 2; ++w) {
        printf("--- %d wave(s) per SIMD ---\n", w);
        printf("  %7.2f  %s\n", run<VALU_ONLY>(w, n_cu, cu_lds, d_out), kNames[VALU_ONLY]);
        printf("  %7.2f  %s\n", run<MFMA_ONLY>(w, n_cu, cu_lds, d_out), kNames[MFMA_ONLY]);
        printf("  %7.2f  %s\n", run<INTERLEAVED>(w, n_cu, cu_lds, d_out), kNames[INTERLEAVED]);
        printf("  %7.2f  %s\n", run<BURST>(w, n_cu, cu_lds, d_out), kNames[BURST]);
        printf("  %7.2f  %s\n", run<HALF_VALU>(w, n_cu, cu_lds, d_out), kNames[HALF_VALU]);
        if (w == 2) printf("  %7.2f  %s\n", run<SPLIT_WAVES>(w, n_cu, cu_lds, d_out), kNames[SPLIT_WAVES]);
    }
    CHECK(hipFree(d_out));
    return 0;
}
