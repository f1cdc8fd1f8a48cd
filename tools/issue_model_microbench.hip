// Issue model of one SIMD of gfx950 for the instruction kinds of the limb arithmetic, written in inline assembly so that the
// order and the dependences are exactly the ones named: how often can ONE wave issue, what does a dependent instruction wait
// for, and does independent work inside the same wave fill the gaps (as a second wave on the SIMD does)?
// tools/lone_wave_microbench.hip measures the product's own functions; this file the model underneath them.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/issue_model_microbench.hip -o tools/issue_model_microbench
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

// The whole unrolled body is ONE asm statement: between two statements the compiler's hazard recogniser puts an s_nop.
#define R2(x) x "\n\t" x
#define R4(x) R2(x) "\n\t" R2(x)
#define R16(x) R4(x) "\n\t" R4(x) "\n\t" R4(x) "\n\t" R4(x)
#define R32(x) R16(x) "\n\t" R16(x)

enum Mix {
    MAD_ACC1,        // one accumulator chain: acc += x*y (depends on the previous multiply through the addend only)
    MAD_ACC2,        // two independent accumulator chains, interleaved
    MAD_ACC4,        // four
    MAD_FULLDEP,     // the multiplicand of each multiply is the low word of the previous result
    AND1,            // x &= y chain
    AND2,            // two independent chains
    AND4,
    MAD_AND_DEP,     // mad; and on its result; the and feeds the next multiply (the chain of tools/lone_wave_microbench.hip)
    MAD_AND_IND,     // accumulator chain interleaved with an independent and chain
    MAD_3AND_IND,    // accumulator chain with three independent full-rate instructions after each multiply
    SHIFT64_1,       // v_lshrrev_b64 chain
    SHIFT64_2,       // two independent
    MAD_SHIFT_IND,   // accumulator chain + independent 64-bit shift chain
    ADDC_1,          // v_add_co_u32 / v_addc_co_u32 pair chain (a 64-bit add)
    MAD_ADDC_IND,    // accumulator chain + independent 64-bit add chain
    MAD2_AND2_IND,   // two accumulator chains + two independent and chains
    MAD4_AND12_BURST,   // four multiplies, then twelve independent full-rate instructions: do runs of simple instructions pair up?
    MAD12_AND4_BURST,   // twelve multiplies (four chains), then four independent full-rate instructions (the product's mix, clustered)
    AND4_E64,        // the 64-bit encoding of the same instruction
    MULLO4,          // v_mul_lo_u32, four chains
    ADD4,            // v_add_u32, four chains
    N_MIX
};
static const char *kNames[N_MIX] = {"mad acc x1", "mad acc x2", "mad acc x4", "mad fully dependent", "and x1", "and x2", "and x4",
                                    "mad -> and -> mad", "mad acc | and (indep)", "mad acc | 3 and (indep)", "shift64 x1", "shift64 x2",
                                    "mad acc | shift64 (indep)", "add64 x1", "mad acc | add64 (indep)", "2 mad acc | 2 and (indep)",
                                    "4 mad, then 12 and", "12 mad, then 4 and", "and x4 (VOP3 encoding)", "mul_lo x4", "add x4"};
static const int kInstr[N_MIX] = {1, 2, 4, 1, 1, 2, 4, 2, 2, 4, 1, 2, 2, 2, 3, 4, 16, 16, 4, 4, 4};

template <int MIX>
__global__ void __launch_bounds__(256) bench(uint32_t *out, unsigned long long *cycles, int trips, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t x0 = threadIdx.x * 2654435761u + seed, x1 = x0 ^ 0x9E3779B9u, x2 = x0 + 77, x3 = x1 + 99;
    uint32_t y = seed | 0x10001u, z0 = x0 ^ 0x55, z1 = x1 ^ 0xaa;
    uint64_t s0 = a0 | (1ull << 62), s1 = a1 | (1ull << 61);
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < trips; ++it) {
        {
            if constexpr (MIX == MAD_ACC1) {
                asm volatile(R32("v_mad_u64_u32 %0, vcc, %1, %2, %0")
                             : "+v"(a0) : "v"(x0), "v"(y) : "vcc");
            } else if constexpr (MIX == MAD_ACC2) {
                asm volatile(R32("v_mad_u64_u32 %0, vcc, %2, %4, %0\n\tv_mad_u64_u32 %1, vcc, %3, %4, %1")
                             : "+v"(a0), "+v"(a1) : "v"(x0), "v"(x1), "v"(y) : "vcc");
            } else if constexpr (MIX == MAD_ACC4) {
                asm volatile(R32("v_mad_u64_u32 %0, vcc, %4, %8, %0\n\tv_mad_u64_u32 %1, vcc, %5, %8, %1\n\tv_mad_u64_u32 %2, vcc, %6, %8, %2\n\tv_mad_u64_u32 %3, vcc, %7, %8, %3")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y) : "vcc");
            } else if constexpr (MIX == MAD_FULLDEP) {
                // multiplicand = low half of the accumulator itself (sub-register 0 of the pair)
                asm volatile(R32("v_mad_u64_u32 v[40:41], vcc, v40, %0, v[40:41]")
                             : : "v"(y) : "vcc", "v40", "v41");
            } else if constexpr (MIX == AND1) {
                asm volatile(R32("v_and_b32 %0, %0, %1")
                             : "+v"(x0) : "v"(y));
            } else if constexpr (MIX == AND2) {
                asm volatile(R32("v_and_b32 %0, %0, %2\n\tv_and_b32 %1, %1, %2")
                             : "+v"(x0), "+v"(x1) : "v"(y));
            } else if constexpr (MIX == AND4) {
                asm volatile(R32("v_and_b32 %0, %0, %4\n\tv_and_b32 %1, %1, %4\n\tv_and_b32 %2, %2, %4\n\tv_and_b32 %3, %3, %4")
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));
            } else if constexpr (MIX == MAD_AND_DEP) {
                asm volatile(R32("v_mad_u64_u32 v[40:41], vcc, v44, %0, v[40:41]\n\tv_and_b32 v44, v40, %0")
                             : : "v"(y) : "vcc", "v40", "v41", "v44");
            } else if constexpr (MIX == MAD_AND_IND) {
                asm volatile(R32("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, %1, %3")
                             : "+v"(a0), "+v"(x1) : "v"(x0), "v"(y) : "vcc");
            } else if constexpr (MIX == MAD_3AND_IND) {
                asm volatile(R32("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_and_b32 %1, %1, %5\n\tv_and_b32 %2, %2, %5\n\tv_and_b32 %3, %3, %5")
                             : "+v"(a0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x0), "v"(y) : "vcc");
            } else if constexpr (MIX == SHIFT64_1) {
                asm volatile(R32("v_lshrrev_b64 %0, 1, %0")
                             : "+v"(s0));
            } else if constexpr (MIX == SHIFT64_2) {
                asm volatile(R32("v_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %1, 1, %1")
                             : "+v"(s0), "+v"(s1));
            } else if constexpr (MIX == MAD_SHIFT_IND) {
                asm volatile(R32("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshrrev_b64 %1, 1, %1")
                             : "+v"(a0), "+v"(s0) : "v"(x0), "v"(y) : "vcc");
            } else if constexpr (MIX == ADDC_1) {
                asm volatile(R32("v_add_co_u32 v42, vcc, v42, %0\n\tv_addc_co_u32 v43, vcc, 0, v43, vcc")
                             : : "v"(y) : "vcc", "v42", "v43");
            } else if constexpr (MIX == MAD_ADDC_IND) {
                asm volatile(R32("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n\tv_add_co_u32 v42, vcc, v42, %2\n\tv_addc_co_u32 v43, vcc, 0, v43, vcc")
                             : "+v"(a0) : "v"(x0), "v"(y) : "vcc", "s10", "s11", "v42", "v43");
            } else if constexpr (MIX == MAD4_AND12_BURST) {
                asm volatile(R32("v_mad_u64_u32 %0, vcc, %8, %10, %0\n\tv_mad_u64_u32 %1, vcc, %9, %10, %1\n\tv_mad_u64_u32 %2, vcc, %8, %10, %2\n\tv_mad_u64_u32 %3, vcc, %9, %10, %3\n\t"
                                 "v_and_b32 %4, %4, %10\n\tv_and_b32 %5, %5, %10\n\tv_and_b32 %6, %6, %10\n\tv_and_b32 %7, %7, %10\n\t"
                                 "v_and_b32 %4, %4, %10\n\tv_and_b32 %5, %5, %10\n\tv_and_b32 %6, %6, %10\n\tv_and_b32 %7, %7, %10\n\t"
                                 "v_and_b32 %4, %4, %10\n\tv_and_b32 %5, %5, %10\n\tv_and_b32 %6, %6, %10\n\tv_and_b32 %7, %7, %10")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(z0), "v"(z1), "v"(y) : "vcc");
            } else if constexpr (MIX == MAD12_AND4_BURST) {
                asm volatile(R32("v_mad_u64_u32 %0, vcc, %8, %10, %0\n\tv_mad_u64_u32 %1, vcc, %9, %10, %1\n\tv_mad_u64_u32 %2, vcc, %8, %10, %2\n\tv_mad_u64_u32 %3, vcc, %9, %10, %3\n\t"
                                 "v_mad_u64_u32 %0, vcc, %8, %10, %0\n\tv_mad_u64_u32 %1, vcc, %9, %10, %1\n\tv_mad_u64_u32 %2, vcc, %8, %10, %2\n\tv_mad_u64_u32 %3, vcc, %9, %10, %3\n\t"
                                 "v_mad_u64_u32 %0, vcc, %8, %10, %0\n\tv_mad_u64_u32 %1, vcc, %9, %10, %1\n\tv_mad_u64_u32 %2, vcc, %8, %10, %2\n\tv_mad_u64_u32 %3, vcc, %9, %10, %3\n\t"
                                 "v_and_b32 %4, %4, %10\n\tv_and_b32 %5, %5, %10\n\tv_and_b32 %6, %6, %10\n\tv_and_b32 %7, %7, %10")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(z0), "v"(z1), "v"(y) : "vcc");
            } else if constexpr (MIX == AND4_E64) {
                asm volatile(R32("v_and_b32_e64 %0, %0, %4\n\tv_and_b32_e64 %1, %1, %4\n\tv_and_b32_e64 %2, %2, %4\n\tv_and_b32_e64 %3, %3, %4")
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));
            } else if constexpr (MIX == MULLO4) {
                asm volatile(R32("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4")
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));
            } else if constexpr (MIX == ADD4) {
                asm volatile(R32("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4")
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));
            } else if constexpr (MIX == MAD2_AND2_IND) {
                asm volatile(R32("v_mad_u64_u32 %0, vcc, %4, %6, %0\n\tv_and_b32 %2, %2, %6\n\tv_mad_u64_u32 %1, vcc, %5, %6, %1\n\tv_and_b32 %3, %3, %6")
                             : "+v"(a0), "+v"(a1), "+v"(x2), "+v"(x3) : "v"(x0), "v"(x1), "v"(y) : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t r = (uint32_t)(a0 ^ a1 ^ a2 ^ a3) ^ (uint32_t)((a0 ^ a1 ^ a2 ^ a3) >> 32) ^ x0 ^ x1 ^ x2 ^ x3 ^ (uint32_t)(s0 ^ s1) ^ (uint32_t)((s0 ^ s1) >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        cycles[3 * w] = t1 - t0;   // s_memtime ticks
        cycles[3 * w + 1] = r0;    // the 100 MHz counter all CUs share: when this wave started and ended
        cycles[3 * w + 2] = r1;
    }
}

// `lds` bytes of dynamic LDS per block pin the residency: with 160 KiB per CU, blocks that ask for 160/k KiB sit k to a CU,
// so a launch of k x (number of CUs) blocks of four waves is exactly k waves on every SIMD (the dispatcher alone does not
// spread a launch evenly).
template <int MIX>
void run(int blocks, int threads, int trips, size_t lds, uint32_t *d_out, unsigned long long *d_cyc) {
    CHECK(hipFuncSetAttribute((const void *)bench<MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(bench<MIX>, dim3(blocks), dim3(threads), lds, 0, d_out, d_cyc, trips / 4 + 1, 1u);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<MIX>, dim3(blocks), dim3(threads), lds, 0, d_out, d_cyc, trips, 1u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    const int waves = blocks * threads / 64;
    std::vector<unsigned long long> cyc(3 * (size_t)waves);
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0, mean_real = 0;
    unsigned long long first = ~0ull, last_start = 0, first_end = ~0ull, last = 0;
    for (int w = 0; w < waves; ++w) {
        mean += (double)cyc[3 * w];
        mean_real += (double)(cyc[3 * w + 2] - cyc[3 * w + 1]);
        first = std::min(first, cyc[3 * w + 1]);
        last_start = std::max(last_start, cyc[3 * w + 1]);
        first_end = std::min(first_end, cyc[3 * w + 2]);
        last = std::max(last, cyc[3 * w + 2]);
    }
    mean /= waves;
    mean_real /= waves;
    const double tick_ghz = mean / (mean_real * 10.0);   // ticks per ns (the shared counter runs at 100 MHz)
    const double steps = 32.0 * trips;
    const double per_wave = mean / steps;
    const double waves_per_simd = blocks * threads / 64.0 / (256.0 * 4.0) < 1 ? 1 : blocks * threads / 64.0 / (256.0 * 4.0);
    // wall clock: instructions one SIMD issued / elapsed (the launch itself is ~10 us of several ms)
    const double ns_per_instr = ms * 1e6 / (steps * kInstr[MIX] * waves_per_simd);
    printf("%-28s %8.2f ticks/step per wave  %7.2f per SIMD  (%d instr/step: %5.2f ticks each per SIMD)  %8.3f ms = %5.2f ns per instr and SIMD; s_memtime at %.3f GHz; waves start within %.1f us, run %.1f us, all end within %.1f us\n",
           kNames[MIX], per_wave, per_wave / waves_per_simd, kInstr[MIX], per_wave / waves_per_simd / kInstr[MIX], ms, ns_per_instr, tick_ghz,
           (last_start - first) * 0.01, mean_real * 0.01, (last - first_end) * 0.01);
}

template <int MIX>
void run_all(int blocks, int threads, int trips, size_t lds, uint32_t *d_out, unsigned long long *d_cyc) {
    run<MIX>(blocks, threads, trips, lds, d_out, d_cyc);
    if constexpr (MIX + 1 < N_MIX) run_all<MIX + 1>(blocks, threads, trips, lds, d_out, d_cyc);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.gcnArchName, n_cu);
    uint32_t *d_out;
    unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 32 * 256 * 4));
    CHECK(hipMalloc(&d_cyc, (size_t)n_cu * 32 * 4 * 8 * 3));
    const int trips = 2000;
    struct Shape { int blocks, threads, per_cu; const char *what; };
    const Shape shapes[] = {{1, 64, 1, "one wave on the whole device"}, {n_cu, 256, 1, "one wave per SIMD"}, {2 * n_cu, 256, 2, "two waves per SIMD"},
                            {3 * n_cu, 256, 3, "three waves per SIMD"}, {4 * n_cu, 256, 4, "four waves per SIMD"}, {8 * n_cu, 256, 8, "eight waves per SIMD"}};
    for (const Shape &s : shapes) {
        const size_t lds = (size_t)(160 * 1024 / s.per_cu) - 1024;   // k blocks fit a CU, k + 1 do not
        printf("---- %s (%zu bytes of LDS per block)\n", s.what, lds);
        run_all<0>(s.blocks, s.threads, trips, lds, d_out, d_cyc);
    }
    return 0;
}
