// What does a matrix-core instruction cost the VALU port when it comes the way the window engines issue it - a BURST of 2 x 17 products
// per row, then ~26 VALU instructions per product (row finish, operand cuts, S-boxes) - and does the accumulator pattern matter?
// tools/mfma_valu_overlap_microbench.hip priced units of 2 products + 16 multiplies; the t = 9 kernel pays ~19 clocks per product where
// that probe paid 9.  Unit of one wave here: 32 v_mfma_i32_32x32x32_i8 and 832 v_mad_u64_u32 (four chains), in five arrangements,
// at exactly 1 and 2 waves per SIMD:
//   alt     d1, d2, d1, d2 ... (the kernel's form: a k-step is one product for states 0-31 and one for states 32-63), then the multiplies
//   chains  d1 x 16, then d2 x 16 (every product accumulates into the result of the one in front of it), then the multiplies
//   four    d1 .. d4 round robin (no product waits for its own predecessor), then the multiplies
//   spread  one product, 26 multiplies, alternating accumulators
//   spread chains  one product, 26 multiplies, d1 for the first half and d2 for the second
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_burst_microbench.hip -o tools/mfma_burst_microbench
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

#define MAD2 "v_mad_u64_u32 %[a0], vcc, %[x0], %[y], %[a0]\n\tv_mad_u64_u32 %[a1], vcc, %[x1], %[y], %[a1]\n\t"
#define MAD4 MAD2 "v_mad_u64_u32 %[a2], vcc, %[x2], %[y], %[a2]\n\tv_mad_u64_u32 %[a3], vcc, %[x3], %[y], %[a3]\n\t"
#define MAD8 MAD4 MAD4
#define MAD26 MAD8 MAD8 MAD8 MAD2
#define R2(x) x x
#define R4(x) R2(R2(x))
#define R8(x) R4(R2(x))
#define R16(x) R4(R4(x))
#define R32(x) R16(R2(x))
#define MADS R32(MAD26)
#define MF(d, a) "v_mfma_i32_32x32x32_i8 %[" #d "], %[" #a "], %[b], %[" #d "]\n\t"
#define MG(d, a) "v_mfma_i32_16x16x64_i8 %[" #d "], %[" #a "], %[b], %[" #d "]\n\t"
#define OPS : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [d4] "+v"(d4), [g1] "+v"(g1), [g2] "+v"(g2), [g3] "+v"(g3), [g4] "+v"(g4) : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [y] "v"(y), [p] "v"(p), [q] "v"(q), [b] "v"(b) : "vcc"

enum Kind { VALU_ONLY, MFMA_ONLY, ALT, CHAINS, FOUR, SPREAD, SPREAD_CHAINS, SMALL_ONLY, SMALL_ALT, SMALL_SPREAD, N_KIND };
static const char *kNames[N_KIND] = {"VALU only: 832 multiplies", "matrix only: 32 products (alternating accumulators)", "alt: (d1, d2) x 16, then 832 multiplies",
                                     "chains: d1 x 16, d2 x 16, then 832 multiplies", "four: (d1, d2, d3, d4) x 8, then 832 multiplies",
                                     "spread: (product, 26 multiplies) x 32, accumulators alternating", "spread chains: (d1, 26 multiplies) x 16, (d2, 26 multiplies) x 16",
                                     "matrix only, 16x16x64: 64 products on four accumulators of 4 registers", "16x16x64 alt: (g1 .. g4) x 16, then 832 multiplies", "16x16x64 spread: (product, 13 multiplies) x 64"};

// PRIO 1: the wave in the odd wave slot of its SIMD (HW_ID.WAVE_ID) runs at priority 3 from start to end; 2: a wave raises its priority
// for its products only; 3: for its multiplies only (both waves alike)
template <int KIND, int PRIO = 0>
__global__ void __launch_bounds__(256) bench(uint32_t *out, int trips, uint32_t seed) {
    if constexpr (PRIO == 1) {
        if (__builtin_amdgcn_s_getreg((4 << 11) | 4) & 1) __builtin_amdgcn_s_setprio(3);   // HW_REG_HW_ID bits 0-3: the wave slot
    }
    uint64_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    uint32_t x0 = threadIdx.x | 1, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6, y = seed * 2654435761u | 1;
    v16i d1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d2 = d1, d3 = d1, d4 = d1;
    v4i g1 = {0, 0, 0, 0}, g2 = g1, g3 = g1, g4 = g1;
    v4i p = {(int)x0, (int)x1, (int)x2, (int)x3}, q = {(int)x3, (int)x2, (int)x1, (int)x0}, b = {(int)y, (int)x1, (int)y, (int)x3};
    lds_pin[threadIdx.x] = make_uint4(x0, x1, x2, x3);
    for (int i = 0; i < trips; ++i) {
        if constexpr (KIND == VALU_ONLY) asm volatile(MADS OPS);
        if constexpr (KIND == MFMA_ONLY) asm volatile(R16(MF(d1, p) MF(d2, p)) OPS);
        if constexpr (KIND == ALT && PRIO < 2) asm volatile(R8(MF(d1, p) MF(d2, p) MF(d1, q) MF(d2, q)) MADS OPS);
        if constexpr (KIND == ALT && PRIO == 2) asm volatile("s_setprio 3\n\t" R8(MF(d1, p) MF(d2, p) MF(d1, q) MF(d2, q)) "s_setprio 0\n\t" MADS OPS);
        if constexpr (KIND == ALT && PRIO == 3) asm volatile(R8(MF(d1, p) MF(d2, p) MF(d1, q) MF(d2, q)) "s_setprio 3\n\t" MADS "s_setprio 0\n\t" OPS);
        if constexpr (KIND == CHAINS) asm volatile(R8(MF(d1, p) MF(d1, q)) R8(MF(d2, p) MF(d2, q)) MADS OPS);
        if constexpr (KIND == FOUR) asm volatile(R8(MF(d1, p) MF(d2, p) MF(d3, q) MF(d4, q)) MADS OPS);
        if constexpr (KIND == SPREAD) asm volatile(R16(MF(d1, p) MAD26 MF(d2, p) MAD26) OPS);
        if constexpr (KIND == SMALL_ONLY) asm volatile(R16(MG(g1, p) MG(g2, p) MG(g3, q) MG(g4, q)) OPS);
        if constexpr (KIND == SMALL_ALT) asm volatile(R16(MG(g1, p) MG(g2, p) MG(g3, q) MG(g4, q)) MADS OPS);
        if constexpr (KIND == SMALL_SPREAD) asm volatile(R16(MG(g1, p) MAD8 MAD4 "v_mad_u64_u32 %[a0], vcc, %[x0], %[y], %[a0]\n\t" MG(g2, p) MAD8 MAD4 "v_mad_u64_u32 %[a1], vcc, %[x1], %[y], %[a1]\n\t" MG(g3, q) MAD8 MAD4 "v_mad_u64_u32 %[a2], vcc, %[x2], %[y], %[a2]\n\t" MG(g4, q) MAD8 MAD4 "v_mad_u64_u32 %[a3], vcc, %[x3], %[y], %[a3]\n\t") OPS);
        if constexpr (KIND == SPREAD_CHAINS) asm volatile(R16(MF(d1, p) MAD26) R16(MF(d2, q) MAD26) OPS);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the matrix results have landed before anything reads them
    uint32_t r = (uint32_t)(a0 ^ a1 ^ a2 ^ a3) ^ (uint32_t)((a0 ^ a1 ^ a2 ^ a3) >> 32);
    for (int k = 0; k < 16; ++k) r ^= (uint32_t)d1[k] ^ (uint32_t)d2[k] ^ (uint32_t)d3[k] ^ (uint32_t)d4[k];
    for (int k = 0; k < 4; ++k) r ^= (uint32_t)g1[k] ^ (uint32_t)g2[k] ^ (uint32_t)g3[k] ^ (uint32_t)g4[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ lds_pin[0].x;
}

template <int KIND, int PRIO = 0>
static double run(int waves, int n_cu, size_t cu_lds, uint32_t *d_out) {
    const int blocks = n_cu * waves;   // blocks of four waves, `waves` of them per CU (pinned by the LDS each asks for)
    size_t lds = cu_lds / waves - 1024;
    lds = std::min(lds, (size_t)64 * 1024);
    if (waves == 1) lds = 64 * 1024;   // (one block per CU is then pinned by launching exactly n_cu blocks)
    CHECK(hipFuncSetAttribute((const void *)bench<KIND, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int trips = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<double> ms;
    for (int rep = 0; rep < 10; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((bench<KIND, PRIO>), dim3(blocks), dim3(256), lds, 0, d_out, trips, 1u + rep);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t = 0;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    std::sort(ms.begin() + 3, ms.end());
    return ms[3 + (ms.size() - 3) / 2] * 1e6 / trips / waves;   // ns of SIMD time per unit
}

// the same with operands in ACCUMULATION registers (the upper half of the unified file): WHICH bit 0 the accumulators (results would come
// back through v_accvgpr_read), bit 1 the A operand (a load can land there directly), bit 2 the B operand (a v_accvgpr_write per word)
#define OPS_X(CC, AC, BC) : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [d1] CC(d1), [d2] CC(d2) : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [y] "v"(y), [p] AC(p), [q] AC(q), [b] BC(b) : "vcc"
template <int KIND, int WHICH>
__global__ void __launch_bounds__(256) bench_acc(uint32_t *out, int trips, uint32_t seed) {
    uint64_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    uint32_t x0 = threadIdx.x | 1, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6, y = seed * 2654435761u | 1;
    v16i d1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d2 = d1;
    v4i p = {(int)x0, (int)x1, (int)x2, (int)x3}, q = {(int)x3, (int)x2, (int)x1, (int)x0}, b = {(int)y, (int)x1, (int)y, (int)x3};
    lds_pin[threadIdx.x] = make_uint4(x0, x1, x2, x3);
#define BODY(OPS_)                                                                                       \
    for (int i = 0; i < trips; ++i) {                                                                    \
        if constexpr (KIND == MFMA_ONLY) asm volatile(R16(MF(d1, p) MF(d2, p)) OPS_);                    \
        if constexpr (KIND == ALT) asm volatile(R8(MF(d1, p) MF(d2, p) MF(d1, q) MF(d2, q)) MADS OPS_);  \
        if constexpr (KIND == SPREAD) asm volatile(R16(MF(d1, p) MAD26 MF(d2, p) MAD26) OPS_);           \
    }
    if constexpr (WHICH == 1) { BODY(OPS_X("+a", "v", "v")) }
    if constexpr (WHICH == 2) { BODY(OPS_X("+v", "a", "v")) }
    if constexpr (WHICH == 6) { BODY(OPS_X("+v", "a", "a")) }
    if constexpr (WHICH == 7) { BODY(OPS_X("+a", "a", "a")) }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    uint32_t r = (uint32_t)(a0 ^ a1 ^ a2 ^ a3) ^ (uint32_t)((a0 ^ a1 ^ a2 ^ a3) >> 32);
    for (int k = 0; k < 16; ++k) r ^= (uint32_t)d1[k] ^ (uint32_t)d2[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ lds_pin[0].x;
}
template <int KIND, int WHICH = 1>
static double run_acc(int waves, int n_cu, size_t cu_lds, uint32_t *d_out) {
    const int blocks = n_cu * waves;
    size_t lds = std::min(cu_lds / waves - 1024, (size_t)64 * 1024);
    if (waves == 1) lds = 64 * 1024;
    CHECK(hipFuncSetAttribute((const void *)bench_acc<KIND, WHICH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int trips = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<double> ms;
    for (int rep = 0; rep < 10; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((bench_acc<KIND, WHICH>), dim3(blocks), dim3(256), lds, 0, d_out, trips, 1u + rep);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t = 0;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    std::sort(ms.begin() + 3, ms.end());
    return ms[3 + (ms.size() - 3) / 2] * 1e6 / trips / waves;
}

// Forced anti-phase: one block of eight waves per CU - waves w and w + 4 share SIMD w -, waves 0-3 run (products, multiplies), waves 4-7
// (half the multiplies, products, the other half), and a workgroup barrier per unit (or per 4 units) keeps the two in step.  SYNC 0: no
// barrier at all (the pair starts in anti-phase and is left alone).
template <int SYNC>
__global__ void __launch_bounds__(512) anti(uint32_t *out, int trips, uint32_t seed) {
    uint64_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    uint32_t x0 = threadIdx.x | 1, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6, y = seed * 2654435761u | 1;
    v16i d1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d2 = d1, d3 = d1, d4 = d1;
    v4i g1 = {0, 0, 0, 0}, g2 = g1, g3 = g1, g4 = g1;
    v4i p = {(int)x0, (int)x1, (int)x2, (int)x3}, q = {(int)x3, (int)x2, (int)x1, (int)x0}, b = {(int)y, (int)x1, (int)y, (int)x3};
    lds_pin[threadIdx.x] = make_uint4(x0, x1, x2, x3);
    const bool second = threadIdx.x >= 256;
    for (int i = 0; i < trips; ++i) {
        if (second) asm volatile(R16(MAD26) R8(MF(d1, p) MF(d2, p) MF(d1, q) MF(d2, q)) R16(MAD26) OPS);
        else asm volatile(R8(MF(d1, p) MF(d2, p) MF(d1, q) MF(d2, q)) MADS OPS);
        if (SYNC > 0 && i % SYNC == SYNC - 1) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    uint32_t r = (uint32_t)(a0 ^ a1 ^ a2 ^ a3) ^ (uint32_t)((a0 ^ a1 ^ a2 ^ a3) >> 32);
    for (int k = 0; k < 16; ++k) r ^= (uint32_t)d1[k] ^ (uint32_t)d2[k] ^ (uint32_t)d3[k] ^ (uint32_t)d4[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ lds_pin[0].x;
}
template <int SYNC>
static double run_anti(int n_cu, uint32_t *d_out) {
    const size_t lds = 64 * 1024;
    CHECK(hipFuncSetAttribute((const void *)anti<SYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int trips = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<double> ms;
    for (int rep = 0; rep < 10; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(anti<SYNC>, dim3(n_cu), dim3(512), lds, 0, d_out, trips, 1u + rep);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t = 0;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    std::sort(ms.begin() + 3, ms.end());
    return ms[3 + (ms.size() - 3) / 2] * 1e6 / trips / 2;
}

// Who yields to whom?  One block of eight waves per CU; the waves of one wave slot only issue products (104 per unit: as long alone as
// the 832 multiplies), those of the other slot only multiplies.  Every wave times itself (wall clock, 100 MHz).
template <int MATRIX_FIRST>
__global__ void __launch_bounds__(512) split(uint32_t *out, unsigned long long *ticks, int trips, uint32_t seed) {
    uint64_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    uint32_t x0 = threadIdx.x | 1, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6, y = seed * 2654435761u | 1;
    v16i d1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d2 = d1, d3 = d1, d4 = d1;
    v4i g1 = {0, 0, 0, 0}, g2 = g1, g3 = g1, g4 = g1;
    v4i p = {(int)x0, (int)x1, (int)x2, (int)x3}, q = {(int)x3, (int)x2, (int)x1, (int)x0}, b = {(int)y, (int)x1, (int)y, (int)x3};
    lds_pin[threadIdx.x] = make_uint4(x0, x1, x2, x3);
    __syncthreads();
    const bool matrix = (threadIdx.x >= 256) != (MATRIX_FIRST != 0);
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < trips; ++i) {
        if (matrix) asm volatile(R8(MF(d1, p) MF(d2, p) MF(d1, q) MF(d2, q)) R16(MF(d1, p) MF(d2, p) MF(d1, q) MF(d2, q)) R2(MF(d1, p) MF(d2, p) MF(d1, q) MF(d2, q)) OPS);
        else asm volatile(MADS OPS);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    uint32_t r = (uint32_t)(a0 ^ a1 ^ a2 ^ a3) ^ (uint32_t)((a0 ^ a1 ^ a2 ^ a3) >> 32);
    for (int k = 0; k < 16; ++k) r ^= (uint32_t)d1[k] ^ (uint32_t)d2[k] ^ (uint32_t)d3[k] ^ (uint32_t)d4[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ lds_pin[0].x;
}
template <int MATRIX_FIRST>
static void run_split(int n_cu, uint32_t *d_out) {
    CHECK(hipFuncSetAttribute((const void *)split<MATRIX_FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    unsigned long long *d_t;
    CHECK(hipMalloc((void **)&d_t, (size_t)n_cu * 8 * 8));
    const int trips = 256;
    std::vector<unsigned long long> h((size_t)n_cu * 8);
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(split<MATRIX_FIRST>, dim3(n_cu), dim3(512), 64 * 1024, 0, d_out, d_t, trips, 1u + rep);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), d_t, h.size() * 8, hipMemcpyDeviceToHost));
    double first = 0, second = 0;
    for (int b = 0; b < n_cu; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? first : second) += (double)h[b * 8 + w] * 10.0 / trips / (4.0 * n_cu);
    printf("  waves 0-3 (%s) %8.1f ns per unit, waves 4-7 (%s) %8.1f ns per unit\n", MATRIX_FIRST ? "104 products" : "832 multiplies", first,
           MATRIX_FIRST ? "832 multiplies" : "104 products", second);
    CHECK(hipFree(d_t));
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    int cu_lds = 0;
    if (hipDeviceGetAttribute(&cu_lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0) != hipSuccess || cu_lds <= 0) cu_lds = 160 * 1024;
    uint32_t *d_out = nullptr;
    CHECK(hipMalloc((void **)&d_out, (size_t)n_cu * 8 * 256 * 4));
    printf("%s, %d CUs.  ns of SIMD time per unit (a unit = 832 multiplies and / or 32 products of one wave)\n", prop.gcnArchName, n_cu);
    for (int i = 0; i < 30; ++i) (void)run<VALU_ONLY>(2, n_cu, cu_lds, d_out);   // spin-up
    for (int w = 1; w <= 2; ++w) {
        printf("--- %d wave(s) per SIMD ---\n", w);
        const double v = run<VALU_ONLY>(w, n_cu, cu_lds, d_out), m = run<MFMA_ONLY>(w, n_cu, cu_lds, d_out);
        printf("  %8.1f  %s  (%.2f ns each)\n", v, kNames[VALU_ONLY], v / 832);
        printf("  %8.1f  %s  (%.2f ns each)\n", m, kNames[MFMA_ONLY], m / 32);
        auto line = [&](int k, double t) { printf("  %8.1f  %-70s a product costs %5.2f ns on top of the multiplies\n", t, kNames[k], (t - v) / 32); };
        line(ALT, run<ALT>(w, n_cu, cu_lds, d_out));
        if (w == 2) {
            printf("    with priorities:\n");
            printf("  %8.1f  alt, the wave in the odd slot of a SIMD at priority 3 throughout: a product costs %5.2f ns\n", run<ALT, 1>(w, n_cu, cu_lds, d_out), (run<ALT, 1>(w, n_cu, cu_lds, d_out) - v) / 32);
            printf("  %8.1f  alt, priority 3 during a wave's products: a product costs %5.2f ns\n", run<ALT, 2>(w, n_cu, cu_lds, d_out), (run<ALT, 2>(w, n_cu, cu_lds, d_out) - v) / 32);
            printf("  %8.1f  alt, priority 3 during a wave's multiplies: a product costs %5.2f ns\n", run<ALT, 3>(w, n_cu, cu_lds, d_out), (run<ALT, 3>(w, n_cu, cu_lds, d_out) - v) / 32);
            printf("  %8.1f  spread, the odd slot at priority 3: a product costs %5.2f ns\n", run<SPREAD, 1>(w, n_cu, cu_lds, d_out), (run<SPREAD, 1>(w, n_cu, cu_lds, d_out) - v) / 32);
        }
        if (w == 2) {
            printf("    forced anti-phase (blocks of eight waves, waves w and w + 4 on one SIMD):\n");
            const double t0 = run_anti<0>(n_cu, d_out), t1 = run_anti<1>(n_cu, d_out), t4 = run_anti<4>(n_cu, d_out);
            printf("  %8.1f  alt, started in anti-phase, no barrier: a product costs %5.2f ns\n", t0, (t0 - v) / 32);
            printf("  %8.1f  alt, a workgroup barrier per unit: a product costs %5.2f ns\n", t1, (t1 - v) / 32);
            printf("  %8.1f  alt, a workgroup barrier per 4 units: a product costs %5.2f ns\n", t4, (t4 - v) / 32);
        }
        line(CHAINS, run<CHAINS>(w, n_cu, cu_lds, d_out));
        line(FOUR, run<FOUR>(w, n_cu, cu_lds, d_out));
        line(SPREAD, run<SPREAD>(w, n_cu, cu_lds, d_out));
        line(SPREAD_CHAINS, run<SPREAD_CHAINS>(w, n_cu, cu_lds, d_out));
        printf("    accumulators in accumulation registers:\n");
        printf("  %8.1f  matrix only\n", run_acc<MFMA_ONLY>(w, n_cu, cu_lds, d_out));
        { const double t = run_acc<ALT>(w, n_cu, cu_lds, d_out); printf("  %8.1f  alt: a product costs %5.2f ns\n", t, (t - v) / 32); }
        { const double t = run_acc<SPREAD>(w, n_cu, cu_lds, d_out); printf("  %8.1f  spread: a product costs %5.2f ns\n", t, (t - v) / 32); }
        printf("    the A operand in accumulation registers (a load can land there): alt %5.2f ns, spread %5.2f ns per product\n",
               (run_acc<ALT, 2>(w, n_cu, cu_lds, d_out) - v) / 32, (run_acc<SPREAD, 2>(w, n_cu, cu_lds, d_out) - v) / 32);
        printf("    A and B there: alt %5.2f, spread %5.2f;  A, B and the accumulators: alt %5.2f, spread %5.2f\n", (run_acc<ALT, 6>(w, n_cu, cu_lds, d_out) - v) / 32,
               (run_acc<SPREAD, 6>(w, n_cu, cu_lds, d_out) - v) / 32, (run_acc<ALT, 7>(w, n_cu, cu_lds, d_out) - v) / 32, (run_acc<SPREAD, 7>(w, n_cu, cu_lds, d_out) - v) / 32);
        const double ms_ = run<SMALL_ONLY>(w, n_cu, cu_lds, d_out);
        printf("  %8.1f  %s  (%.2f ns per two = the multiply-adds of one 32x32x32)\n", ms_, kNames[SMALL_ONLY], ms_ / 32);
        line(SMALL_ALT, run<SMALL_ALT>(w, n_cu, cu_lds, d_out));
        line(SMALL_SPREAD, run<SMALL_SPREAD>(w, n_cu, cu_lds, d_out));
    }
    printf("--- who yields to whom: a wave that only issues products beside a wave that only multiplies, on one SIMD (each alone: ~1460 ns per unit)\n");
    run_split<0>(n_cu, d_out);
    run_split<1>(n_cu, d_out);
    CHECK(hipFree(d_out));
    return 0;
}
