// Do simple VALU instructions regain their 2.2-clock slot in long multiply-free phases?  tools/issue_model_microbench.hip found that in any
// stream that HOLDS multiplies a SIMD takes one VALU instruction per ~4 clocks whatever the instruction is - with multiplies and simple
// instructions interleaved 4 + 12 or 12 + 4.  The kernels' simple instructions come in longer runs (a row finish: ~90, an operand cut: 27
// per element) between S-boxes of ~600 instructions of which 414 multiply.  This probe alternates PHASES of N v_mad_u64_u32 and N simple
// instructions (v_and_b32 / v_add_u32), N = 1 ... 1024, with exactly k waves resident per SIMD (pinned by the LDS a block asks for), and
// prints the nanoseconds per instruction and SIMD next to what the two pure streams would give if each kept its own rate.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/phase_issue_microbench.hip -o tools/phase_issue_microbench
#include <hip/hip_runtime.h>

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

extern __shared__ uint4 lds_pin[];

#define MAD4 "v_mad_u64_u32 %[a0], vcc, %[x0], %[y], %[a0]\n\tv_mad_u64_u32 %[a1], vcc, %[x1], %[y], %[a1]\n\tv_mad_u64_u32 %[a2], vcc, %[x2], %[y], %[a2]\n\tv_mad_u64_u32 %[a3], vcc, %[x3], %[y], %[a3]\n\t"
#define AND4 "v_and_b32 %[s0], %[s0], %[y]\n\tv_add_u32 %[s1], %[s1], %[y]\n\tv_and_b32 %[s2], %[s2], %[y]\n\tv_add_u32 %[s3], %[s3], %[y]\n\t"
#define R4(x) x x x x
#define OPS : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [s0] "+v"(s0), [s1] "+v"(s1), [s2] "+v"(s2), [s3] "+v"(s3) : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [y] "v"(y) : "vcc"

#define R2(x) x x
#define R8(x) R4(R2(x))
#define R16(x) R4(R4(x))
#define R32(x) R8(R4(x))
#define R64(x) R16(R4(x))
#define R128(x) R32(R4(x))
#define R256(x) R64(R4(x))
#define MAD1A "v_mad_u64_u32 %[a0], vcc, %[x0], %[y], %[a0]\n\tv_and_b32 %[s0], %[s0], %[y]\n\tv_mad_u64_u32 %[a1], vcc, %[x1], %[y], %[a1]\n\tv_add_u32 %[s1], %[s1], %[y]\n\t"
#define MAD1B "v_mad_u64_u32 %[a2], vcc, %[x2], %[y], %[a2]\n\tv_and_b32 %[s2], %[s2], %[y]\n\tv_mad_u64_u32 %[a3], vcc, %[x3], %[y], %[a3]\n\tv_add_u32 %[s3], %[s3], %[y]\n\t"

// One loop body is 1024 instructions whatever the phase length (2048 for N = 1024), so the loop's own branch weighs the same on every
// line.  MODE = the phase length N (1, 4, 16, 64, 256, 1024); -1: multiplies only; -2: simple instructions only.
template <int MODE>
__global__ void __launch_bounds__(256) bench(uint32_t *out, int bodies) {
    uint64_t a0 = threadIdx.x + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    uint32_t x0 = threadIdx.x | 1, x1 = x0 + 2, x2 = x0 + 4, x3 = x0 + 6, y = 0x9e3779b9u * (blockIdx.x + 1) | 1;
    uint32_t s0 = x0, s1 = x1, s2 = x2, s3 = x3;
    lds_pin[threadIdx.x] = make_uint4(x0, x1, x2, x3);
    for (int p = 0; p < bodies; ++p) {
        if constexpr (MODE == -1) asm volatile(R256(MAD4) OPS);
        if constexpr (MODE == -2) asm volatile(R256(AND4) OPS);
        if constexpr (MODE == 1) asm volatile(R128(MAD1A MAD1B) OPS);
        if constexpr (MODE == 4) asm volatile(R128(MAD4 AND4) OPS);
        if constexpr (MODE == 16) asm volatile(R32(R4(MAD4) R4(AND4)) OPS);
        if constexpr (MODE == 64) asm volatile(R8(R16(MAD4) R16(AND4)) OPS);
        if constexpr (MODE == 256) asm volatile(R2(R64(MAD4) R64(AND4)) OPS);
        if constexpr (MODE == 1024) asm volatile(R256(MAD4) R256(AND4) OPS);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3) ^ (uint32_t)((a0 ^ a1 ^ a2 ^ a3) >> 32) ^ s0 ^ s1 ^ s2 ^ s3 ^ lds_pin[0].x;
}

template <int MODE>
static double run(int blocks, size_t lds, uint32_t *d_out, int bodies) {
    CHECK(hipFuncSetAttribute((const void *)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), lds, 0, d_out, bodies);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;   // (the first launches run under the clock ramp)
    }
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    int cu_lds = 160 * 1024;
    (void)hipDeviceGetAttribute(&cu_lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0);
    uint32_t *d_out;
    CHECK(hipMalloc((void **)&d_out, (size_t)n_cu * 8 * 256 * 4));
    printf("device %s, %d CUs.  ns per VALU instruction and SIMD; a phase pair = N multiplies (4 chains) then N simple instructions\n", prop.gcnArchName, n_cu);
    for (int i = 0; i < 40; ++i) (void)run<-1>(n_cu * 4, 39 * 1024, d_out, 256);   // spin-up: the clock ramps over the first ~0.5 s of load
    for (int waves : {2, 3, 4}) {
        const int blocks = n_cu * waves;
        size_t lds = (size_t)(cu_lds / waves) - 1024;
        int block_lds = 0;
        if (hipDeviceGetAttribute(&block_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, 0) == hipSuccess && block_lds > 0 && lds > (size_t)block_lds) lds = (size_t)block_lds;
        printf("--- %d waves per SIMD\n", waves);
        const int bodies = 512;   // 512 x 1024 = 524288 instructions per lane and launch
        const double instr = 1024.0 * bodies;
        auto ns = [&](double ms, double scale) { return ms * 1e6 / (instr * scale * waves); };
        const double t_mad = ns(run<-1>(blocks, lds, d_out, bodies), 1), t_and = ns(run<-2>(blocks, lds, d_out, bodies), 1);
        printf("    multiplies only                         %6.3f ns\n", t_mad);
        printf("    simple instructions only                %6.3f ns\n", t_and);
        printf("    (if each phase kept its own rate: %6.3f ns per instruction of a 1 : 1 mix)\n", (t_mad + t_and) / 2);
        auto line = [&](int n, double t) { printf("    phases of N = %4d multiplies, %4d simple %6.3f ns   -> a simple instruction costs %6.3f ns\n", n, n, t, 2 * t - t_mad); };
        line(1, ns(run<1>(blocks, lds, d_out, bodies), 1));
        line(4, ns(run<4>(blocks, lds, d_out, bodies), 1));
        line(16, ns(run<16>(blocks, lds, d_out, bodies), 1));
        line(64, ns(run<64>(blocks, lds, d_out, bodies), 1));
        line(256, ns(run<256>(blocks, lds, d_out, bodies), 1));
        line(1024, ns(run<1024>(blocks, lds, d_out, bodies / 2), 1));
        line(16, ns(run<16>(blocks, lds, d_out, bodies), 1));   // (again: the order must not matter)
    }
    return 0;
}
