// What does ONE wave alone on its SIMD pay per instruction?  The narrow levels of a Merkle tree (and a single sponge) run
// that way: one dependent chain per wave, nothing else resident.  The kernel times (s_memtime) unrolled dependent chains
// written in plain C++ - compiled like the product kernels, carry-out of the multiplies in an allocator-chosen SGPR pair -
// with `waves` waves per SIMD on every CU, for: a chain of v_mad_u64_u32; the same chain with a full-rate ALU instruction,
// a DPP move or a 64-bit shift after every multiply; and the product's own mont_mul / mont_sqr / tab_dot<3>.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -opt-disable=reassociate -I sponge_amd/csrc tools/lone_wave_microbench.hip -o tools/lone_wave_microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pmx_field.hpp"

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

using namespace pmx;

enum Mix { MAD, MAD_AND, MAD_DPP, MAD_SHIFT64, MAD_MULLO, MAD_AND2, MONT_MUL, MONT_SQR, TAB_DOT3, N_MIX };
static const char *kNames[N_MIX] = {"mad chain", "mad + v_and", "mad + dpp mov", "mad + v_lshrrev_b64", "mad + v_mul_lo_u32",
                                    "mad + 2 x v_and", "mont_mul chain", "mont_sqr chain", "tab_dot<3> chain"};
// VALU instructions per unrolled step (for the table): multiply first, then the riders
static const int kInstr[N_MIX] = {1, 2, 2, 2, 2, 3, 0, 0, 0};

struct Args {
    FieldRt f;
    uint32_t tab[kN * 32 + 32];   // enough for tab_row_words(3)
};

template <int MIX>
__global__ void __launch_bounds__(256) bench(const Args args, const uint32_t *__restrict__ io, uint32_t *out, unsigned long long *cycles, int trips, uint32_t seed) {
    FieldRt f = args.f;
    f.io = io;
    uint64_t acc = seed + threadIdx.x;
    uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9E3779B9u, z = seed;
    Fe s;
#pragma unroll
    for (int i = 0; i < kN; ++i) s.l[i] = (x + i * 977u) & kMask;
    Fe c = s;
    c.l[0] ^= 5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < trips; ++it) {
        if constexpr (MIX == MONT_MUL) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s = mont_mul(s, c, f);
        } else if constexpr (MIX == MONT_SQR) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s = mont_sqr(s, f);
        } else if constexpr (MIX == TAB_DOT3) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                Fe zz[3] = {s, c, s};
                s = tab_dot<3, false>(zz, io + 64, s, f);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 128; ++u) {
                acc += (uint64_t)x * y;
                if constexpr (MIX == MAD_AND) x = (uint32_t)acc & y;
                if constexpr (MIX == MAD_AND2) { x = (uint32_t)acc & y; z = (x & z) + 1; y ^= z; }
                if constexpr (MIX == MAD_DPP) x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)acc, 0x55, 0xf, 0xf, false);
                if constexpr (MIX == MAD_SHIFT64) acc = (acc >> 29) | 1;
                if constexpr (MIX == MAD_MULLO) x = (uint32_t)acc * y;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = (uint32_t)acc ^ (uint32_t)(acc >> 32) ^ x ^ z;
#pragma unroll
    for (int i = 0; i < kN; ++i) r ^= s.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MIX>
void run(int blocks, int threads, int trips, const Args &a, const uint32_t *d_io, uint32_t *d_out, unsigned long long *d_cyc, double clock_per_tick) {
    hipLaunchKernelGGL(bench<MIX>, dim3(blocks), dim3(threads), 0, 0, a, d_io, d_out, d_cyc, trips / 4 + 1, 1u);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<MIX>, dim3(blocks), dim3(threads), 0, 0, a, d_io, d_out, d_cyc, trips, 1u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int waves = blocks * threads / 64;
    std::vector<unsigned long long> cyc(waves);
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : cyc) mean += (double)c;
    mean /= cyc.size();
    const bool field = MIX >= MONT_MUL;
    const double steps = (field ? 4.0 : 128.0) * trips;
    printf("%-22s blocks=%5d x %3d  %8.3f ms  %9.1f ticks/step  = %8.1f shader cycles/step", kNames[MIX], blocks, threads, ms, mean / steps,
           mean / steps * clock_per_tick);
    if (!field) printf("  (%d VALU instr/step: %.2f cycles each)", kInstr[MIX], mean / steps * clock_per_tick / kInstr[MIX]);
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, n_cu, prop.clockRate);
    // s_memtime counts shader clocks on this part (csrc/pmx_diag.hip derives the clock from its ratio to s_memrealtime)
    const double clock_per_tick = 1.0;
    Args a{};
    // BLS12-381 Fr in 29-bit limbs (values only matter for timing: any odd p works)
    const uint32_t p29[kN] = {0x00000001, 0x1ffffff8, 0x1f96ffbf, 0x1b4805ff, 0x04ec0404, 0x0fa91a33, 0x14a6533b, 0x1d3a9d4c, 0x0073eda7};
    for (int i = 0; i < kN; ++i) a.f.p[i] = p29[i];
    a.f.pinv = kMask;   // -1 mod 2^29
    a.f.unit = 1;
    std::vector<uint32_t> io(64 + 9 * 32 + 32, 0x01234567u & kMask);
    uint32_t *d_io, *d_out;
    unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_io, io.size() * 4));
    CHECK(hipMemcpy(d_io, io.data(), io.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 16 * 256 * 4));
    CHECK(hipMalloc(&d_cyc, (size_t)n_cu * 16 * 4 * 8));
    const int trips = 2000;
    struct Shape { int blocks, threads; const char *what; };
    const Shape shapes[] = {{1, 64, "one wave on the whole device"}, {n_cu, 256, "one wave per SIMD"}, {2 * n_cu, 256, "two waves per SIMD"},
                            {4 * n_cu, 256, "four waves per SIMD"}};
    for (const Shape &s : shapes) {
        printf("---- %s\n", s.what);
        run<MAD>(s.blocks, s.threads, trips, a, d_io, d_out, d_cyc, clock_per_tick);
        run<MAD_AND>(s.blocks, s.threads, trips, a, d_io, d_out, d_cyc, clock_per_tick);
        run<MAD_AND2>(s.blocks, s.threads, trips, a, d_io, d_out, d_cyc, clock_per_tick);
        run<MAD_DPP>(s.blocks, s.threads, trips, a, d_io, d_out, d_cyc, clock_per_tick);
        run<MAD_SHIFT64>(s.blocks, s.threads, trips, a, d_io, d_out, d_cyc, clock_per_tick);
        run<MAD_MULLO>(s.blocks, s.threads, trips, a, d_io, d_out, d_cyc, clock_per_tick);
        run<MONT_MUL>(s.blocks, s.threads, trips, a, d_io, d_out, d_cyc, clock_per_tick);
        run<MONT_SQR>(s.blocks, s.threads, trips, a, d_io, d_out, d_cyc, clock_per_tick);
        run<TAB_DOT3>(s.blocks, s.threads, trips, a, d_io, d_out, d_cyc, clock_per_tick);
    }
    return 0;
}
