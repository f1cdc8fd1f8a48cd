// Instruction-throughput microbenchmark for the integer/FP64 VALU ops a 256-bit Montgomery multiplier can
// be built from on gfx950.  It calibrates the "int-VALU roofline" DESIGN.md prices the permutation kernel
// against (SURVEY.md section 8d asks for this measurement instead of an assumed quarter rate).
//
//   hipcc -O3 --offload-arch=gfx950 tools/valu_microbench.hip -o tools/valu_microbench && tools/valu_microbench
//
// For every op: 8 independent dependency chains per lane, 64 instructions per loop trip, `waves` waves per
// SIMD on every CU.  Reports cycles per wave-instruction per SIMD (from s_memtime, so DVFS-independent)
// and chip-wide G lane-ops/s (from hipEvents).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));             \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

enum Op { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, ADD_CO_ADDC, LSHL_ADD_U64, FMA_F64, FMA_F32, MAD_U32_U24,
          MUL_HI_U32_U24, ADD_U32, ADD_F64, MUL_F64, CVT_F64_U32, MAD_U64_SGPR, N_OPS };
static const char *kNames[N_OPS] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_co+v_addc_co (pair)",
                                    "v_lshl_add_u64", "v_fma_f64", "v_fma_f32", "v_mad_u32_u24",
                                    "v_mul_hi_u32_u24", "v_add_u32", "v_add_f64", "v_mul_f64", "v_cvt_f64_u32",
                                    "v_mad_u64_u32 (sgpr src)"};
static const int kInstrPerStep[N_OPS] = {1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

template <int OP>
__device__ __forceinline__ void step(unsigned long long &a64, unsigned &a32, double &d, float &f, unsigned x, unsigned y,
                                     double dx, unsigned sx) {
    if constexpr (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a64) : "v"(x), "v"(y) : "vcc");
    if constexpr (OP == MAD_U64_SGPR) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a64) : "v"(x), "s"(sx) : "vcc");
    if constexpr (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a32) : "v"(x));
    if constexpr (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a32) : "v"(x));
    if constexpr (OP == ADD_CO_ADDC) {
        unsigned lo = (unsigned)a64, hi = (unsigned)(a64 >> 32);
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(x), "v"(y) : "vcc");
        a64 = ((unsigned long long)hi << 32) | lo;
    }
    if constexpr (OP == LSHL_ADD_U64) {
        unsigned long long b = ((unsigned long long)y << 32) | x;
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a64) : "v"(b));
    }
    if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d) : "v"(dx));
    if constexpr (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(dx));
    if constexpr (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(dx));
    if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(f) : "v"(f));
    if constexpr (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a32) : "v"(x), "v"(y));
    if constexpr (OP == MUL_HI_U32_U24) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a32) : "v"(x));
    if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a32) : "v"(x));
    if constexpr (OP == CVT_F64_U32) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d) : "v"(a32));
}

template <int OP>
__global__ void __launch_bounds__(256) bench(unsigned *out, unsigned long long *cycles, int trips, unsigned seed, unsigned sx) {
    unsigned long long a64[8];
    unsigned a32[8];
    double d[8];
    float f[8];
    const unsigned x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9E3779B9u;
    const double dx = 1.0 + 1e-9 * threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a64[k] = x + k;
        a32[k] = y + k;
        d[k] = dx + k;
        f[k] = (float)k + 0.5f;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < trips; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) step<OP>(a64[k], a32[k], d[k], f[k], x, y, dx, sx);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= (unsigned)a64[k] ^ (unsigned)(a64[k] >> 32) ^ a32[k] ^ (unsigned)d[k] ^ (unsigned)f[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
void run(int waves_per_simd, int n_cu, int trips, unsigned *d_out, unsigned long long *d_cyc) {
    const int blocks = n_cu * waves_per_simd;  // 256 threads = 4 waves = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, trips / 8, 1u, 12345u);  // warm
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, trips, 1u, 12345u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(blocks * 4);
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : cyc) mean += (double)c;
    mean /= cyc.size();
    const double steps = 64.0 * trips;                          // steps per lane
    const double instr = steps * kInstrPerStep[OP];
    // s_memtime ticks at 100 MHz-independent "shader clock" on gfx9: report ticks per wave-instruction per SIMD
    const double ticks_per_instr_simd = mean / instr / waves_per_simd;
    const double lane_ops = (double)blocks * 256 * instr;
    printf("%-28s waves/SIMD=%d  %8.3f ms  %7.2f ticks/instr/SIMD  %8.1f G lane-instr/s\n", kNames[OP], waves_per_simd, ms,
           ticks_per_instr_simd, lane_ops / ms / 1e6);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, n_cu, prop.clockRate);
    unsigned *d_out;
    unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4));
    CHECK(hipMalloc(&d_cyc, (size_t)n_cu * 8 * 4 * 8));
    const int trips = 4096;
    for (int w : {1, 2, 4}) {
        run<MAD_U64_U32>(w, n_cu, trips, d_out, d_cyc);
        run<MAD_U64_SGPR>(w, n_cu, trips, d_out, d_cyc);
        run<MUL_LO_U32>(w, n_cu, trips, d_out, d_cyc);
        run<MUL_HI_U32>(w, n_cu, trips, d_out, d_cyc);
        run<ADD_CO_ADDC>(w, n_cu, trips, d_out, d_cyc);
        run<LSHL_ADD_U64>(w, n_cu, trips, d_out, d_cyc);
        run<ADD_U32>(w, n_cu, trips, d_out, d_cyc);
        run<MAD_U32_U24>(w, n_cu, trips, d_out, d_cyc);
        run<MUL_HI_U32_U24>(w, n_cu, trips, d_out, d_cyc);
        run<FMA_F32>(w, n_cu, trips, d_out, d_cyc);
        run<FMA_F64>(w, n_cu, trips, d_out, d_cyc);
        run<ADD_F64>(w, n_cu, trips, d_out, d_cyc);
        run<MUL_F64>(w, n_cu, trips, d_out, d_cyc);
        run<CVT_F64_U32>(w, n_cu, trips, d_out, d_cyc);
    }
    return 0;
}
