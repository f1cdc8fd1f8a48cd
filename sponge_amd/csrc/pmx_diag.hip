// Diagnostics: the integer-VALU roofline of THIS device, measured when asked.
//
// The permutation kernels are bound by the issue rate of v_mad_u64_u32 (32 x 32 + 64 -> 64), the instruction every limb
// product is.  That rate depends on the clock the chip holds under load, which differs from
// box to box and run to run, so bench.py does not price its kernels against a constant: it calls this before timing
// anything.  The loop is the densest possible stream of the instruction - 8 independent chains per lane, 4 waves per
// SIMD on every CU, nothing else in the loop body - so no kernel built from that instruction can issue it faster.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include <cstdarg>
#include <cstdio>
#include <exception>

#include "../../include/poseidon_mi355x.h"        // (the status codes only)
#include "../../include/poseidon_mi355x_diag.h"

// libposeidon_mi355x_diag.so: a library of its own (round 6) - the shipped libposeidon_mi355x.so carries no benchmark instrumentation.
namespace {

thread_local char g_diag_error[512] = "";
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_diag_error, sizeof g_diag_error, fmt, ap);
    va_end(ap);
    return code;
}
int hip_fail(hipError_t e, const char *what) { return set_error(PMX_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); }
#define PMX_HIP(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)
int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
// the caller's current device is put back on the way out
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
// nothing unwinds through the C ABI (std::vector below)
template <class F>
int guarded(const char *who, F &&body) noexcept {
    try {
        return body();
    } catch (const std::exception &e) {
        return set_error(PMX_ERR_HOST, "%s: %s", who, e.what());
    } catch (...) {
        return set_error(PMX_ERR_HOST, "%s: unknown exception", who);
    }
}

// CARRY_IN_VCC: where the instruction's (unused) carry-out goes - VCC, or an SGPR pair the register allocator picks, which
// is what compiled kernels use.  Both forms are timed; the faster one is the peak.
template <bool CARRY_IN_VCC>
__global__ void __launch_bounds__(256) mad_chain_kernel(unsigned *out, unsigned long long *stamps, int trips, unsigned seed) {
    unsigned long long a[8];
    const unsigned x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9E3779B9u;
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = x + k;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < trips; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (CARRY_IN_VCC) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y) : "vcc");
                } else {
                    unsigned long long carry;
                    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(a[k]), "=s"(carry) : "v"(x), "v"(y));
                }
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= (unsigned)a[k] ^ (unsigned)(a[k] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;       // shader-clock ticks
        stamps[2 * blockIdx.x + 1] = r1 - r0;   // 100 MHz ticks
    }
}

}  // namespace

extern "C" const char *pmx_diag_last_error(void) { return g_diag_error; }

extern "C" int pmx_diag_int_valu_peak(int device, double seconds, pmx_valu_peak *out) {
    return guarded("pmx_diag_int_valu_peak", [&]() -> int {
    if (!out) return set_error(PMX_ERR_ARG, "pmx_diag_int_valu_peak: null pointer");
    *out = pmx_valu_peak{};
    const int ndev = device_count();
    if (ndev == 0) return set_error(PMX_ERR_HIP, "no HIP device available; this library has no CPU fallback");
    if (device < 0 || device >= ndev) return set_error(PMX_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    if (!(seconds > 0)) seconds = 0.02;
    if (seconds > 5) seconds = 5;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipDeviceProp_t prop;
    PMX_HIP(hipGetDeviceProperties(&prop, device));
    const int n_cu = prop.multiProcessorCount, blocks = n_cu * 4;   // 256 threads = one wave per SIMD of a CU; x4
    const int trips = 1024;                                         // 65,536 multiplies per lane and launch (~0.6 ms)
    unsigned *d_out = nullptr;
    unsigned long long *d_stamps = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void **)&d_out, (size_t)blocks * 256 * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_stamps, (size_t)blocks * 16);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    std::vector<double> rates[2];
    int launches = 0;
    for (int form = 0; form < 2 && e == hipSuccess; ++form) {
        const auto t_begin = std::chrono::steady_clock::now();
        int n_form = 0;
        while (e == hipSuccess) {
            e = hipEventRecord(e0, st);
            if (form == 0) hipLaunchKernelGGL(mad_chain_kernel<true>, dim3(blocks), dim3(256), 0, st, d_out, d_stamps, trips, 1u + launches);
            else hipLaunchKernelGGL(mad_chain_kernel<false>, dim3(blocks), dim3(256), 0, st, d_out, d_stamps, trips, 1u + launches);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(e1, st);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) break;
            rates[form].push_back((double)blocks * 256 * 64.0 * trips / (ms * 1e-3));
            ++launches;
            ++n_form;
            const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
            if (elapsed >= seconds / 2 && n_form >= 4) break;
        }
    }
    std::vector<unsigned long long> stamps((size_t)blocks * 2);
    if (e == hipSuccess) e = hipMemcpy(stamps.data(), d_stamps, stamps.size() * 8, hipMemcpyDeviceToHost);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (d_out) (void)hipFree(d_out);
    if (d_stamps) (void)hipFree(d_stamps);
    if (e != hipSuccess) return hip_fail(e, "pmx_diag_int_valu_peak");
    // the clock ramps over the first launches: per form the figure is the median of the second half of its window
    double med[2] = {0, 0}, best = 0;
    for (int form = 0; form < 2; ++form) {
        std::vector<double> tail(rates[form].begin() + rates[form].size() / 2, rates[form].end());
        std::sort(tail.begin(), tail.end());
        med[form] = tail[tail.size() / 2];
        best = std::max(best, *std::max_element(rates[form].begin(), rates[form].end()));
    }
    out->lane_mads_per_s = std::max(med[0], med[1]);
    out->lane_mads_per_s_vcc = med[0];
    out->lane_mads_per_s_sgpr = med[1];
    out->best_lane_mads_per_s = best;
    std::vector<double> clocks;
    for (int b = 0; b < blocks; ++b)
        if (stamps[2 * b + 1]) clocks.push_back((double)stamps[2 * b] / (double)stamps[2 * b + 1] * 100e6);
    std::sort(clocks.begin(), clocks.end());
    out->shader_clock_hz = clocks.empty() ? 0.0 : clocks[clocks.size() / 2];   // of the last launch
    out->compute_units = n_cu;
    out->launches = launches;
    // 4 SIMDs per CU, each retiring 16 lanes of this half-rate instruction per clock
    out->theoretical_lane_mads_per_s = (double)n_cu * 4 * 16 * out->shader_clock_hz;
    return PMX_OK;
    });
}

// ---- the VALU issue slot, measured in the run ----------------------------------------------------------------------------
// A SIMD of this part takes one VALU instruction per ~4 shader clocks from any stream that contains multiplies, whatever
// the instruction is (tools/issue_model_microbench.hip), so a kernel's floor is its VALU instruction
// count times that slot.  The slot depends on the clock the chip holds under the stream, so - like the multiply peak above -
// it is measured on this device in this run, at the occupancy of the kernel it prices: exactly `waves_per_simd` waves resident
// on every SIMD (pinned by the LDS each block asks for; the dispatcher alone does not spread a launch evenly), three
// calibration streams written as single asm statements, timed by the wall clock.
namespace {
#define PMX_R2(x) x "\n\t" x
#define PMX_R4(x) PMX_R2(x) "\n\t" PMX_R2(x)
#define PMX_R16(x) PMX_R4(x) "\n\t" PMX_R4(x) "\n\t" PMX_R4(x) "\n\t" PMX_R4(x)
#define PMX_MAD4 "v_mad_u64_u32 %0, vcc, %8, %10, %0\n\tv_mad_u64_u32 %1, vcc, %9, %10, %1\n\tv_mad_u64_u32 %2, vcc, %8, %10, %2\n\tv_mad_u64_u32 %3, vcc, %9, %10, %3"
#define PMX_AND4 "v_and_b32 %4, %4, %10\n\tv_and_b32 %5, %5, %10\n\tv_and_b32 %6, %6, %10\n\tv_and_b32 %7, %7, %10"
// MIX 0: 12 multiplies then 4 simple instructions (the permutation kernels' own mix: three multiplies per other instruction)
// MIX 1: 4 multiplies then 12 simple instructions;  MIX 2: 16 multiplies.  16 instructions per step, 16 steps per trip.
template <int MIX>
__global__ void __launch_bounds__(256) issue_stream_kernel(unsigned *out, int trips, unsigned seed) {
    extern __shared__ unsigned pin[];   // only its size matters
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    unsigned x0 = threadIdx.x * 2654435761u + seed, x1 = x0 ^ 0x9E3779B9u, x2 = x0 + 77, x3 = x1 + 99;
    const unsigned y = seed | 0x10001u, z0 = x0 ^ 0x55, z1 = x1 ^ 0xaa;
    for (int it = 0; it < trips; ++it) {
        if constexpr (MIX == 0) {
            asm volatile(PMX_R16(PMX_MAD4 "\n\t" PMX_MAD4 "\n\t" PMX_MAD4 "\n\t" PMX_AND4)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(z0), "v"(z1), "v"(y) : "vcc");
        } else if constexpr (MIX == 1) {
            asm volatile(PMX_R16(PMX_MAD4 "\n\t" PMX_AND4 "\n\t" PMX_AND4 "\n\t" PMX_AND4)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(z0), "v"(z1), "v"(y) : "vcc");
        } else {
            asm volatile(PMX_R16(PMX_MAD4 "\n\t" PMX_MAD4 "\n\t" PMX_MAD4 "\n\t" PMX_MAD4)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(z0), "v"(z1), "v"(y) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a0 ^ a1 ^ a2 ^ a3) ^ (unsigned)((a0 ^ a1 ^ a2 ^ a3) >> 32) ^ x0 ^ x1 ^ x2 ^ x3 ^ pin[0];
}
}  // namespace

extern "C" int pmx_diag_issue_slot(int device, int waves_per_simd, double seconds, pmx_issue_slot *out) {
    return guarded("pmx_diag_issue_slot", [&]() -> int {
    if (!out) return set_error(PMX_ERR_ARG, "pmx_diag_issue_slot: null pointer");
    *out = pmx_issue_slot{};
    const int ndev = device_count();
    if (ndev == 0) return set_error(PMX_ERR_HIP, "no HIP device available; this library has no CPU fallback");
    if (device < 0 || device >= ndev) return set_error(PMX_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    if (waves_per_simd < 1 || waves_per_simd > 8) return set_error(PMX_ERR_ARG, "waves_per_simd %d out of range [1,8]", waves_per_simd);
    if (!(seconds > 0)) seconds = 0.03;
    if (seconds > 5) seconds = 5;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipDeviceProp_t prop;
    PMX_HIP(hipGetDeviceProperties(&prop, device));
    const int n_cu = prop.multiProcessorCount, blocks = n_cu * waves_per_simd;   // a block = one wave on each SIMD of a CU
    int cu_lds = 0;
    if (hipDeviceGetAttribute(&cu_lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) != hipSuccess || cu_lds <= 0) cu_lds = 160 * 1024;
    size_t lds = (size_t)(cu_lds / waves_per_simd) - 1024;                       // k blocks fit a CU, k + 1 do not
    int block_lds = 0;                                                           // (one block per CU: no more than a workgroup may ask for - still more than half the CU's)
    if (hipDeviceGetAttribute(&block_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && block_lds > 0 && lds > (size_t)block_lds) lds = (size_t)block_lds;
    const int trips = 256;                                                       // 65,536 instructions per lane and launch
    unsigned *d_out = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void **)&d_out, (size_t)blocks * 256 * 4);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    const void *kernels[3] = {(const void *)issue_stream_kernel<0>, (const void *)issue_stream_kernel<1>, (const void *)issue_stream_kernel<2>};
    for (int m = 0; m < 3 && e == hipSuccess; ++m) e = hipFuncSetAttribute(kernels[m], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<double> ns[3];
    int launches = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    while (e == hipSuccess) {   // the three streams in turn, so that all of them see the same clock ramp
        for (int m = 0; m < 3 && e == hipSuccess; ++m) {
            e = hipEventRecord(e0, st);
            if (m == 0) hipLaunchKernelGGL(issue_stream_kernel<0>, dim3(blocks), dim3(256), lds, st, d_out, trips, 1u + launches);
            else if (m == 1) hipLaunchKernelGGL(issue_stream_kernel<1>, dim3(blocks), dim3(256), lds, st, d_out, trips, 1u + launches);
            else hipLaunchKernelGGL(issue_stream_kernel<2>, dim3(blocks), dim3(256), lds, st, d_out, trips, 1u + launches);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(e1, st);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) break;
            // instructions one SIMD issued: waves_per_simd waves x trips x 16 steps x 16 instructions
            ns[m].push_back(ms * 1e6 / ((double)waves_per_simd * trips * 256.0));
            ++launches;
        }
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        if (elapsed >= seconds && ns[0].size() >= 4) break;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return hip_fail(e, "pmx_diag_issue_slot");
    double med[3];
    for (int m = 0; m < 3; ++m) {
        std::vector<double> tail(ns[m].begin() + ns[m].size() / 2, ns[m].end());   // past the clock ramp
        std::sort(tail.begin(), tail.end());
        med[m] = tail[tail.size() / 2];
    }
    out->ns_12mad_4simple = med[0];
    out->ns_4mad_12simple = med[1];
    out->ns_16mad = med[2];
    out->ns_floor = std::min(med[0], std::min(med[1], med[2]));
    out->waves_per_simd = waves_per_simd;
    out->compute_units = n_cu;
    out->launches = launches;
    return PMX_OK;
    });
}
