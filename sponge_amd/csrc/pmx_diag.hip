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

#include "../../include/poseidon_mi355x.h"
#include "pmx_ctx.hpp"

using namespace pmx;

namespace {

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

extern "C" int pmx_diag_int_valu_peak(int device, double seconds, pmx_valu_peak *out) {
    if (!out) return set_error(PMX_ERR_ARG, "pmx_diag_int_valu_peak: null pointer");
    *out = pmx_valu_peak{};
    const int ndev = pmx_device_count();
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
}
