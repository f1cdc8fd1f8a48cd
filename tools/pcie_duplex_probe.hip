// What the link gives a round trip of 96 MiB each way (2^20 t = 3 states): SDMA copies and kernels that read / write a page-locked host
// buffer themselves, alone and in pairs.  The host path's pipeline (pmx_api.cpp: host_pipeline) is priced against the best pair.
//   hipcc -O2 --offload-arch=gfx950 tools/pcie_duplex_probe.hip -o tools/pcie_duplex_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void copy16(const uint4 *in, uint4 *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main() {
    const size_t bytes = 96ull << 20, n = bytes / 16;
    void *h_in, *h_out, *d_a, *d_b;
    CK(hipHostMalloc(&h_in, bytes, hipHostMallocDefault)); CK(hipHostMalloc(&h_out, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d_a, bytes)); CK(hipMalloc(&d_b, bytes));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); double t_h2d = now() - t0;
        t0 = now(); CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); double t_d2h = now() - t0;
        for (int blocks : {256, 1024, 4096}) {
            t0 = now(); copy16<<<blocks, 256, 0, s0>>>((const uint4 *)d_b, (uint4 *)h_out, n); CK(hipStreamSynchronize(s0)); double t_kw = now() - t0;
            t0 = now(); copy16<<<blocks, 256, 0, s0>>>((const uint4 *)h_in, (uint4 *)d_a, n); CK(hipStreamSynchronize(s0)); double t_kr = now() - t0;
            t0 = now(); CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s1)); copy16<<<blocks, 256, 0, s0>>>((const uint4 *)d_b, (uint4 *)h_out, n);
            CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); double t_both = now() - t0;
            printf("rep %d blocks %4d: kernel writes host %.2f ms, kernel reads host %.2f ms, SDMA upload beside kernel writing host %.2f ms\n", rep, blocks, t_kw * 1e3, t_kr * 1e3, t_both * 1e3);
        }
        t0 = now(); CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s1)); copy16<<<1024, 256, 0, s0>>>((const uint4 *)h_in, (uint4 *)d_a, n);
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); printf("rep %d: SDMA download beside kernel reading host %.2f ms\n", rep, (now() - t0) * 1e3);
        t0 = now(); copy16<<<1024, 256, 0, s1>>>((const uint4 *)d_b, (uint4 *)h_out, n); copy16<<<1024, 256, 0, s0>>>((const uint4 *)h_in, (uint4 *)d_a, n);
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); printf("rep %d: kernel writing host beside kernel reading host %.2f ms\n", rep, (now() - t0) * 1e3);
        for (int chunks : {4, 8, 16}) {
            t0 = now();
            for (int k = 0; k < chunks; ++k) {
                const size_t off = bytes / chunks * k;
                CK(hipMemcpyAsync((char *)d_a + off, (char *)h_in + off, bytes / chunks, hipMemcpyHostToDevice, s1));
                CK(hipMemcpyAsync((char *)h_out + off, (char *)d_b + off, bytes / chunks, hipMemcpyDeviceToHost, s0));
            }
            CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); printf("rep %d: SDMA both ways in %d chunks each %.2f ms\n", rep, chunks, (now() - t0) * 1e3);
        }
        t0 = now(); CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s0));
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); double t_dup = now() - t0;
        printf("rep %d: SDMA upload %.2f ms, download %.2f ms, both at once %.2f ms (96 MiB each)\n", rep, t_h2d * 1e3, t_d2h * 1e3, t_dup * 1e3);
    }
    return 0;
}
