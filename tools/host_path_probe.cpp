// Probe for the PCIe-inclusive host path: (a) pmx_permute_batch as is (pageable copies), (b) hipHostRegister + one
// async round trip, (c) registered memory, chunked over two streams so H2D / kernel / D2H overlap.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../include/poseidon_mi355x.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const uint64_t bls[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    std::vector<uint64_t> ark(39 * 3 * 4), mds(9 * 4);
    if (pmx_find_poseidon_ark_and_mds(bls, 255, 2, 8, 31, 0, ark.data(), mds.data())) return 1;
    pmx_config cfg{}; cfg.full_rounds = 8; cfg.partial_rounds = 31; cfg.alpha = 5; cfg.rate = 2; cfg.capacity = 1;
    memcpy(cfg.modulus, bls, 32); cfg.ark = ark.data(); cfg.mds = mds.data();
    pmx_ctx *ctx; if (pmx_ctx_create(&cfg, 0, &ctx)) { printf("%s\n", pmx_last_error()); return 1; }
    const size_t n = 1 << 20, bytes = n * 96;
    std::vector<uint64_t> host(n * 12);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (i % 4 == 3) ? 0x1234567 + i : 0x9e3779b97f4a7c15ull * (i + 1);
    void *dev; CK(hipMalloc(&dev, bytes));
    hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); pmx_permute_batch(ctx, host.data(), n); double ta = now() - t0;
        t0 = now(); CK(hipHostRegister(host.data(), bytes, hipHostRegisterDefault)); double treg = now() - t0;
        t0 = now();
        CK(hipMemcpyAsync(dev, host.data(), bytes, hipMemcpyHostToDevice, st[0]));
        pmx_permute_batch_dev(ctx, (uint64_t *)dev, n, st[0]);
        CK(hipMemcpyAsync(host.data(), dev, bytes, hipMemcpyDeviceToHost, st[0]));
        CK(hipStreamSynchronize(st[0])); double tb = now() - t0;
        for (int chunks : {4, 8, 16}) {
            t0 = now();
            const size_t cn = n / chunks;
            for (int k = 0; k < chunks; ++k) {
                hipStream_t s = st[k & 1];
                char *h = (char *)host.data() + (size_t)k * cn * 96, *d = (char *)dev + (size_t)k * cn * 96;
                CK(hipMemcpyAsync(d, h, cn * 96, hipMemcpyHostToDevice, s));
                pmx_permute_batch_dev(ctx, (uint64_t *)d, cn, s);
                CK(hipMemcpyAsync(h, d, cn * 96, hipMemcpyDeviceToHost, s));
            }
            CK(hipStreamSynchronize(st[0])); CK(hipStreamSynchronize(st[1]));
            printf("rep %d chunks %2d: %.2f ms\n", rep, chunks, (now() - t0) * 1e3);
        }
        {
            static hipStream_t up = nullptr, run = nullptr, down = nullptr;
            static hipEvent_t eu[64], ek[64];
            if (!up) {
                CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&run, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
                for (int i = 0; i < 64; ++i) { CK(hipEventCreateWithFlags(&eu[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ek[i], hipEventDisableTiming)); }
            }
            for (int chunks : {4, 8, 16, 32}) {
                t0 = now();
                const size_t cn = n / chunks;
                for (int k = 0; k < chunks; ++k) {
                    char *h = (char *)host.data() + (size_t)k * cn * 96, *d = (char *)dev + (size_t)k * cn * 96;
                    CK(hipMemcpyAsync(d, h, cn * 96, hipMemcpyHostToDevice, up));
                    CK(hipEventRecord(eu[k], up)); CK(hipStreamWaitEvent(run, eu[k], 0));
                    pmx_permute_batch_dev(ctx, (uint64_t *)d, cn, run);
                    CK(hipEventRecord(ek[k], run)); CK(hipStreamWaitEvent(down, ek[k], 0));
                    CK(hipMemcpyAsync(h, d, cn * 96, hipMemcpyDeviceToHost, down));
                }
                CK(hipStreamSynchronize(down)); CK(hipStreamSynchronize(run)); CK(hipStreamSynchronize(up));
                printf("rep %d three streams, chunks %2d: %.2f ms\n", rep, chunks, (now() - t0) * 1e3);
            }
        }
        {   // graded chunks: small first and last pieces (the pipeline's fill and drain are one upload + one kernel and one kernel + one download of those)
            static hipStream_t up = nullptr, run = nullptr, down = nullptr;
            static hipEvent_t eu[64], ek[64];
            if (!up) {
                CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&run, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
                for (int i = 0; i < 64; ++i) { CK(hipEventCreateWithFlags(&eu[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ek[i], hipEventDisableTiming)); }
            }
            const std::vector<std::vector<int>> shapes = {{1, 2, 4, 9, 9, 4, 2, 1}, {1, 3, 6, 6, 6, 6, 3, 1}, {2, 4, 5, 5, 5, 5, 4, 2}, {1, 1, 2, 4, 8, 8, 4, 2, 1, 1}, {1, 2, 3, 4, 6, 6, 4, 3, 2, 1}, {1, 1, 2, 2, 4, 6, 6, 4, 2, 2, 1, 1}};
            for (const auto &shape : shapes) {
                int total = 0; for (int w : shape) total += w;
                t0 = now();
                size_t at = 0;
                for (size_t k = 0; k < shape.size(); ++k) {
                    const size_t cn = k + 1 == shape.size() ? n - at : n * shape[k] / total;
                    char *h = (char *)host.data() + at * 96, *d = (char *)dev + at * 96;
                    CK(hipMemcpyAsync(d, h, cn * 96, hipMemcpyHostToDevice, up));
                    CK(hipEventRecord(eu[k], up)); CK(hipStreamWaitEvent(run, eu[k], 0));
                    pmx_permute_batch_dev(ctx, (uint64_t *)d, cn, run);
                    CK(hipEventRecord(ek[k], run)); CK(hipStreamWaitEvent(down, ek[k], 0));
                    CK(hipMemcpyAsync(h, d, cn * 96, hipMemcpyDeviceToHost, down));
                    at += cn;
                }
                CK(hipStreamSynchronize(down)); CK(hipStreamSynchronize(run)); CK(hipStreamSynchronize(up));
                printf("rep %d three streams, graded chunks", rep); for (int w : shape) printf(" %d", w); printf(" / %d: %.2f ms\n", total, (now() - t0) * 1e3);
            }
        }
        {   // zero copy: the kernel itself reads the states out of the page-locked host buffer and writes them back there (both directions
            // of the link at once, no staging, no chunks) - the registered range's device address IS the host address on this platform
            void *mapped = nullptr;
            CK(hipHostGetDevicePointer(&mapped, host.data(), 0));
            for (int chunks : {1, 2, 4}) {
                t0 = now();
                const size_t cn = n / chunks;
                for (int k = 0; k < chunks; ++k) pmx_permute_batch_dev(ctx, (uint64_t *)((char *)mapped + (size_t)k * cn * 96), cn, st[k & 1]);
                CK(hipStreamSynchronize(st[0])); CK(hipStreamSynchronize(st[1]));
                printf("rep %d zero copy (kernel on the mapped host buffer), %d launch(es): %.2f ms\n", rep, chunks, (now() - t0) * 1e3);
            }
        }
        t0 = now(); CK(hipHostUnregister(host.data())); double tun = now() - t0;
        printf("rep %d: pageable %.2f ms | register %.2f ms, one round trip %.2f ms, unregister %.2f ms\n", rep, ta * 1e3, treg * 1e3, tb * 1e3, tun * 1e3);
    }
    return 0;
}
