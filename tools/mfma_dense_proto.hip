// Prototype (not part of the product): one DENSE layer of a t = 9 state on the matrix cores.
// Tables and check: tools/mfma_dense_proto.py.  Every lane holds one state; a wave's 64 states are the N dimension of two
// v_mfma_i32_32x32x32_i8 per k-step (states 0-31 and 32-63), the 32 balanced bytes of an output residue the M dimension,
// the bytes of the nine input elements the K dimension (lanes 32-63 of an MFMA feed the second half of every k-step, so
// half of each lane's digit registers is exchanged with lane +-32 by v_permlane32_swap, once per layer).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_dense_proto.hip -o tools/mfma_dense_proto
//   python3 tools/mfma_dense_proto.py gen <dir> 18 && tools/mfma_dense_proto <dir> && python3 tools/mfma_dense_proto.py check <dir>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

#ifndef PROTO_BLOCK
#define PROTO_BLOCK 512
#endif
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kT = 9, kN = 9, kW = 29, kNQ = 11, kWords = 88;
constexpr uint32_t kMask = (1u << kW) - 1;

struct Params {
    uint32_t p[kN];
    uint32_t pinv;
};

__device__ __forceinline__ void swap32(uint32_t &x, uint32_t &y) {
    // lanes 32-63 of x <-> lanes 0-31 of y
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}
__device__ __forceinline__ void swap32(int &x, int &y) {
    auto r = __builtin_amdgcn_permlane32_swap((uint32_t)x, (uint32_t)y, false, false);
    x = (int)r[0];
    y = (int)r[1];
}

__global__ void probe_swap(uint32_t *out) {
    uint32_t x = threadIdx.x, y = 100 + threadIdx.x;
    swap32(x, y);
    out[threadIdx.x] = x;
    out[64 + threadIdx.x] = y;
}

// one layer, `reps` times (the output of one pass is the input of the next: every pass re-cuts its digits).
// The 11 KiB of table one output row needs are staged in LDS once per BLOCK (double-buffered, one barrier per row): read
// straight from global memory by every wave they are 101 KiB per wave and layer, 415 MB per layer of 2^18 states - the
// L2 -> L1 path, not the arithmetic, then sets the time (measured: 80 us per layer).
// FOLD (timing only): the rows of every pass are XOR-folded into one element instead of replacing the state, so that the cost of
// the row loop is seen without the 81 registers of a second state (the product keeps new rows in its LDS scratch)
template <int BLOCK, bool FOLD>
__global__ void __launch_bounds__(BLOCK, 512 / BLOCK) layer_kernel(const Params prm, const v4i *__restrict__ tab, const long long *__restrict__ corr,
                                                                   const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n, int reps) {
    __shared__ v4i tile[2][kNQ * 64];
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // n is a multiple of BLOCK
    const int lane = threadIdx.x & 63;
    auto stage = [&](int row, int buf) {
        for (int e = threadIdx.x; e < kNQ * 64; e += BLOCK) tile[buf][e] = tab[(size_t)row * kNQ * 64 + e];
    };
    const uint32_t *src = in + gid * kT * kN;
    uint32_t *dst = out + gid * kT * kN;
    uint32_t st[kT][kN];   // the state stays in registers between passes, as it would between the rounds of a permutation
#pragma unroll
    for (int j = 0; j < kT; ++j) {
#pragma unroll
        for (int k = 0; k < kN; ++k) st[j][k] = src[j * kN + k];
    }
    for (int rep = 0; rep < reps; ++rep) {
        uint32_t W[kWords];
#pragma unroll
        for (int j = 0; j < kT; ++j) {
            uint32_t l[kN];
#pragma unroll
            for (int k = 0; k < kN; ++k) l[k] = FOLD ? (st[j][k] & kMask) : st[j][k];
#pragma unroll
            for (int w = 0; w < 9; ++w) {
                const int bit = 32 * w, li = bit / kW, sh = bit % kW;
                uint64_t v = (uint64_t)l[li] >> sh;
                if (li + 1 < kN) v |= (uint64_t)l[li + 1] << (kW - sh);
                if (li + 2 < kN && 2 * kW - sh < 32) v |= (uint64_t)l[li + 2] << (2 * kW - sh);
                W[9 * j + w] = (uint32_t)v ^ 0x80808080u;   // bytes enter as u - 128
            }
        }
#pragma unroll
        for (int w = 81; w < kWords; ++w) W[w] = 0x80808080u;
#pragma unroll
        for (int q = 0; q < kNQ; ++q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) swap32(W[8 * q + u], W[8 * q + 4 + u]);
        }
        stage(0, 0);
#pragma clang loop unroll(disable)
        for (int i = 0; i < kT; ++i) {
            __syncthreads();                                  // row i is staged (and row i - 1's readers are done with the other buffer)
            if (i + 1 < kT) stage(i + 1, (i + 1) & 1);
            v16i d1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d2 = d1;
            const v4i *ap = &tile[i & 1][lane];
#pragma unroll
            for (int q = 0; q < kNQ; ++q) {
                const v4i a = ap[q * 64];
                const v4i b1 = {(int)W[8 * q + 0], (int)W[8 * q + 1], (int)W[8 * q + 2], (int)W[8 * q + 3]};
                const v4i b2 = {(int)W[8 * q + 4], (int)W[8 * q + 5], (int)W[8 * q + 6], (int)W[8 * q + 7]};
                d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b2, d2, 0, 0, 0);
            }
            // rows 8g + r of every state now sit in d1[4g + r] of lanes < 32 (own state) and lanes >= 32 (state lane - 32), rows
            // 8g + 4 + r likewise in d2 for states 32-63: one exchange per register gives every lane all 32 rows of ITS state
            int D1[16], D2[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                D1[v] = d1[v];
                D2[v] = d2[v];
                swap32(D1[v], D2[v]);
            }
            // D1[4g + r] = row 8g + r, D2[4g + r] = row 8g + 4 + r: word w (rows 4w .. 4w+3) = (w even ? D1 : D2)[4 (w / 2) ..]
            uint32_t wd[9];
            long long c = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int *d = (w & 1) ? D2 : D1;
                const int b = 4 * (w >> 1);
                long long t = (long long)d[b] + ((long long)d[b + 1] << 8) + ((long long)d[b + 2] << 16) + ((long long)d[b + 3] << 24) + corr[i * 8 + w] + c;
                wd[w] = (uint32_t)t;
                c = t >> 32;
            }
            wd[8] = (uint32_t)c;
            uint32_t L[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const int bit = kW * k, wi = bit / 32, sh = bit % 32;
                uint64_t pair = wd[wi];
                if (wi + 1 < 9) pair |= (uint64_t)wd[wi + 1] << 32;
                L[k] = (uint32_t)(pair >> sh) & kMask;
            }
            uint64_t acc = L[0];
            const uint32_t m0 = ((uint32_t)acc * prm.pinv) & kMask;
            acc += (uint64_t)m0 * prm.p[0];
            acc >>= kW;
            acc += L[1];
            acc += (uint64_t)m0 * prm.p[1];
            const uint32_t m1 = ((uint32_t)acc * prm.pinv) & kMask;
            acc += (uint64_t)m1 * prm.p[0];
            acc >>= kW;
            uint32_t o[kN];
#pragma unroll
            for (int k = 2; k <= 10; ++k) {
                if (k < 10) acc += L[k];
                if (k <= 8) acc += (uint64_t)m0 * prm.p[k];
                if (k - 1 <= 8) acc += (uint64_t)m1 * prm.p[k - 1];
                o[k - 2] = k < 10 ? ((uint32_t)acc & kMask) : (uint32_t)acc;
                acc >>= kW;
            }
#pragma unroll
            for (int k = 0; k < kN; ++k) {
                if constexpr (FOLD) st[0][k] ^= o[k];
                else dst[i * kN + k] = o[k];   // (one pass only: the verification run)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    if constexpr (FOLD) {
#pragma unroll
        for (int k = 0; k < kN; ++k) dst[k] = st[0][k];
    }
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <dir> [reps for the timed run = 8]\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[1];
    const int reps = argc > 2 ? atoi(argv[2]) : 8;
    FILE *f = fopen((dir + "/proto_in.bin").c_str(), "rb");
    if (!f) {
        perror("proto_in.bin");
        return 1;
    }
    uint32_t hdr[8];
    if (fread(hdr, 4, 8, f) != 8 || hdr[0] != 0x50524F54u || hdr[2] != (uint32_t)kT || hdr[3] != (uint32_t)kNQ) {
        fprintf(stderr, "bad header\n");
        return 1;
    }
    const size_t n = hdr[1];
    Params prm;
    uint32_t pp[10];
    if (fread(pp, 4, 10, f) != 10) return 1;
    for (int k = 0; k < kN; ++k) prm.p[k] = pp[k];
    prm.pinv = pp[9];
    std::vector<int8_t> tab((size_t)kT * kNQ * 64 * 16);
    std::vector<long long> corr(kT * 8);
    std::vector<uint32_t> st(n * kT * kN);
    if (fread(tab.data(), 1, tab.size(), f) != tab.size() || fread(corr.data(), 8, corr.size(), f) != corr.size() ||
        fread(st.data(), 4, st.size(), f) != st.size()) {
        fprintf(stderr, "short file\n");
        return 1;
    }
    fclose(f);

    uint32_t *d_probe;
    CHECK(hipMalloc(&d_probe, 128 * 4));
    hipLaunchKernelGGL(probe_swap, dim3(1), dim3(64), 0, 0, d_probe);
    uint32_t pr[128];
    CHECK(hipMemcpy(pr, d_probe, sizeof pr, hipMemcpyDeviceToHost));
    printf("permlane32_swap(x = lane, y = 100 + lane): x[0]=%u x[31]=%u x[32]=%u x[63]=%u | y[0]=%u y[31]=%u y[32]=%u y[63]=%u\n", pr[0], pr[31], pr[32],
           pr[63], pr[64], pr[95], pr[96], pr[127]);

    v4i *d_tab;
    long long *d_corr;
    uint32_t *d_in, *d_out;
    CHECK(hipMalloc(&d_tab, tab.size()));
    CHECK(hipMalloc(&d_corr, corr.size() * 8));
    CHECK(hipMalloc(&d_in, st.size() * 4));
    CHECK(hipMalloc(&d_out, st.size() * 4));
    CHECK(hipMemcpy(d_tab, tab.data(), tab.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_corr, corr.data(), corr.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_in, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    constexpr int kBlock = PROTO_BLOCK;
    const dim3 grid((unsigned)(n / kBlock)), block(kBlock);
    hipLaunchKernelGGL((layer_kernel<kBlock, false>), grid, block, 0, 0, prm, d_tab, d_corr, d_in, d_out, n, 1);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> res(st.size());
    CHECK(hipMemcpy(res.data(), d_out, res.size() * 4, hipMemcpyDeviceToHost));
    f = fopen((dir + "/proto_out.bin").c_str(), "wb");
    fwrite(res.data(), 4, res.size(), f);
    fclose(f);

    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r : {1, reps, reps}) {
        hipLaunchKernelGGL((layer_kernel<kBlock, true>), grid, block, 0, 0, prm, d_tab, d_corr, d_in, d_out, n, r);   // warm
        CHECK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((layer_kernel<kBlock, true>), grid, block, 0, 0, prm, d_tab, d_corr, d_in, d_out, n, r);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("reps %2d: %.3f ms per launch of %zu states = %.3f us per layer and 2^18 states\n", r, ms / 10, n, ms / 10 / r * 1e3 * (262144.0 / n));
    }
    return 0;
}
