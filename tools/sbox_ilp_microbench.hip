// Is the S-box of the two-waves-per-SIMD kernels (t >= 6) short of instruction-level parallelism?  A Montgomery product is ONE dependent
// chain (every v_mad_u64_u32 of a column adds into the accumulator of the one before, the columns hang together through the carry and the
// quotient digit); tools/issue_model_microbench.hip measured a fully dependent multiply chain at 5.7 clocks per instruction with two waves on
// a SIMD against 4.5 for two chains.  This probe runs the product's own x^5 (pmx_field.hpp) at exactly 1, 2, 3 and 4 waves per SIMD:
//   one    x = S(x + c), one element per lane
//   two    two elements per lane, one S-box after the other in the source (the compiler's schedule)
//   pair   two elements per lane, the two chains written side by side instruction by instruction (mont_sqr2 / mont_mul2 below)
// and prints nanoseconds per S-box and SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -opt-disable=reassociate -Isponge_amd/csrc tools/sbox_ilp_microbench.hip -o tools/sbox_ilp_microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pmx_field.hpp"

using namespace pmx;

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

extern __shared__ uint4 lds_pin[];

// two Montgomery squarings side by side
__device__ __forceinline__ void mont_sqr2(const Fe &a, const Fe &b, Fe &oa, Fe &ob, const FieldRt &f) {
    uint32_t da[kN], db[kN], ma[kN], mb[kN];
#pragma unroll
    for (int i = 0; i < kN; ++i) da[i] = a.l[i] << 1, db[i] = b.l[i] << 1;
    uint64_t xa = 0, xb = 0;
#pragma unroll
    for (int k = 0; k < 2 * kN - 1; ++k) {
#pragma unroll
        for (int i = 0; i < kN; ++i) {
            const int j = k - i;
            if (j > i && j < kN) {
                xa += (uint64_t)a.l[i] * da[j];
                xb += (uint64_t)b.l[i] * db[j];
            }
        }
        if ((k & 1) == 0) {
            xa += (uint64_t)a.l[k / 2] * a.l[k / 2];
            xb += (uint64_t)b.l[k / 2] * b.l[k / 2];
        }
        if (k < kN) {
#pragma unroll
            for (int j = 0; j < k; ++j) {
                xa += (uint64_t)ma[j] * f.p[k - j];
                xb += (uint64_t)mb[j] * f.p[k - j];
            }
            ma[k] = ((uint32_t)xa * f.pinv) & kMask;
            mb[k] = ((uint32_t)xb * f.pinv) & kMask;
            xa += (uint64_t)ma[k] * f.p[0];
            xb += (uint64_t)mb[k] * f.p[0];
            xa >>= kW;
            xb >>= kW;
        } else {
#pragma unroll
            for (int j = k - (kN - 1); j < kN; ++j) {
                xa += (uint64_t)ma[j] * f.p[k - j];
                xb += (uint64_t)mb[j] * f.p[k - j];
            }
            oa.l[k - kN] = (uint32_t)xa & kMask;
            ob.l[k - kN] = (uint32_t)xb & kMask;
            xa >>= kW;
            xb >>= kW;
        }
    }
    oa.l[kN - 1] = (uint32_t)xa;
    ob.l[kN - 1] = (uint32_t)xb;
}
// two Montgomery products side by side
__device__ __forceinline__ void mont_mul2(const Fe &a, const Fe &c, const Fe &b, const Fe &d, Fe &oa, Fe &ob, const FieldRt &f) {
    uint32_t ma[kN], mb[kN];
    uint64_t xa = 0, xb = 0;
#pragma unroll
    for (int k = 0; k < 2 * kN - 1; ++k) {
        const int lo_i = k < kN ? 0 : k - (kN - 1), hi_i = k < kN ? k : kN - 1;
#pragma unroll
        for (int i = lo_i; i <= hi_i; ++i) {
            xa += (uint64_t)a.l[i] * c.l[k - i];
            xb += (uint64_t)b.l[i] * d.l[k - i];
        }
#pragma unroll
        for (int j = lo_i; j <= hi_i; ++j) {
            if (j < k || k >= kN) {
                xa += (uint64_t)ma[j] * f.p[k - j];
                xb += (uint64_t)mb[j] * f.p[k - j];
            }
        }
        if (k < kN) {
            ma[k] = ((uint32_t)xa * f.pinv) & kMask;
            mb[k] = ((uint32_t)xb * f.pinv) & kMask;
            xa += (uint64_t)ma[k] * f.p[0];
            xb += (uint64_t)mb[k] * f.p[0];
            xa >>= kW;
            xb >>= kW;
        } else {
            oa.l[k - kN] = (uint32_t)xa & kMask;
            ob.l[k - kN] = (uint32_t)xb & kMask;
            xa >>= kW;
            xb >>= kW;
        }
    }
    oa.l[kN - 1] = (uint32_t)xa;
    ob.l[kN - 1] = (uint32_t)xb;
}
__device__ __forceinline__ void sbox5_pair(Fe &x, Fe &y, const FieldRt &f) {
    Fe x2, y2, x4, y4;
    mont_sqr2(x, y, x2, y2, f);
    mont_sqr2(x2, y2, x4, y4, f);
    mont_mul2(x4, x, y4, y, x, y, f);
}

struct Params {
    FieldRt f;
    Fe one, c;
};

template <int MODE>
__global__ void __launch_bounds__(256) bench(Params P, uint32_t *out, int iters) {
    lds_pin[threadIdx.x] = make_uint4(1, 2, 3, 4);
    const FieldRt f = P.f;
    Fe x = P.c, y = P.one;
    x.l[0] ^= threadIdx.x;
    y.l[1] ^= threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {
            x = fe_sbox<5>(fe_add_lazy(x, P.c), 5, P.one, f);
        } else if constexpr (MODE == 1) {
            x = fe_sbox<5>(fe_add_lazy(x, P.c), 5, P.one, f);
            y = fe_sbox<5>(fe_add_lazy(y, P.c), 5, P.one, f);
        } else {
            x = fe_add_lazy(x, P.c);
            y = fe_add_lazy(y, P.c);
            sbox5_pair(x, y, f);
        }
    }
    uint32_t r = lds_pin[0].x;
    for (int k = 0; k < kN; ++k) r ^= x.l[k] ^ y.l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
static double run(const Params &P, int waves, int n_cu, size_t cu_lds, uint32_t *d_out, std::vector<uint32_t> *first = nullptr) {
    const int blocks = n_cu * waves;
    size_t lds = std::min(cu_lds / waves - 1024, (size_t)64 * 1024);
    if (waves == 1) lds = 64 * 1024;
    CHECK(hipFuncSetAttribute((const void *)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int iters = 400;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 8; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), lds, 0, P, d_out, iters);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t = 0;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 3 && t < best) best = t;
    }
    if (first) {
        first->resize(64);
        CHECK(hipMemcpy(first->data(), d_out, 64 * 4, hipMemcpyDeviceToHost));
    }
    const double sboxes = (MODE == 0 ? 1.0 : 2.0) * iters * waves;   // per SIMD
    return best * 1e6 / sboxes;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    int cu_lds = 0;
    if (hipDeviceGetAttribute(&cu_lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0) != hipSuccess || cu_lds <= 0) cu_lds = 160 * 1024;
    // BLS12-381 Fr
    const uint64_t p64[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    Params P{};
    for (int i = 0; i < kN; ++i) {
        const int bit = kW * i, wi = bit / 64, sh = bit % 64;
        unsigned __int128 pair = p64[wi];
        if (wi + 1 < 4) pair |= (unsigned __int128)p64[wi + 1] << 64;
        P.f.p[i] = (uint32_t)(pair >> sh) & kMask;
    }
    uint32_t inv = 1;   // p^-1 mod 2^32 by Newton, then -p^-1 mod 2^29
    for (int i = 0; i < 6; ++i) inv *= 2 - P.f.p[0] * inv;
    P.f.pinv = (0u - inv) & kMask;
    P.f.unit = 1;
    P.f.io = nullptr;
    for (int i = 0; i < kN; ++i) P.one.l[i] = (0x1234567u * (i + 3)) & kMask, P.c.l[i] = (0x7654321u * (i + 5)) & kMask;
    P.one.l[kN - 1] &= 0x3fffff;
    P.c.l[kN - 1] &= 0x3fffff;
    uint32_t *d_out = nullptr;
    CHECK(hipMalloc((void **)&d_out, (size_t)n_cu * 8 * 256 * 4));
    for (int i = 0; i < 20; ++i) (void)run<0>(P, 4, n_cu, cu_lds, d_out);   // spin-up
    printf("%s, %d CUs: x^5 on 9 x 29-bit limbs (pmx_field.hpp), nanoseconds per S-box and SIMD\n", prop.gcnArchName, n_cu);
    std::vector<uint32_t> a, b;
    for (int w : {1, 2, 3, 4}) {
        const double one = run<0>(P, w, n_cu, cu_lds, d_out), two = run<1>(P, w, n_cu, cu_lds, d_out, &a), pair = run<2>(P, w, n_cu, cu_lds, d_out, &b);
        printf("  %d wave(s) per SIMD: one element per lane %7.1f | two, one after the other in the source %7.1f (%+5.1f %%) | two, side by side %7.1f (%+5.1f %%)%s\n", w, one, two,
               100 * (one / two - 1), pair, 100 * (one / pair - 1), a == b ? "" : "  RESULTS DIFFER");
    }
    return 0;
}
