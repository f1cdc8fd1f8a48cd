// Which workgroups share a CU, and what tells them apart?  The window engines of t >= 6 run two workgroups of four waves per CU
// (80 KiB of LDS each): wave w of both sits on SIMD w.  This probe launches a grid of that shape (256 threads, 80 KiB of LDS,
// 4 x the resident capacity), has every wave record HW_ID / XCC_ID and its start time, and prints per CU which blocks ran there, with
// which wave slot (HW_ID.WAVE_ID) and which workgroup slot (HW_ID.TG_ID), in start order.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/hwid_probe.hip -o tools/hwid_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
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

struct Rec {
    uint32_t hw_id, xcc_id;
    uint64_t t0, t1;
};

__global__ void __launch_bounds__(256, 2) probe(Rec *out, uint32_t spin_ticks) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();             // 100 MHz
    lds_pin[threadIdx.x] = make_uint4(hw, xcc, 0, 0);
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        Rec r{hw, xcc, t0, __builtin_amdgcn_s_memrealtime()};
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;
    }
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 2048;
    const uint32_t spin = argc > 2 ? (uint32_t)atoi(argv[2]) : 3000;   // 30 us
    Rec *d;
    CHECK(hipMalloc((void **)&d, sizeof(Rec) * blocks * 4));
    CHECK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    probe<<<blocks, 256, 80 * 1024>>>(d, spin);
    CHECK(hipDeviceSynchronize());
    std::vector<Rec> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), d, sizeof(Rec) * blocks * 4, hipMemcpyDeviceToHost));
    uint64_t tmin = ~0ull;
    for (auto &r : h) tmin = std::min(tmin, r.t0);
    // key: (xcc, se, sh, cu) -> list of (start, block, wave, wave_id, simd, tg_id)
    struct E { uint64_t t0; int block, wave, wave_id, simd, tg; };
    std::map<uint32_t, std::vector<E>> cus;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 4; ++w) {
            const Rec &r = h[b * 4 + w];
            const uint32_t hw = r.hw_id;
            const uint32_t key = ((r.xcc_id & 0xf) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
            cus[key].push_back(E{r.t0 - tmin, b, w, (int)(hw & 0xf), (int)((hw >> 4) & 3), (int)((hw >> 16) & 0xf)});
        }
    printf("%zu distinct (xcc, se, sh, cu) keys for %d blocks\n", cus.size(), blocks);
    int shown = 0, simd_is_wave = 0, total = 0;
    std::map<int, int> first_gen_waveid, first_gen_tg;
    int pair_distinct_wave = 0, pair_distinct_tg = 0, pairs = 0, same_xcd_mod8 = 0;
    for (auto &kv : cus) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end(), [](const E &a, const E &b) { return a.t0 < b.t0 || (a.t0 == b.t0 && a.block < b.block); });
        for (auto &e : v) { total++; simd_is_wave += e.simd == e.wave; }
        // the first generation: the first 8 records (2 blocks x 4 waves)
        std::map<int, std::vector<E>> by_block;
        for (size_t i = 0; i < v.size() && i < 8; ++i) by_block[v[i].block].push_back(v[i]);
        if (by_block.size() == 2) {
            auto it = by_block.begin();
            const auto &a = it->second; ++it; const auto &b = it->second;
            pairs++;
            pair_distinct_wave += (a[0].wave_id & 1) != (b[0].wave_id & 1);
            pair_distinct_tg += (a[0].tg & 1) != (b[0].tg & 1);
            same_xcd_mod8 += (a[0].block % 8) == (b[0].block % 8);
        }
        if (shown < 6) {
            printf("cu key %05x:", kv.first);
            for (size_t i = 0; i < v.size() && i < 24; ++i)
                printf(" [t=%llu b%d w%d simd%d slot%d tg%d]", (unsigned long long)v[i].t0, v[i].block, v[i].wave, v[i].simd, v[i].wave_id, v[i].tg);
            printf("\n");
            shown++;
        }
    }
    printf("wave w on SIMD w: %d of %d records\n", simd_is_wave, total);
    printf("first-generation pairs on a CU: %d; wave slot parity differs in %d, TG_ID parity differs in %d; same block %% 8 in %d\n", pairs, pair_distinct_wave,
           pair_distinct_tg, same_xcd_mod8);
    // how the first generation's blocks map: block index -> which of the two slots (by TG parity)
    int lo_half_tg0 = 0, lo_half = 0;
    for (auto &kv : cus)
        for (size_t i = 0; i < kv.second.size() && i < 8; ++i)
            if (kv.second[i].wave == 0 && kv.second[i].block < 256) { lo_half++; lo_half_tg0 += (kv.second[i].tg & 1) == 0; }
    printf("blocks 0..255 (wave 0 records in a first generation): %d, of which TG_ID even: %d\n", lo_half, lo_half_tg0);
    return 0;
}
