// CPU-side check of the KERNEL ALGORITHMS (test infrastructure, not a product path): the field arithmetic
// and permutation templates of sponge_amd/csrc/pmx_field.hpp / pmx_permute.hpp are compiled for the host and
// exported so that tests/test_hostcheck.py can compare them with the oracle without a GPU.  The product
// library never links this file; libposeidon_mi355x.so has no CPU data path.
//
// Build: g++ -O2 -std=c++17 -fPIC -shared -I sponge_amd/csrc tests/hostcheck/pmx_hostcheck.cpp -o tests/hostcheck/libpmx_hostcheck.so
#include <cstdint>
#include <cstring>
#include <string>

#define PMX_HOSTCHECK 1
#include "pmx_prepare.hpp"
#include "pmx_sponge_plan.hpp"

using namespace pmx;

// ---- bound tracker (PMX_TRACK in pmx_field.hpp / pmx_permute.hpp): largest limb and largest value / p seen per tag
static uint32_t g_max_limb[4];
static double g_max_b[4];
void pmx::hostcheck_track(int tag, const Fe &x, const FieldRt &f) {
    long double v = 0, pv = 0;
    for (int i = kN - 1; i >= 0; --i) {
        v = v * 536870912.0L + x.l[i];
        pv = pv * 536870912.0L + f.p[i];
        if (x.l[i] > g_max_limb[tag]) g_max_limb[tag] = x.l[i];
    }
    const double b = (double)(v / pv);
    if (b > g_max_b[tag]) g_max_b[tag] = b;
}
// inputs of the matrix-core layers (pmx_mfma.hpp: mfma_state_words): each must be norm and below 2^256 - the 32-byte form drops the ninth word
static unsigned long long g_layer_inputs = 0, g_layer_inputs_too_large = 0;
void pmx::hostcheck_below_2_256(const Fe &x) {
    ++g_layer_inputs;
    bool bad = (x.l[kN - 1] >> (256 - kW * (kN - 1))) != 0;      // bits 256 .. 260 live in the top limb
    for (int i = 0; i < kN; ++i) bad |= x.l[i] > kMask;
    if (bad) ++g_layer_inputs_too_large;
}
extern "C" void hc_layer_inputs(unsigned long long *seen, unsigned long long *too_large) {
    *seen = g_layer_inputs;
    *too_large = g_layer_inputs_too_large;
}
extern "C" void hc_track_reset() {
    for (int t = 0; t < 4; ++t) { g_max_limb[t] = 0; g_max_b[t] = 0; }
}
extern "C" void hc_track_get(int tag, uint32_t *max_limb, double *max_b) {
    *max_limb = g_max_limb[tag];
    *max_b = g_max_b[tag];
}

static Abi load_abi(const uint64_t *p) {
    Abi a;
    std::memcpy(a.w, p, 32);
    return a;
}
static void store_abi(uint64_t *p, const Abi &a) { std::memcpy(p, a.w, 32); }

// cooperative t = 3 schedule, the four lanes of a quad simulated in turn (lane 3: the squaring lane of the folded sparse
// rounds; in the uniform rounds it only shadows lane 2)
template <int ALPHA>
static void coop_permute_one(Fe (&s)[4], const uint32_t *coop, const Prepared &pp) {
    const uint32_t first_partial = pp.c.half_full, last_partial = pp.c.half_full + pp.c.partial_rounds - 1;
    for (uint32_t r = 0; r < pp.c.total_rounds; ++r) {
        auto entry = [&](int q) { return coop + ((size_t)r * 3 + (q < 3 ? q : 2)) * kCoopElems * kFeStride; };
        if (kCoopFolded<ALPHA> && r >= first_partial && r < last_partial) {
            const Fe x = fe_add_lazy(s[0], fe_const(entry(0)));
            Fe res_a[4], res_b[4], xpow;
            for (int q = 0; q < 4; ++q) res_a[q] = coop_fold_a(q, x, entry(q), pp.f);
            for (int q = 0; q < 4; ++q) res_b[q] = coop_fold_b(q, s[q], res_a[q], entry(q), pp.f);
            xpow = res_b[3];
            for (int k = 0; k < kCoopExtraSquarings<ALPHA>; ++k) xpow = mont_sqr(xpow, pp.f);
            for (int q = 0; q < 4; ++q) s[q] = coop_fold_c(q, s[q], xpow, res_a[q], res_b[1], res_b[2], pp.f);
            for (int q = 0; q < 3; ++q) PMX_TRACK(q == 0 ? 0 : 1, s[q], pp.f);
            continue;
        }
        Fe z[3], nxt[4];
        for (int q = 0; q < 3; ++q) z[q] = coop_pre<ALPHA>(s[q], entry(q), is_full_round(r, pp.c) || q == 0, pp.c, pp.one, pp.f);
        for (int q = 0; q < 4; ++q) nxt[q] = coop_layer_is_norm(r, pp.c) ? coop_post_norm(z, entry(q), pp.f) : coop_post(z, entry(q), pp.f);
        for (int q = 0; q < 4; ++q) s[q] = nxt[q];
    }
}

extern "C" int hc_permute_coop(const pmx_config *cfg, uint64_t *states, size_t n) {
    Prepared pp;
    std::string err;
    int rc = prepare(cfg, pp, err);
    if (rc) return rc;
    if (!pp.has_opt || pp.t != 3) return PMX_ERR_UNSUPPORTED;
    const uint32_t *coop = pp.consts.data() + pp.coop_offset;
    for (size_t k = 0; k < n; ++k) {
        Fe s[4];
        for (int i = 0; i < 3; ++i) s[i] = fe_from_abi_scaled(load_abi(states + (k * 3 + i) * 4));
        s[3] = fe_zero();
        if (pp.c.alpha == 5) coop_permute_one<5>(s, coop, pp);
        else if (pp.c.alpha == 17) coop_permute_one<17>(s, coop, pp);
        else coop_permute_one<0>(s, coop, pp);
        for (int i = 0; i < 3; ++i) store_abi(states + (k * 3 + i) * 4, fe_to_abi_scaled(s[i], pp.f));
    }
    return PMX_OK;
}

template <int T>
struct HostScratch {
    Fe slot[T - 1];   // the engines give the scratch T-1 slots: index T-1 must never be used (ASan checks it)
    Fe get(uint32_t i) const { return slot[i]; }
    void set(uint32_t i, const Fe &x) { slot[i] = x; }
};

// The window engines (HybridEngine<3..9, alpha>): the round loop of pmx_permute.hpp with its dense layers - and the partial section as
// windows closed by one layer each (pmx_mfma.hpp: PMX_MFMA_WINDOW) - through pmx_mfma.hpp's tables, byte strings and row finish; the GEMM
// itself as plain integer sums (no matrix cores on the host).
template <int T>
static int permute_hybrid_mfma_t(const Prepared &pp, uint64_t *states, size_t n) {
    OptTables tb;
    tb.ark = pp.consts.data() + pp.opt_offset;
    tb.mfma = pp.consts.data() + pp.mfma_offset;
    tb.win = pp.consts.data() + pp.win_offset;
    constexpr int KW = mfma_window_for(T);   // the partial section as windows when the width takes them
    if ((int)pp.mfma_window != KW) return PMX_ERR_UNSUPPORTED;
    for (size_t k = 0; k < n; ++k) {
        Fe s[T];
        HostScratch<T> sc;
        for (int i = 0; i < T; ++i) s[i] = fe_from_abi_scaled(load_abi(states + (k * T + i) * 4));
        // a state whose lane 0 is zero takes the shortcut the compress / hash kernels take for a fresh sponge (lane0_zero)
        const uint64_t *lane0 = states + (k * T) * 4;
        const bool z0 = KW > 0 && !(lane0[0] | lane0[1] | lane0[2] | lane0[3]);
        if (pp.c.alpha == 5) permute_hybrid<T, 5, HostScratch<T>, KW>(s, sc, tb, pp.c, pp.one, pp.f, 0, T, z0);
        else permute_hybrid<T, 0, HostScratch<T>, KW>(s, sc, tb, pp.c, pp.one, pp.f, 0, T, z0);
        for (int i = 0; i < T; ++i) store_abi(states + (k * T + i) * 4, fe_to_abi_scaled(s[i], pp.f));
    }
    return PMX_OK;
}
extern "C" int hc_permute_hybrid_mfma(const pmx_config *cfg, uint64_t *states, size_t n) {
    Prepared pp;
    std::string err;
    int rc = prepare(cfg, pp, err);
    if (rc) return rc;
    if (!pp.has_opt || !pp.mfma_dense) return PMX_ERR_UNSUPPORTED;
    switch (pp.t) {
        case 3: return permute_hybrid_mfma_t<3>(pp, states, n);
        case 4: return permute_hybrid_mfma_t<4>(pp, states, n);
        case 5: return permute_hybrid_mfma_t<5>(pp, states, n);
        case 6: return permute_hybrid_mfma_t<6>(pp, states, n);
        case 7: return permute_hybrid_mfma_t<7>(pp, states, n);
        case 8: return permute_hybrid_mfma_t<8>(pp, states, n);
        case 9: return permute_hybrid_mfma_t<9>(pp, states, n);
        default: return PMX_ERR_UNSUPPORTED;
    }
}
// how many windows of a t = 3 config carry their history constant as a small integer (pmx_prepare.hpp: the window's free scale);
// out[0] = windows with a history term, out[1 .. 4] = those with the constant 1 .. 4
extern "C" int hc_window_small_history(const pmx_config *cfg, uint32_t out[5]) {
    Prepared pp;
    std::string err;
    int rc = prepare(cfg, pp, err);
    if (rc) return rc;
    for (int i = 0; i < 5; ++i) out[i] = 0;
    if (!pp.mfma_dense || pp.t != 3 || pp.mfma_window != 3) return PMX_ERR_UNSUPPORTED;
    const size_t kLayer = (size_t)mfma_layer_words_io(3 - 1 + 3, 3), kPer = kLayer + (size_t)mfma_window_hist_words(3, 3);
    const uint32_t n_win = (pp.c.partial_rounds + 2) / 3, first = pp.c.partial_rounds - (n_win - 1) * 3;
    for (uint32_t w = 0; w < n_win; ++w) {
        if ((w == 0 ? first : 3u) < 3) continue;
        out[0] += 1;
        const uint32_t *hist = pp.consts.data() + pp.win_offset + mfma_layer_words(3) + (size_t)w * kPer + kLayer;
        if (hist[0] == kMfmaHistSmallMarker && hist[1] >= 1 && hist[1] <= 4) out[hist[1]] += 1;
    }
    return PMX_OK;
}
// the window size this library was compiled with (tests build one library per size)
extern "C" int hc_mfma_window(int t) { return mfma_window_for(t); }

// run-time-width path (what LdsEngine runs), state in two plain arrays
struct HostState {
    Fe cur[PMX_MAX_WIDTH], nxt[PMX_MAX_WIDTH];
    Fe get(uint32_t i) const { return cur[i]; }
    void set(uint32_t i, const Fe &x) { cur[i] = x; }
    void set_next(uint32_t i, const Fe &x) { nxt[i] = x; }
    void swap() { for (int i = 0; i < PMX_MAX_WIDTH; ++i) { Fe t = cur[i]; cur[i] = nxt[i]; nxt[i] = t; } }
};

extern "C" int hc_permute_rt(const pmx_config *cfg, uint64_t *states, size_t n) {
    Prepared pp;
    std::string err;
    int rc = prepare(cfg, pp, err);
    if (rc) return rc;
    const uint32_t t = pp.t;
    const uint32_t *ark = pp.consts.data();
    const uint32_t *mds = pp.consts.data() + pp.mds_offset;
    for (size_t k = 0; k < n; ++k) {
        HostState st;
        for (uint32_t i = 0; i < t; ++i) st.cur[i] = fe_from_abi(load_abi(states + (k * t + i) * 4), pp.f);
        if (pp.c.alpha == 5) permute_dense_rt<5>(st, t, ark, mds, pp.c, pp.one, pp.f);
        else if (pp.c.alpha == 17) permute_dense_rt<17>(st, t, ark, mds, pp.c, pp.one, pp.f);
        else permute_dense_rt<0>(st, t, ark, mds, pp.c, pp.one, pp.f);
        for (uint32_t i = 0; i < t; ++i) store_abi(states + (k * t + i) * 4, fe_to_abi(st.cur[i], pp.f));
    }
    return PMX_OK;
}

// (the reference's dense schedule exists once, at run-time width: what LdsEngine runs)
extern "C" int hc_permute(const pmx_config *cfg, uint64_t *states, size_t n) { return hc_permute_rt(cfg, states, n); }

// op: 0 mul, 1 sqr(a), 2 dot3(a[0..3), b[0..3)), 3 round trip abi->internal->abi
extern "C" int hc_field_op(const uint64_t modulus[4], int op, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    pmx_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    std::memcpy(cfg.modulus, modulus, 32);
    cfg.full_rounds = 2; cfg.partial_rounds = 0; cfg.rate = 1; cfg.capacity = 0; cfg.alpha = 5;
    uint64_t zeros[8] = {0};
    cfg.ark = zeros; cfg.mds = zeros;
    Prepared pp;
    std::string err;
    int rc = prepare(&cfg, pp, err);
    if (rc) return rc;
    const FieldRt &f = pp.f;
    if (op == 0) {
        store_abi(out, fe_to_abi(mont_mul(fe_from_abi(load_abi(a), f), fe_from_abi(load_abi(b), f), f), f));
    } else if (op == 1) {
        store_abi(out, fe_to_abi(mont_sqr(fe_from_abi(load_abi(a), f), f), f));
    } else if (op == 2) {
        Fe x[3], y[3];
        for (int i = 0; i < 3; ++i) { x[i] = fe_from_abi(load_abi(a + 4 * i), f); y[i] = fe_from_abi(load_abi(b + 4 * i), f); }
        store_abi(out, fe_to_abi(mont_dot<3>(x, y, f), f));
    } else if (op == 3) {
        store_abi(out, fe_to_abi(fe_from_abi(load_abi(a), f), f));
    } else if (op == 4) {
        // the absorb step of the pass kernels (sponge_walk, AbsorbAdjust): a + b on the ABI residues, no multiplication (pmx_device.hip)
        store_abi(out, abi_add_mod(load_abi(a), load_abi(b), f.io + kIoP32));
    } else {
        return PMX_ERR_ARG;
    }
    return PMX_OK;
}

// Products by constants through shifted tables (pmx_field.hpp: tab_dot and the streamed forms), checked one
// operation at a time.  a: n variable elements, c: n constants (both ABI Montgomery), s: addend.
//   form 0: tab_dot<n, false>          sum_i a_i * c_i            n in {1, 3, 6, 9}
// out_i = s_i + a * c_i for n single constants through shifted tables (tab_lanes_stream<n>: the history term of a t = 3 window is n = 1)
template <int N>
static void tab_case(const Prepared &pp, const HostField &hf, const uint64_t *a, const uint64_t *c, const uint64_t *s, uint64_t *out) {
    const FieldRt &f = pp.f;
    Fe add[N];
    U256 cm[N];
    const Fe z = fe_from_abi(load_abi(a), f);
    for (int i = 0; i < N; ++i) {
        add[i] = fe_from_abi(load_abi(s + 4 * i), f);
        std::memcpy(cm[i].l, c + 4 * i, 32);
    }
    std::vector<uint32_t> tab((size_t)N * kTabOneWords, 0u);
    for (int i = 0; i < N; ++i) put_shifted_row(hf, &cm[i], 1, &tab[(size_t)i * kTabOneWords]);
    tab_lanes_stream<N>(z, tab.data(), add, f);
    for (int i = 0; i < N; ++i) store_abi(out + 4 * i, fe_to_abi(add[i], f));
}

extern "C" int hc_tab_op(const uint64_t modulus[4], int n, const uint64_t *a, const uint64_t *c, const uint64_t *s, uint64_t *out) {
    pmx_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    std::memcpy(cfg.modulus, modulus, 32);
    cfg.full_rounds = 2; cfg.partial_rounds = 0; cfg.rate = 1; cfg.capacity = 0; cfg.alpha = 5;
    uint64_t zeros[8] = {0};
    cfg.ark = zeros; cfg.mds = zeros;
    Prepared pp;
    std::string err;
    int rc = prepare(&cfg, pp, err);
    if (rc) return rc;
    if (n == 1) tab_case<1>(pp, pp.hf, a, c, s, out);
    else if (n == 2) tab_case<2>(pp, pp.hf, a, c, s, out);
    else if (n == 8) tab_case<8>(pp, pp.hf, a, c, s, out);
    else return PMX_ERR_ARG;
    return PMX_OK;
}

// Largest value a column accumulator of a table product can reach: `terms` terms of nine products each (every column
// of a table product is full), operand limbs `zmax`, table words < 2^29, plus the two reduction products and the carry.
extern "C" void hc_worst_tab_column(int terms, uint32_t zmax, uint64_t *hi, uint64_t *lo) {
    unsigned __int128 worst = 0, acc = 0;
    for (int k = 0; k < kN + kTabSteps; ++k) {
        if (k < kN) acc += (unsigned __int128)terms * kN * zmax * kMask;
        acc += (unsigned __int128)kTabSteps * kMask * kMask + kMask;   // reduction products (+ the addend of the ADD form)
        if (acc > worst) worst = acc;
        acc >>= kW;
    }
    *hi = (uint64_t)(worst >> 64);
    *lo = (uint64_t)worst;
}

// Largest value any 64-bit column accumulator can reach in mont_dot<T> when every a-limb is `amax`,
// every b-limb `bmax` (worst case over all inputs, ignoring field semantics).  Returned as hi:lo.
extern "C" void hc_worst_column(int terms, uint32_t amax, uint32_t bmax, uint64_t *hi, uint64_t *lo) {
    unsigned __int128 worst = 0, acc = 0;
    for (int k = 0; k < 2 * kN - 1; ++k) {
        const int nprod = (k < kN) ? k + 1 : 2 * kN - 1 - k;
        acc += (unsigned __int128)terms * nprod * amax * bmax;
        const int nred = (k < kN) ? k + 1 : 2 * kN - 1 - k;   // m_j * p_{k-j} terms incl. m_k * p_0
        acc += (unsigned __int128)nred * kMask * kMask;
        if (acc > worst) worst = acc;
        acc >>= kW;
    }
    *hi = (uint64_t)(worst >> 64);
    *lo = (uint64_t)worst;
}

// One row of permute_dense_rt (run-time width, lazily added operands: limbs < 2^30 against constants < 2^29) replayed the
// same way: a full compression before every fourth term, the last <= 3 terms reduced uncompressed.
extern "C" void hc_worst_dense_rt_row(int terms, uint64_t *hi, uint64_t *lo) {
    unsigned __int128 c[2 * kN] = {0}, worst = 0;
    auto note = [&]() {
        for (int k = 0; k < 2 * kN; ++k)
            if (c[k] > worst) worst = c[k];
    };
    const unsigned __int128 prod = (unsigned __int128)((1u << 30) - 1) * kMask, red = (unsigned __int128)kMask * kMask;
    uint32_t pending = 0;
    for (int j = 0; j < terms; ++j) {
        if (pending == kRtLazyTerms) {
            for (int k = 0; k < 2 * kN - 1; ++k) {
                c[k + 1] += c[k] >> kW;
                c[k] &= kMask;
            }
            pending = 0;
        }
        for (int i = 0; i < kN; ++i)
            for (int l = 0; l < kN; ++l) c[i + l] += prod;
        ++pending;
        note();
    }
    for (int k = 0; k < kN; ++k) {
        for (int jj = 0; jj < kN; ++jj) c[k + jj] += red;
        note();
        c[k + 1] += c[k] >> kW;
        note();
    }
    for (int k = kN; k < 2 * kN - 1; ++k) {
        c[k + 1] += c[k] >> kW;
        note();
    }
    *hi = (uint64_t)(worst >> 64);
    *lo = (uint64_t)worst;
}

// Same for mont_sqr with every limb of the operand equal to `amax` (cross products use the doubled limb).
extern "C" void hc_worst_sqr_column(uint32_t amax, uint64_t *hi, uint64_t *lo) {
    unsigned __int128 worst = 0, acc = 0;
    for (int k = 0; k < 2 * kN - 1; ++k) {
        int cross = 0;
        for (int i = 0; i < kN; ++i) {
            const int j = k - i;
            if (j > i && j < kN) ++cross;
        }
        acc += (unsigned __int128)cross * amax * (2ull * amax);
        if ((k & 1) == 0) acc += (unsigned __int128)amax * amax;
        const int nred = (k < kN) ? k + 1 : 2 * kN - 1 - k;
        acc += (unsigned __int128)nred * kMask * kMask;
        if (acc > worst) worst = acc;
        acc >>= kW;
    }
    *hi = (uint64_t)(worst >> 64);
    *lo = (uint64_t)worst;
}

// ---- the absorb / squeeze driver as passes (pmx_sponge_plan.hpp): the plan of one sponge and one pass -------------------
// out = {permute, state_pos, first, count, end_index}
extern "C" void hc_sponge_pass(int squeeze, uint32_t tag, uint32_t index, size_t len, uint32_t rate, uint32_t capacity, size_t pass, uint64_t out[5]) {
    const SpongePass sp = squeeze ? squeeze_pass(tag, index, (uint32_t)len, rate, capacity, (uint32_t)pass) : absorb_pass(tag, index, (uint32_t)len, rate, capacity, (uint32_t)pass);
    out[0] = sp.permute ? 1 : 0;
    out[1] = sp.state_pos;
    out[2] = sp.first;
    out[3] = sp.count;
    out[4] = sp.end_index;
}
extern "C" size_t hc_sponge_passes(int squeeze, size_t len, uint32_t rate) { return squeeze ? squeeze_passes(len, rate) : absorb_passes(len, rate); }
