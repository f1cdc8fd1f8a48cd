// Layers of products by CONSTANTS on the matrix cores (gfx950: v_mfma_i32_32x32x32_i8): the dense layers of the full rounds (round 3,
// t = 7..9) and - round 4 - the linear part of the partial rounds gathered into windows (below: PMX_MFMA_WINDOW), at every width from 3.
//
// A dense layer multiplies the state by a matrix of CONSTANTS: out_i = sum_j c_ij z_j.  On the VALU that is t rows of
// 81 t + 81 limb products (pmx_permute.hpp: matrix_rows_rolled); here it is an int8 GEMM whose N dimension is the 64 states
// of the wave (one per lane), whose K dimension is the bytes of the state and whose M dimension is the bytes of the result:
//
//   * every element z_j a layer takes in is an S-box output (a Montgomery product: below 1.3 p) or a row of the layer before
//     (below 2^248 + p), so with p < 2^255 it is below 2^256: it is re-cut into 32 bytes u_{j,b} - eight words, exactly ONE k-step of
//     the matrix-core instruction per element (round 5; rounds 3-4 carried 36 bytes per element, 33 used, for values up to 2^261:
//     -12 ... -25 % products per row, 16 registers fewer at t = 9; the bound holds for every exponent: alpha = 1 is formed as the
//     product x * 1, alpha = 0 is the constant 1 - pmx_field.hpp: fe_sbox);
//   * for output row i the host stores Y_{j,b} = c_ij * 2^(8 b + 24) mod p in 32 BALANCED signed bytes y_e in [-128, 127]
//     (pmx_prepare.hpp: put_mfma_layer; the modulus' top byte must be <= 126 for 32 of them to do), laid out as the A
//     operand: 16 bytes per lane and k-step, lane l = row e (l & 31) and half (l >> 5) of the k-step;
//   * MFMA bytes are signed, state bytes are not: they enter as u - 128 (one v_xor per register) and the host adds the
//     constant 128 * sum_k Y_k back (per output row, as eight 64-bit word sums);
//   * lanes 32-63 of an MFMA feed the SECOND half of every k-step for the columns (states) of lanes 0-31, not states of
//     their own.  So a k-step is two MFMAs - states 0-31 and states 32-63 of the wave - and before them each lane hands the
//     half of its byte registers its partner lane (+-32) must feed to that partner (v_permlane32_swap, once per layer);
//     afterwards the 32 sums of a state sit half on its own lane and half on the partner: sixteen more swaps per row;
//   * the 32 sums S_e (|S_e| < 2^25) are the integer V = sum_e S_e 2^(8e) = sum u Y < 2^272: eight 64-bit word sums, a
//     carry pass with ONE Montgomery step of 24 bits inside it (the table carries the 2^24: one multiply-add per word on top of the
//     word sum, mfma_row_acc) and a re-cut of the words into nine 29-bit limbs give the row below 2^248 + p - instead of 810 multiplies.
//
// The table of one row (n_in KiB: 9 for a dense row of t = 9, 14 for a row of its window layers) passes through an LDS tile once per
// WORKGROUP, in stages (read per wave from L2 the L2 -> L1 path sets the time), which is why the engines that use this run several waves
// per workgroup (pmx_device.hip: kMfmaWaves, HybridEngine::kTileSteps).  tools/mfma_dense_proto.{py,hip} is the stand-alone form of the same code with its check against Python
// integers; profiles/r03/g_mfma_dense_proto.txt its measurements.
#pragma once

#include "pmx_field.hpp"

namespace pmx {

// Widths whose dense layers go to the matrix cores.  A row costs ~85 VALU instructions of finish plus its share of the state's
// re-cut (27 per element) and 2 t MFMA issue slots (58 clocks of the matrix pipe each), against 81 t + 81 multiplies and their carries on the VALU.
#ifndef PMX_MFMA_MIN_T
#define PMX_MFMA_MIN_T 3   // (10: the library never selects these engines nor builds their tables - INTEGRATION.md section 8)
#endif
#define PMX_MFMA_MAX_T 9   // (a row's mid-column budget and the 36 t / 32 k-steps are laid out for t <= 9)
constexpr int kMfmaElemBytes = 32;   // K bytes per element: every input of a layer is below 2^256 (see above), one k-step each
constexpr int kMfmaShift = 24;       // the tables hold c 2^(8 b + kMfmaShift): the row finish divides by 2^24 (one Montgomery step inside the word sums)
PMX_FN constexpr int mfma_k_steps(int t) { return (t * kMfmaElemBytes + 31) / 32; }
PMX_FN constexpr int mfma_row_words(int t) { return mfma_k_steps(t) * 64 * 4; }             // A operand of one output row
PMX_FN constexpr int mfma_layer_words(int t) { return t * mfma_row_words(t) + t * 16; }     // t rows, then t x 8 int64 corrections
// a layer with n_in input elements and n_out rows (mfma_k_steps / mfma_row_words count INPUT elements)
PMX_FN constexpr int mfma_layer_words_io(int n_in, int n_out) { return n_out * mfma_row_words(n_in) + n_out * 16; }

// Windows of partial rounds (round 4).  A partial round applies its S-box to lane 0 only, so over K of them everything but the K
// S-box outputs is LINEAR: with the carried lanes u in a basis chosen by the host (pmx_prepare.hpp: derive_window_layers) the
// S-box inputs are   x_1 (carried),  x_{k+1} = z_k + u_k + sum_{i<k} h_{k,i} z_i   - additions and (K-1)(K-2)/2 products by
// constants - and the whole linear part of the K rounds (2t-1 products by constants each on the VALU) is ONE layer on the
// matrix cores: inputs (u_1 .. u_{t-1}, z_1 .. z_K), outputs (x_1, u_1 .. u_{t-1}) of the next window, or the state the full
// rounds after the partial section expect.  The first window is the short one when K does not divide the number of partial rounds.
// K = the width (capped at 9, the widest state of these engines): with the history terms as rows on the matrix cores a longer window
// costs one row finish per S-box more and saves a whole layer of t rows per window fewer - 6 was the optimum of round 4, when a history
// term was 81 (k - 1) + 90 multiplies on the VALU (profiles/r05/u_ab_window_as_long_as_the_width.txt: C3 +2.3 %, t = 8 +1.4 %, d9 +2.4 %).
#ifndef PMX_MFMA_WINDOW
#define PMX_MFMA_WINDOW 9
#endif
// window size of a width (0: its partial rounds keep their sparse layers on the VALU)
PMX_FN constexpr int mfma_window_for(int t) {
    return (t >= PMX_MFMA_MIN_T && t <= PMX_MFMA_MAX_T) ? (PMX_MFMA_WINDOW < t ? PMX_MFMA_WINDOW : t) : 0;
}
PMX_FN constexpr int mfma_window_hist(int k) { return (k - 1) * (k - 2) / 2; }   // history constants per window
// The history terms of a window's S-box inputs, x_{k+1} - z_k = u_k + sum_{i<k} h_{k,i} z_i (k = 2 .. K - 1): at t = 3 - ONE term per window -
// a product by a shifted table on the VALU (pmx_field.hpp: tab_dot; profiles/r04/q_ab_history_tables.txt), from t = 4 rows on the matrix
// cores (below; profiles/r05/k_ab_history_rows_on_the_matrix_cores.txt, l_ab_history_rows_t4_t5.txt: t = 4 +1.8 %, 5 +3.7 %, 6 +6.3 %, 7 +4.7 %,
// 8 +4.9 %, 9 +3.6 % over tables (t <= 5) / elements (t >= 6)).
#ifndef PMX_MFMA_HIST_TAB_MAX_T
#define PMX_MFMA_HIST_TAB_MAX_T 3
#endif
PMX_FN constexpr bool mfma_hist_tab(int t) { return t <= PMX_MFMA_HIST_TAB_MAX_T; }
// first word of a window's history table when its single constant is a small integer (second word) and there is no table: the host
// picks the window's free scale for that where it can (pmx_prepare.hpp: derive_window_layers; a table's first word is a 29-bit limb)
constexpr uint32_t kMfmaHistSmallMarker = 0xffffffffu;
// ROWS ON THE MATRIX CORES (round 5): the history term is itself a product by constants of values that are cut into bytes for the window's
// layer anyway - one row of k inputs (z_1 .. z_{k-1}, u_k), its A operand read straight from global memory a whole S-box ahead (k KiB per
// wave: no tile, no barrier), its 2 k products issued right behind S-box k.  One row finish + 2 k products instead of the
// 81 (k - 1) + 90 multiplies of the element form (t = 9: 4 rows and 28 products per window instead of 1170 multiplies).
PMX_FN constexpr bool mfma_hist_rows(int t) { return !mfma_hist_tab(t); }
// With the history terms as rows, the only outputs of a layer inside the partial section that are ever needed as field ELEMENTS are row 0
// (x_1, the first S-box input) and row 1 (u_1: x_2 = z_1 + u_1); the other carried lanes only ever enter matrix-core rows again.  Those
// rows are finished in OPERAND form (mfma_row_finish_operand: the row - below 2^248 + p - read out of the accumulators as eight 32-bit
// words IS the eight operand words) and travel between the layers as such: no re-cut into limbs, no byte cut.
PMX_FN constexpr int mfma_fe_rows(int t) { return mfma_hist_rows(t) ? 2 : t; }
PMX_FN constexpr int mfma_hist_row_words(int k) { return mfma_k_steps(k) * 64 * 4 + 16; }          // the row of x_{k+1}: k inputs, eight correction words
PMX_FN constexpr int mfma_hist_rows_offset(int k) {                                               // words in front of it
    int w = 0;
    for (int j = 2; j < k; ++j) w += mfma_hist_row_words(j);
    return w;
}
// input q of that row, as an index into the window layer's inputs (u_1 .. u_{t-1}, z_1 .. z_K): z_1 .. z_{k-1}, then u_k
PMX_FN constexpr int mfma_hist_row_input(int t, int k, int q) { return q < k - 1 ? t - 1 + q : k - 1; }
PMX_FN constexpr int mfma_hist_tab_offset(int k) {   // words in front of the row of x_{k+1}
    int w = 0;
    for (int j = 2; j < k; ++j) w += tab_row_words(j - 1);
    return w;
}
PMX_FN constexpr int mfma_window_hist_words(int t, int k) { return mfma_hist_tab(t) ? mfma_hist_tab_offset(k) : mfma_hist_rows_offset(k); }
// words of the window tables of a config: the entry layer (t -> t), then per window its layer (t - 1 + K -> t) and its history constants
PMX_FN constexpr size_t mfma_window_words(int t, int k, size_t windows) {
    return (size_t)mfma_layer_words(t) + windows * ((size_t)mfma_layer_words_io(t - 1 + k, t) + (size_t)mfma_window_hist_words(t, k));
}

// one element's K bytes: eight 32-bit words (u - 128 in every byte); x norm and below 2^256
PMX_FN void mfma_cut_element(const Fe &x, uint32_t *w8) {
#if defined(PMX_HOSTCHECK) && !defined(__HIPCC__)
    hostcheck_below_2_256(x);   // tests/hostcheck: the bound the 32-byte form rests on, checked on every element of every layer
#endif
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int bit = 32 * w, li = bit / kW, sh = bit % kW;
        uint64_t v = (uint64_t)x.l[li] >> sh;
        if (li + 1 < kN) v |= (uint64_t)x.l[li + 1] << (kW - sh);
        if (li + 2 < kN && 2 * kW - sh < 32) v |= (uint64_t)x.l[li + 2] << (2 * kW - sh);
        w8[w] = (uint32_t)v ^ 0x80808080u;
    }
}
template <int T>
PMX_FN void mfma_state_words(const Fe *s, uint32_t (&W)[8 * mfma_k_steps(T)]) {
    static_assert(kMfmaElemBytes == 32 && mfma_k_steps(T) == T, "one k-step per element");
    static_for<0, T>([&](auto jj) { mfma_cut_element(s[decltype(jj)::value], &W[8 * decltype(jj)::value]); });
}

// One output row from its 32 sums: R[w][r] = S_{4w + r}, the sum for residue byte 4w + r.  V = sum_e S_e 2^(8e) + the row's correction
// (V < 2^272) is formed as eight 64-bit word sums with carries, and ONE Montgomery step of 24 bits runs in the SAME accumulators:
// m = V (-p^-1) mod 2^24 is known after word 0, and every word takes m p_w (below 2^56) on top of its sum (|t| < 2^50) before its
// carry leaves - one v_mad_u64_u32 per word, where a step of its own behind the carry pass paid four (two zero-extensions, the product,
// a 64-bit add: round 5's first form, 2^32) and the 29-bit step of rounds 3-4 five per limb.  a[0 .. 8] are the words of V + m p: the low
// 24 bits are zero, (V + m p) / 2^24 < 2^248 + p < 2^256 is the row (the table carries the 2^24), read out of a[] at bit 24.
PMX_FN void mfma_row_acc(const int32_t (&R)[8][4], const long long *corr, const FieldRt &f, uint32_t (&a)[9]) {
    // Every term of a word sum is ONE v_mad_i64_i32: the weights (and the 1 that brings in the carry and the first byte) come from
    // registers the compiler cannot see through, or it would turn each into a sign extension, a 64-bit shift and a 64-bit add - and
    // from VECTOR registers, so that the row's correction, a wave-uniform 64-bit value in a scalar pair, can be the addend of the word's
    // first mad as it stands (an instruction takes one scalar operand: with the weights in SGPRs every correction word cost two
    // v_mov, 16 per row).  The accumulator stays inside the signed range (|t| < 2^57), so the carry into the next word (t >> 32) is the
    // high register as it stands.
    int w1, w8, w16, w24;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mov_b32 %0, 1" : "=v"(w1));
    asm("v_mov_b32 %0, 0x100" : "=v"(w8));
    asm("v_mov_b32 %0, 0x10000" : "=v"(w16));
    asm("v_mov_b32 %0, 0x1000000" : "=v"(w24));
#else
    w1 = 1, w8 = 1 << 8, w16 = 1 << 16, w24 = 1 << 24;
#endif
    int carry = 0;
    uint32_t m = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        long long t = corr[w] + (long long)carry * w1;
        t += (long long)R[w][0] * w1;
        t += (long long)R[w][1] * w8;
        t += (long long)R[w][2] * w16;
        t += (long long)R[w][3] * w24;
        if (w == 0) m = ((uint32_t)t * f.pinv) & 0xffffffu;   // (-1/p mod 2^29 serves modulo 2^24 - and lives in a scalar register for the S-boxes anyway)
        t = (long long)((uint64_t)m * f.io[kIoP32 + w] + (uint64_t)t);
        a[w] = (uint32_t)t;
        carry = (int)(t >> 32);
    }
    a[8] = (uint32_t)carry;   // V + m p >= 0: the top carry is not negative
}
// 32 bits of the row from bit `bit` of a[] on (a funnel shift on the device)
PMX_FN uint32_t mfma_row_bits(const uint32_t (&a)[9], int bit) {
    const int wi = bit / 32, sh = bit % 32;
    const uint32_t lo = a[wi], hi = wi + 1 < 9 ? a[wi + 1] : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    return sh == 0 ? lo : __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}
// A row as a field ELEMENT: nine 29-bit limbs, one funnel shift and one mask each (norm, below 2^248 + p).
PMX_FN Fe mfma_row_finish(const int32_t (&R)[8][4], const long long *corr, const FieldRt &f) {
    uint32_t a[9];
    mfma_row_acc(R, corr, f, a);
    Fe row;
#pragma unroll
    for (int k = 0; k < kN; ++k) {
        const uint32_t v = mfma_row_bits(a, kMfmaShift + kW * k);
        row.l[k] = k + 1 < kN ? (v & kMask) : v;   // (the top limb is the top carry)
    }
    return row;
}

#if defined(__HIPCC__)
__device__ __forceinline__ void lane32_swap(uint32_t &x, uint32_t &y);
#endif
// A row in OPERAND form (mfma_fe_rows): its eight words, u - 128 per byte (and on the device the second half handed to the partner
// lane), ARE the operand words of a layer input - no re-cut into limbs, no byte cut.  They travel in the first eight words of an Fe-sized
// container (the scratch slots hold nine words either way).
PMX_FN Fe mfma_row_finish_operand(const int32_t (&R)[8][4], const long long *corr, const FieldRt &f) {
    uint32_t a[9];
    mfma_row_acc(R, corr, f, a);
    Fe row;
#pragma unroll
    for (int k = 0; k < 8; ++k) row.l[k] = mfma_row_bits(a, kMfmaShift + 32 * k) ^ 0x80808080u;
    row.l[8] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int u = 0; u < 4; ++u) lane32_swap(row.l[u], row.l[4 + u]);
#endif
    return row;
}
// the operand words of an input that arrives in operand form already
PMX_FN void mfma_copy_operand(const Fe &x, uint32_t *w8) {
#pragma unroll
    for (int w = 0; w < 8; ++w) w8[w] = x.l[w];
}

#if !defined(__HIPCC__)
// Host form for tests/hostcheck (g++, no matrix cores): the same tables, bytes and finish, the GEMM as plain integer sums.
// the 32 sums of row i of a layer with NQ k-steps over the operand words W, finished
template <int NQ>
inline Fe mfma_row_host(const uint32_t *W, const uint32_t *layer, size_t n_rows, uint32_t i, const FieldRt &f, bool as_element = true) {
    const int8_t *bytes = reinterpret_cast<const int8_t *>(layer);
    const long long *corr = reinterpret_cast<const long long *>(layer + n_rows * (size_t)(NQ * 64 * 4));
    int32_t R[8][4];
    for (int e = 0; e < 32; ++e) {
        long long sum = 0;
        for (int k = 0; k < 32 * NQ; ++k) {
            const int u = (int)(int8_t)((W[k / 4] >> (8 * (k % 4))) & 0xff);
            const int q = k / 32, h = (k % 32) / 16, byte = k % 16;
            sum += (long long)u * bytes[(((size_t)i * NQ + q) * 64 + 32 * h + e) * 16 + byte];
        }
        R[e / 4][e % 4] = (int32_t)sum;
    }
    return as_element ? mfma_row_finish(R, corr + (size_t)i * 8, f) : mfma_row_finish_operand(R, corr + (size_t)i * 8, f);
}
// the operand form of an element (device: the second half of the words goes to the partner lane)
inline void mfma_cut_operand(const Fe &x, uint32_t *w8) { mfma_cut_element(x, w8); }
// rows [lo, hi) of NOUT over the operand words of NIN inputs (NOUT - 1 <= the scratch's slots)
template <int NIN, int NOUT, int THREADS, int TILE_STEPS, class Scratch>
inline void matrix_rows_mfma_w(const uint32_t (&W)[8 * NIN], Fe *out, Scratch &sc, const uint32_t *layer, void * /*tile*/, const FieldRt &f, uint32_t lo, uint32_t hi,
                               uint32_t fe_rows = NOUT) {
    Fe last = out[NOUT - 1];
    for (uint32_t i = lo; i < hi; ++i) {
        const Fe row = mfma_row_host<NIN>(W, layer, NOUT, i, f, i < fe_rows);
        if (i + 1 < (uint32_t)NOUT) sc.set(i, row);
        else last = row;
    }
    static_for<0, NOUT - 1>([&](auto i) { out[i] = sc.get(i); });
    out[NOUT - 1] = last;
}
template <int NIN, int NOUT, int THREADS, int TILE_STEPS, class Scratch>
inline void matrix_rows_mfma_io(const Fe *in, Fe *out, Scratch &sc, const uint32_t *layer, void *tile, const FieldRt &f, uint32_t lo, uint32_t hi,
                                uint32_t fe_rows = NOUT) {
    uint32_t W[8 * NIN];
    mfma_state_words<NIN>(in, W);
    matrix_rows_mfma_w<NIN, NOUT, THREADS, TILE_STEPS>(W, out, sc, layer, tile, f, lo, hi, fe_rows);
}
template <int T, int THREADS, int TILE_STEPS, class Scratch>
inline void matrix_rows_mfma(Fe (&s)[T], Scratch &sc, const uint32_t *layer, void *tile, const FieldRt &f, uint32_t lo, uint32_t hi, uint32_t fe_rows = T) {
    matrix_rows_mfma_io<T, T, THREADS, TILE_STEPS>(s, s, sc, layer, tile, f, lo, hi, fe_rows);
}
// one history row of a window (mfma_hist_rows): load / products / finish, as the device issues them around an S-box
template <int T>
struct MfmaHistRow {
    const uint32_t *loaded = nullptr, *table = nullptr;   // (the next row is loaded while this one is still to be finished)
    uint32_t Wc[8 * PMX_MFMA_WINDOW];
    template <int KK>
    void load(const uint32_t *tab) { loaded = tab; }
    template <int KK, int NW>
    void products(const uint32_t (&W)[NW]) {
        table = loaded;
        static_for<0, KK>([&](auto qq) {
            constexpr int q = decltype(qq)::value, g = mfma_hist_row_input(T, KK, q);
            for (int w = 0; w < 8; ++w) Wc[8 * q + w] = W[8 * g + w];
        });
    }
    template <int KK>
    Fe finish(const FieldRt &f) { return mfma_row_host<KK>(Wc, table, 1, 0, f); }
};
#endif

#if defined(__HIPCC__)   // (tests/hostcheck compiles the headers with g++: no matrix cores there)

typedef int mfma_v16i __attribute__((ext_vector_type(16)));
typedef int mfma_v4i __attribute__((ext_vector_type(4)));

// lanes 32-63 of x <-> lanes 0-31 of y
__device__ __forceinline__ void lane32_swap(uint32_t &x, uint32_t &y) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// Rows [lo, hi) of the layer whose tables start at `layer` (global memory; mfma_layer_words(T) words); the other rows of s
// come back unspecified, like matrix_rows_rolled.  s norm.  `tile`: TILE_STEPS KiB of LDS shared by the workgroup's THREADS
// threads, all of which must arrive here together (two barriers per stage of a row) with every lane active.
// General form: NIN input elements at `in`, rows [lo, hi) of NOUT into out (which may alias in: the inputs are consumed first).
// k-steps of the A operand in flight between the tile and the matrix cores (4 registers each): as many as a stage has, up to 8 - 4 at
// t = 5, whose kernels sit on the 168 registers of three waves per SIMD (8 ahead spilled there: -5.5 %, profiles/r05/d_ab_lds_tile_read_ahead.txt)
PMX_FN constexpr int mfma_lds_ahead(int t) { return t == 5 ? 4 : 8; }
// the operand form of an element: its eight words, the second half handed to the partner lane (+-32) - lanes 32-63 of a product feed
// the second half of every k-step for the states of lanes 0-31
__device__ __forceinline__ void mfma_cut_operand(const Fe &x, uint32_t *w8) {
    mfma_cut_element(x, w8);
#pragma unroll
    for (int u = 0; u < 4; ++u) lane32_swap(w8[u], w8[4 + u]);
}
// the rows over the operand words W of the NIN inputs (mfma_cut_operand each)
// (fe_rows < NOUT only where the width has operand-form rows at all - mfma_fe_rows: elsewhere the operand finish is not even compiled in,
// or the t = 3 kernel, which has none, pays its registers: 32 bytes of scratch per lane at four waves per SIMD, HBM writes 1.34 x)
template <int NIN, int NOUT, int THREADS, int TILE_STEPS, class Scratch>
__device__ __forceinline__ void matrix_rows_mfma_w(const uint32_t (&W)[8 * NIN], Fe *out, Scratch &sc, const uint32_t *layer, mfma_v4i *tile, const FieldRt &f,
                                                   uint32_t lo, uint32_t hi, uint32_t fe_rows = NOUT) {
    constexpr bool kOperandRows = mfma_fe_rows(NOUT) < NOUT;
    constexpr int T = NOUT;
    constexpr int NQ = mfma_k_steps(NIN);
    constexpr int NS = (NQ + TILE_STEPS - 1) / TILE_STEPS;   // the tile holds TILE_STEPS k-steps: a row passes through it in NS stages
    constexpr int SPS = (NQ + NS - 1) / NS;                  // ... of SPS k-steps each (the last one the remainder): balanced
    const uint32_t lane = threadIdx.x & 63;
    const long long *corr = reinterpret_cast<const long long *>(layer + (size_t)T * mfma_row_words(NIN));
    Fe last = out[T - 1];
    // This thread's share of every stage of a row's table, one register buffer per stage: a buffer is refilled with the SAME stage of the
    // NEXT row as soon as its contents are in the tile, so a fetch has a whole row's time to land.  (With one buffer refilled a stage
    // ahead - round 4 - the fetch for the second stage of a two-stage row had the 16 products of the first to hide behind: 512 clocks
    // against an L2 round trip of more; a lone workgroup spent 29 % of its time waiting, profiles/r05/f_c3_by_batch_size.txt.)
    constexpr int kPer = (SPS * 64 + THREADS - 1) / THREADS;
    mfma_v4i pre[NS][kPer];
    auto fetch = [&](uint32_t row, auto st) {
        constexpr int stage = decltype(st)::value;
        const mfma_v4i *src = reinterpret_cast<const mfma_v4i *>(layer) + ((size_t)row * NQ + (size_t)stage * SPS) * 64;
        constexpr uint32_t count = (uint32_t)((stage == NS - 1 ? NQ - stage * SPS : SPS) * 64);
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const uint32_t e = threadIdx.x + q * THREADS;
            if (e < count) pre[stage][q] = src[e];
        }
    };
    if (lo < hi) static_for<0, NS>([&](auto st) { fetch(lo, st); });
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (uint32_t i = lo; i < hi; ++i) {
        mfma_v16i d1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d2 = d1;
        static_for<0, NS>([&](auto st) {
            constexpr int stage = decltype(st)::value, steps = stage == NS - 1 ? NQ - stage * SPS : SPS;
            __syncthreads();   // the readers of the stage before are done with the tile
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const uint32_t e = threadIdx.x + q * THREADS;
                if (e < (uint32_t)steps * 64) tile[e] = pre[stage][q];
            }
            __syncthreads();
            if (i + 1 < hi) fetch(i + 1, st);
            // The A operand of up to kAhead k-steps is read out of the tile BEFORE the first product of the stage, and a slot is
            // refilled as soon as its pair of products has been issued: left to itself the compiler reads one or two k-steps ahead
            // and then waits the whole LDS round trip (~120 clocks, against the 64 a pair of products keeps the matrix pipe busy)
            // in front of every pair - a wave spent 22 % of its cycles in s_waitcnt that way (profiles/r04/s_pmc_window_kernels_c3_c2.txt).
            constexpr int kAhead = steps < mfma_lds_ahead(NOUT) ? steps : mfma_lds_ahead(NOUT);
            mfma_v4i a[kAhead];
#pragma unroll
            for (int qq = 0; qq < kAhead; ++qq) a[qq] = tile[qq * 64 + lane];
            PMX_SCHED_FENCE();   // (the reads stay in front: the scheduler would sink them next to their products again)
#pragma unroll
            for (int qq = 0; qq < steps; ++qq) {
                constexpr int q0 = stage * SPS;
                const int q = q0 + qq;
                const mfma_v4i b1 = {(int)W[8 * q + 0], (int)W[8 * q + 1], (int)W[8 * q + 2], (int)W[8 * q + 3]};
                const mfma_v4i b2 = {(int)W[8 * q + 4], (int)W[8 * q + 5], (int)W[8 * q + 6], (int)W[8 * q + 7]};
                d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[qq % kAhead], b1, d1, 0, 0, 0);   // states 0-31 of the wave
                d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[qq % kAhead], b2, d2, 0, 0, 0);   // states 32-63
                if (qq + kAhead < steps) {
                    PMX_SCHED_FENCE();
                    a[qq % kAhead] = tile[(qq + kAhead) * 64 + lane];
                    PMX_SCHED_FENCE();
                }
            }
        });
        // register v of d1 / d2 holds row 8 (v / 4) + 4 (lane / 32) + v % 4 of the column lane % 32: after the exchange
        // d1[4g + r] is row 8g + r and d2[4g + r] row 8g + 4 + r of THIS lane's state, i.e. word 2g of the row is d1[4g ..], word 2g + 1 d2[4g ..]
        int32_t R[8][4];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            uint32_t x = (uint32_t)d1[v], y = (uint32_t)d2[v];
            lane32_swap(x, y);
            R[2 * (v / 4)][v % 4] = (int32_t)x;
            R[2 * (v / 4) + 1][v % 4] = (int32_t)y;
        }
        // (rows from fe_rows on - wave-uniform - stay in operand form: they only ever enter matrix-core rows again, mfma_fe_rows)
        Fe row;
        if (!kOperandRows || i < fe_rows) row = mfma_row_finish(R, corr + (size_t)i * 8, f);
        else row = mfma_row_finish_operand(R, corr + (size_t)i * 8, f);
        if (i + 1 < (uint32_t)T) sc.set(i, row);
        else last = row;
    }
    static_for<0, T - 1>([&](auto i) { out[i] = sc.get(i); });
    out[T - 1] = last;
}
template <int NIN, int NOUT, int THREADS, int TILE_STEPS, class Scratch>
__device__ __forceinline__ void matrix_rows_mfma_io(const Fe *in, Fe *out, Scratch &sc, const uint32_t *layer, mfma_v4i *tile, const FieldRt &f,
                                                    uint32_t lo, uint32_t hi, uint32_t fe_rows = NOUT) {
    uint32_t W[8 * NIN];
    static_for<0, NIN>([&](auto j) { mfma_cut_operand(in[decltype(j)::value], &W[8 * decltype(j)::value]); });
    matrix_rows_mfma_w<NIN, NOUT, THREADS, TILE_STEPS>(W, out, sc, layer, tile, f, lo, hi, fe_rows);
}
template <int T, int THREADS, int TILE_STEPS, class Scratch>
__device__ __forceinline__ void matrix_rows_mfma(Fe (&s)[T], Scratch &sc, const uint32_t *layer, mfma_v4i *tile, const FieldRt &f, uint32_t lo,
                                                 uint32_t hi, uint32_t fe_rows = T) {
    matrix_rows_mfma_io<T, T, THREADS, TILE_STEPS>(s, s, sc, layer, tile, f, lo, hi, fe_rows);
}

// One history row of a window (mfma_hist_rows): load<KK> fetches the row's A operand (KK k-steps, 16 bytes per lane each) straight from
// global memory - issued a whole S-box before products<KK> consumes it -, products<KK> issues the 2 KK matrix-core instructions over the
// operand words of its inputs, finish<KK> brings the sums home and reduces them.  Every lane of the wave must be active.
template <int T>
struct MfmaHistRow {
    mfma_v4i a[PMX_MFMA_WINDOW > 2 ? PMX_MFMA_WINDOW - 1 : 1];
    mfma_v16i d1, d2;
    const long long *loaded_corr, *corr;   // (the next row is loaded while this one is still to be finished)
    template <int KK>
    __device__ __forceinline__ void load(const uint32_t *table) {
        const mfma_v4i *src = reinterpret_cast<const mfma_v4i *>(table) + (threadIdx.x & 63);
#pragma unroll
        for (int q = 0; q < KK; ++q) a[q] = src[q * 64];
        loaded_corr = reinterpret_cast<const long long *>(table + (size_t)mfma_k_steps(KK) * 64 * 4);
    }
    template <int KK, int NW>
    __device__ __forceinline__ void products(const uint32_t (&W)[NW]) {
        const mfma_v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        d1 = zero, d2 = zero;
        corr = loaded_corr;
        static_for<0, KK>([&](auto qq) {
            constexpr int q = decltype(qq)::value, g = mfma_hist_row_input(T, KK, q);
            const mfma_v4i b1 = {(int)W[8 * g + 0], (int)W[8 * g + 1], (int)W[8 * g + 2], (int)W[8 * g + 3]};
            const mfma_v4i b2 = {(int)W[8 * g + 4], (int)W[8 * g + 5], (int)W[8 * g + 6], (int)W[8 * g + 7]};
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q], b1, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q], b2, d2, 0, 0, 0);
        });
    }
    template <int KK>
    __device__ __forceinline__ Fe finish(const FieldRt &f) {
        int32_t R[8][4];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            uint32_t x = (uint32_t)d1[v], y = (uint32_t)d2[v];
            lane32_swap(x, y);
            R[2 * (v / 4)][v % 4] = (int32_t)x;
            R[2 * (v / 4) + 1][v % 4] = (int32_t)y;
        }
        return mfma_row_finish(R, corr, f);
    }
};

#endif  // __HIPCC__

}  // namespace pmx
