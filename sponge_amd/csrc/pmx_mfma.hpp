// Layers of products by CONSTANTS on the matrix cores (gfx950: v_mfma_i32_32x32x32_i8): the dense layers of the full rounds (round 3,
// t = 7..9) and - round 4 - the linear part of the partial rounds gathered into windows (below: PMX_MFMA_WINDOW), at every width from 3.
//
// A dense layer multiplies the state by a matrix of CONSTANTS: out_i = sum_j c_ij z_j.  On the VALU that is t rows of
// 81 t + 81 limb products (rounds 1-3 did that); here it is an int8 GEMM whose N dimension is the 64 states
// of the wave (one per lane), whose K dimension is the bytes of the state and whose M dimension is the bytes of the result:
//
//   * every element z_j a layer takes in is an S-box output (a Montgomery product: below 1.3 p) or a row of the layer before
//     (below 2^248.1 + p), so with p < 2^255 it is below 2^256: it is re-cut into 32 bytes u_{j,b} - eight words, exactly ONE k-step of
//     the matrix-core instruction per element (round 5; rounds 3-4 carried 36 bytes per element, 33 used, for values up to 2^261:
//     -12 ... -25 % products per row, 16 registers fewer at t = 9; the bound holds for every exponent: alpha = 1 is formed as the
//     product x * 1, alpha = 0 is the constant 1 - pmx_field.hpp: fe_sbox);
//   * for output row i the host stores Y_{j,b} = c_ij * 2^(8 b + 24) mod p in 32 BALANCED signed bytes y_e in [-128, 127]
//     (pmx_prepare.hpp: put_mfma_layer; a residue above what 32 of them reach - moduli near 2^255 - is stored as Y - p), laid out as the A
//     operand: 16 bytes per lane and k-step, lane l = row e (l & 31) and half (l >> 5) of the k-step;
//   * MFMA bytes are signed, state bytes are not: they enter as u - 128 (one v_xor per register) and the host adds the
//     constant 128 * sum_k Y_k back (per output row, as eight 64-bit word sums);
//   * lanes 32-63 of an MFMA feed the SECOND half of every k-step for the columns (states) of lanes 0-31, not states of
//     their own.  So a k-step is two MFMAs - states 0-31 and states 32-63 of the wave - and before them each lane hands the
//     half of its byte registers its partner lane (+-32) must feed to that partner (v_permlane32_swap, once per layer);
//     afterwards the 32 sums of a state sit half on its own lane and half on the partner: sixteen more swaps per row;
//   * the 32 sums S_e (|S_e| <= 32 n_in * 128 * 128 < 2^24 for the n_in <= 17 inputs of the longest layer) are the integer
//     V = sum_e S_e 2^(8e) + the row's correction = sum u Y < n_in * 32 * 255 * p (2^272.1 at n_in = 17, p near 2^255): eight 64-bit word
//     sums, a carry pass with ONE Montgomery step of 24 bits inside it (the table carries the 2^24: one multiply-add per word on top of the
//     word sum, mfma_row_acc) and a re-cut of the words into nine 29-bit limbs give the row (V + m p) / 2^24 < 2^248.1 + p < 2^256 -
//     instead of 810 multiplies.  kMfmaMaxInputs below ties the window size and the widest state to these budgets.
//
// The table of one row (n_in KiB: 9 for a dense row of t = 9, 17 for a row of its window layers) is streamed by every wave for itself,
// from global memory (L2 / L1) into a ring of registers at least six k-steps ahead of its use (matrix_rows_mfma_w below; rounds 3-5 passed
// it through an LDS tile shared by the workgroup, two barriers per stage).  tools/mfma_dense_proto.{py,hip} is the stand-alone form of the
// first version with its check against Python integers; profiles/r03/g_mfma_dense_proto.txt its measurements.
#pragma once

#include "pmx_field.hpp"

namespace pmx {

// Widths whose dense layers go to the matrix cores.  A row costs ~85 VALU instructions of finish plus its share of the state's
// re-cut (27 per element) and 2 t MFMA issue slots (58 clocks of the matrix pipe each), against 81 t + 81 multiplies and their carries on the VALU.
#define PMX_MFMA_MIN_T 3
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
// The longest row of any layer: t - 1 carried lanes + K S-box outputs.  Budgets that rest on it (pmx_mfma.hpp, head): a sum S_e of
// 32 n_in byte products stays inside 2^25 (the row finish adds four of them, shifted by up to 24 bits, into a 64-bit word: |t| < 2^57), and
// the row, (n_in * 32 * 255 * p + p) / 2^24 with p < 2^255, stays below 2^256 - what the 32-byte operand cut of the next layer needs.
constexpr int kMfmaMaxInputs = PMX_MFMA_MAX_T - 1 + (PMX_MFMA_WINDOW < PMX_MFMA_MAX_T ? PMX_MFMA_WINDOW : PMX_MFMA_MAX_T);
static_assert(32 * kMfmaMaxInputs * 128 * 128 < (1 << 25), "a sum of byte products must stay inside the accumulator budget of mfma_row_acc");
static_assert((unsigned long long)kMfmaMaxInputs * 32 * 255 < (1ull << kMfmaShift), "a row must stay below 2^256: n_in * 32 * 255 * p / 2^24 + p < 2 p");
// The history terms of a window's S-box inputs, x_{k+1} - z_k = u_k + sum_{i<k} h_{k,i} z_i (k = 2 .. K - 1): at t = 3 - ONE term per window -
// a product by a shifted table on the VALU (pmx_field.hpp: tab_lanes_stream; profiles/r04/q_ab_history_tables.txt), from t = 4 rows on the matrix
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
// rows are finished in OPERAND form (mfma_row_finish_operand: the row - below 2^248.1 + p - read out of the accumulators as eight 32-bit
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
// (V < 2^272.1) is formed as eight 64-bit word sums with carries, and ONE Montgomery step of 24 bits runs in the SAME accumulators:
// m = V (-p^-1) mod 2^24 is known after word 0, and every word takes m p_w (below 2^56) on top of its sum (|t| < 2^50) before its
// carry leaves - one v_mad_u64_u32 per word, where a step of its own behind the carry pass paid four (two zero-extensions, the product,
// a 64-bit add: round 5's first form, 2^32) and the 29-bit step of rounds 3-4 five per limb.  a[0 .. 8] are the words of V + m p: the low
// 24 bits are zero, (V + m p) / 2^24 < 2^248.1 + p < 2^256 is the row (the table carries the 2^24), read out of a[] at bit 24.
PMX_FN void mfma_row_acc_p(const int32_t (&R)[8][4], const long long *corr, const uint32_t *p32, uint32_t pinv, uint32_t (&a)[9]);
PMX_FN void mfma_row_acc(const int32_t (&R)[8][4], const long long *corr, const FieldRt &f, uint32_t (&a)[9]) {
    mfma_row_acc_p(R, corr, f.io + kIoP32, f.pinv, a);
}
// (corr: the row's eight correction words, p32: the modulus as eight 32-bit words - both wave-uniform; the streamed rows fetch them in front of
// their products, so that the finish does not open with a scalar-cache round trip)
PMX_FN void mfma_row_acc_p(const int32_t (&R)[8][4], const long long *corr, const uint32_t *p32, uint32_t pinv, uint32_t (&a)[9]) {
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
        if (w == 0) m = ((uint32_t)t * pinv) & 0xffffffu;   // (-1/p mod 2^29 serves modulo 2^24 - and lives in a scalar register for the S-boxes anyway)
        t = (long long)((uint64_t)m * p32[w] + (uint64_t)t);
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
// A row as a field ELEMENT: nine 29-bit limbs, one funnel shift and one mask each (norm, below 2^248.1 + p).
PMX_FN Fe mfma_row_limbs(const uint32_t (&a)[9]) {
    Fe row;
#pragma unroll
    for (int k = 0; k < kN; ++k) {
        const uint32_t v = mfma_row_bits(a, kMfmaShift + kW * k);
        row.l[k] = k + 1 < kN ? (v & kMask) : v;   // (the top limb is the top carry)
    }
    return row;
}
PMX_FN Fe mfma_row_finish(const int32_t (&R)[8][4], const long long *corr, const FieldRt &f) {
    uint32_t a[9];
    mfma_row_acc(R, corr, f, a);
    return mfma_row_limbs(a);
}

#if defined(__HIPCC__)
__device__ __forceinline__ void lane32_swap(uint32_t &x, uint32_t &y);
#endif
// A row in OPERAND form (mfma_fe_rows): its eight words, u - 128 per byte (and on the device the second half handed to the partner
// lane), ARE the operand words of a layer input - no re-cut into limbs, no byte cut.  They travel in the first eight words of an Fe-sized
// container (the scratch slots hold nine words either way).
PMX_FN Fe mfma_row_operand(const uint32_t (&a)[9]);
PMX_FN Fe mfma_row_finish_operand(const int32_t (&R)[8][4], const long long *corr, const FieldRt &f) {
    uint32_t a[9];
    mfma_row_acc(R, corr, f, a);
    return mfma_row_operand(a);
}
PMX_FN Fe mfma_row_operand(const uint32_t (&a)[9]) {
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
template <int NIN, int NOUT, class Scratch>
inline void matrix_rows_mfma_w(const uint32_t (&W)[8 * NIN], Fe *out, Scratch &sc, const uint32_t *layer, const FieldRt &f, uint32_t lo, uint32_t hi,
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
template <int NIN, int NOUT, class Scratch>
inline void matrix_rows_mfma_io(const Fe *in, Fe *out, Scratch &sc, const uint32_t *layer, const FieldRt &f, uint32_t lo, uint32_t hi,
                                uint32_t fe_rows = NOUT) {
    uint32_t W[8 * NIN];
    mfma_state_words<NIN>(in, W);
    matrix_rows_mfma_w<NIN, NOUT>(W, out, sc, layer, f, lo, hi, fe_rows);
}
template <int T, class Scratch>
inline void matrix_rows_mfma(Fe (&s)[T], Scratch &sc, const uint32_t *layer, const FieldRt &f, uint32_t lo, uint32_t hi, uint32_t fe_rows = T) {
    matrix_rows_mfma_io<T, T>(s, s, sc, layer, f, lo, hi, fe_rows);
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
// come back unspecified.  s norm.  Every lane of the wave must be active.
// General form: NIN input elements at `in`, rows [lo, hi) of NOUT into out (which may alias in: the inputs are consumed first).
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
// Every wave streams the A operand of its rows straight from global memory (L2 / L1: a layer's table is one contiguous run of 1 KiB
// k-steps, the same for every wave of the launch) into a ring of registers - no LDS tile, no workgroup barrier (round 6; rounds 3-5 passed
// the table through a tile shared by the workgroup's four waves, two barriers per stage: C3 +5.8 %, h9 +6.6 %, d9 +7 %, t = 8 ... 4
// +3.4 / +2.8 / +3.4 / +1.6 / +0.8 %, profiles/r06/c_ab_streamed_rows.txt).
// Slots 0 .. R-1 serve k-steps [0, FULL) of a row (slot q % R, refilled R k-steps ahead; the last R of them with the NEXT row's first R),
// the REM = NQ - FULL k-steps behind them have a slot each, refilled with the next row's same k-step: every load is issued at least R
// k-steps (R pairs of products = 64 R clocks of the matrix pipe) before its use, and the assignment is the same for every row.
// R for a row of nq k-steps: the whole row where it is short (nq <= 9: every k-step is fetched a row ahead), else the ring of 6 .. 9 slots
// that leaves the fewest slots in all (R + nq % R; ties to the longer ring): 17 -> 8 + 1, 15 -> 7 + 1, 13 -> 6 + 1, 11 -> 9 + 2.
PMX_FN constexpr int mfma_ring(int nq) {
    if (nq <= 9) return nq;
    int best = 6;
    for (int r = 7; r <= 9; ++r)
        if (r + nq % r <= best + nq % best) best = r;
    return best;
}
template <int NIN, int NOUT, class Scratch>
__device__ __forceinline__ void matrix_rows_mfma_w(const uint32_t (&W)[8 * NIN], Fe *out, Scratch &sc, const uint32_t *layer, const FieldRt &f,
                                                   uint32_t lo, uint32_t hi, uint32_t fe_rows = NOUT) {
    constexpr bool kOperandRows = mfma_fe_rows(NOUT) < NOUT;
    constexpr int T = NOUT;
    constexpr int NQ = mfma_k_steps(NIN);
    constexpr int R = mfma_ring(NQ), FULL = NQ / R * R, REM = NQ - FULL, SLOTS = R + REM;
    const uint32_t lane = threadIdx.x & 63;
    const long long *corr = reinterpret_cast<const long long *>(layer + (size_t)T * mfma_row_words(NIN));
    // k-step q of row i: the 1 KiB at layer + (i NQ + q) KiB, lane l its 16 bytes at 16 l.  Buffer loads: the table's base in a resource
    // descriptor (four scalar registers), the row in the scalar offset, the k-step in the instruction's immediate where it fits, the lane
    // in ONE vector register - the stream costs no vector registers for addresses (global loads took a 64-bit pair per 4 KiB of reach).
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(layer), 0, 0x7fffffff, 0x00020000);
    const uint32_t lane_off = lane * 16u;
    auto kstep = [&](uint32_t row, int q) -> mfma_v4i {
        const uint32_t soff = (row * (uint32_t)NQ + (uint32_t)(q & ~3)) * 1024u;   // (wave-uniform: scalar arithmetic)
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off + (uint32_t)(q & 3) * 1024u, soff, 0);
        return mfma_v4i{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
    };
    Fe last = out[T - 1];
    mfma_v4i a[SLOTS];
    if (lo < hi) {   // (issued in the order the loop refills the slots: the compiler's wait counts at the top of the loop are the steady state's)
#pragma unroll
        for (int q = 0; q < R; ++q) {
            a[q] = kstep(lo, q);
            PMX_SCHED_FENCE();
        }
#pragma unroll
        for (int e = 0; e < REM; ++e) {
            a[R + e] = kstep(lo, FULL + e);
            PMX_SCHED_FENCE();
        }
    }
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t nxt = i + 1 < hi ? i + 1 : i;   // (the last row refetches itself: no branch in the stream, the loads are never used)
        mfma_v16i d1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d2 = d1;
        // the row's correction words and the modulus words of its finish: fetched (scalar loads) in front of the products, so that the
        // finish does not open with a scalar-cache round trip (t = 8 +2 %, h9 +1.2 %; it also lets the register allocation of the t = 9
        // kernel come out without a spill)
        long long cw[8];
        uint32_t pw[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) cw[w] = corr[(size_t)i * 8 + w], pw[w] = f.io[kIoP32 + w];
        PMX_SCHED_FENCE();
        static_for<0, NQ>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            constexpr int slot = q < FULL ? q % R : R + (q - FULL);
            const mfma_v4i b1 = {(int)W[8 * q + 0], (int)W[8 * q + 1], (int)W[8 * q + 2], (int)W[8 * q + 3]};
            const mfma_v4i b2 = {(int)W[8 * q + 4], (int)W[8 * q + 5], (int)W[8 * q + 6], (int)W[8 * q + 7]};
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[slot], b1, d1, 0, 0, 0);   // states 0-31 of the wave
            d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[slot], b2, d2, 0, 0, 0);   // states 32-63
            PMX_SCHED_FENCE();
            if constexpr (q + R < FULL) a[slot] = kstep(i, q + R);
            else if constexpr (q < FULL) a[slot] = kstep(nxt, q % R);
            else a[slot] = kstep(nxt, q);
            PMX_SCHED_FENCE();
        });
        int32_t R4[8][4];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            uint32_t x = (uint32_t)d1[v], y = (uint32_t)d2[v];
            lane32_swap(x, y);
            R4[2 * (v / 4)][v % 4] = (int32_t)x;
            R4[2 * (v / 4) + 1][v % 4] = (int32_t)y;
        }
        Fe row;
        uint32_t acc9[9];
        mfma_row_acc_p(R4, cw, pw, f.pinv, acc9);
        if (!kOperandRows || i < fe_rows) row = mfma_row_limbs(acc9);
        else row = mfma_row_operand(acc9);
        if (i + 1 < (uint32_t)T) sc.set(i, row);
        else last = row;
    }
    static_for<0, T - 1>([&](auto i) { out[i] = sc.get(i); });
    out[T - 1] = last;
}
template <int NIN, int NOUT, class Scratch>
__device__ __forceinline__ void matrix_rows_mfma_io(const Fe *in, Fe *out, Scratch &sc, const uint32_t *layer, const FieldRt &f,
                                                    uint32_t lo, uint32_t hi, uint32_t fe_rows = NOUT) {
    uint32_t W[8 * NIN];
    static_for<0, NIN>([&](auto j) { mfma_cut_operand(in[decltype(j)::value], &W[8 * decltype(j)::value]); });
    matrix_rows_mfma_w<NIN, NOUT>(W, out, sc, layer, f, lo, hi, fe_rows);
}
template <int T, class Scratch>
__device__ __forceinline__ void matrix_rows_mfma(Fe (&s)[T], Scratch &sc, const uint32_t *layer, const FieldRt &f, uint32_t lo,
                                                 uint32_t hi, uint32_t fe_rows = T) {
    matrix_rows_mfma_io<T, T>(s, s, sc, layer, f, lo, hi, fe_rows);
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
