// The Poseidon permutation on the internal field form, shared by the device engines and the host-side
// algorithm check (tests/hostcheck/pmx_hostcheck.cpp).
//
// Reference: PoseidonSponge::permute, src/poseidon/mod.rs:95-118 - every round is
//   apply_ark (:76-80)  ->  apply_s_box (:63-74; partial rounds touch state[0] only)  ->  apply_mds (:82-93),
// rounds [0, RF/2) and [RF/2+RP, RF+RP) full, the RP in between partial; MDS also after the last round.
#pragma once
#include "pmx_field.hpp"
#include "pmx_mfma.hpp"

namespace pmx {

// Wave-uniform scalars of a config (by value: SGPRs).
struct Rounds {
    uint32_t rate, capacity, half_full, partial_rounds, total_rounds;
    uint64_t alpha;
};

PMX_FN bool is_full_round(uint32_t r, const Rounds &c) {
    return r < c.half_full || r >= c.half_full + c.partial_rounds;
}

// Dense schedule, width T in registers.
//   ark: [total_rounds][T][kFeStride] words, mds: [T][T][kFeStride] words (internal form, mds[i][j] row-major)
// State elements are norm with B < 1.3 on entry and on exit.
template <int T, int ALPHA>
PMX_FN void permute_dense(Fe (&s)[T], const uint32_t *ark, const uint32_t *mds, const Rounds &c, const Fe &one,
                          const FieldRt &f) {
    for (uint32_t r = 0; r < c.total_rounds; ++r) {
        const uint32_t *rk = ark + (size_t)r * T * kFeStride;
        Fe y[T];
        static_for<0, T>([&](auto i) { y[i] = fe_add_lazy(s[i], fe_const(rk + i * kFeStride)); });   // lazy, B < 2.3
        y[0] = fe_sbox<ALPHA>(y[0], c.alpha, one, f);
        if (is_full_round(r, c)) {
            static_for<1, T>([&](auto i) { y[i] = fe_sbox<ALPHA>(y[i], c.alpha, one, f); });
        }
        static_for<0, T>([&](auto i) {
            Fe row[T];
            static_for<0, T>([&](auto j) { row[j] = fe_const(mds + ((size_t)i * T + j) * kFeStride); });
            s[i] = mont_dot<T>(y, row, f);   // new[i] = sum_j mds[i][j] * y[j], one reduction
        });
    }
}

// ------------------------------------------------------------------------------------------------------------
// Optimised schedule: two exact algebraic rewrites of the same permutation, both done on the constants by the host
// (pmx_prepare.hpp: derive_opt_tables has the derivation).
//  (1) Basis change (Poseidon paper, appendix B): the RP partial rounds only touch state[0] non-linearly, so lanes 1..t-1
//      are carried in a rotated basis in which every linear layer from the one after the last full round of the first
//      half (the "entrance" round) to the one after the second-to-last partial round is SPARSE,
//          s_0' = ONE z_0 + v . u,      u_i' = u_i + w_i z_0         (z_0: S-box output of lane 0, u: lanes 1..t-1),
//      and the round constants of lanes 1..t-1 leave the partial section (they re-enter through lane 0 and through the
//      first full round after it).
//  (2) Diagonal scalings: x -> x^alpha commutes with a diagonal matrix up to its alpha-th power, so the state between two
//      rounds is carried scaled lane by lane, chosen so that one entry per row is exactly ONE: column 0 of every dense
//      layer that feeds a full S-box layer, the coefficient of z_0 in row 0 of every sparse layer.  A NORMALISED row is
//      z_0 + sum_{j>=1} c_j z_j: an addend and t-1 products.  Only the last round's matrix stays fully dense.
// Tables (elements, kFeStride words each; the same matrices as shifted tables next to them):
//   ark      [total_rounds][T]   scaled constants; partial round k keeps only lane 0, the first full round after the
//                                partial section has the deferred constants folded in
//   full     [RF-1][T][T]        one matrix per full round except the entrance round, in round order (full_ordinal);
//                                column 0 is ONE except in the last one
//   sparse   [RP][2T-1]          layer 0 follows the entrance round, layer j partial round j-1: row0[T] = (ONE, v), then w[T-1]
//   bdense   [T][T]              layer after the last partial round (normalised)
// Products by constants per permutation at t = 3, 8 + 31: 175 (reference schedule: 351).  Outputs are identical mod p.
struct OptTables {
    const uint32_t *ark, *mds, *full, *sparse, *bdense;   // elements (mds: the reference matrix, dense schedule only)
    const uint32_t *tab_full, *tab_sparse, *tab_bdense;   // shifted tables (pmx_prepare.hpp layout)
    const uint32_t *mfma;                                 // int8 tables of the dense layers (pmx_mfma.hpp), or null
    const uint32_t *win;                                  // window tables of the partial rounds (pmx_mfma.hpp: mfma_window_words), or null
};

PMX_FN uint32_t full_ordinal(uint32_t r, const Rounds &c) { return r < c.half_full ? r : r - c.partial_rounds - 1; }

// Identity lanes of the sparse layers:  s_i <- s_i + w_i * z0  is one mont_mul_add, which leaves the
// magnitude of s_i uncapped: it grows by at most (1 + B_z p / 2^261) p per layer (1.02 p for the usual S-boxes) from
// B < 1.3 at the entrance round.  Nothing downstream depends on B being small - the lanes are only
// ever multiplied by constants inside reductions that return  T / 2^261 + p  - except that s_i must stay below
// 2^261 (nine normalised limbs).  pmx_prepare.hpp (opt_schedule_lane_headroom) evaluates that condition per config,
// and configs with more partial rounds than it allows (61 to 66 for a 255-bit modulus, depending on 2^261 / p)
// run on the dense schedule instead.

// one row of a dense layer in element form; NORM: column 0 is ONE, the row is z_0 + sum_{j>=1} c_j z_j
template <int T, bool NORM>
PMX_FN Fe row_elem(const Fe (&z)[T], const uint32_t *row, const FieldRt &f) {
    Fe c[T];
    static_for<(NORM ? 1 : 0), T>([&](auto j) { c[j] = fe_const(row + j * kFeStride); });
    if constexpr (NORM && T > 1) return mont_dot_add<T - 1>(&z[1], &c[1], z[0], f);
    else return mont_dot<T>(z, c, f);
}

// a sparse layer in element form: s = (z_0, u) in, the next state out
template <int T>
PMX_FN void sparse_layer_elem(Fe (&s)[T], const uint32_t *sp, const FieldRt &f) {
    const Fe z0 = s[0];
    Fe v[T];
    static_for<1, T>([&](auto j) { v[j] = fe_const(sp + j * kFeStride); });
    if constexpr (T > 1) s[0] = mont_dot_add<T - 1>(&s[1], &v[1], z0, f);   // z_0 + v . u
    PMX_TRACK(0, s[0], f);
    static_for<1, T>([&](auto i) { s[i] = mont_mul_add(z0, fe_const(sp + (T + i - 1) * kFeStride), s[i], f); });
    static_for<1, T>([&](auto i) { PMX_TRACK(1, s[i], f); });
}

// One loop over the rounds: a non-linear stage (every lane, or lane 0 alone) followed by the round's linear layer - sparse,
// normalised dense, or the last round's dense one - so that each block of code exists once in a kernel.
// want_lo / want_hi: lanes [want_lo, want_hi) of the RESULT the caller will read (a fixed-shape hash squeezing its last
// elements, a 2-to-1 compression): the other rows of the last round's matrix are skipped and those lanes hold garbage.
template <int T, int ALPHA>
PMX_FN void permute_opt(Fe (&s)[T], const OptTables &tb, const Rounds &c, const Fe &one, const FieldRt &f, uint32_t want_lo = 0,
                        uint32_t want_hi = T) {
    const uint32_t first_partial = c.half_full, last_partial = c.half_full + c.partial_rounds - 1;
    for (uint32_t r = 0; r < c.total_rounds; ++r) {
        const uint32_t *rk = tb.ark + (size_t)r * T * kFeStride;
        const bool full = r < first_partial || r > last_partial;
        // lanes 1..T-1 of a partial round stay norm (mont_mul_add, see opt_schedule_lane_headroom); lane 0 is re-derived every round
        s[0] = fe_sbox<ALPHA>(fe_add_lazy(s[0], fe_const(rk)), c.alpha, one, f);
        if (full) static_for<1, T>([&](auto i) { s[i] = fe_sbox<ALPHA>(fe_add_lazy(s[i], fe_const(rk + i * kFeStride)), c.alpha, one, f); });
        if (r + 1 >= first_partial && r < last_partial) {   // sparse layer: after the entrance round and every partial round but the last
            sparse_layer_elem<T>(s, tb.sparse + (size_t)(r + 1 - first_partial) * (2 * T - 1) * kFeStride, f);
        } else {
            const uint32_t *mat = full ? tb.full + (size_t)full_ordinal(r, c) * T * T * kFeStride : tb.bdense;
            Fe z[T];
            static_for<0, T>([&](auto i) { z[i] = s[i]; });
            if (r + 1 == c.total_rounds) {
                static_for<0, T>([&](auto i) {
                    if ((uint32_t)i >= want_lo && (uint32_t)i < want_hi) s[i] = row_elem<T, false>(z, mat + (size_t)i * T * kFeStride, f);
                });
            } else {
                static_for<0, T>([&](auto i) { s[i] = row_elem<T, true>(z, mat + (size_t)i * T * kFeStride, f); });
            }
        }
    }
}

// The same schedule with every matrix consumed as SHIFTED TABLES (pmx_field.hpp: tab_dot): all multiplications
// except the S-box are by constants, and a product by a constant whose nine residues C * 2^(29 j + 58) mod p are
// precomputed needs 81 + 18 multiplies instead of 81 + 81 (an N-term row 81 N + 18 instead of 81 N + 81).  Per
// permutation at t = 3: 44,361 multiplies instead of 51,498.  The tables are 9x larger (54 KiB at t = 3) and stream
// through the scalar cache, 81 SGPR operands per product.  Used by every t = 3 kernel for alpha = 5 and 17; what
// makes that stream fit the 100-odd SGPRs of a wave is that FieldRt carries only p, -p^-1 and `unit` by value.
// Scalar-cache warm-up for a table that is about to be streamed: one word of every 64-byte line is loaded (scalar
// loads, no VALU work) and folded into a value the caller keeps alive, so the loads are real and are issued here -
// a whole S-box ahead of the products that consume the table, which then find their lines in the scalar cache
// instead of paying an L2 round trip per chunk with only two waves per SIMD to hide it.
// (kept on: +1..4 % at t = 4..9 on the VALU-row engines.  Tried and not kept, DESIGN.md section 8: warming the element-form row 0 of
// t >= 6 as well, C3 -1 %; row 0 of the wide sparse layers as a streamed shifted table, t = 8, 9 -4..-5 %.)
template <int WORDS>
PMX_FN uint32_t table_touch(const uint32_t *tab) {
    uint32_t x = 0;
#pragma unroll
    for (int w = 0; w < WORDS; w += 16) x ^= tab[w];
    return x ^ tab[WORDS - 1];   // the table need not start on a line boundary: its last words may sit in one more line
}

// (Tried for t = 3 as well and not kept, DESIGN.md section 8: the same warm-up, C2 -0.5 %, hash +0.8 % - four waves per SIMD already
// hide the misses; the explicitly pipelined stream forms for permute_opt_tab, C2 -3 %.)
// shifted tables of one sparse round: row 0 over its T-1 constants v, then the T-1 single constants w
PMX_FN constexpr int sparse_tab_words(int t) { return tab_row_words(t - 1) + (t - 1) * kTabOneWords; }
// one row of a dense layer as a shifted table; NORM: z_0 + sum_{j>=1} z_j c_j (the table holds c_1 ..)
template <int T, bool NORM>
PMX_FN Fe row_tab(const Fe (&z)[T], const uint32_t *tab, const FieldRt &f) {
    if constexpr (NORM && T > 1) return tab_dot<T - 1, true>(&z[1], tab, z[0], f);
    else return tab_dot<T, false>(z, tab, z[0], f);
}

// a sparse layer on shifted tables: s = (z_0, u) in, the next state out
template <int T>
PMX_FN void sparse_layer_tab(Fe (&s)[T], const uint32_t *sp, const FieldRt &f) {
    const Fe z0 = s[0];
    if constexpr (T > 1) s[0] = tab_dot<T - 1, true>(&s[1], sp, z0, f);   // z_0 + v . u
    PMX_TRACK(0, s[0], f);
    static_for<1, T>([&](auto i) { s[i] = tab_dot<1, true>(&z0, sp + tab_row_words(T - 1) + (i - 1) * kTabOneWords, s[i], f); });
    static_for<1, T>([&](auto i) { PMX_TRACK(1, s[i], f); });
}

template <int T, int ALPHA>
PMX_FN void permute_opt_tab(Fe (&s)[T], const OptTables &tb, const Rounds &c, const Fe &one, const FieldRt &f, uint32_t want_lo = 0,
                            uint32_t want_hi = T) {
    const uint32_t first_partial = c.half_full, last_partial = c.half_full + c.partial_rounds - 1;
    for (uint32_t r = 0; r < c.total_rounds; ++r) {
        const uint32_t *rk = tb.ark + (size_t)r * T * kFeStride;
        const bool full = r < first_partial || r > last_partial;
        const bool sparse_layer = r + 1 >= first_partial && r < last_partial;
        s[0] = fe_sbox<ALPHA>(fe_add_lazy(s[0], fe_const(rk)), c.alpha, one, f);
        if (full) static_for<1, T>([&](auto i) { s[i] = fe_sbox<ALPHA>(fe_add_lazy(s[i], fe_const(rk + i * kFeStride)), c.alpha, one, f); });
        if (sparse_layer) {
            sparse_layer_tab<T>(s, tb.tab_sparse + (size_t)(r + 1 - first_partial) * sparse_tab_words(T), f);
        } else {
            const uint32_t *mat = full ? tb.tab_full + (size_t)full_ordinal(r, c) * T * tab_row_words(T) : tb.tab_bdense;
            Fe z[T];
            static_for<0, T>([&](auto i) { z[i] = s[i]; });
            if (r + 1 == c.total_rounds) {
                static_for<0, T>([&](auto i) {
                    if ((uint32_t)i >= want_lo && (uint32_t)i < want_hi) s[i] = row_tab<T, false>(z, mat + (size_t)i * tab_row_words(T), f);
                });
            } else {
                static_for<0, T>([&](auto i) { s[i] = row_tab<T, true>(z, mat + (size_t)i * tab_row_words(T), f); });
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Optimised schedule for wider states (t = 4..9): the state lives in registers, but loops over ELEMENTS in the
// full rounds (t S-boxes, t matrix rows) are rolled to keep the code inside the instruction cache; a rolled
// loop needs dynamic indexing, which goes through `Scratch` (one LDS slot array per lane on the device):
//   sc.set(i, x) / sc.get(i)   i may be a run-time value
// Matrix rows accumulate term by term into explicit 64-bit columns (only one 9-limb constant live at a time),
// re-compressed every 5 terms.  The sparse partial rounds are fully unrolled and never touch the scratch.
// The scratch holds elements 0..T-2 only: the last element / matrix row is peeled off the rolled loops and handled
// with a static index, which keeps the array at 2.25 (T-1) KiB - at t = 9 that is what lets 8 waves (2 per SIMD)
// share a CU's 160 KiB of LDS instead of 7.
// Column budget of a T-term row on normalised operands (every limb < 2^29, a product < 2^58, 64 products per 64-bit
// column): column k receives min(k+1, 17-k) products per term and, in the reduction, as many m_j p_i products, so up to
// six terms need no carry propagation at all (col 8: 54 + 9 = 63), and a longer row overflows only its middle columns
// 6 .. 10.  Those five are compressed once, after kRowMidTerm terms; everything else - including the final carry
// propagation - is left to cols_redc, which consumes each column together with the carry from below.  Per 9-term row:
// 20 carry instructions instead of 136.  tests/test_hostcheck.py replays the worst case (all limbs 2^29 - 1).
constexpr int kRowFreeTerms = 6, kRowMidTerm = 5, kRowMidLo = 6, kRowMidHi = 10;
template <int T>
PMX_FN Fe matrix_row(const Fe (&s)[T], const uint32_t *row, const FieldRt &f) {
    static_assert(T <= 9, "one mid-row compression covers up to 9 terms");
    Cols acc;
    static_for<0, T>([&](auto j) {
        if constexpr (j == 0) cols_mul_init(acc, s[j], fe_const(row + j * kFeStride));
        else cols_mul_acc(acc, s[j], fe_const(row + j * kFeStride));
        if constexpr (T > kRowFreeTerms && j == kRowMidTerm - 1) cols_compress_range<kRowMidLo, kRowMidHi>(acc);
    });
    return cols_redc(acc, f);
}

// Row 0 of a sparse round on the wide engines:  addend + sum_{j < N} s[j] * row[j]  (N = T-1 terms; the addend is the
// S-box output, whose coefficient is one): the addend enters the upper nine columns (addend * 2^261) inside the reduction.
template <int N>
PMX_FN Fe matrix_row_add(const Fe *s, const uint32_t *row, const Fe &addend, const FieldRt &f) {
    static_assert(N <= 9, "one mid-row compression covers up to 9 terms");
    Cols acc;
    static_for<0, N>([&](auto j) {
        if constexpr (j == 0) cols_mul_init(acc, s[j], fe_const(row + j * kFeStride));
        else cols_mul_acc(acc, s[j], fe_const(row + j * kFeStride));
        if constexpr (N > kRowFreeTerms && j == kRowMidTerm - 1) cols_compress_range<kRowMidLo, kRowMidHi>(acc);
    });
    return cols_redc<true>(acc, f, &addend);
}

// Which products of the hybrid engines take shifted tables (pmx_field.hpp: tab_dot, or streamed: tab_dot_stream /
// tab_lanes_stream).  Measured on the default tables: with every matrix as tables t = 4 gains 10 % and t = 5 3-9 %, t = 6 nothing, and
// t = 7..9 LOSE 6-10 % - at 2 waves per SIMD the latency of a 400 KiB table that misses the 16 KiB scalar cache on
// every load is not covered by a one-chunk look-ahead, and the SGPR file has no room for a deeper one.  So up to
// kHybridTabMaxT everything is tables; above it only the identity lanes are (171 -> 108 multiplies each, half
// the stream of the whole round), while the t-term rows, where one reduction is already shared by t products and a
// table would save 63 of 810 multiplies, stay on the element form: +3..7 % at t = 6..9.
constexpr int kHybridTabMaxT = 5;
// (Below that width the tables are consumed through the hand-pipelined stream forms: the compiler's own placement of the scalar loads
// was 1 % (t = 4) and 6 % (t = 5) faster in round 2 and collapsed in round 3 - 538 SGPR spills in the t = 5 sparse layer after an
// unrelated change, 2.6 -> 1.9e8 /s.  The normalised dense layers of t >= 6 skip the product by ONE in a second rolled row block:
// through the last round's t-term row code on the same table they were slower at every width, C3 -0.9 %, t = 7 -2.4 %.)

// The t rows of one dense layer with the element loop rolled (dynamic indexing through the scratch): NORM rows are
// s_0 + sum_{j>=1} c_j s_j - the same code for every row, which is why the normalised entry is column 0 and not the
// diagonal - the last round's rows are t-term dot products.
// Rows [lo, hi) only (the rest of s is left as it was / unspecified): the last round of a permutation whose caller reads
// only those lanes.
template <int T, bool NORM, class Scratch>
PMX_FN void matrix_rows_rolled_tab(Fe (&s)[T], Scratch &sc, const uint32_t *mat, const FieldRt &f, uint32_t lo = 0, uint32_t hi = T) {
    auto row = [&](uint32_t i) {
        const uint32_t *tab = mat + (size_t)i * tab_row_words(T);
        if constexpr (NORM) return tab_dot_stream<T - 1, true>(&s[1], tab, f, &s[0]);
        else return tab_dot_stream<T>(s, tab, f);
    };
    const uint32_t end = hi < (uint32_t)T ? hi : (uint32_t)T - 1;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (uint32_t i = lo; i < end; ++i) sc.set(i, row(i));
    Fe last = s[T - 1];
    if (hi == (uint32_t)T) last = row(T - 1);
    static_for<0, T - 1>([&](auto i) { s[i] = sc.get(i); });
    s[T - 1] = last;
}

template <int T, bool NORM, class Scratch>
PMX_FN void matrix_rows_rolled(Fe (&s)[T], Scratch &sc, const uint32_t *mat, const FieldRt &f, uint32_t lo = 0, uint32_t hi = T) {
    auto row = [&](uint32_t i) {
        const uint32_t *rc = mat + (size_t)i * T * kFeStride;
        if constexpr (NORM) return matrix_row_add<T - 1>(&s[1], rc + kFeStride, s[0], f);
        else return matrix_row<T>(s, rc, f);
    };
    const uint32_t end = hi < (uint32_t)T ? hi : (uint32_t)T - 1;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (uint32_t i = lo; i < end; ++i) sc.set(i, row(i));
    Fe last = s[T - 1];
    if (hi == (uint32_t)T) last = row(T - 1);
    static_for<0, T - 1>([&](auto i) { s[i] = sc.get(i); });
    s[T - 1] = last;
}

// One loop over the rounds, each a non-linear stage (all lanes through the rolled S-box loop, or lane 0 alone) followed by
// a linear stage chosen by the round - sparse layer, normalised dense layer, the last round's dense layer - so that every
// block of code exists once (the sparse layer is used by the entrance round and by the partial rounds alike; two inlined
// copies would not fit the instruction cache at t = 9).
// MFMA: the dense layers run on the matrix cores (pmx_mfma.hpp) - every lane of the wave must be here (the rows exchange operands between
// the lanes of a pair).
// MFMA_WINDOW = K > 0 (with MFMA): the partial rounds run as windows of K (pmx_mfma.hpp) - the layer after the entrance
// round is the windows' entry layer on the matrix cores, and the sparse layers are not in the kernel at all.
// lane0_zero (window engines): the caller knows that s[0] is zero on entry - the capacity lane of a fresh sponge - so the S-box of that
// lane in round 0 is a constant of the config, stored behind the window tables: one S-box of 55 fewer per 2-to-1 compression at t = 3.
template <int T, int ALPHA, class Scratch, bool MFMA = false, int MFMA_WINDOW = 0>
PMX_FN void permute_hybrid(Fe (&s)[T], Scratch &sc, const OptTables &tb, const Rounds &c, const Fe &one,
                           const FieldRt &f, uint32_t want_lo = 0, uint32_t want_hi = T, bool lane0_zero = false) {
    const uint32_t first_partial = c.half_full, last_partial = c.half_full + c.partial_rounds - 1;
    uint32_t guard = 0;   // keeps table_touch's loads alive (see the end of the function)
    for (uint32_t r = 0; r < c.total_rounds; ++r) {
        if constexpr (MFMA && MFMA_WINDOW > 0) {
            if (r == first_partial) {   // the whole partial section: windows, each closed by one layer on the matrix cores
                constexpr int K = MFMA_WINDOW, NIN = T - 1 + K;
                constexpr size_t kLayer = (size_t)mfma_layer_words_io(NIN, T), kPer = kLayer + (size_t)mfma_window_hist_words(T, K);
                const uint32_t n_win = (c.partial_rounds + K - 1) / K;
                uint32_t kw = c.partial_rounds - (n_win - 1) * K;   // the first window is the short one
                const uint32_t *wt = tb.win + mfma_layer_words(T);
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (uint32_t w = 0; w < n_win; ++w, wt += kPer, kw = K) {
                    const uint32_t *hist = wt + kLayer;
                    if constexpr (mfma_hist_rows(T)) {
                        // History terms as rows on the matrix cores (pmx_mfma.hpp: mfma_hist_rows).  The inputs of the window's layer are cut into
                        // operand words as they appear - the carried lanes now, every S-box output when it exists - and the row of x_{k+1}
                        // (inputs z_1 .. z_{k-1}, u_k) is formed right behind S-box k from a table fetched in front of it.
                        // (the carried lanes behind the first arrive in operand form from the layer before - pmx_mfma.hpp: mfma_fe_rows)
                        uint32_t W[8 * NIN];
                        static_for<1, T>([&](auto i) {
                            if constexpr ((int)i < mfma_fe_rows(T)) mfma_cut_operand(s[i], &W[8 * (i - 1)]);
                            else mfma_copy_operand(s[i], &W[8 * (i - 1)]);
                        });
                        MfmaHistRow<T> hr;
                        constexpr bool kFetchAhead = T != 5 && T != 9;   // (t = 5 sits on the 168 registers of three waves per SIMD, t = 9 carries 136 operand words: the 16 of a fetched-ahead table spill)
                        if constexpr (K > 2 && kFetchAhead) hr.template load<2>(hist);
                        Fe z = fe_sbox<ALPHA>(s[0], c.alpha, one, f);                 // z_1 = x_1^alpha: x_1 came whole out of the layer before
                        mfma_cut_operand(z, &W[8 * (T - 1)]);
                        Fe x = fe_add_lazy(s[1], z);                                  // x_2 = z_1 + u_1
                        static_for<2, K + 1>([&](auto kk) {
                            constexpr int k = decltype(kk)::value;                    // S-box k, and in front of it the row of x_{k+1}
                            if ((uint32_t)k <= kw) {
                                const bool row = k < K && (uint32_t)k < kw;           // (wave-uniform)
                                z = fe_sbox<ALPHA>(x, c.alpha, one, f);
                                mfma_cut_operand(z, &W[8 * (T - 2 + k)]);
                                if constexpr (k < K) {
                                    // The row's products are issued BEHIND S-box k, its A operand having been fetched in front of it: issued in
                                    // front as well (they need none of z_k) the 32 sums stay live across the S-box and the kernels of t = 8, 9
                                    // spill (C3 -3.3 %, t = 8 -1.9 %, t = 6, 7 the same: profiles/r05/k_ab_history_rows_in_front_of_or_behind_their_sbox.txt)
                                    if (row) {
                                        if constexpr (!kFetchAhead) hr.template load<k>(hist + mfma_hist_rows_offset(k));
                                        hr.template products<k>(W);
                                        if constexpr (k + 1 < K && kFetchAhead) hr.template load<k + 1>(hist + mfma_hist_rows_offset(k + 1));
                                        x = fe_add_lazy(hr.template finish<k>(f), z);   // x_{k+1} = z_k + (u_k + sum_{i<k} h_{k,i} z_i)
                                    }
                                }
                            } else {
                                mfma_cut_operand(fe_zero(), &W[8 * (T - 2 + k)]);     // a short first window: no such round (its table columns are zero)
                            }
                        });
                        const uint32_t fe_rows = w + 1 < n_win ? (uint32_t)mfma_fe_rows(T) : (uint32_t)T;   // (the last layer feeds S-boxes on every lane)
                        matrix_rows_mfma_w<NIN, T>(W, s, sc, wt, f, 0u, (uint32_t)T, fe_rows);
                    } else {
                    Fe in[NIN];
                    static_for<1, T>([&](auto i) { in[i - 1] = s[i]; });
                    in[T - 1] = fe_sbox<ALPHA>(s[0], c.alpha, one, f);               // z_1 = x_1^alpha: x_1 came whole out of the layer before
                    static_for<1, K>([&](auto kk) {
                        constexpr int k = decltype(kk)::value;                        // z_{k+1} from x_{k+1} = z_k + u_k + sum_{i<k} h_{k,i} z_i
                        if ((uint32_t)k < kw) {
                            Fe x = s[k];
                            bool small_sum = false;                    // (wave-uniform)
                            if constexpr (k == 2) {                    // one constant: the compact single-constant table
                                // ... or no product at all: where the host found a scale of the window under which the constant is 1 .. 4
                                // (pmx_prepare.hpp: the window's free scale) the term is that many lazy additions, normalised with the rest
                                const uint32_t small = hist[0] == kMfmaHistSmallMarker ? hist[1] : 0u;
                                if (small) {
                                    small_sum = true;
                                    x = fe_add_lazy(x, in[T - 1]);
                                    if (small >= 2) x = fe_add_lazy(x, in[T - 1]);
                                    if (small >= 3) x = fe_add_lazy(x, in[T - 1]);
                                    if (small >= 4) x = fe_add_lazy(x, in[T - 1]);     // limbs < 5 * 2^29; with z_k below: < 2^32 - 8 (fe_normalize)
                                } else {
                                    PMX_SCHED_FENCE();
                                    tab_lanes_stream<1>(in[T - 1], hist, &x, f);
                                    PMX_SCHED_FENCE();
                                }
                            } else if constexpr (k >= 3) {
                                const Fe x0 = x;
                                PMX_SCHED_FENCE();
                                x = tab_dot_stream<k - 1, true>(&in[T - 1], hist + mfma_hist_tab_offset(k), f, &x0);
                                PMX_SCHED_FENCE();
                            }
                            x = fe_add_lazy(x, in[T - 2 + k]);
                            if (small_sum) x = fe_normalize(x);        // (below 7.6 p: (7.6 p)^2 < p 2^261 for every p < 2^255)
                            in[T - 1 + k] = fe_sbox<ALPHA>(x, c.alpha, one, f);
                        } else {
                            in[T - 1 + k] = fe_zero();
                        }
                    });
                    matrix_rows_mfma_io<NIN, T>(in, s, sc, wt, f, 0u, (uint32_t)T);
                    }
                }
                r = last_partial;
                continue;
            }
        }
        const uint32_t *rk = tb.ark + (size_t)r * T * kFeStride;
        const bool full = r < first_partial || r > last_partial;
        const bool sparse_layer = r + 1 >= first_partial && r < last_partial;   // entrance round and all partial rounds but the last
        const uint32_t layer = r + 1 - first_partial;                            // index into `sparse` when sparse_layer
        if (full) {            // S-box on every lane
            static_for<0, T - 1>([&](auto i) { sc.set(i, s[i]); });
            uint32_t first = 0;
            if constexpr (MFMA && MFMA_WINDOW > 0) {
                if (r == 0 && lane0_zero && c.half_full > 0) {   // (wave-uniform) lane 0 came in as zero: its S-box output is a constant
                    const uint32_t n_win = (c.partial_rounds + MFMA_WINDOW - 1) / MFMA_WINDOW;
                    sc.set(0, fe_const(tb.win + mfma_window_words(T, MFMA_WINDOW, n_win)));
                    first = 1;
                }
            }
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (uint32_t i = first; i + 1 < (uint32_t)T; ++i)
                sc.set(i, fe_sbox<ALPHA>(fe_add_lazy(sc.get(i), fe_const(rk + i * kFeStride)), c.alpha, one, f));
            s[T - 1] = fe_sbox<ALPHA>(fe_add_lazy(s[T - 1], fe_const(rk + (T - 1) * kFeStride)), c.alpha, one, f);
            static_for<0, T - 1>([&](auto i) { s[i] = sc.get(i); });
        } else if constexpr (!(MFMA && MFMA_WINDOW > 0)) {   // partial round: S-box on lane 0; lanes 1..T-1 stay norm (mont_mul_add, see opt_schedule_lane_headroom)
            if (sparse_layer) {   // warm the scalar cache for this round's tables, a whole S-box ahead of their use
                const uint32_t *rt = tb.tab_sparse + (size_t)layer * sparse_tab_words(T);
                if constexpr (T <= kHybridTabMaxT) guard ^= table_touch<tab_row_words(T - 1)>(rt);
                guard ^= table_touch<(T - 1) * kTabOneWords>(rt + tab_row_words(T - 1));
            }
            s[0] = fe_sbox<ALPHA>(fe_add_lazy(s[0], fe_const(rk)), c.alpha, one, f);
        }
        constexpr bool kWindows = MFMA && MFMA_WINDOW > 0;
        if (sparse_layer && !kWindows) {
            const uint32_t *spt = tb.tab_sparse + (size_t)layer * sparse_tab_words(T);
            const Fe z0 = s[0];
            if constexpr (T <= kHybridTabMaxT) {
                PMX_SCHED_FENCE();
                s[0] = tab_dot_stream<T - 1, true>(&s[1], spt, f, &z0);
                PMX_TRACK(0, s[0], f);
                tab_lanes_stream<T - 1>(z0, spt + tab_row_words(T - 1), &s[1], f);
            } else {
                const uint32_t *sp = tb.sparse + (size_t)layer * (2 * T - 1) * kFeStride;
                s[0] = matrix_row_add<T - 1>(&s[1], sp + kFeStride, z0, f);   // z_0 + v . u
                PMX_TRACK(0, s[0], f);
                // wide states: only the identity lanes take tables - that is where they pay (108 instead of 171 multiplies
                // each); a 9-term row saves 63 of 810 and would double the constant stream
                PMX_SCHED_FENCE();
                tab_lanes_stream<T - 1>(z0, spt + tab_row_words(T - 1), &s[1], f);
            }
            static_for<1, T>([&](auto i) { PMX_TRACK(1, s[i], f); });
        } else {
            // dense layer: a full round's own matrix or B after the last partial round - normalised (z_0 + sum_{j>=1} c_j z_j),
            // except the last round's, whose output is the permutation's
            const bool last = r + 1 == c.total_rounds;
            const uint32_t o = full ? full_ordinal(r, c) : 0;
            if constexpr (MFMA) {
                const uint32_t n_full = c.total_rounds - c.partial_rounds - 1;   // the layer after the last partial round follows the full rounds' own
                const uint32_t *lay = tb.mfma + (size_t)(full ? o : n_full) * mfma_layer_words(T);
                uint32_t fe_rows = (uint32_t)T;
                if (kWindows && sparse_layer) {   // the entrance round's layer leads into the first window (same code, other table)
                    lay = tb.win;
                    fe_rows = (uint32_t)mfma_fe_rows(T);
                }
                matrix_rows_mfma<T>(s, sc, lay, f, last ? want_lo : 0u, last ? want_hi : (uint32_t)T, fe_rows);
            } else if constexpr (T <= kHybridTabMaxT) {
                const uint32_t *mat = full ? tb.tab_full + (size_t)o * T * tab_row_words(T) : tb.tab_bdense;
                if (last) matrix_rows_rolled_tab<T, false>(s, sc, mat, f, want_lo, want_hi);
                else matrix_rows_rolled_tab<T, true>(s, sc, mat, f);
            } else {
                const uint32_t *mat = full ? tb.full + (size_t)o * T * T * kFeStride : tb.bdense;
                if (last) matrix_rows_rolled<T, false>(s, sc, mat, f, want_lo, want_hi);
                else matrix_rows_rolled<T, true>(s, sc, mat, f);
            }
        }
    }
    if (guard == 0x9e3779b9u && f.unit == 0) s[0].l[0] ^= 1;   // never true (unit is 1): the compiler cannot know
}

// ------------------------------------------------------------------------------------------------------------
// Cooperative form of the optimised t = 3 schedule: ONE state is spread over three lanes (lane q holds element
// q), which shortens the dependent chain of a single permutation from 51k to 29k multiplies.  It exists for
// latency-bound launches (the narrow upper levels of a Merkle tree), where lanes are plentiful and the time of
// one permutation on a lone wave is what is paid.  Every round is the same instruction stream for all lanes:
//     x  = s + ark'[r][q]                      (0 for lanes 1, 2 in partial rounds: their constants are deferred)
//     z  = S-box(x) if (full round or q == 0) else x
//     s  = dot3((z_0, z_1, z_2), row[r][q])    after gathering the three z across the lanes
// with row[r][q] = M[q] in full rounds, B[q] in the last partial round, and in the sparse partial rounds
//     q = 0: (m00, v_1, v_2)      q = 1: (w_1, ONE, 0)      q = 2: (w_2, 0, ONE)
// i.e. the identity lanes take "+ u_i" as a product with ONE = 2^261 mod p inside the same reduction, so their
// magnitude is re-normalised every round and the code is uniform across lanes.
// Table (pmx_prepare.hpp): coop[r][q][0] = ark'[r][q], coop[r][q][1..3] = row[r][q][0..2].
constexpr int kCoopElems = 4;   // elements per (round, lane)

template <int ALPHA>
PMX_FN Fe coop_pre(const Fe &s, const uint32_t *entry, bool apply_sbox, const Rounds &c, const Fe &one, const FieldRt &f) {
    const Fe x = fe_add_lazy(s, fe_const(entry));
    const Fe y = fe_sbox<ALPHA>(x, c.alpha, one, f);   // computed by every lane (uniform stream), kept where it applies
    Fe z;
#pragma unroll
    for (int w = 0; w < kN; ++w) z.l[w] = apply_sbox ? y.l[w] : x.l[w];
    return z;
}

PMX_FN Fe coop_post(const Fe (&z)[3], const uint32_t *entry, const FieldRt &f) {
    Fe row[3];
    static_for<0, 3>([&](auto j) { row[j] = fe_const(entry + (1 + j) * kFeStride); });
    return mont_dot<3>(z, row, f);
}
// the same for a normalised dense layer (every lane's row has ONE in column 0): z_0 + c_1 z_1 + c_2 z_2
PMX_FN Fe coop_post_norm(const Fe (&z)[3], const uint32_t *entry, const FieldRt &f) {
    Fe row[3];
    static_for<1, 3>([&](auto j) { row[j] = fe_const(entry + (1 + j) * kFeStride); });
    return mont_dot_add<2>(&z[1], &row[1], z[0], f);
}
// normalised dense layers of the optimised schedule: every full round's but the entrance round's (sparse) and the last
// round's (dense), and the one after the last partial round
PMX_FN bool coop_layer_is_norm(uint32_t r, const Rounds &c) {
    const uint32_t last_partial = c.half_full + c.partial_rounds - 1;
    return r == last_partial || (is_full_round(r, c) && r + 1 != c.half_full && r + 1 != c.total_rounds);
}

// ---- folded sparse rounds (S-box exponents with alpha - 1 a power of two: 5 and 17) ---------------------------------
// What the narrow tree levels pay is the LENGTH of the dependent chain of one permutation, and in the uniform form above
// a sparse round is four multiplications deep: x^2, x^4, x^5, then the row.  The fourth lane of the quad is idle, and
//     s_0' = m00 x^5 + v_1 s_1 + v_2 s_2 = x^4 (x m00) + (v_1 s_1 + v_2 s_2),      s_i' = s_i + x^4 (x w_i)
// so the constants can be multiplied in WHILE the spare lane squares - three multiplications deep instead of four:
//     stage A   lane 3: x x           lanes 0, 1, 2: x m00, x w_1, x w_2                        (one multiplication)
//     stage B   lane 3: x^2 x^2       lanes 1, 2: v_1 s_1, v_2 s_2                              (one multiplication)
//     (alpha = 17: lane 3 squares twice more)
//     stage C   lanes 0, 1, 2: x^(alpha-1) (stage A) + (v_1 s_1 + v_2 s_2 | s_1 | s_2)          (one multiply-add)
// 495 multiplies per sparse round instead of 738 (alpha = 5).  The identity lanes are then plain accumulators as in
// permute_opt (uncapped, bounded by opt_schedule_lane_headroom: each round adds less than (1 + 1.2 / Q) p).
// Table entry of a sparse round (pmx_prepare.hpp): lane 0: [e_k, m00, 0, 0], lane q = 1, 2: [0, w_q, v_q, 0].
template <int ALPHA>
constexpr bool kCoopFolded = ALPHA == 5 || ALPHA == 17;
template <int ALPHA>
constexpr int kCoopExtraSquarings = ALPHA == 17 ? 2 : 0;

PMX_FN Fe fe_select(bool c, const Fe &a, const Fe &b) {
    Fe r;
#pragma unroll
    for (int w = 0; w < kN; ++w) r.l[w] = c ? a.l[w] : b.l[w];
    return r;
}

// x = s_0 + e_k (lazy: limbs < 2^30, below 6.1 p) as seen by every lane of the quad
PMX_FN Fe coop_fold_a(uint32_t q, const Fe &x, const uint32_t *entry, const FieldRt &f) {
    return mont_mul(x, fe_select(q == 3, x, fe_const(entry + kFeStride)), f);
}
PMX_FN Fe coop_fold_b(uint32_t q, const Fe &s, const Fe &res_a, const uint32_t *entry, const FieldRt &f) {
    return mont_mul(fe_select(q == 3, res_a, s), fe_select(q == 3, res_a, fe_const(entry + 2 * kFeStride)), f);
}
// xpow = x^(alpha-1) from lane 3, p1 / p2 = stage B of lanes 1 / 2
PMX_FN Fe coop_fold_c(uint32_t q, const Fe &s, const Fe &xpow, const Fe &res_a, const Fe &p1, const Fe &p2, const FieldRt &f) {
    return mont_mul_add(xpow, res_a, fe_select(q == 0, fe_add_lazy(p1, p2), s), f);
}

// Dense schedule, width known only at run time.  `State` provides get(i) / set(i, x) on the current state and
// set_next(i, x) / swap() on a second buffer (LDS on the device).  Element loops are rolled.
constexpr uint32_t kRtLazyTerms = 3;
template <int ALPHA, class State>
PMX_FN void permute_dense_rt(State &st, uint32_t t, const uint32_t *ark, const uint32_t *mds, const Rounds &c,
                             const Fe &one, const FieldRt &f) {
    for (uint32_t r = 0; r < c.total_rounds; ++r) {
        const uint32_t *rk = ark + (size_t)r * t * kFeStride;
        const uint32_t n_sbox = is_full_round(r, c) ? t : 1;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (uint32_t i = 0; i < t; ++i) {
            Fe x = fe_add_lazy(st.get(i), fe_const(rk + i * kFeStride));
            if (i < n_sbox) x = fe_sbox<ALPHA>(x, c.alpha, one, f);
            st.set(i, x);
        }
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (uint32_t i = 0; i < t; ++i) {
            Cols acc;
            cols_zero(acc);
            uint32_t pending = 0;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (uint32_t j = 0; j < t; ++j) {
                if (pending == kRtLazyTerms) {   // at most 3 lazy terms (limbs < 2^30) per 64-bit column before re-compressing
                    cols_compress(acc);
                    pending = 0;
                }
                cols_mul_acc(acc, st.get(j), fe_const(mds + ((size_t)i * t + j) * kFeStride));
                ++pending;
            }
            st.set_next(i, cols_redc(acc, f));   // the last <= 3 terms go into the reduction uncompressed (27 * 2^59 + 9 * 2^58 + carry < 2^64)
        }
        st.swap();
    }
}

}  // namespace pmx
