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

// ------------------------------------------------------------------------------------------------------------
// Optimised schedule: exact algebraic rewrites of the same permutation, all done on the constants by the host (pmx_prepare.hpp:
// derive_opt_tables and derive_window_layers have the derivations).
//  (1) Basis change (Poseidon paper, appendix B): the RP partial rounds only touch state[0] non-linearly, so lanes 1..t-1 are carried in a
//      rotated basis in which the round constants of those lanes leave the partial section (they re-enter through lane 0 and through the
//      first full round after it).
//  (2) Diagonal scalings: x -> x^alpha commutes with a diagonal matrix up to its alpha-th power, so the state between two rounds is
//      carried scaled lane by lane (2^-5 at both ends of the permutation makes the ABI residue the internal form: pmx_field.hpp).
//  (3) Windows: over K partial rounds everything but the K S-box outputs is linear - one layer of products by constants per window
//      (pmx_mfma.hpp), on the matrix cores like every other layer.
// Tables the window engines read (pmx_prepare.hpp layout): ark' (scaled round constants, kFeStride words each: [total_rounds][T]),
// the int8 tables of the dense layers (`mfma`) and of the windows (`win`).  The quad engine of t = 3 has its own table (`coop`, below).
// Outputs are identical mod p.
struct OptTables {
    const uint32_t *ark;    // elements
    const uint32_t *mfma;   // int8 tables of the dense layers (pmx_mfma.hpp)
    const uint32_t *win;    // window tables of the partial rounds (pmx_mfma.hpp: mfma_window_words)
};

PMX_FN uint32_t full_ordinal(uint32_t r, const Rounds &c) { return r < c.half_full ? r : r - c.partial_rounds - 1; }

// The permutation of the window engines (t = 3 .. 9): the state lives in registers, the loop over the ELEMENTS of a full round's S-boxes
// is rolled to keep the code inside the instruction cache (a rolled loop needs dynamic indexing, which goes through `Scratch`: one LDS
// slot array per lane on the device - sc.set(i, x) / sc.get(i), i may be a run-time value; it holds elements 0 .. T-2 only, the last one
// is peeled off and handled with a static index: 2.25 (T - 1) KiB per wave), and every linear layer is a layer of products by constants
// on the matrix cores (pmx_mfma.hpp):
//   full rounds    S-box on every lane, then the round's dense layer (tb.mfma; the layer of the last full round of the first half - the
//                  "entrance" round - is the windows' entry layer, tb.win)
//   partial rounds as windows of K = MFMA_WINDOW: K S-boxes on lane 0, their inputs x_1 (carried whole), x_{k+1} = z_k + u_k + sum_{i<k}
//                  h_{k,i} z_i - the history terms rows on the matrix cores too (t >= 4) or one shifted-table product / a few additions
//                  (t = 3) - then ONE layer for the whole window; the first window is the short one
//   want_lo / want_hi: lanes [want_lo, want_hi) of the RESULT the caller will read (a fixed-shape hash squeezing its last elements, a
//                  2-to-1 compression): the other rows of the last round's layer are skipped and those lanes hold garbage.
//   lane0_zero     the caller knows that s[0] is zero on entry - the capacity lane of a fresh sponge - so the S-box of that lane in
//                  round 0 is a constant of the config, stored behind the window tables: one S-box of 55 fewer per 2-to-1 compression at t = 3.
// Every lane of the wave must be here (the rows exchange operands between the lanes of a pair).
template <int T, int ALPHA, class Scratch, int MFMA_WINDOW>
PMX_FN void permute_hybrid(Fe (&s)[T], Scratch &sc, const OptTables &tb, const Rounds &c, const Fe &one,
                           const FieldRt &f, uint32_t want_lo = 0, uint32_t want_hi = T, bool lane0_zero = false) {
    static_assert(MFMA_WINDOW > 0 && MFMA_WINDOW <= T, "a window is at most as long as the state is wide");
    const uint32_t first_partial = c.half_full, last_partial = c.half_full + c.partial_rounds - 1;
    for (uint32_t r = 0; r < c.total_rounds; ++r) {
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
                    // t = 3: a window of three S-boxes has ONE history term, h_{2,1} z_1 - a product by a shifted table on the VALU, or no
                    // product at all: where the host found a scale of the window under which the constant is 1 .. 4 (pmx_prepare.hpp: the
                    // window's free scale) the term is that many lazy additions, normalised with the rest
                    static_assert(K <= 3, "the table form of the history terms covers the single term of a window of three");
                    Fe in[NIN];
                    static_for<1, T>([&](auto i) { in[i - 1] = s[i]; });
                    in[T - 1] = fe_sbox<ALPHA>(s[0], c.alpha, one, f);               // z_1 = x_1^alpha: x_1 came whole out of the layer before
                    static_for<1, K>([&](auto kk) {
                        constexpr int k = decltype(kk)::value;                        // z_{k+1} from x_{k+1} = z_k + u_k + sum_{i<k} h_{k,i} z_i
                        if ((uint32_t)k < kw) {
                            Fe x = s[k];
                            bool small_sum = false;                    // (wave-uniform)
                            if constexpr (k == 2) {
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
        // a full round: S-box on every lane (rolled through the scratch), then its dense layer
        const uint32_t *rk = tb.ark + (size_t)r * T * kFeStride;
        static_for<0, T - 1>([&](auto i) { sc.set(i, s[i]); });
        uint32_t first = 0;
        if (r == 0 && lane0_zero && c.half_full > 0) {   // (wave-uniform) lane 0 came in as zero: its S-box output is a constant
            const uint32_t n_win = (c.partial_rounds + MFMA_WINDOW - 1) / MFMA_WINDOW;
            sc.set(0, fe_const(tb.win + mfma_window_words(T, MFMA_WINDOW, n_win)));
            first = 1;
        }
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (uint32_t i = first; i + 1 < (uint32_t)T; ++i)
            sc.set(i, fe_sbox<ALPHA>(fe_add_lazy(sc.get(i), fe_const(rk + i * kFeStride)), c.alpha, one, f));
        s[T - 1] = fe_sbox<ALPHA>(fe_add_lazy(s[T - 1], fe_const(rk + (T - 1) * kFeStride)), c.alpha, one, f);
        static_for<0, T - 1>([&](auto i) { s[i] = sc.get(i); });
        // its layer: the round's own matrix - or, behind the entrance round, the windows' entry layer (same code, other table);
        // the last round's output is the permutation's
        const bool last = r + 1 == c.total_rounds, entrance = r + 1 == first_partial;
        const uint32_t *lay = tb.mfma + (size_t)full_ordinal(r, c) * mfma_layer_words(T);
        uint32_t fe_rows = (uint32_t)T;
        if (entrance) {
            lay = tb.win;
            fe_rows = (uint32_t)mfma_fe_rows(T);
        }
        matrix_rows_mfma<T>(s, sc, lay, f, last ? want_lo : 0u, last ? want_hi : (uint32_t)T, fe_rows);
    }
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
// the optimised schedule's sparse rounds (uncapped, bounded by opt_schedule_lane_headroom: each round adds less than (1 + 1.2 / Q) p).
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
