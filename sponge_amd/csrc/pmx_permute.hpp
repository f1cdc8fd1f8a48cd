// The Poseidon permutation on the internal field form, shared by the device engines and the host-side
// algorithm check (tools/host_field_check.cpp).
//
// Reference: PoseidonSponge::permute, src/poseidon/mod.rs:95-118 - every round is
//   apply_ark (:76-80)  ->  apply_s_box (:63-74; partial rounds touch state[0] only)  ->  apply_mds (:82-93),
// rounds [0, RF/2) and [RF/2+RP, RF+RP) full, the RP in between partial; MDS also after the last round.
#pragma once
#include "pmx_field.hpp"

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
#pragma unroll
        for (int i = 0; i < T; ++i) y[i] = fe_add_lazy(s[i], fe_const(rk + i * kFeStride));   // lazy, B < 2.3
        y[0] = fe_sbox<ALPHA>(y[0], c.alpha, one, f);
        if (is_full_round(r, c)) {
#pragma unroll
            for (int i = 1; i < T; ++i) y[i] = fe_sbox<ALPHA>(y[i], c.alpha, one, f);
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
            Fe row[T];
#pragma unroll
            for (int j = 0; j < T; ++j) row[j] = fe_const(mds + ((size_t)i * T + j) * kFeStride);
            s[i] = mont_dot<T>(y, row, f);   // new[i] = sum_j mds[i][j] * y[j], one reduction
        }
    }
}

// Dense schedule, width known only at run time.  `State` provides get(i) / set(i, x) on the current state and
// set_next(i, x) / swap() on a second buffer (LDS on the device).  Element loops are rolled.
template <int ALPHA, class State>
PMX_FN void permute_dense_rt(State &st, uint32_t t, const uint32_t *ark, const uint32_t *mds, const Rounds &c,
                             const Fe &one, const FieldRt &f) {
    for (uint32_t r = 0; r < c.total_rounds; ++r) {
        const uint32_t *rk = ark + (size_t)r * t * kFeStride;
        const uint32_t n_sbox = is_full_round(r, c) ? t : 1;
        for (uint32_t i = 0; i < t; ++i) {
            Fe x = fe_add_lazy(st.get(i), fe_const(rk + i * kFeStride));
            if (i < n_sbox) x = fe_sbox<ALPHA>(x, c.alpha, one, f);
            st.set(i, x);
        }
        for (uint32_t i = 0; i < t; ++i) {
            Cols acc;
            cols_zero(acc);
            uint32_t pending = 0;
            for (uint32_t j = 0; j < t; ++j) {
                cols_mul_acc(acc, st.get(j), fe_const(mds + ((size_t)i * t + j) * kFeStride));
                if (++pending == 3) {   // at most 3 lazy terms per 64-bit column before re-compressing
                    cols_compress(acc);
                    pending = 0;
                }
            }
            if (pending) cols_compress(acc);
            st.set_next(i, cols_redc(acc, f));
        }
        st.swap();
    }
}

}  // namespace pmx
