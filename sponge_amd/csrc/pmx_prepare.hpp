// Host-side preparation of a validated config for the kernels: modulus view, conversion constants and the
// round constants / MDS matrix rewritten into the internal field form (29-bit limbs, x * 2^261 mod p).
// Shared by pmx_api.cpp (which uploads the table) and the CPU-side algorithm check in tests/hostcheck.
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/poseidon_mi355x.h"
#include "pmx_field.hpp"
#include "pmx_host_field.hpp"
#include "pmx_permute.hpp"

namespace pmx {

inline void to_limbs29(const U256 &x, uint32_t out[kN]) {
    for (int i = 0; i < kN; ++i) {
        const int bit = kW * i, wi = bit / 64, sh = bit % 64;
        unsigned __int128 pair = x.l[wi];
        if (wi + 1 < 4) pair |= (unsigned __int128)x.l[wi + 1] << 64;
        out[i] = (uint32_t)(pair >> sh) & kMask;
    }
}

inline U256 times_pow2(const HostField &f, U256 v, int k) {  // v * 2^k mod p
    for (int i = 0; i < k; ++i) v = f.add(v, v);
    return v;
}

struct Prepared {
    HostField hf;
    FieldRt f;
    Fe one;                         // 2^261 mod p
    Rounds c;
    uint32_t t;
    // one table, kFeStride words per element, offsets in words:
    //   [0, mds_offset)            ark      [rounds][t]      dense schedule
    //   [mds_offset, opt_offset)   mds      [t][t]
    //   -- only when has_opt (see pmx_permute.hpp: OptTables) --
    //   opt_offset                 ark'     [rounds][t]
    //   -- t == 3 only: the element form of the optimised schedule's matrices, the source of the quad engine's table (coop) --
    //   opt_full_offset            full     [RF-1][t][t]     one matrix per full round except the entrance round
    //   opt_sparse_offset          sparse   [RP][2t-1]       layer 0 follows the entrance round, layer j partial round j-1
    //   opt_bdense_offset          bdense   [t][t]           layer after the last partial round
    std::vector<uint32_t> consts;
    size_t mds_offset;
    bool has_opt;
    size_t opt_offset, opt_full_offset, opt_sparse_offset, opt_bdense_offset;
    //   -- only when has_opt and t == 3 (pmx_permute.hpp: cooperative schedule) --
    //   coop_offset                coop     [rounds][3][4]
    size_t coop_offset;
    //   -- only when mfma_dense: the dense layers as int8 GEMM operands (pmx_mfma.hpp); RF - 1 layers of `full`, then bdense --
    //   mfma_offset                [RF][mfma_layer_words(t)]
    size_t mfma_offset;
    bool mfma_dense;
    //   -- only when mfma_dense and mfma_window_for(t) > 0: the partial section as windows (pmx_mfma.hpp: mfma_window_words) --
    size_t win_offset;
    uint32_t mfma_window;   // K, or 0: no windows (the sparse layers run on the VALU)
    size_t io_offset;   // kIoWords words behind FieldRt::io
};

// One dense layer as the A operands of pmx_mfma.hpp: `rows` = t rows of t constants (ABI Montgomery residues, row-major).
// Row i, k-step q, lane l: 16 bytes = bytes e = l & 31 of the residues  Y = c_ij * 2^(8 b + 24) mod p  for the 16 positions
// k = 32 q + 16 (l >> 5) + 0..15 of the state's byte string (k = 32 j + b), as balanced signed bytes; then per row the eight
// word sums of 128 * sum_k Y_k (the state's bytes enter as u - 128).
// General form: n_out rows of n_in constants; aff (may be null): one constant per row added to the row's value.
// (the table carries the 2^24 the row's finish divides by - element or operand form, the Montgomery step is the same: pmx_mfma.hpp, mfma_row_acc)
inline void put_mfma_layer_io(const HostField &hf, const U256 *rows, size_t n_in, size_t n_out, const U256 *aff, uint32_t *dst) {
    const size_t nq = (size_t)mfma_k_steps((int)n_in), row_words = (size_t)mfma_row_words((int)n_in);
    int8_t *bytes = reinterpret_cast<int8_t *>(dst);
    long long *corr = reinterpret_cast<long long *>(dst + n_out * row_words);
    // 32 balanced digits reach from -128 S to 127 S, S = (256^32 - 1) / 255: every residue of a modulus whose top byte is at most 126,
    // but not the residues above 127 S = 0.996 * 2^255 of a larger one (2^255 - 19).  Such a residue Y is stored as Y - p instead (congruent,
    // above -2^255 > -128 S), and the row's correction takes 255 p for it - the term then contributes U Y + (255 - U) p for the state byte
    // U in [0, 255]: non-negative, congruent, and at most 255 p like every other term, so every bound of pmx_mfma.hpp stands as it is
    // (round 6; until then those moduli had no tables and ran on VALU rows).
    for (size_t i = 0; i < n_out; ++i) {
        long long colsum[32] = {0};
        size_t negatives = 0;
        constexpr int shift = kMfmaShift;
        for (size_t j = 0; j < n_in; ++j) {
            U256 y = times_pow2(hf, hf.from_mont(rows[i * n_in + j]), shift);
            for (size_t b = 0; b < (size_t)kMfmaElemBytes; ++b) {   // (the inputs of a layer are below 2^256: pmx_mfma.hpp)
                const size_t k = j * kMfmaElemBytes + b, q = k / 32, r = k % 32, h = r / 16, byte = r % 16;
                // balanced bytes of the 256-bit pattern v: digit e in [-128, 127], carry into the next; returns the carry out of digit 31
                auto digits = [&](const U256 &v, int8_t *out32) {
                    unsigned carry = 0;
                    for (size_t e = 0; e < 32; ++e) {
                        int dgt = (int)((v.l[e / 8] >> (8 * (e % 8))) & 0xff) + (int)carry;
                        carry = 0;
                        if (dgt >= 128) {
                            dgt -= 256;
                            carry = 1;
                        }
                        out32[e] = (int8_t)dgt;
                    }
                    return carry;
                };
                int8_t d32[32];
                if (digits(y, d32)) {          // Y > 127 S: the digits of the two's-complement pattern of Y - p (its carry out is its sign)
                    U256 neg;
                    unsigned __int128 borrow = 0;
                    for (int w = 0; w < 4; ++w) {
                        const unsigned __int128 dif = (unsigned __int128)y.l[w] - hf.p.l[w] - borrow;
                        neg.l[w] = (uint64_t)dif;
                        borrow = (dif >> 64) & 1;
                    }
                    const unsigned sign = digits(neg, d32);
                    if (!borrow || !sign) std::abort();   // (cannot happen: Y < p, and Y - p > -2^255 has 32 balanced digits)
                    ++negatives;
                }
                for (size_t e = 0; e < 32; ++e) {
                    bytes[(((i * nq + q) * 64 + 32 * h + e) * 16) + byte] = d32[e];
                    colsum[e] += d32[e];
                }
                y = times_pow2(hf, y, 8);
            }
        }
        // the row's constant in the units of V: its internal form (x 2^261) times the 2^kMfmaShift the finish divides by
        U256 add = {{0, 0, 0, 0}};
        if (aff) add = times_pow2(hf, hf.from_mont(aff[i]), 261 + shift);
        for (size_t w = 0; w < 8; ++w) {
            long long v = 0;
            for (size_t tt = 0; tt < 4; ++tt) v += (128 * colsum[4 * w + tt]) * (1ll << (8 * tt));
            const long long p_word = (long long)((hf.p.l[w / 2] >> (32 * (w % 2))) & 0xffffffffull);   // (negatives * 255 < 2^18: the product stays below 2^50)
            corr[i * 8 + w] = v + (long long)((add.l[w / 2] >> (32 * (w % 2))) & 0xffffffffull) + (long long)(negatives * 255) * p_word;
        }
    }
}
inline void put_mfma_layer(const HostField &hf, const U256 *rows, size_t t, uint32_t *dst) { put_mfma_layer_io(hf, rows, t, t, nullptr, dst); }

// shifted table of a ROW of n constants (given as ABI Montgomery residues C_i * 2^256) in the chunked layout of
// pmx_field.hpp (tab_index): limb k of C_i * 2^(29 j + 58) mod p; tab_row_words(n) words, padding left 0
inline void put_shifted_row(const HostField &hf, const U256 *c_mont, size_t n, uint32_t *tab) {
    for (size_t i = 0; i < n; ++i) {
        U256 tj = times_pow2(hf, hf.from_mont(c_mont[i]), 58);
        for (int j = 0; j < kN; ++j) {
            uint32_t limbs[kN];
            to_limbs29(tj, limbs);
            for (int k = 0; k < kN; ++k) {
                const size_t ng = (n + kTabChunk - 1) / kTabChunk;
                const size_t at = n == 1 ? (size_t)(k / kTabChunk) * kTabChunkWords + (k % kTabChunk) * kN + j
                                         : ((size_t)k * ng + i / kTabChunk) * kTabChunkWords + (i % kTabChunk) * kN + j;
                tab[at] = limbs[k];
            }
            tj = times_pow2(hf, tj, kW);
        }
    }
}

// ---- small dense linear algebra over the ABI Montgomery form (host only) ----------------------------------------
typedef std::vector<std::vector<U256>> HostMat;

inline HostMat mat_identity(const HostField &f, size_t n) {
    HostMat m(n, std::vector<U256>(n, U256{{0, 0, 0, 0}}));
    for (size_t i = 0; i < n; ++i) m[i][i] = f.r;
    return m;
}

inline HostMat mat_mul(const HostField &f, const HostMat &a, const HostMat &b) {
    const size_t n = a.size(), k = b.size(), m = b[0].size();
    HostMat r(n, std::vector<U256>(m, U256{{0, 0, 0, 0}}));
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < m; ++j)
            for (size_t l = 0; l < k; ++l) r[i][j] = f.add(r[i][j], f.mul(a[i][l], b[l][j]));
    return r;
}

inline std::vector<U256> mat_vec(const HostField &f, const HostMat &a, const std::vector<U256> &v) {
    std::vector<U256> r(a.size(), U256{{0, 0, 0, 0}});
    for (size_t i = 0; i < a.size(); ++i)
        for (size_t j = 0; j < v.size(); ++j) r[i] = f.add(r[i], f.mul(a[i][j], v[j]));
    return r;
}

// Gauss-Jordan inverse; false if singular
inline bool mat_inverse(const HostField &f, HostMat a, HostMat &inv) {
    const size_t n = a.size();
    inv = mat_identity(f, n);
    for (size_t col = 0; col < n; ++col) {
        size_t piv = col;
        while (piv < n && u256_is_zero(a[piv][col])) ++piv;
        if (piv == n) return false;
        std::swap(a[piv], a[col]);
        std::swap(inv[piv], inv[col]);
        const U256 d = f.inverse(a[col][col]);
        for (size_t j = 0; j < n; ++j) {
            a[col][j] = f.mul(a[col][j], d);
            inv[col][j] = f.mul(inv[col][j], d);
        }
        for (size_t i = 0; i < n; ++i) {
            if (i == col || u256_is_zero(a[i][col])) continue;
            const U256 k = a[i][col];
            for (size_t j = 0; j < n; ++j) {
                a[i][j] = f.sub(a[i][j], f.mul(k, a[col][j]));
                inv[i][j] = f.sub(inv[i][j], f.mul(k, inv[col][j]));
            }
        }
    }
    return true;
}

// The optimised schedule updates its identity lanes without a magnitude cap (pmx_permute.hpp: mont_mul_add).
// With Q = 2^261 / p (>= 64) and every lane below Q p (nine normalised limbs), row 0 of a sparse layer returns at most
// (t-1) + B_z + 1 <= 10.3 p for t <= 9, so the S-box input x = row0 + constant is below 11.3 p and the S-box output z_0
// below B_z p with
//     alpha >= 4: the chain ends in a product with x of a value < 1.05 p  ->  B_z < 1.05 * 11.3 / 64 + 1 < 1.2
//     alpha = 3: x^2 * x, x^2 < 3 p -> B_z < 1.6;   alpha = 2: B_z < 3;   alpha = 1: z_0 = x as the product x * 1 (fe_sbox), B_z < 1.2;
//     alpha = 0: z_0 = 1.
// A lane starts as an S-box output of the entrance round (< 1.3 p), takes its first update there and one more per sparse
// partial round, growing by at most (B_z / Q + 1) p each time; the dense layer after the last partial round adds its own
// + p.  The schedule is used when all of that stays below Q.
inline bool opt_schedule_lane_headroom(long double two_261_over_p, uint32_t partial_rounds, uint64_t alpha) {
    const long double bz = alpha == 0 ? 1.0L : alpha == 2 ? 3.0L : alpha == 3 ? 1.6L : 1.3L;
    const long double growth = 1.0L + bz / two_261_over_p;
    const long double worst = 1.4L + growth * (long double)partial_rounds + 1.5L;
    return worst < two_261_over_p;
}

inline U256 host_pow(const HostField &f, const U256 &x, uint64_t e) {   // x^e in the Montgomery domain
    U256 acc = f.r;
    for (int bit = 63; bit >= 0; --bit) {
        acc = f.mul(acc, acc);
        if ((e >> bit) & 1) acc = f.mul(acc, x);
    }
    return acc;
}

// index of a full round's matrix in the `full` table: every full round but the entrance round (the last one of the first
// half, whose linear layer is sparse[0]) has its own, in round order
inline size_t full_matrix_ordinal(uint32_t r, uint32_t half_full, uint32_t rp) { return r < half_full ? r : r - rp - 1; }

// Derives the tables of the optimised schedule from (ark, mds).  Two exact rewrites of the same permutation:
//
// (1) Basis change on lanes 1..t-1 (Poseidon paper, appendix B).  The partial rounds apply no S-box to those lanes, so they
//     may be carried in any basis N_j = diag(1, Nh_j).  With B_j = M N_j the layer after S-box layer j becomes
//     sparse_j = N_{j+1}^-1 B_j = [[b00, bv], [Bh^-1 bw, I]]  for Nh_{j+1} = Bh_j (lower-right block of B_j): 2t-1 products
//     instead of t^2.  Layer j = 0 is the one after the LAST FULL ROUND OF THE FIRST HALF (the "entrance" round: the state
//     it produces is only ever used in the new basis), j = 1..RP-1 follow the partial rounds, and the layer after the last
//     partial round is dense (B_RP: the full rounds need every lane back).  Round constants of lanes 1.. are deferred
//     (vector D) and re-enter through lane 0 (e_j) and through the first full round after the partial section.
//
// (2) Diagonal scalings.  x -> x^alpha commutes with a diagonal matrix up to its alpha-th power, S(D x) = D^alpha S(x), so
//     the state between two rounds may be carried as D_r s_r for any invertible diagonal D_r (the same D at both ends): every
//     constant is rescaled on the host, nothing else changes.  One free scalar per lane and round boundary makes one matrix
//     entry per row equal to ONE: column 0 of every dense layer that feeds a full S-box layer, the coefficient of the
//     S-box output in row 0 of every sparse layer (the lanes keep their unit coefficients).  A normalised row is
//     z_0 + sum_{j >= 1} c_j z_j: t-1 products and an addend.  Only the last round's matrix stays fully dense (its
//     output is the permutation's, unscaled).
//
// Products by constants per permutation: (RF-2) t (t-1) + t^2 + RP (2t-2) + t (t-1), e.g. t = 3, 8 + 31: 175 (dense
// schedule 351); t = 9, 8 + 57: 1497 (5265).
inline bool derive_opt_tables(const HostField &f, uint32_t t, uint32_t half_full, uint32_t rp, uint32_t rounds, uint64_t alpha,
                              const std::vector<U256> &ark, const HostMat &M, std::vector<U256> &ark_opt,
                              std::vector<U256> &fullmat, std::vector<U256> &sparse, std::vector<U256> &bdense,
                              std::vector<U256> *entrance_scale = nullptr, std::vector<U256> *exit_scale = nullptr) {
    if (rp == 0 || half_full == 0 || t < 2 || half_full + rp >= rounds) return false;
    const size_t n = t - 1, per = 2 * (size_t)t - 1;
    const uint32_t rf_total = rounds - rp, entrance = half_full - 1, last_partial = half_full + rp - 1;
    const U256 zero = {{0, 0, 0, 0}};
    // ---- (1) basis change ----------------------------------------------------------------------------------------
    ark_opt = ark;
    sparse.assign((size_t)rp * per, zero);
    HostMat Nh = mat_identity(f, n);
    HostMat B;
    std::vector<U256> D(n, zero);   // deferred constants of lanes 1.., in the current basis
    for (uint32_t j = 0; j <= rp; ++j) {
        // B = M * diag(1, Nh)
        B.assign(t, std::vector<U256>(t, zero));
        for (size_t i = 0; i < t; ++i) {
            B[i][0] = M[i][0];
            for (size_t c = 0; c < n; ++c)
                for (size_t l = 0; l < n; ++l) B[i][1 + c] = f.add(B[i][1 + c], f.mul(M[i][1 + l], Nh[l][c]));
        }
        if (j == rp) break;
        HostMat Bh(n, std::vector<U256>(n)), Bh_inv;
        for (size_t i = 0; i < n; ++i)
            for (size_t c = 0; c < n; ++c) Bh[i][c] = B[1 + i][1 + c];
        if (!mat_inverse(f, Bh, Bh_inv)) return false;
        U256 *sp = &sparse[(size_t)j * per];
        for (size_t c = 0; c < t; ++c) sp[c] = B[0][c];                  // row0 = (b00, bv)
        std::vector<U256> bw(n);
        for (size_t i = 0; i < n; ++i) bw[i] = B[1 + i][0];
        const std::vector<U256> w = mat_vec(f, Bh_inv, bw);              // Bh^-1 bw
        for (size_t i = 0; i < n; ++i) sp[t + i] = w[i];
        // constants of the round that follows (partial round j) in the new basis
        const size_t rn = (size_t)(half_full + j) * t;
        std::vector<U256> c1(n);
        for (size_t i = 0; i < n; ++i) c1[i] = ark[rn + 1 + i];
        const std::vector<U256> ct = mat_vec(f, Bh_inv, c1);
        U256 e = ark[rn];
        for (size_t c = 0; c < n; ++c) e = f.add(e, f.mul(B[0][1 + c], D[c]));   // + bv . D
        ark_opt[rn] = e;
        for (size_t i = 0; i < n; ++i) { ark_opt[rn + 1 + i] = zero; D[i] = f.add(D[i], ct[i]); }
        Nh = Bh;
    }
    bdense.assign((size_t)t * t, zero);
    for (size_t i = 0; i < t; ++i)
        for (size_t c = 0; c < t; ++c) bdense[i * t + c] = B[i][c];
    // E = B [0; D] joins the constants of the first full round after the partial section
    const size_t rf = (size_t)(half_full + rp) * t;
    for (size_t i = 0; i < t; ++i) {
        U256 e = zero;
        for (size_t c = 0; c < n; ++c) e = f.add(e, f.mul(B[i][1 + c], D[c]));
        ark_opt[rf + i] = f.add(ark[rf + i], e);
    }
    fullmat.assign((size_t)(rf_total - 1) * t * t, zero);
    for (size_t o = 0; o + 1 < rf_total; ++o)
        for (size_t i = 0; i < t; ++i)
            for (size_t c = 0; c < t; ++c) fullmat[(o * t + i) * t + c] = M[i][c];
    // ---- (2) diagonal scalings -------------------------------------------------------------------------------------
    // D at both ends of the permutation is 2^-5, not 1: the internal form of x / 32 is (x / 32) 2^261 = x 2^256, the ABI
    // residue itself, so states, absorbed and squeezed elements enter and leave the kernels without a multiplication
    // (pmx_field.hpp: fe_from_abi_scaled / fe_to_abi_scaled).  Between two permutations of a sponge the state stays in
    // these coordinates.
    const U256 thirty_two = {{32, 0, 0, 0}};
    const U256 io_scale = f.inverse(f.to_mont(thirty_two));
    std::vector<U256> d(t, io_scale), dn(t), e(t), inv_e(t);
    for (uint32_t r = 0; r < rounds; ++r) {
        const bool full = r < half_full || r > last_partial;
        for (size_t i = 0; i < t; ++i) ark_opt[(size_t)r * t + i] = f.mul(ark_opt[(size_t)r * t + i], d[i]);   // D_r c_r
        for (size_t i = 0; i < t; ++i) {                // scaling of what enters the linear layer
            e[i] = (full || i == 0) ? host_pow(f, d[i], alpha) : d[i];
            if (u256_is_zero(e[i])) return false;
            inv_e[i] = f.inverse(e[i]);
        }
        if (r == entrance && entrance_scale) *entrance_scale = e;       // what multiplies the S-box outputs of the entrance round
        if (r == last_partial + 1 && exit_scale) *exit_scale = d;       // what multiplies the state that enters the first full round after the partial section
        if (r == entrance || (!full && r < last_partial)) {            // sparse layer
            U256 *sp = &sparse[(size_t)(r - entrance) * per];
            if (u256_is_zero(sp[0])) return false;
            dn[0] = f.mul(e[0], f.inverse(sp[0]));
            for (size_t c = 1; c < t; ++c) sp[c] = f.mul(f.mul(sp[c], dn[0]), inv_e[c]);
            sp[0] = f.r;                                               // the coefficient of the S-box output: one
            for (size_t i = 1; i < t; ++i) {
                dn[i] = e[i];                                          // the lanes keep their unit coefficients
                sp[t + i - 1] = f.mul(f.mul(sp[t + i - 1], dn[i]), inv_e[0]);
            }
        } else {
            const bool last = r + 1 == rounds;
            U256 *L = full ? &fullmat[full_matrix_ordinal(r, half_full, rp) * t * t] : bdense.data();
            for (size_t i = 0; i < t; ++i) {
                if (last) {
                    dn[i] = io_scale;
                } else {
                    if (u256_is_zero(L[i * t])) return false;
                    dn[i] = f.mul(e[0], f.inverse(L[i * t]));          // column 0 becomes one
                }
                for (size_t c = 0; c < t; ++c) L[i * t + c] = f.mul(f.mul(L[i * t + c], dn[i]), inv_e[c]);
                if (!last) L[i * t] = f.r;
            }
        }
        d = dn;
    }
    return true;
}

// ---- the partial section as windows (pmx_mfma.hpp, pmx_permute.hpp: MFMA_WINDOW) ----------------------------------------------
// Exact algebra on the reference's rounds (s <- M (s + c_r with lane 0 raised to alpha)), per window of kw <= K partial rounds:
// with sigma = lanes 1.. of the true state at the window's start and z_1 .. z_kw its S-box outputs, every later S-box input and the
// state at the window's end are affine in (sigma, z).  The window carries  x^_1 = x_1  and  u^ = Psi sigma + psi  where the first
// kw - 1 rows of Psi are the sigma-parts (and constants) of x_2 .. x_kw, scaled by delta_{k+1} = delta_k^alpha / M_00 so that
//     x^_{k+1} = delta_{k+1} x_{k+1} = z^_k + u^_k + sum_{i<k} h_{k,i} z^_i,      z^_k = x^_k^alpha = delta_k^alpha z_k,
// and the other rows are unit rows that keep Psi invertible.  Layer w maps (u^, z^) of window w to (x^_1, u^) of window w + 1 (the
// last one to what the first full round after the section expects: D (s + c) minus the constant the kernel adds there), the entry
// layer maps the scaled S-box outputs of the entrance round to window 0.
struct WindowPlan {
    size_t n_win = 0;
    std::vector<U256> entry_rows, entry_aff;                 // t x t, t
    std::vector<std::vector<U256>> rows, aff;                // per window: t x (t - 1 + K), t
    std::vector<U256> hist;                                  // per window mfma_window_hist(K) constants
    std::vector<uint32_t> hist_small;                        // per window: h_{2,1} as a small integer (1 .. 4), or 0 (derive_window_layers: the window's free scale)
};

inline size_t mat_rank(const HostField &f, HostMat a) {
    size_t rank = 0;
    const size_t n = a.size(), m = n ? a[0].size() : 0;
    for (size_t col = 0; col < m && rank < n; ++col) {
        size_t piv = rank;
        while (piv < n && u256_is_zero(a[piv][col])) ++piv;
        if (piv == n) continue;
        std::swap(a[piv], a[rank]);
        const U256 d = f.inverse(a[rank][col]);
        for (size_t i = rank + 1; i < n; ++i) {
            if (u256_is_zero(a[i][col])) continue;
            const U256 k = f.mul(a[i][col], d);
            for (size_t j = col; j < m; ++j) a[i][j] = f.sub(a[i][j], f.mul(k, a[rank][j]));
        }
        ++rank;
    }
    return rank;
}

// ---- a window's free scale ------------------------------------------------------------------------------------------
// x^_1 of a window may be carried scaled by any lambda (the layer in front multiplies row 0 by it): the single history constant of a
// window of three S-boxes becomes h lambda^(alpha^2 - alpha), and where s / h has such a root for a small integer s the product h z^_1
// - 81 + 18 multiplies from a shifted table - is s lazy additions (pmx_permute.hpp).  alpha = 5: 20 = 5 * 4 and gcd(5, p - 1) = 1 for
// a config whose S-box is a permutation - a fifth root always exists, a fourth root for one value in four; alpha = 17: one in sixteen.
inline U256 host_pow_u256(const HostField &f, const U256 &x, const U256 &e) {
    U256 acc = f.r;
    for (int bit = 255; bit >= 0; --bit) {
        acc = f.mul(acc, acc);
        if ((e.l[bit / 64] >> (bit % 64)) & 1) acc = f.mul(acc, x);
    }
    return acc;
}
inline uint64_t u256_divmod_small(U256 &q, const U256 &a, uint64_t d) {   // q = a / d, returns a % d
    u128 rem = 0;
    for (int i = 3; i >= 0; --i) {
        const u128 cur = (rem << 64) | a.l[i];
        q.l[i] = (uint64_t)(cur / d);
        rem = cur % d;
    }
    return (uint64_t)rem;
}
inline U256 u256_shr(const U256 &a, unsigned k) {   // k < 64
    U256 r;
    for (int i = 0; i < 4; ++i) r.l[i] = k ? ((a.l[i] >> k) | (i + 1 < 4 ? a.l[i + 1] << (64 - k) : 0)) : a.l[i];
    return r;
}
inline bool u256_eq(const U256 &a, const U256 &b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }
// square root in the Montgomery domain (Tonelli-Shanks); false: a is not a square
inline bool host_sqrt(const HostField &f, const U256 &a, U256 &root) {
    if (u256_is_zero(a)) { root = a; return true; }
    U256 pm1 = f.p;
    pm1.l[0] -= 1;                                   // p odd: no borrow
    unsigned S = 0;
    U256 q = pm1;
    while (!(q.l[0] & 1)) {
        q = u256_shr(q, 1);
        ++S;
    }
    const U256 half = u256_shr(pm1, 1), minus_one = f.neg(f.r);
    if (!u256_eq(host_pow_u256(f, a, half), f.r)) return false;
    U256 z = f.r;                                    // a non-residue: 2, 3, 4 ... in the Montgomery domain
    for (int tries = 0; tries < 200; ++tries) {
        z = f.add(z, f.r);
        if (u256_eq(host_pow_u256(f, z, half), minus_one)) break;
        if (tries == 199) return false;
    }
    U256 qp1 = q;                                    // (q + 1) / 2
    qp1.l[0] += 1;                                   // q odd: q + 1 even; no carry unless q = 2^64k - 1 (not for a 225 .. 255-bit p - 1 with S >= 1)
    qp1 = u256_shr(qp1, 1);
    U256 c = host_pow_u256(f, z, q), r = host_pow_u256(f, a, qp1), tt = host_pow_u256(f, a, q);
    unsigned m = S;
    while (!u256_eq(tt, f.r)) {
        unsigned i = 0;
        U256 t2 = tt;
        while (!u256_eq(t2, f.r)) {
            t2 = f.mul(t2, t2);
            if (++i == m) return false;
        }
        U256 b = c;
        for (unsigned k = 0; k + i + 1 < m; ++k) b = f.mul(b, b);
        r = f.mul(r, b);
        c = f.mul(b, b);
        tt = f.mul(tt, c);
        m = i;
    }
    root = r;
    return true;
}
// x^(1/e) for a 64-bit e prime to p - 1 (the map is a bijection): x^d with d = e^-1 mod (p - 1) = (1 + k (p - 1)) / e
inline bool host_root_coprime(const HostField &f, const U256 &x, uint64_t e, U256 &root) {
    if (e == 1) { root = x; return true; }
    U256 pm1 = f.p, Q;
    pm1.l[0] -= 1;
    const uint64_t R = u256_divmod_small(Q, pm1, e);
    // k = -R^-1 mod e (extended Euclid on 64-bit values; no inverse: e shares a factor with p - 1)
    __int128 r0 = (__int128)e, r1 = (__int128)R, t0 = 0, t1 = 1;
    while (r1 != 0) {
        const __int128 q = r0 / r1, r2 = r0 - q * r1, t2 = t0 - q * t1;
        r0 = r1; r1 = r2; t0 = t1; t1 = t2;
    }
    if (r0 != 1) return false;
    __int128 inv = t0 % (__int128)e;
    if (inv < 0) inv += (__int128)e;
    const uint64_t k = (uint64_t)(((__int128)e - inv) % (__int128)e);      // k R = -1 (mod e)
    U256 d = {{0, 0, 0, 0}};                                               // d = k Q + (1 + k R) / e   (k Q < p - 1: no overflow)
    u128 carry = 0;
    for (int i = 0; i < 4; ++i) {
        const u128 v = (u128)Q.l[i] * k + carry;
        d.l[i] = (uint64_t)v;
        carry = v >> 64;
    }
    const u128 rest = ((u128)k * R + 1) / e;
    const U256 add = {{(uint64_t)rest, (uint64_t)(rest >> 64), 0, 0}};
    u256_add(d, d, add);
    root = host_pow_u256(f, x, d);
    return true;
}
// is x a 2^a-th power?  (the 2-part of p - 1 is 2^S: x^((p - 1) / 2^min(a, S)) = 1)
inline bool host_is_2power_residue(const HostField &f, const U256 &x, unsigned a) {
    if (a == 0 || u256_is_zero(x)) return true;
    U256 e = f.p;
    e.l[0] -= 1;
    for (unsigned i = 0; i < a && !(e.l[0] & 1); ++i) e = u256_shr(e, 1);
    return u256_eq(host_pow_u256(f, x, e), f.r);
}
// lambda with lambda^(alpha (alpha - 1)) = y, or false: the alpha-th root and the odd part of alpha - 1 by inverting the exponent, the
// power of two in alpha - 1 by square roots (alpha = 5: 20 = 5 * 4, alpha = 17: 272 = 17 * 16, alpha = 257: 257 * 256)
inline bool host_root_window_scale(const HostField &f, const U256 &y, uint64_t alpha, U256 &lambda) {
    if (alpha < 2 || alpha > ((uint64_t)1 << 32)) return false;
    uint64_t m = alpha - 1;
    unsigned a = 0;
    while (!(m & 1)) {
        m >>= 1;
        ++a;
    }
    U256 u;
    if (!host_root_coprime(f, y, alpha, u) || !host_root_coprime(f, u, m, u)) return false;
    if (!host_is_2power_residue(f, u, a)) return false;
    for (unsigned left = a; left > 0; --left) {            // of the two square roots the one that is still a 2^(left-1)-th power
        U256 r;
        if (!host_sqrt(f, u, r)) return false;
        if (!host_is_2power_residue(f, r, left - 1)) r = f.neg(r);
        if (!host_is_2power_residue(f, r, left - 1)) return false;
        u = r;
    }
    lambda = u;
    U256 chk = host_pow(f, host_pow(f, lambda, alpha), alpha - 1);
    return u256_eq(chk, y);
}

inline bool derive_window_layers(const HostField &f, uint32_t t, uint32_t half, uint32_t rp, uint64_t alpha, uint32_t K,
                                 const std::vector<U256> &ark, const HostMat &M, const std::vector<U256> &entrance_scale,
                                 const std::vector<U256> &exit_scale, const U256 *arkopt_exit, WindowPlan &plan) {
    if (K < 1 || K > t || rp == 0 || half == 0 || entrance_scale.size() != t || exit_scale.size() != t) return false;
    const U256 zero = {{0, 0, 0, 0}};
    const size_t n = t - 1, G = n + K;                       // generators: sigma (n), z_1..z_K; index G = the constant
    const size_t n_win = (rp + K - 1) / K;
    typedef std::vector<U256> Form;                          // G + 1 coefficients
    auto unit = [&](size_t g) { Form v(G + 1, zero); v[g] = f.r; return v; };
    plan = WindowPlan();
    plan.n_win = n_win;
    plan.rows.resize(n_win);
    plan.aff.resize(n_win);
    plan.hist.assign(n_win * (size_t)mfma_window_hist((int)K), zero);
    plan.hist_small.assign(n_win, 0u);
    bool memo_valid = false;                                 // the last history constant searched for a small multiple, and what was found
    U256 memo_h1 = zero, memo_lambda = zero;
    uint32_t memo_small = 0;
    std::vector<U256> lane0(n_win, f.r);                     // delta_1 of each window: x^_1 = lane0 x_1 (1 unless a better one exists, below)
    // what the layer BEFORE window w has to produce, from the true state at the window's start: x^_1 = s_0 + c, u^ = Psi s_1.. + psi
    std::vector<HostMat> Psi(n_win), PsiInv(n_win);
    std::vector<std::vector<U256>> psi(n_win);
    std::vector<std::vector<Form>> Send(n_win);              // true state after the window, over (sigma, z, 1)
    std::vector<std::vector<U256>> zscale(n_win);            // delta_j^-alpha: z_j = zscale_j z^_j
    uint32_t r1 = half;
    for (size_t w = 0; w < n_win; ++w) {
        const uint32_t kw = w == 0 ? rp - (uint32_t)(n_win - 1) * K : K;
        std::vector<Form> S(t, Form(G + 1, zero)), X(kw + 1);
        for (size_t i = 1; i < t; ++i) S[i] = unit(i - 1);
        for (uint32_t j = 1; j <= kw; ++j) {
            const U256 *c = &ark[(size_t)(r1 + j - 1) * t];
            std::vector<Form> y = S;
            for (size_t i = 0; i < t; ++i) y[i][G] = f.add(y[i][G], c[i]);
            if (j >= 2) X[j] = y[0];
            y[0] = unit(n + j - 1);
            for (size_t i = 0; i < t; ++i) {
                Form acc(G + 1, zero);
                for (size_t l = 0; l < t; ++l)
                    for (size_t g = 0; g <= G; ++g)
                        if (!u256_is_zero(y[l][g])) acc[g] = f.add(acc[g], f.mul(M[i][l], y[l][g]));
                S[i] = acc;
            }
        }
        Send[w] = S;
        // the scales of the S-box inputs, x^_j = delta_j x_j, for a given delta_1 (the window's free scale: lane0[w])
        auto scales = [&](const U256 &delta1) -> bool {
            std::vector<U256> delta(kw + 1, f.r), dpow(kw + 1, f.r);   // delta_j, delta_j^alpha
            delta[1] = delta1;
            dpow[1] = host_pow(f, delta1, alpha);
            if (u256_is_zero(dpow[1])) return false;
            zscale[w].assign(K, zero);
            zscale[w][0] = f.inverse(dpow[1]);
            Psi[w].assign(n, std::vector<U256>(n, zero));
            psi[w].assign(n, zero);
            for (uint32_t j = 2; j <= kw; ++j) {
                const U256 a = X[j][n + j - 2];                  // coefficient of z_{j-1} in x_j
                if (u256_is_zero(a)) return false;
                delta[j] = f.mul(dpow[j - 1], f.inverse(a));
                dpow[j] = host_pow(f, delta[j], alpha);
                if (u256_is_zero(dpow[j])) return false;
                zscale[w][j - 1] = f.inverse(dpow[j]);
                const size_t k = j - 1;                          // x_{k+1}: coordinate u^_k, history h_{k,i}
                for (size_t g = 0; g < n; ++g) Psi[w][k - 1][g] = f.mul(delta[j], X[j][g]);
                psi[w][k - 1] = f.mul(delta[j], X[j][G]);
                for (size_t i = 1; i < k; ++i)
                    plan.hist[w * (size_t)mfma_window_hist((int)K) + (size_t)mfma_window_hist((int)k) + (i - 1)] =
                        f.mul(f.mul(delta[j], X[j][n + i - 1]), zscale[w][i - 1]);
            }
            return true;
        };
        lane0[w] = f.r;
        if (!scales(f.r)) return false;
        // a window of three S-boxes whose history term is a table product (t = 3): delta_1 with h_{2,1} = 1 .. 4 where one exists
        if (alpha >= 2 && K == 3 && kw == 3 && mfma_hist_tab((int)t)) {
            const U256 h1 = plan.hist[w * (size_t)mfma_window_hist((int)K)];
            if (!u256_is_zero(h1)) {
                if (!memo_valid || !u256_eq(memo_h1, h1)) {      // (the constant depends on M and alpha only: one search per config)
                    memo_valid = true;
                    memo_h1 = h1;
                    memo_small = 0;
                    const U256 h1_inv = f.inverse(h1);
                    U256 small = zero;
                    for (uint32_t sm = 1; sm <= 4 && !memo_small; ++sm) {
                        small = f.add(small, f.r);               // sm in the Montgomery domain
                        if (host_root_window_scale(f, f.mul(small, h1_inv), alpha, memo_lambda)) memo_small = sm;
                    }
                }
                if (memo_small) {
                    U256 small = zero;
                    for (uint32_t i = 0; i < memo_small; ++i) small = f.add(small, f.r);
                    if (scales(memo_lambda) && u256_eq(plan.hist[w * (size_t)mfma_window_hist((int)K)], small)) {
                        lane0[w] = memo_lambda;
                        plan.hist_small[w] = memo_small;
                    } else if (!scales(f.r)) {
                        return false;
                    }
                }
            }
        }
        // the other coordinates: lanes of sigma themselves, chosen so that Psi stays invertible
        size_t have = kw - 1, cand = 0;
        while (have < n) {
            if (cand >= n) return false;
            HostMat trial(Psi[w].begin(), Psi[w].begin() + (long)have);
            trial.push_back(std::vector<U256>(n, zero));
            trial.back()[cand] = f.r;
            if (mat_rank(f, trial) == have + 1) {
                Psi[w][have] = trial.back();
                ++have;
            }
            ++cand;
        }
        if (!mat_inverse(f, Psi[w], PsiInv[w])) return false;
        r1 += kw;
    }
    // rows of a producing layer: P[i] = true lane i at the start of window w over the producer's inputs (n_in + 1 coefficients)
    auto emit = [&](const std::vector<Form> &P, size_t n_in, size_t w, uint32_t first_round, std::vector<U256> &rows, std::vector<U256> &aff) {
        rows.assign((size_t)t * n_in, zero);
        aff.assign(t, zero);
        for (size_t g = 0; g < n_in; ++g) rows[g] = f.mul(lane0[w], P[0][g]);
        aff[0] = f.mul(lane0[w], f.add(P[0][n_in], ark[(size_t)first_round * t]));
        for (size_t k = 0; k < n; ++k) {
            for (size_t g = 0; g <= n_in; ++g) {
                U256 acc = zero;
                for (size_t l = 0; l < n; ++l) acc = f.add(acc, f.mul(Psi[w][k][l], P[1 + l][g]));
                if (g < n_in) rows[(1 + k) * n_in + g] = acc;
                else aff[1 + k] = f.add(acc, psi[w][k]);
            }
        }
    };
    {   // entry layer: inputs = scaled S-box outputs of the entrance round, true z_j = in_j / e_j, state = M z
        std::vector<Form> P(t, Form(t + 1, zero));
        for (size_t j = 0; j < t; ++j) {
            if (u256_is_zero(entrance_scale[j])) return false;
            const U256 inv = f.inverse(entrance_scale[j]);
            for (size_t i = 0; i < t; ++i) P[i][j] = f.mul(M[i][j], inv);
        }
        emit(P, t, 0, half, plan.entry_rows, plan.entry_aff);
    }
    r1 = half;
    for (size_t w = 0; w < n_win; ++w) {
        const uint32_t kw = w == 0 ? rp - (uint32_t)(n_win - 1) * K : K;
        r1 += kw;                                            // first round after this window
        // the window's end state over its own inputs (u^, z^): sigma = PsiInv (u^ - psi), z_j = zscale_j z^_j
        std::vector<U256> shift = mat_vec(f, PsiInv[w], psi[w]);   // PsiInv psi
        std::vector<Form> P(t, Form(G + 1, zero));
        for (size_t i = 0; i < t; ++i) {
            const Form &S = Send[w][i];
            for (size_t m = 0; m < n; ++m) {
                U256 acc = zero;
                for (size_t l = 0; l < n; ++l) acc = f.add(acc, f.mul(S[l], PsiInv[w][l][m]));
                P[i][m] = acc;
            }
            for (size_t j = 0; j < K; ++j) P[i][n + j] = f.mul(S[n + j], zscale[w][j]);   // (zero columns for the rounds a short window lacks)
            U256 cst = S[G];
            for (size_t l = 0; l < n; ++l) cst = f.sub(cst, f.mul(S[l], shift[l]));
            P[i][G] = cst;
        }
        if (w + 1 < n_win) {
            emit(P, G, w + 1, r1, plan.rows[w], plan.aff[w]);
        } else {   // into the full rounds: D (s + c) - (the constant the kernel adds at that round)
            plan.rows[w].assign((size_t)t * G, zero);
            plan.aff[w].assign(t, zero);
            for (size_t i = 0; i < t; ++i) {
                for (size_t g = 0; g < G; ++g) plan.rows[w][i * G + g] = f.mul(exit_scale[i], P[i][g]);
                plan.aff[w][i] = f.sub(f.mul(exit_scale[i], f.add(P[i][G], ark[(size_t)r1 * t + i])), arkopt_exit[i]);
            }
        }
    }
    return true;
}

// Returns PMX_OK or an error code with a message in `err`.
inline int prepare(const pmx_config *cfg, Prepared &out, std::string &err) {
    const uint64_t t64 = (uint64_t)cfg->rate + cfg->capacity;
    if (cfg->rate == 0) { err = "rate must be >= 1"; return PMX_ERR_CONFIG; }
    if (t64 > PMX_MAX_WIDTH) { err = "width exceeds PMX_MAX_WIDTH"; return PMX_ERR_UNSUPPORTED; }
    const uint64_t rounds = (uint64_t)cfg->full_rounds + cfg->partial_rounds;
    if (rounds == 0 || rounds > 4096) { err = "round count out of range"; return PMX_ERR_CONFIG; }
    HostField &hf = out.hf;
    if (!hf.init(cfg->modulus)) { err = "modulus must be odd and > 2"; return PMX_ERR_CONFIG; }
    // the unsaturated 9 x 29-bit arithmetic needs 6 spare bits below 2^261 (pmx_field.hpp)
    if (hf.bits() > 255) { err = "modulus must be < 2^255 (BLS12-381 Fr and BN254 Fr are 255 and 254 bits)"; return PMX_ERR_UNSUPPORTED; }
    if (hf.bits() < 225) { err = "modulus must be at least 225 bits"; return PMX_ERR_UNSUPPORTED; }
    const uint32_t t = (uint32_t)t64;
    const size_t n_ark = (size_t)rounds * t, n_mds = (size_t)t * t;
    out.t = t;
    out.consts.assign((n_ark + n_mds) * kFeStride, 0u);
    out.mds_offset = n_ark * kFeStride;
    for (size_t k = 0; k < n_ark + n_mds; ++k) {
        const uint64_t *src = k < n_ark ? cfg->ark + 4 * k : cfg->mds + 4 * (k - n_ark);
        U256 v;
        std::memcpy(v.l, src, sizeof v.l);
        if (u256_geq(v, hf.p)) {  // every constant must be a reduced residue (ark-ff's invariant)
            err = std::string(k < n_ark ? "ark" : "mds") + " constant " + std::to_string(k < n_ark ? k : k - n_ark) + " is not reduced";
            return PMX_ERR_CONFIG;
        }
        to_limbs29(times_pow2(hf, v, 5), &out.consts[k * kFeStride]);   // x*2^256 -> x*2^261
    }
    // optimised schedule
    std::vector<U256> src_full, src_sparse, src_bdense, entrance_scale, exit_scale, arkopt_exit, ark_true;
    U256 arkopt_first = {{0, 0, 0, 0}};   // ark'[0][0]
    HostMat M_true;
    const uint32_t half = cfg->full_rounds / 2, rp = cfg->partial_rounds;
    const size_t n_full = cfg->full_rounds ? (size_t)cfg->full_rounds - 1 : 0;   // matrices in `full`
    {
        std::vector<U256> ark(n_ark), ark_opt;
        for (size_t k = 0; k < n_ark; ++k) std::memcpy(ark[k].l, cfg->ark + 4 * k, 32);
        HostMat M(t, std::vector<U256>(t));
        for (size_t i = 0; i < t; ++i)
            for (size_t j = 0; j < t; ++j) std::memcpy(M[i][j].l, cfg->mds + 4 * (i * t + j), 32);
        long double pv = 0;
        for (int i = 3; i >= 0; --i) pv = pv * 18446744073709551616.0L + (long double)hf.p.l[i];
        long double two_261 = 1;
        for (int i = 0; i < 261; ++i) two_261 *= 2;
        out.has_opt = opt_schedule_lane_headroom(two_261 / pv, rp, cfg->alpha) &&
                      derive_opt_tables(hf, t, half, rp, (uint32_t)rounds, cfg->alpha, ark, M, ark_opt, src_full, src_sparse, src_bdense,
                                        &entrance_scale, &exit_scale);
        if (out.has_opt) {
            arkopt_exit.assign(ark_opt.begin() + (long)((size_t)(half + rp) * t), ark_opt.begin() + (long)((size_t)(half + rp) * t + t));
            arkopt_first = ark_opt[0];
            ark_true = ark;
            M_true = M;
        }
        out.opt_offset = out.consts.size();
        out.opt_full_offset = out.opt_sparse_offset = out.opt_bdense_offset = out.opt_offset;
        if (out.has_opt) {
            // ark' for every engine of the optimised schedule; the matrices in element form only where the quad engine's table is
            // built from them (t = 3) - the window engines read the same matrices as int8 tables (below)
            const bool elems = t == 3;
            out.opt_full_offset = out.opt_offset + n_ark * kFeStride;
            out.opt_sparse_offset = out.opt_full_offset + (elems ? src_full.size() : 0) * kFeStride;
            out.opt_bdense_offset = out.opt_sparse_offset + (elems ? src_sparse.size() : 0) * kFeStride;
            out.consts.resize(out.opt_bdense_offset + (elems ? src_bdense.size() : 0) * kFeStride, 0u);
            size_t k = out.opt_offset / kFeStride;
            for (const U256 &v : ark_opt) to_limbs29(times_pow2(hf, v, 5), &out.consts[(k++) * kFeStride]);
            if (elems)
                for (const auto *vec : {&src_full, &src_sparse, &src_bdense})
                    for (const U256 &v : *vec) to_limbs29(times_pow2(hf, v, 5), &out.consts[(k++) * kFeStride]);
        }
    }
    // cooperative t = 3 table: per (round, lane): ark' element, then the lane's matrix row of that round
    out.coop_offset = out.consts.size();
    if (out.has_opt && t == 3) {
        out.consts.resize(out.coop_offset + (size_t)rounds * 3 * kCoopElems * kFeStride, 0u);
        const uint32_t *ark_o = &out.consts[out.opt_offset];
        const uint32_t *full_i = &out.consts[out.opt_full_offset];
        const uint32_t *sparse_i = &out.consts[out.opt_sparse_offset];
        const uint32_t *bdense_i = &out.consts[out.opt_bdense_offset];
        uint32_t one29[kN];
        to_limbs29(times_pow2(hf, hf.r, 5), one29);
        auto put = [&](size_t r, size_t q, size_t slot, const uint32_t *src) {
            std::memcpy(&out.consts[out.coop_offset + ((r * 3 + q) * kCoopElems + slot) * kFeStride], src, kN * 4);
        };
        const bool folded = cfg->alpha == 5 || cfg->alpha == 17;   // folded sparse rounds (pmx_permute.hpp: coop_fold_*)
        for (size_t r = 0; r < rounds; ++r) {
            const bool partial = r >= half && r < (size_t)half + rp;
            for (size_t q = 0; q < 3; ++q) {
                put(r, q, 0, ark_o + (r * 3 + q) * kFeStride);
                if (r + 1 == half || (partial && r + 1 < (size_t)half + rp)) {       // sparse layer: row0[3] = (ONE, v_1, v_2), w[2]
                    const uint32_t *sp = sparse_i + (r + 1 - half) * 5 * kFeStride;
                    if (partial && folded) {
                        if (q == 0) {
                            put(r, q, 1, sp);                             // stage A: x * ONE
                        } else {
                            put(r, q, 1, sp + (3 + q - 1) * kFeStride);   // stage A: x * w_q
                            put(r, q, 2, sp + q * kFeStride);             // stage B: s_q * v_q
                        }
                    } else if (q == 0) {                                  // uniform rounds (the entrance round is one: full S-box layer)
                        for (size_t j = 0; j < 3; ++j) put(r, q, 1 + j, sp + j * kFeStride);
                    } else {
                        put(r, q, 1, sp + (3 + q - 1) * kFeStride);       // w_q * z_0
                        put(r, q, 1 + q, one29);                          // + ONE * z_q   (other slot stays 0)
                    }
                } else {
                    const uint32_t *mat = partial ? bdense_i : full_i + full_matrix_ordinal((uint32_t)r, half, rp) * 9 * kFeStride;
                    for (size_t j = 0; j < 3; ++j) put(r, q, 1 + j, mat + (q * 3 + j) * kFeStride);
                }
            }
        }
    }
    // the dense layers as int8 GEMM operands
    while (out.consts.size() % 4) out.consts.push_back(0u);   // 16-byte operands
    out.mfma_offset = out.consts.size();
    // (a layer's inputs must be below 2^256: S-box outputs ARE products for every alpha - pmx_field.hpp: fe_sbox; p < 2^255 is a limit of the build)
    out.mfma_dense = out.has_opt && t >= PMX_MFMA_MIN_T && t <= PMX_MFMA_MAX_T && n_full >= 1;
    if (out.mfma_dense) {
        const size_t lw = (size_t)mfma_layer_words((int)t);
        out.consts.resize(out.mfma_offset + (n_full + 1) * lw, 0u);
        for (size_t o = 0; o < n_full; ++o) put_mfma_layer(hf, &src_full[o * t * t], t, &out.consts[out.mfma_offset + o * lw]);
        put_mfma_layer(hf, src_bdense.data(), t, &out.consts[out.mfma_offset + n_full * lw]);
    }
    // ... and the partial section as windows closed by one such layer each.  A width that takes windows has no sparse layers in
    // its matrix-core kernels: if the windows cannot be derived (a zero where the algebra divides) the config runs on the VALU rows.
    out.win_offset = out.consts.size();
    out.mfma_window = 0;
    if (out.mfma_dense && mfma_window_for((int)t) > 0) {
        const uint32_t K = (uint32_t)mfma_window_for((int)t);
        WindowPlan plan;
        if (K <= t && derive_window_layers(hf, t, half, rp, cfg->alpha, K, ark_true, M_true, entrance_scale, exit_scale, arkopt_exit.data(), plan)) {
            const size_t lw_in = (size_t)mfma_layer_words_io((int)(t - 1 + K), (int)t), nh = (size_t)mfma_window_hist((int)K);
            out.consts.resize(out.win_offset + mfma_window_words((int)t, (int)K, plan.n_win) + kFeStride, 0u);
            uint32_t *dst = &out.consts[out.win_offset];
            // (rows that only feed matrix-core inputs - every carried lane but the first, where the history terms are rows too - stay in
            // operand form between the layers: pmx_mfma.hpp, mfma_fe_rows; the last window's layer feeds S-boxes on every lane)
            put_mfma_layer_io(hf, plan.entry_rows.data(), t, t, plan.entry_aff.data(), dst);
            dst += mfma_layer_words((int)t);
            for (size_t w = 0; w < plan.n_win; ++w) {
                put_mfma_layer_io(hf, plan.rows[w].data(), t - 1 + K, t, plan.aff[w].data(), dst);
                dst += lw_in;
                if (mfma_hist_tab((int)t)) {
                    for (uint32_t k = 2; k < K; ++k) {
                        if (k == 2 && plan.hist_small[w]) {   // h_{2,1} = 1 .. 4: no table - a marker no limb can be, and the integer (pmx_permute.hpp)
                            dst[mfma_hist_tab_offset(2)] = kMfmaHistSmallMarker;
                            dst[mfma_hist_tab_offset(2) + 1] = plan.hist_small[w];
                            continue;
                        }
                        put_shifted_row(hf, &plan.hist[w * nh + (size_t)mfma_window_hist((int)k)], k - 1, dst + mfma_hist_tab_offset((int)k));
                    }
                } else {   // the history terms as matrix-core rows: the row of x_{k+1} over (z_1 .. z_{k-1}, u_k), coefficients (h_{k,.}, ONE)
                    for (uint32_t k = 2; k < K; ++k) {
                        std::vector<U256> row(&plan.hist[w * nh + (size_t)mfma_window_hist((int)k)], &plan.hist[w * nh + (size_t)mfma_window_hist((int)k)] + (k - 1));
                        row.push_back(hf.r);
                        put_mfma_layer_io(hf, row.data(), k, 1, nullptr, dst + mfma_hist_rows_offset((int)k));
                    }
                }
                dst += (size_t)mfma_window_hist_words((int)t, (int)K);
            }
            // behind the windows: S-box(ark'[0][0]) - what lane 0 of round 0 is when that lane comes in as zero (the capacity lane of a fresh
            // sponge: every 2-to-1 compression, the first permutation of every hash row), so those kernels skip that S-box (pmx_permute.hpp)
            to_limbs29(times_pow2(hf, host_pow(hf, arkopt_first, cfg->alpha), 5), dst);
            out.mfma_window = K;
        } else {
            out.mfma_dense = false;
            out.consts.resize(out.mfma_offset);
            out.win_offset = out.consts.size();
        }
    }
    FieldRt &f = out.f;
    f.unit = 1;
    to_limbs29(hf.p, f.p);
    f.pinv = (uint32_t)hf.inv & kMask;
    // constants of the ABI conversions (FieldRt::io)
    out.io_offset = out.consts.size();
    out.consts.resize(out.io_offset + kIoWords, 0u);
    std::memcpy(&out.consts[out.io_offset + kIoP32], hf.p.l, 32);
    to_limbs29(times_pow2(hf, hf.r, 10), &out.consts[out.io_offset + kIoToInt]);   // 2^266 mod p
    to_limbs29(hf.r, &out.consts[out.io_offset + kIoToAbi]);                       // 2^256 mod p
    f.io = out.consts.data() + out.io_offset;   // host view; valid while `out` is neither copied nor resized
    to_limbs29(times_pow2(hf, hf.r, 5), out.one.l);     // 2^261 mod p
    out.c.rate = cfg->rate;
    out.c.capacity = cfg->capacity;
    out.c.half_full = half;   // odd RF: RF/2 full rounds before the partial section, RF - RF/2 after (mod.rs:96-116)
    out.c.partial_rounds = cfg->partial_rounds;
    out.c.total_rounds = (uint32_t)rounds;
    out.c.alpha = cfg->alpha;
    return PMX_OK;
}

}  // namespace pmx
