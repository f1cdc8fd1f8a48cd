/*
 * CPU restatement ("port") of the arkworks-rs/sponge Poseidon hot path  --  TEST INFRASTRUCTURE ONLY.
 *
 * This is the bulk comparator for the HIP product path and bench.py's cpu_baseline ("kind":"port").
 * Nothing under sponge_amd/ links or loads it.  It keeps the reference's operation sequence:
 *   - dense MDS, row-major, accumulator starting at zero       src/poseidon/mod.rs:82-93
 *   - S-box through a generic MSB-first square-and-multiply pow src/poseidon/mod.rs:63-74 (ark-ff Field::pow)
 *   - ARK -> S-box -> MDS per round, RF/2 | RP | RF/2 schedule  src/poseidon/mod.rs:95-118
 *   - absorb_internal / squeeze_internal / mode machine         src/poseidon/mod.rs:121-182, 232-254, 321-341
 * Field arithmetic (third-party ark-ff, not in the reference tree): fully reduced Montgomery residues
 * x*2^256 mod p in 4 little-endian u64 limbs; here a textbook CIOS Montgomery product.
 *
 * Parity pin: validated limb-for-limb against oracle/poseidon_oracle.py (which reproduces the three
 * reference KATs) in tests/test_oracle_c.py, including the KAT of src/poseidon/mod.rs:376-399.
 *
 * Build: make -C oracle   ->  oracle/libposeidon_oracle.so
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

#define PREF_MAX_T 16

typedef struct {
    uint32_t t, rate, capacity, full_rounds, partial_rounds;
    uint32_t pad_;
    uint64_t alpha;
    uint64_t modulus[4];
    uint64_t inv;            /* -p^-1 mod 2^64 */
    const uint64_t *ark;     /* [RF+RP][t][4] Montgomery */
    const uint64_t *mds;     /* [t][t][4]    Montgomery, mds[i][j] */
} pref_config;

typedef struct { uint64_t l[4]; } fe;

static inline __attribute__((always_inline)) int geq(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return 1;
}

static inline __attribute__((always_inline)) void sub_n(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}

/* r = a + b mod p   (Fp::add_assign) */
static inline __attribute__((always_inline)) void fe_add(fe *r, const fe *a, const fe *b, const uint64_t p[4]) {
    uint64_t s[4], carry = 0;
    for (int i = 0; i < 4; ++i) {
        u128 v = (u128)a->l[i] + b->l[i] + carry;
        s[i] = (uint64_t)v;
        carry = (uint64_t)(v >> 64);
    }
    if (carry || geq(s, p)) sub_n(s, s, p);
    memcpy(r->l, s, sizeof s);
}

/* r = a * b * 2^-256 mod p   (Fp::mul on Montgomery residues), CIOS */
static inline __attribute__((always_inline)) void fe_mul(fe *r, const fe *a, const fe *b, const uint64_t p[4], uint64_t inv) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        uint64_t c = 0;
#pragma GCC unroll 4
        for (int j = 0; j < 4; ++j) {
            u128 v = (u128)a->l[j] * b->l[i] + t[j] + c;
            t[j] = (uint64_t)v;
            c = (uint64_t)(v >> 64);
        }
        u128 v = (u128)t[4] + c;
        t[4] = (uint64_t)v;
        t[5] = (uint64_t)(v >> 64);
        uint64_t m = t[0] * inv;
        v = (u128)m * p[0] + t[0];
        c = (uint64_t)(v >> 64);
#pragma GCC unroll 4
        for (int j = 1; j < 4; ++j) {
            v = (u128)m * p[j] + t[j] + c;
            t[j - 1] = (uint64_t)v;
            c = (uint64_t)(v >> 64);
        }
        v = (u128)t[4] + c;
        t[3] = (uint64_t)v;
        t[4] = t[5] + (uint64_t)(v >> 64);
    }
    if (t[4] || geq(t, p)) sub_n(t, t, p);
    memcpy(r->l, t, 4 * sizeof(uint64_t));
}

/* generic pow, bits of e from the most significant set bit down: res = res^2; if bit: res *= x.
 * Starts from one (= R mod p in Montgomery form), like ark-ff's Field::pow. */
static inline void fe_pow(fe *r, const fe *x, uint64_t e, const fe *one, const uint64_t p[4], uint64_t inv) {
    fe res = *one;
    int started = 0;
    for (int bit = 63; bit >= 0; --bit) {
        int b = (int)((e >> bit) & 1);
        if (!started && !b) continue;
        started = 1;
        fe_mul(&res, &res, &res, p, inv);
        if (b) fe_mul(&res, &res, x, p, inv);
    }
    *r = res;
}

static void mont_one(fe *one, const uint64_t p[4]) {
    /* R mod p by repeated doubling of 1: 256 modular doublings */
    fe v = {{1, 0, 0, 0}};
    for (int i = 0; i < 256; ++i) fe_add(&v, &v, &v, p);
    *one = v;
}

static void permute_one(const pref_config *c, fe *s, const fe *one) {
    const uint32_t t = c->t;
    const uint32_t half = c->full_rounds / 2;
    const uint32_t total = c->full_rounds + c->partial_rounds;
    const fe *ark = (const fe *)c->ark;
    const fe *mds = (const fe *)c->mds;
    fe ns[PREF_MAX_T];
    for (uint32_t r = 0; r < total; ++r) {
        for (uint32_t i = 0; i < t; ++i) fe_add(&s[i], &s[i], &ark[r * t + i], c->modulus);
        if (r < half || r >= half + c->partial_rounds) {
            for (uint32_t i = 0; i < t; ++i) fe_pow(&s[i], &s[i], c->alpha, one, c->modulus, c->inv);
        } else {
            fe_pow(&s[0], &s[0], c->alpha, one, c->modulus, c->inv);
        }
        for (uint32_t i = 0; i < t; ++i) {
            fe cur = {{0, 0, 0, 0}};
            for (uint32_t j = 0; j < t; ++j) {
                fe term;
                fe_mul(&term, &s[j], &mds[i * t + j], c->modulus, c->inv);
                fe_add(&cur, &cur, &term, c->modulus);
            }
            ns[i] = cur;
        }
        memcpy(s, ns, t * sizeof(fe));
    }
}

int pref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* states: [n][t][4] Montgomery limbs, permuted in place.  threads<=0: all cores. */
int pref_permute_batch(const pref_config *c, uint64_t *states, size_t n, int threads) {
    if (c->t == 0 || c->t > PREF_MAX_T) return 1;
    fe one;
    mont_one(&one, c->modulus);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (long long k = 0; k < (long long)n; ++k) {
        permute_one(c, (fe *)(states + (size_t)k * c->t * 4), &one);
    }
    return 0;
}

/* ---- duplex sponge on one explicit (state, mode, index) triple ------------------------------- */
enum { PREF_ABSORBING = 0, PREF_SQUEEZING = 1 };

static void absorb_internal(const pref_config *c, fe *st, uint32_t *mode, uint32_t *index,
                            uint32_t start, const fe *in, size_t len, const fe *one) {
    for (;;) {
        if (start + len <= c->rate) {
            for (size_t i = 0; i < len; ++i)
                fe_add(&st[c->capacity + start + i], &st[c->capacity + start + i], &in[i], c->modulus);
            *mode = PREF_ABSORBING;
            *index = (uint32_t)(start + len);
            return;
        }
        uint32_t take = c->rate - start;
        for (uint32_t i = 0; i < take; ++i)
            fe_add(&st[c->capacity + start + i], &st[c->capacity + start + i], &in[i], c->modulus);
        permute_one(c, st, one);
        in += take;
        len -= take;
        start = 0;
    }
}

static void squeeze_internal(const pref_config *c, fe *st, uint32_t *mode, uint32_t *index,
                             uint32_t start, fe *out, size_t len, const fe *one) {
    for (;;) {
        if (start + len <= c->rate) {
            memcpy(out, &st[c->capacity + start], len * sizeof(fe));
            *mode = PREF_SQUEEZING;
            *index = (uint32_t)(start + len);
            return;
        }
        uint32_t take = c->rate - start;
        memcpy(out, &st[c->capacity + start], take * sizeof(fe));
        if (len != c->rate) permute_one(c, st, one); /* src/poseidon/mod.rs:175 */
        out += take;
        len -= take;
        start = 0;
    }
}

int pref_sponge_absorb(const pref_config *c, uint64_t *state, uint32_t *mode, uint32_t *index,
                       const uint64_t *in, size_t len) {
    fe one;
    mont_one(&one, c->modulus);
    if (len == 0) return 0;
    uint32_t idx = 0;
    if (*mode == PREF_ABSORBING) {
        idx = *index;
        if (idx == c->rate) { permute_one(c, (fe *)state, &one); idx = 0; }
    } else {
        permute_one(c, (fe *)state, &one);
    }
    absorb_internal(c, (fe *)state, mode, index, idx, (const fe *)in, len, &one);
    return 0;
}

int pref_sponge_squeeze(const pref_config *c, uint64_t *state, uint32_t *mode, uint32_t *index,
                        uint64_t *out, size_t len) {
    fe one;
    mont_one(&one, c->modulus);
    uint32_t idx = 0;
    if (*mode == PREF_ABSORBING) {
        permute_one(c, (fe *)state, &one);
    } else {
        idx = *index;
        if (idx == c->rate) { permute_one(c, (fe *)state, &one); idx = 0; }
    }
    squeeze_internal(c, (fe *)state, mode, index, idx, (fe *)out, len, &one);
    return 0;
}

/* per row: new; absorb(L elements); squeeze_native(k).   in [n][L][4], out [n][k][4] */
int pref_hash_batch(const pref_config *c, const uint64_t *in, size_t L, uint64_t *out, size_t k,
                    size_t n, int threads) {
    if (c->t == 0 || c->t > PREF_MAX_T) return 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (long long r = 0; r < (long long)n; ++r) {
        fe st[PREF_MAX_T];
        memset(st, 0, sizeof st);
        uint32_t mode = PREF_ABSORBING, index = 0;
        pref_sponge_absorb(c, (uint64_t *)st, &mode, &index, in + (size_t)r * L * 4, L);
        pref_sponge_squeeze(c, (uint64_t *)st, &mode, &index, out + (size_t)r * k * 4, k);
    }
    return 0;
}

/* nodes: [2m-1][4]; nodes[0..m) = leaves on entry, then level by level, root last. */
int pref_merkle_2to1(const pref_config *c, uint64_t *nodes, size_t m, int threads) {
    if (m == 0 || (m & (m - 1))) return 1;
    size_t src = 0, width = m;
    while (width > 1) {
        int rc = pref_hash_batch(c, nodes + src * 4, 2, nodes + (src + width) * 4, 1, width / 2, threads);
        if (rc) return rc;
        src += width;
        width /= 2;
    }
    return 0;
}

/* canonical <-> Montgomery, elementwise over n field elements; r2 = R^2 mod p */
int pref_to_mont(const uint64_t modulus[4], uint64_t inv, const uint64_t r2[4], uint64_t *x, size_t n) {
    fe R2;
    memcpy(R2.l, r2, sizeof R2.l);
    for (size_t i = 0; i < n; ++i) fe_mul((fe *)(x + 4 * i), (fe *)(x + 4 * i), &R2, modulus, inv);
    return 0;
}

int pref_from_mont(const uint64_t modulus[4], uint64_t inv, uint64_t *x, size_t n) {
    fe one = {{1, 0, 0, 0}};
    for (size_t i = 0; i < n; ++i) fe_mul((fe *)(x + 4 * i), (fe *)(x + 4 * i), &one, modulus, inv);
    return 0;
}
