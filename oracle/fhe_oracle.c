/*
 * fhe_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see fhe_oracle.h for the pinning statement).
 *
 * Written from the behaviour of the reference, not from its text: 64-bit limbs with explicit
 * carries via unsigned __int128 stand where the reference strings PTX add.cc/addc/madc together
 * (kernels/ptx_bigint.cuh:34-117).
 */
#include "fhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------ */
/* 256-bit wrap-around add / sub; return carry / borrow out of limb 3 (the reference drops it). */
static inline unsigned add256(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (unsigned)c;
}
static inline unsigned sub256(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    unsigned borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (unsigned)(d >> 64) & 1u;
    }
    return borrow;
}

/* include/bigint.cuh:27-48.  NOTE the borrow test: the reference does not look at the borrow
 * flag, it compares the top limbs (`temp.limbs[3] > result.limbs[3]`, :45) -- SURVEY D15. */
void orc_add_mod(orc_u256 *r, const orc_u256 *a, const orc_u256 *b, const orc_u256 *q) {
    uint64_t s[4], t[4];
    add256(s, a->limbs, b->limbs);                 /* carry out of limb 3 discarded (:32-35) */
    sub256(t, s, q->limbs);                        /* (:39-42) */
    int underflow = t[3] > s[3];                   /* (:45) */
    memcpy(r->limbs, underflow ? s : t, 32);
}

/* include/bigint.cuh:50-73 ; borrow detected as result.limbs[3] > a.limbs[3] (:61). */
void orc_sub_mod(orc_u256 *r, const orc_u256 *a, const orc_u256 *b, const orc_u256 *q) {
    uint64_t d[4], t[4];
    sub256(d, a->limbs, b->limbs);
    int borrow = d[3] > a->limbs[3];
    if (borrow) { add256(t, d, q->limbs); memcpy(r->limbs, t, 32); }
    else memcpy(r->limbs, d, 32);
}

/* include/bigint.cuh:76-140 : separated-operand-scanning Montgomery, R = 2^256.
 * Product rows (:83-98), four reduction rounds m = t[i]*inv0 (:100-122) with the ripple stopping at
 * limb 7 (carry out of the 512-bit accumulator is lost), result = upper half (:125-129), one
 * conditional subtraction with the top-limb borrow test (:131-139). */
void orc_mont_mul(orc_u256 *r, const orc_u256 *a, const orc_u256 *b, const orc_u256 *q, uint64_t inv0) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 acc = (u128)a->limbs[i] * b->limbs[j] + carry + t[i + j];
            t[i + j] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        t[i + 4] = carry;
    }
    for (int i = 0; i < 4; i++) {
        uint64_t m = t[i] * inv0;
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 acc = (u128)m * q->limbs[j] + carry + t[i + j];
            t[i + j] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        for (int j = 4; j < 8 - i; j++) {
            u128 acc = (u128)t[i + j] + carry;
            t[i + j] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
    }
    uint64_t u[4] = { t[4], t[5], t[6], t[7] }, d[4];
    sub256(d, u, q->limbs);
    int underflow = d[3] > u[3];
    memcpy(r->limbs, underflow ? u : d, 32);
}

/* src/bigint.cu:23-40 : six Newton steps from x = 1 on the low limb, then negate. */
uint64_t orc_mont_inverse(const orc_u256 *q) {
    uint64_t inv = 1, n0 = q->limbs[0];
    for (int i = 0; i < 6; i++) inv = inv * (2 - n0 * inv);
    return (uint64_t)0 - inv;
}

/* include/ntt.cuh:147-155 : b is formed from the OLD a. */
void orc_ct_butterfly(orc_u256 *a, orc_u256 *b, const orc_u256 *w, const orc_u256 *q, uint64_t inv0) {
    orc_u256 t, na, nb;
    orc_mont_mul(&t, b, w, q, inv0);
    orc_sub_mod(&nb, a, &t, q);
    orc_add_mod(&na, a, &t, q);
    *a = na; *b = nb;
}

/* include/ntt.cuh:158-167 */
void orc_gs_butterfly(orc_u256 *a, orc_u256 *b, const orc_u256 *w, const orc_u256 *q, uint64_t inv0) {
    orc_u256 s, d, nb;
    orc_add_mod(&s, a, b, q);
    orc_sub_mod(&d, a, b, q);
    orc_mont_mul(&nb, &d, w, q, inv0);
    *a = s; *b = nb;
}

void orc_batch_add(orc_u256 *r, const orc_u256 *a, const orc_u256 *b, const orc_u256 *q, size_t count) {
    for (size_t i = 0; i < count; i++) orc_add_mod(&r[i], &a[i], &b[i], q);
}
void orc_batch_sub(orc_u256 *r, const orc_u256 *a, const orc_u256 *b, const orc_u256 *q, size_t count) {
    for (size_t i = 0; i < count; i++) orc_sub_mod(&r[i], &a[i], &b[i], q);
}
void orc_batch_mont(orc_u256 *r, const orc_u256 *a, const orc_u256 *b, const orc_u256 *q, uint64_t inv0, size_t count) {
    for (size_t i = 0; i < count; i++) orc_mont_mul(&r[i], &a[i], &b[i], q, inv0);
}

/* ------------------------------------------------------------------------------------------ */
/* L1: the kernels as written.  Within one stage every thread touches a private (idx1, idx2)
 * pair, so running the threads one after another between two barriers is equivalent. */
static uint32_t ref_log_n(uint32_t n) { return (uint32_t)__builtin_popcount(n - 1) + 1; } /* ntt_kernels.cu:28 (sic: log2(n)+1) */

void orc_ref_forward_kernel(orc_u256 *data, const orc_u256 *tw, const orc_u256 *q, uint64_t inv0, uint32_t n) {
    uint32_t log_n = ref_log_n(n);
    for (uint32_t stage = 0; stage < log_n; stage++) {
        uint32_t m = 1u << stage, m2 = m << 1;
        for (uint32_t tid = 0; tid < n; tid++) {
            uint32_t k = tid / m, j = tid % m;
            if ((uint64_t)k * m2 + j + m < n) {
                uint32_t idx1 = k * m2 + j, idx2 = idx1 + m;
                uint32_t tw_idx = j << (log_n - stage - 1);
                orc_u256 u = data[idx1], v;
                orc_mont_mul(&v, &data[idx2], &tw[tw_idx], q, inv0);
                orc_add_mod(&data[idx1], &u, &v, q);
                orc_sub_mod(&data[idx2], &u, &v, q);
            }
        }
    }
}

void orc_ref_inverse_kernel(orc_u256 *data, const orc_u256 *itw, const orc_u256 *q, uint64_t inv0,
                            const orc_u256 *n_inv, uint32_t n) {
    uint32_t log_n = ref_log_n(n);
    for (int stage = (int)log_n - 1; stage >= 0; stage--) {
        uint32_t m = 1u << stage, m2 = m << 1;
        for (uint32_t tid = 0; tid < n; tid++) {
            uint32_t k = tid / m, j = tid % m;
            if ((uint64_t)k * m2 + j + m < n) {
                uint32_t idx1 = k * m2 + j, idx2 = idx1 + m;
                uint32_t tw_idx = j << (log_n - (uint32_t)stage - 1);
                orc_u256 u = data[idx1], v = data[idx2], d;
                orc_add_mod(&data[idx1], &u, &v, q);
                orc_sub_mod(&d, &u, &v, q);
                orc_mont_mul(&data[idx2], &d, &itw[tw_idx], q, inv0);
            }
        }
    }
    for (uint32_t i = 0; i < n; i++) { orc_u256 x = data[i]; orc_mont_mul(&data[i], &x, n_inv, q, inv0); }
}

void orc_ref_pointwise_kernel(orc_u256 *r, const orc_u256 *a, const orc_u256 *b, const orc_u256 *q,
                              uint64_t inv0, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) orc_mont_mul(&r[i], &a[i], &b[i], q, inv0);
}

/* ntt_stockham_kernel (kernels/ntt_kernels.cu:213-243), the n/2 in-bounds butterflies of one out-of-place stage. */
void orc_ref_stockham_stage(orc_u256 *output, const orc_u256 *input, const orc_u256 *tw, const orc_u256 *q, uint64_t inv0, uint32_t n, uint32_t stage) {
    uint32_t m = 1u << stage, m2 = m << 1;
    for (uint32_t idx = 0; idx < n / 2; idx++) {
        uint32_t k = idx / m, j = idx % m, idx1 = k * m2 + j, idx2 = idx1 + m;
        uint32_t tw_idx = j * (n / m2);
        orc_u256 u = input[idx1], v;
        orc_mont_mul(&v, &input[idx2], &tw[tw_idx], q, inv0);
        orc_add_mod(&output[idx1], &u, &v, q);
        orc_sub_mod(&output[idx2], &u, &v, q);
    }
}

void orc_ref_placeholder_table(orc_u256 *tw, uint32_t n) {
    memset(tw, 0, (size_t)n * sizeof(orc_u256));
    tw[0].limbs[0] = 1;                                      /* src/ntt.cu:86,93 */
    for (uint32_t i = 1; i < n; i++) tw[i].limbs[0] = i;     /* src/ntt.cu:87-90,94-96 */
}

/* ------------------------------------------------------------------------------------------ */
/* L2 plan: host maths built on the primitives above. */
struct orc_plan {
    uint32_t n, log_n;
    orc_u256 q;
    uint64_t inv0;
    orc_u256 r1;        /* R mod q   (Montgomery form of 1) */
    orc_u256 r2;        /* R^2 mod q */
    orc_u256 psi;       /* plain */
    orc_u256 n_inv_m;   /* n^-1 * R mod q */
    orc_u256 *tw_m;     /* psi^bitrev(k) * R mod q, k in [0,n) */
    orc_u256 *itw_m;    /* psi^-bitrev(k) * R mod q */
};

static int u256_is_zero(const orc_u256 *a) { return !(a->limbs[0] | a->limbs[1] | a->limbs[2] | a->limbs[3]); }
static int u256_eq(const orc_u256 *a, const orc_u256 *b) { return memcmp(a, b, 32) == 0; }
static orc_u256 u256_from(uint64_t v) { orc_u256 r = {{v, 0, 0, 0}}; return r; }
static void u256_shr(orc_u256 *a, unsigned s) { /* 0 < s < 64 */
    for (int i = 0; i < 4; i++) {
        uint64_t hi = (i < 3) ? a->limbs[i + 1] : 0;
        a->limbs[i] = (a->limbs[i] >> s) | (hi << (64 - s));
    }
}
static uint32_t bitrev(uint32_t x, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

static void to_mont(const orc_plan *p, orc_u256 *r, const orc_u256 *a) { orc_mont_mul(r, a, &p->r2, &p->q, p->inv0); }
static void from_mont(const orc_plan *p, orc_u256 *r, const orc_u256 *a) { orc_u256 one = u256_from(1); orc_mont_mul(r, a, &one, &p->q, p->inv0); }

/* base^e, everything in Montgomery form except the exponent. */
static void pow_mont(const orc_plan *p, orc_u256 *r, const orc_u256 *base_m, const orc_u256 *e) {
    orc_u256 acc = p->r1;
    for (int i = 255; i >= 0; i--) {
        orc_u256 t; orc_mont_mul(&t, &acc, &acc, &p->q, p->inv0); acc = t;
        if ((e->limbs[i / 64] >> (i % 64)) & 1) { orc_mont_mul(&t, &acc, base_m, &p->q, p->inv0); acc = t; }
    }
    *r = acc;
}

orc_plan *orc_plan_create(uint32_t n, const orc_u256 *q) {
    if (n < 2 || (n & (n - 1)) || !(q->limbs[0] & 1) || (q->limbs[3] >> 63)) return NULL;
    orc_plan *p = (orc_plan *)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->n = n; p->q = *q; p->inv0 = orc_mont_inverse(q);
    while ((1u << p->log_n) < n) p->log_n++;

    /* R mod q and R^2 mod q by 512 modular doublings of 1 (the reference sets r_squared = 1, src/bigint.cu:49). */
    orc_u256 x = u256_from(1), one = x, t;
    {   /* 1 mod q (q > 1 assumed) */
        for (int i = 0; i < 512; i++) { orc_add_mod(&t, &x, &x, q); x = t; if (i == 255) p->r1 = x; }
        p->r2 = x;
    }
    /* q = 1 mod 2n ? */
    orc_u256 qm1; { orc_u256 z = u256_from(0); (void)z; qm1 = *q; qm1.limbs[0] -= 1; }
    uint64_t two_n = 2ull * n;
    if (qm1.limbs[0] & (two_n - 1)) { free(p); return NULL; }
    orc_u256 e = qm1; u256_shr(&e, p->log_n + 1);            /* (q-1)/2n */
    orc_u256 n_u = u256_from(n);
    orc_u256 qm1_m; to_mont(p, &qm1_m, &qm1);                /* (q-1)*R mod q == -R */

    /* psi search: first x = 2,3,... with (x^e)^n == -1. */
    int found = 0;
    for (uint64_t g = 2; g < 100000 && !found; g++) {
        orc_u256 g_u = u256_from(g), g_m, c_m, c_n;
        to_mont(p, &g_m, &g_u);
        pow_mont(p, &c_m, &g_m, &e);
        pow_mont(p, &c_n, &c_m, &n_u);
        if (u256_eq(&c_n, &qm1_m)) { from_mont(p, &p->psi, &c_m); found = 1; }
    }
    if (!found) { free(p); return NULL; }

    /* inverse of psi = psi^(2n-1); n^-1 = n^(q-2) (Fermat; q prime). */
    orc_u256 psi_m, ipsi_m, ex;
    to_mont(p, &psi_m, &p->psi);
    ex = u256_from(two_n - 1);
    pow_mont(p, &ipsi_m, &psi_m, &ex);
    orc_u256 n_m; to_mont(p, &n_m, &n_u);
    ex = *q; ex.limbs[0] -= 2;                               /* q odd > 2: no borrow */
    pow_mont(p, &p->n_inv_m, &n_m, &ex);
    /* sanity: n * n^-1 == 1 */
    orc_mont_mul(&t, &n_m, &p->n_inv_m, q, p->inv0);
    if (!u256_eq(&t, &p->r1)) { free(p); return NULL; }      /* q not prime */

    p->tw_m = (orc_u256 *)malloc((size_t)n * sizeof(orc_u256));
    p->itw_m = (orc_u256 *)malloc((size_t)n * sizeof(orc_u256));
    if (!p->tw_m || !p->itw_m) { orc_plan_destroy(p); return NULL; }
    /* natural powers, scattered to bit-reversed slots */
    orc_u256 pw = p->r1, ipw = p->r1;
    for (uint32_t k = 0; k < n; k++) {
        uint32_t s = bitrev(k, p->log_n);
        p->tw_m[s] = pw; p->itw_m[s] = ipw;
        orc_mont_mul(&t, &pw, &psi_m, q, p->inv0); pw = t;
        orc_mont_mul(&t, &ipw, &ipsi_m, q, p->inv0); ipw = t;
    }
    (void)one; (void)u256_is_zero;
    return p;
}

void orc_plan_destroy(orc_plan *p) { if (!p) return; free(p->tw_m); free(p->itw_m); free(p); }
void orc_plan_psi(const orc_plan *p, orc_u256 *psi) { *psi = p->psi; }
uint64_t orc_plan_inv0(const orc_plan *p) { return p->inv0; }
void orc_plan_twiddle(const orc_plan *p, uint32_t k, orc_u256 *w) { from_mont(p, w, &p->tw_m[k % p->n]); }

/* Merged-twiddle Cooley-Tukey, natural in -> bit-reversed out.  Twiddles are in Montgomery form,
 * data in plain form, so ct_butterfly's mont(b, w) is the exact plain product. */
void orc_ntt_forward(const orc_plan *p, orc_u256 *x) {
    uint32_t n = p->n, t = n;
    for (uint32_t m = 1; m < n; m <<= 1) {
        t >>= 1;
        for (uint32_t i = 0; i < m; i++) {
            uint32_t j1 = 2 * i * t;
            const orc_u256 *w = &p->tw_m[m + i];
            for (uint32_t j = j1; j < j1 + t; j++) orc_ct_butterfly(&x[j], &x[j + t], w, &p->q, p->inv0);
        }
    }
}

/* Gentleman-Sande, bit-reversed in -> natural out, then the n^-1 scaling. */
void orc_ntt_inverse(const orc_plan *p, orc_u256 *x) {
    uint32_t n = p->n, t = 1;
    for (uint32_t m = n >> 1; m >= 1; m >>= 1) {
        for (uint32_t i = 0; i < m; i++) {
            uint32_t j1 = 2 * i * t;
            const orc_u256 *w = &p->itw_m[m + i];
            for (uint32_t j = j1; j < j1 + t; j++) orc_gs_butterfly(&x[j], &x[j + t], w, &p->q, p->inv0);
        }
        t <<= 1;
    }
    for (uint32_t j = 0; j < n; j++) { orc_u256 v = x[j]; orc_mont_mul(&x[j], &v, &p->n_inv_m, &p->q, p->inv0); }
}

void orc_ntt_pointwise(const orc_plan *p, orc_u256 *r, const orc_u256 *a, const orc_u256 *b) {
    for (uint32_t j = 0; j < p->n; j++) {
        orc_u256 t; orc_mont_mul(&t, &a[j], &b[j], &p->q, p->inv0);   /* a*b*R^-1 */
        orc_mont_mul(&r[j], &t, &p->r2, &p->q, p->inv0);              /* back to plain */
    }
}

void orc_polymul_ntt(const orc_plan *p, orc_u256 *r, const orc_u256 *a, const orc_u256 *b) {
    size_t bytes = (size_t)p->n * sizeof(orc_u256);
    orc_u256 *ta = (orc_u256 *)malloc(bytes), *tb = (orc_u256 *)malloc(bytes);
    memcpy(ta, a, bytes); memcpy(tb, b, bytes);                       /* src/ntt.cu:51-58 */
    orc_ntt_forward(p, ta); orc_ntt_forward(p, tb);
    orc_ntt_pointwise(p, r, ta, tb);
    orc_ntt_inverse(p, r);
    free(ta); free(tb);
}

void orc_polymul_schoolbook(const orc_plan *p, orc_u256 *r, const orc_u256 *a, const orc_u256 *b) {
    uint32_t n = p->n;
    orc_u256 *bm = (orc_u256 *)malloc((size_t)n * sizeof(orc_u256));
    for (uint32_t j = 0; j < n; j++) to_mont(p, &bm[j], &b[j]);       /* mont(a, bR) = a*b */
    memset(r, 0, (size_t)n * sizeof(orc_u256));
    for (uint32_t i = 0; i < n; i++) {
        if (u256_is_zero(&a[i])) continue;
        for (uint32_t j = 0; j < n; j++) {
            orc_u256 prod, acc;
            orc_mont_mul(&prod, &a[i], &bm[j], &p->q, p->inv0);
            uint32_t k = i + j;
            if (k < n) orc_add_mod(&acc, &r[k], &prod, &p->q);
            else { k -= n; orc_sub_mod(&acc, &r[k], &prod, &p->q); }   /* x^n = -1 */
            r[k] = acc;
        }
    }
    free(bm);
}

/* ------------------------------------------------------------------------------------------ */
static int u256_bits(const orc_u256 *a) {
    for (int i = 3; i >= 0; i--) if (a->limbs[i]) return 64 * i + 64 - __builtin_clzll(a->limbs[i]);
    return 0;
}
uint32_t orc_relin_num_digits(orc_plan *const *plans, uint32_t L, uint32_t decomp_bits) {
    int mx = 0;
    for (uint32_t l = 0; l < L; l++) { int b = u256_bits(&plans[l]->q); if (b > mx) mx = b; }
    return ((uint32_t)mx + decomp_bits - 1) / decomp_bits;
}
/* bits [lo, lo+w) of a, w <= 64 */
static uint64_t u256_extract(const orc_u256 *a, uint32_t lo, uint32_t w) {
    if (lo >= 256) return 0;
    uint32_t limb = lo / 64, sh = lo % 64;
    uint64_t v = a->limbs[limb] >> sh;
    if (sh && limb + 1 < 4) v |= a->limbs[limb + 1] << (64 - sh);
    return w >= 64 ? v : (v & ((1ull << w) - 1));
}
static void reduce_small(const orc_plan *p, orc_u256 *r, uint64_t d) {   /* d mod q for a 64-bit d */
    orc_u256 x = u256_from(d), t;
    if (p->q.limbs[1] | p->q.limbs[2] | p->q.limbs[3]) { *r = x; return; }
    x.limbs[0] = d % p->q.limbs[0]; *r = x; (void)t;
}
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int clamp_threads(int threads) {
    int mx = orc_max_threads();
    if (threads < 1) threads = 1;
    return threads > mx ? mx : threads;
}

int orc_rns_forward(orc_plan *const *plans, uint32_t L, orc_u256 *data, uint32_t batch, int threads) {
    threads = clamp_threads(threads);
    long total = (long)batch * L;
    #pragma omp parallel for num_threads(threads) schedule(dynamic) if (threads > 1)
    for (long u = 0; u < total; u++) {
        const orc_plan *p = plans[u % L];
        orc_ntt_forward(p, data + (size_t)u * p->n);
    }
    return threads;
}

int orc_rns_inverse(orc_plan *const *plans, uint32_t L, orc_u256 *data, uint32_t batch, int threads) {
    threads = clamp_threads(threads);
    long total = (long)batch * L;
    #pragma omp parallel for num_threads(threads) schedule(dynamic) if (threads > 1)
    for (long u = 0; u < total; u++) {
        const orc_plan *p = plans[u % L];
        orc_ntt_inverse(p, data + (size_t)u * p->n);
    }
    return threads;
}

int orc_rns_polymul(orc_plan *const *plans, uint32_t L, orc_u256 *r, const orc_u256 *a, const orc_u256 *b,
                    uint32_t batch, int threads) {
    threads = clamp_threads(threads);
    long total = (long)batch * L;
    #pragma omp parallel for num_threads(threads) schedule(dynamic) if (threads > 1)
    for (long u = 0; u < total; u++) {
        const orc_plan *p = plans[u % L];
        size_t off = (size_t)u * p->n;
        orc_polymul_ntt(p, r + off, a + off, b + off);
    }
    return threads;
}

int orc_ct_multiply(orc_plan *const *plans, uint32_t L, orc_u256 *c0, orc_u256 *c1, orc_u256 *c2,
                    const orc_u256 *a0, const orc_u256 *a1, const orc_u256 *b0, const orc_u256 *b1,
                    uint32_t batch, int threads) {
    threads = clamp_threads(threads);
    long total = (long)batch * L;
    #pragma omp parallel for num_threads(threads) schedule(dynamic) if (threads > 1)
    for (long u = 0; u < total; u++) {
        const orc_plan *p = plans[u % L];
        size_t off = (size_t)u * p->n, bytes = (size_t)p->n * sizeof(orc_u256);
        orc_u256 *t1 = (orc_u256 *)malloc(bytes), *t2 = (orc_u256 *)malloc(bytes);
        orc_polymul_ntt(p, c0 + off, a0 + off, b0 + off);              /* src/fhe.cu:208 */
        orc_polymul_ntt(p, t1, a0 + off, b1 + off);                    /* :211-212 */
        orc_polymul_ntt(p, t2, a1 + off, b0 + off);                    /* :213 */
        orc_batch_add(c1 + off, t1, t2, &p->q, p->n);                  /* :214 */
        orc_polymul_ntt(p, c2 + off, a1 + off, b1 + off);              /* :217 */
        free(t1); free(t2);
    }
    return threads;
}

int orc_relinearize(orc_plan *const *plans, uint32_t L, uint32_t decomp_bits, orc_u256 *c0, orc_u256 *c1, const orc_u256 *c2,
                    const orc_u256 *const *keys_b, const orc_u256 *const *keys_a, uint32_t batch, int threads) {
    if (decomp_bits < 1 || decomp_bits > 64) return -1;
    threads = clamp_threads(threads);
    const uint32_t K = orc_relin_num_digits(plans, L, decomp_bits), n = plans[0]->n;
    /* keys in NTT form, once */
    size_t polyb = (size_t)n * sizeof(orc_u256);
    orc_u256 *kb = (orc_u256 *)malloc(polyb * L * L * K), *ka = (orc_u256 *)malloc(polyb * L * L * K);
    for (uint32_t jk = 0; jk < L * K; jk++)
        for (uint32_t i = 0; i < L; i++) {
            orc_u256 *db = kb + ((size_t)jk * L + i) * n, *da = ka + ((size_t)jk * L + i) * n;
            memcpy(db, keys_b[jk] + (size_t)i * n, polyb); memcpy(da, keys_a[jk] + (size_t)i * n, polyb);
            orc_ntt_forward(plans[i], db); orc_ntt_forward(plans[i], da);
        }
    long total = (long)batch * L;
    #pragma omp parallel for num_threads(threads) schedule(dynamic) if (threads > 1)
    for (long u = 0; u < total; u++) {
        const uint32_t b = (uint32_t)(u / L), i = (uint32_t)(u % L);
        const orc_plan *p = plans[i];
        orc_u256 *acc0 = (orc_u256 *)calloc(n, sizeof(orc_u256)), *acc1 = (orc_u256 *)calloc(n, sizeof(orc_u256));
        orc_u256 *d = (orc_u256 *)malloc(polyb), *t = (orc_u256 *)malloc(polyb);
        for (uint32_t j = 0; j < L; j++) {
            const orc_u256 *src = c2 + ((size_t)b * L + j) * n;
            for (uint32_t k = 0; k < K; k++) {
                for (uint32_t x = 0; x < n; x++) reduce_small(p, &d[x], u256_extract(&src[x], k * decomp_bits, decomp_bits));
                orc_ntt_forward(p, d);
                const uint32_t jk = j * K + k;
                orc_ntt_pointwise(p, t, d, kb + ((size_t)jk * L + i) * n); orc_batch_add(acc0, acc0, t, &p->q, n);
                orc_ntt_pointwise(p, t, d, ka + ((size_t)jk * L + i) * n); orc_batch_add(acc1, acc1, t, &p->q, n);
            }
        }
        orc_ntt_inverse(p, acc0); orc_ntt_inverse(p, acc1);
        orc_u256 *o0 = c0 + ((size_t)b * L + i) * n, *o1 = c1 + ((size_t)b * L + i) * n;
        orc_batch_add(o0, o0, acc0, &p->q, n); orc_batch_add(o1, o1, acc1, &p->q, n);
        free(acc0); free(acc1); free(d); free(t);
    }
    free(kb); free(ka);
    return threads;
}

/* ------------------------------------------------------------------------------------------ */
/* values mod q through the Montgomery primitive: mont(mont(x, R^2), 1) = x mod q for any 256-bit x
 * (u = (x*R2 + m*q)/R < 2q, one conditional subtraction). */
void orc_to_rns(orc_plan *const *plans, uint32_t L, orc_u256 *rns, const orc_u256 *values, uint32_t batch) {
    const uint32_t n = plans[0]->n;
    for (uint32_t b = 0; b < batch; b++)
        for (uint32_t l = 0; l < L; l++) {
            const orc_plan *p = plans[l];
            for (uint32_t x = 0; x < n; x++) {
                orc_u256 t;
                orc_mont_mul(&t, &values[(size_t)b * n + x], &p->r2, &p->q, p->inv0);
                from_mont(p, &rns[((size_t)b * L + l) * n + x], &t);
            }
        }
}

/* a*b for multi-limb a (4 limbs) and a 256-bit b, low 256 bits; returns 0 on overflow. */
static int mul256_checked(orc_u256 *r, const orc_u256 *a, const orc_u256 *b) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 acc = (u128)a->limbs[i] * b->limbs[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)acc; carry = (uint64_t)(acc >> 64);
        }
        t[i + 4] = carry;
    }
    memcpy(r->limbs, t, 32);
    return !(t[4] | t[5] | t[6] | t[7]);
}

int orc_from_rns(orc_plan *const *plans, uint32_t L, orc_u256 *values, const orc_u256 *rns, uint32_t batch) {
    const uint32_t n = plans[0]->n;
    /* Q and the CRT constants, with the arithmetic mod Q done by a throw-away plan-like context */
    orc_u256 Q = u256_from(1);
    for (uint32_t l = 0; l < L; l++) { orc_u256 t; if (!mul256_checked(&t, &Q, &plans[l]->q)) return -1; Q = t; }
    if (Q.limbs[3] >> 63) return -1;
    orc_plan big; memset(&big, 0, sizeof big);
    big.q = Q; big.inv0 = orc_mont_inverse(&Q);
    { orc_u256 x = u256_from(1), t; for (int i = 0; i < 512; i++) { orc_add_mod(&t, &x, &x, &Q); x = t; if (i == 255) big.r1 = x; } big.r2 = x; }
    orc_u256 *Mi_m = (orc_u256 *)malloc(L * sizeof(orc_u256)), *inv_m = (orc_u256 *)malloc(L * sizeof(orc_u256));
    for (uint32_t l = 0; l < L; l++) {
        orc_u256 Mi = u256_from(1), t;
        for (uint32_t k = 0; k < L; k++) if (k != l) { mul256_checked(&t, &Mi, &plans[k]->q); Mi = t; }
        to_mont(&big, &Mi_m[l], &Mi);                                   /* M_l * R_Q mod Q */
        const orc_plan *p = plans[l];
        orc_u256 mi_mod, mi_m, e = p->q; e.limbs[0] -= 2;
        orc_mont_mul(&t, &Mi, &p->r2, &p->q, p->inv0); mi_m = t;         /* (M_l mod q_l) * R, any 256-bit M_l */
        (void)mi_mod;
        pow_mont(p, &inv_m[l], &mi_m, &e);                              /* (M_l^-1 mod q_l) * R */
    }
    for (uint32_t b = 0; b < batch; b++)
        for (uint32_t x = 0; x < n; x++) {
            orc_u256 acc = u256_from(0);
            for (uint32_t l = 0; l < L; l++) {
                const orc_plan *p = plans[l];
                orc_u256 tl, term, s;
                orc_mont_mul(&tl, &rns[((size_t)b * L + l) * n + x], &inv_m[l], &p->q, p->inv0);   /* [r_l * M_l^-1]_{q_l} */
                orc_mont_mul(&term, &tl, &Mi_m[l], &Q, big.inv0);                                 /* * M_l mod Q */
                orc_add_mod(&s, &acc, &term, &Q); acc = s;
            }
            values[(size_t)b * n + x] = acc;
        }
    free(Mi_m); free(inv_m);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
static int u256_gt(const orc_u256 *a, const orc_u256 *b) {        /* a > b */
    for (int i = 3; i >= 0; i--) { if (a->limbs[i] != b->limbs[i]) return a->limbs[i] > b->limbs[i]; }
    return 0;
}
void orc_rescale_drop_last(orc_plan *const *plans, uint32_t L, orc_u256 *out, const orc_u256 *in, uint32_t batch) {
    const uint32_t n = plans[0]->n;
    const orc_plan *pl = plans[L - 1];
    orc_u256 half = pl->q; u256_shr(&half, 1);                       /* floor(q_last / 2) */
    for (uint32_t l = 0; l + 1 < L; l++) {
        const orc_plan *p = plans[l];
        /* q_last^-1 mod q_l in Montgomery form: (q_last mod q_l)^(q_l - 2) */
        orc_u256 ql_m, inv_m, e = p->q; e.limbs[0] -= 2;
        orc_mont_mul(&ql_m, &pl->q, &p->r2, &p->q, p->inv0);
        pow_mont(p, &inv_m, &ql_m, &e);
        for (uint32_t b = 0; b < batch; b++)
            for (uint32_t x = 0; x < n; x++) {
                const orc_u256 *cl = &in[((size_t)b * L + (L - 1)) * n + x];
                orc_u256 mag, t, r_l, d;
                const int neg = u256_gt(cl, &half);                /* centred residue r = cl - q_last < 0 */
                if (neg) sub256(mag.limbs, pl->q.limbs, cl->limbs); else mag = *cl;
                orc_mont_mul(&t, &mag, &p->r2, &p->q, p->inv0); from_mont(p, &r_l, &t);     /* |r| mod q_l */
                if (neg && !u256_is_zero(&r_l)) { orc_u256 z; sub256(z.limbs, p->q.limbs, r_l.limbs); r_l = z; }
                orc_sub_mod(&d, &in[((size_t)b * L + l) * n + x], &r_l, &p->q);
                orc_mont_mul(&out[((size_t)b * (L - 1) + l) * n + x], &d, &inv_m, &p->q, p->inv0);
            }
    }
}

/* ------------------------------------------------------------------------------------------ */
void orc_fast_base_convert(orc_plan *const *src, uint32_t L, orc_plan *const *dst, uint32_t Lp, orc_u256 *out, const orc_u256 *in,
                           uint32_t batch) {
    const uint32_t n = src[0]->n;
    /* (M_i^-1 mod q_i) * R_i, and for every target j: (M_i mod p_j) * R_j   with M_i = prod_{k != i} q_k taken mod the modulus at hand */
    orc_u256 *inv_m = (orc_u256 *)malloc(L * sizeof(orc_u256)), *Mij_m = (orc_u256 *)malloc((size_t)L * Lp * sizeof(orc_u256));
    for (uint32_t i = 0; i < L; i++) {
        const orc_plan *p = src[i];
        orc_u256 acc = p->r1, t, e = p->q; e.limbs[0] -= 2;
        for (uint32_t k = 0; k < L; k++) if (k != i) { orc_u256 qk_m; orc_mont_mul(&qk_m, &src[k]->q, &p->r2, &p->q, p->inv0); orc_mont_mul(&t, &acc, &qk_m, &p->q, p->inv0); acc = t; }
        pow_mont(p, &inv_m[i], &acc, &e);
        for (uint32_t j = 0; j < Lp; j++) {
            const orc_plan *d = dst[j];
            orc_u256 a = d->r1;
            for (uint32_t k = 0; k < L; k++) if (k != i) { orc_u256 qk_m; orc_mont_mul(&qk_m, &src[k]->q, &d->r2, &d->q, d->inv0); orc_mont_mul(&t, &a, &qk_m, &d->q, d->inv0); a = t; }
            Mij_m[(size_t)i * Lp + j] = a;
        }
    }
    for (uint32_t b = 0; b < batch; b++)
        for (uint32_t x = 0; x < n; x++)
            for (uint32_t j = 0; j < Lp; j++) {
                const orc_plan *d = dst[j];
                orc_u256 acc = u256_from(0);
                for (uint32_t i = 0; i < L; i++) {
                    const orc_plan *p = src[i];
                    orc_u256 ti, ti_red, ti_m, term, s2;
                    orc_mont_mul(&ti, &in[((size_t)b * L + i) * n + x], &inv_m[i], &p->q, p->inv0);        /* [x_i * M_i^-1]_{q_i} */
                    orc_mont_mul(&ti_m, &ti, &d->r2, &d->q, d->inv0);                                     /* (t_i mod p_j) * R_j  */
                    orc_mont_mul(&term, &ti_m, &Mij_m[(size_t)i * Lp + j], &d->q, d->inv0);               /* t_i * M_i * R_j      */
                    from_mont(d, &ti_red, &term);
                    orc_add_mod(&s2, &acc, &ti_red, &d->q); acc = s2;
                }
                out[((size_t)b * Lp + j) * n + x] = acc;
            }
    free(inv_m); free(Mij_m);
}

/* ------------------------------------------------------------------------------------------ */
void orc_monomial_mul_sub(orc_plan *const *plans, uint32_t L, orc_u256 *out, const orc_u256 *in, const uint32_t *shifts, uint32_t batch) {
    const uint32_t n = plans[0]->n;
    for (uint32_t b = 0; b < batch; b++) {
        const uint32_t a = shifts[b] % (2 * n);
        for (uint32_t l = 0; l < L; l++) {
            const orc_plan *p = plans[l];
            const orc_u256 *src = in + ((size_t)b * L + l) * n;
            orc_u256 *dst = out + ((size_t)b * L + l) * n;
            for (uint32_t x = 0; x < n; x++) {
                /* coefficient x of X^a * p: +-p[(x - a) mod 2n folded into n] */
                uint32_t k = (x + 2 * n - a) % (2 * n);
                int neg = k >= n; if (neg) k -= n;
                orc_u256 zero = u256_from(0), v = src[k], t;
                if (neg) { orc_sub_mod(&t, &zero, &v, &p->q); v = t; }
                orc_sub_mod(&dst[x], &v, &src[x], &p->q);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Row N4: samplers, single-modulus modulus switch, negacyclic fold                             */
/* ------------------------------------------------------------------------------------------ */
/* sample_uniform_kernel (src/polynomial.cu:130-143), literal: ((seed + idx) * 1103515245 + 12345) % modulus.limbs[0],
 * uint64 wrap-around, idx a uint32_t. */
void orc_sample_uniform_lcg(orc_u256 *out, const orc_u256 *q, uint64_t seed, size_t count) {
    for (size_t i = 0; i < count; i++) {
        uint32_t idx = (uint32_t)i;
        out[i] = u256_from(((seed + idx) * 1103515245ull + 12345ull) % q->limbs[0]);
    }
}
/* sample_gaussian_kernel (src/polynomial.cu:113-128), literal placeholder: (seed + idx) % modulus.limbs[0]. */
void orc_sample_gaussian_placeholder(orc_u256 *out, const orc_u256 *q, uint64_t seed, size_t count) {
    for (size_t i = 0; i < count; i++) out[i] = u256_from((seed + (uint32_t)i) % q->limbs[0]);
}

/* Counter-based generator shared (by specification, not by code) with the device samplers: SplitMix64's output function over
 * (seed, element index, draw number). */
static uint64_t sm64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t ctr_rand(uint64_t seed, uint64_t index, uint64_t draw) { return sm64(sm64(seed ^ (index * 0xD1342543DE82EF95ull)) + draw); }
uint64_t orc_ctr_rand(uint64_t seed, uint64_t index, uint64_t draw) { return ctr_rand(seed, index, draw); }

static void embed_small(orc_plan *const *plans, uint32_t L, orc_u256 *out, size_t b, size_t x, uint64_t mag, int neg) {
    const uint32_t n = plans[0]->n;
    for (uint32_t l = 0; l < L; l++) {
        orc_u256 v = u256_from(mag);
        if (neg && mag) { orc_u256 t; sub256(t.limbs, plans[l]->q.limbs, v.limbs); v = t; }
        out[(b * L + l) * n + x] = v;
    }
}
/* sample_ternary_kernel (include/polynomial.cuh:129-135, declared only): nonzero with probability thr / 2^32, sign uniform. */
void orc_sample_ternary(orc_plan *const *plans, uint32_t L, orc_u256 *out, uint64_t thr, uint64_t seed, uint32_t batch) {
    const uint32_t n = plans[0]->n;
    for (size_t g = 0; g < (size_t)batch * n; g++) {
        uint64_t r = ctr_rand(seed, g, 0);
        embed_small(plans, L, out, g / n, g % n, (r & 0xffffffffull) < thr ? 1 : 0, (int)(r >> 63));
    }
}
/* Cumulative table of the discrete Gaussian cut at 12 sigma: table[k] = floor(2^64 P(|X| <= k)); returns len = ceil(12 sigma). */
uint32_t orc_gaussian_cdt(double sigma, uint64_t *table, uint32_t capacity) {
    uint32_t len = (uint32_t)ceil(sigma * 12.0);
    if (len < 1) len = 1;
    if (!table || capacity < len) return len;
    const double den = (2.0 * sigma) * sigma;
    double *w = (double *)malloc((len + 1) * sizeof(double));
    for (uint32_t k = 0; k <= len; k++) w[k] = exp(-((double)k * (double)k) / den);
    double Z = w[0];
    for (uint32_t k = 1; k <= len; k++) Z += 2.0 * w[k];
    double cum = 0;
    for (uint32_t k = 0; k < len; k++) {
        const double term = (k == 0 ? w[0] : 2.0 * w[k]) / Z;
        cum += term;
        const double scaled = cum * 18446744073709551616.0;
        table[k] = scaled >= 18446744073709551615.0 ? ~0ull : (uint64_t)scaled;
    }
    free(w);
    return len;
}
void orc_sample_gaussian(orc_plan *const *plans, uint32_t L, orc_u256 *out, const uint64_t *cdt, uint32_t len, uint64_t seed, uint32_t batch) {
    const uint32_t n = plans[0]->n;
    for (size_t g = 0; g < (size_t)batch * n; g++) {
        uint64_t r = ctr_rand(seed, g, 1), m = 0;
        for (uint32_t j = 0; j < len; j++) m += r >= cdt[j];
        embed_small(plans, L, out, g / n, g % n, m, (int)(ctr_rand(seed, g, 2) >> 63));
    }
}
/* Uniform residues in [0, q_l): rejection on draws masked to the bit length of q_l; draw t uses words 16 + 4t .. 16 + 4t + 3. */
void orc_sample_uniform(orc_plan *const *plans, uint32_t L, orc_u256 *out, uint64_t seed, uint32_t batch) {
    const uint32_t n = plans[0]->n;
    for (size_t g = 0; g < (size_t)batch * L * n; g++) {
        const orc_u256 *q = &plans[(g / n) % L]->q;
        int top = 3; while (top > 0 && q->limbs[top] == 0) top--;
        int bits = 64 - __builtin_clzll(q->limbs[top]);
        uint64_t mask = bits == 64 ? ~0ull : ((1ull << bits) - 1);
        orc_u256 v;
        for (uint32_t t = 0;; t++) {
            for (int i = 0; i < 4; i++) v.limbs[i] = i <= top ? ctr_rand(seed, g, 16 + 4 * (uint64_t)t + i) : 0;
            v.limbs[top] &= mask;
            if (u256_gt(q, &v)) break;
            if (t == 63) { v.limbs[top] &= mask >> 1; break; }
        }
        out[g] = v;
    }
}
/* poly_mod_switch_kernel (include/polynomial.cuh:96-103, declared; FHEContext::decrypt src/fhe.cu:181-184):
 * round(a * new_q / old_q) mod new_q, half up, a < old_q < 2^255, new_q < 2^64.  Restoring division of the 320-bit numerator. */
void orc_poly_mod_switch(orc_u256 *out, const orc_u256 *in, const orc_u256 *old_q, uint64_t new_q, size_t count) {
    orc_u256 half = *old_q; u256_shr(&half, 1);
    for (size_t g = 0; g < count; g++) {
        uint64_t p[5]; unsigned __int128 c = 0;
        for (int i = 0; i < 4; i++) { c += (unsigned __int128)in[g].limbs[i] * new_q + half.limbs[i]; p[i] = (uint64_t)c; c >>= 64; }
        p[4] = (uint64_t)c;
        orc_u256 rem = u256_from(0); unsigned __int128 quo = 0;
        for (int bit = 319; bit >= 0; bit--) {
            for (int i = 3; i > 0; i--) rem.limbs[i] = (rem.limbs[i] << 1) | (rem.limbs[i - 1] >> 63);
            rem.limbs[0] = (rem.limbs[0] << 1) | ((p[bit >> 6] >> (bit & 63)) & 1);
            quo <<= 1;
            if (!u256_gt(old_q, &rem)) { orc_u256 t; sub256(t.limbs, rem.limbs, old_q->limbs); rem = t; quo |= 1; }
        }
        out[g] = u256_from((uint64_t)(quo % new_q));
    }
}
/* negacyclic_reduce_kernel (include/polynomial.cuh:105-110, declared): data[i] = sub_mod(data[i], data[i + n]), i < n. */
void orc_negacyclic_reduce(orc_u256 *data, const orc_u256 *q, size_t n) {
    for (size_t i = 0; i < n; i++) { orc_u256 t; orc_sub_mod(&t, &data[i], &data[i + n], q); data[i] = t; }
}

/* ------------------------------------------------------------------------------------------ */
/* Word-sized CPU port of the same polymul (q < 2^62): 64-bit residues, Shoup-style constant    */
/* multiplication through unsigned __int128.  Same transforms, same outputs as orc_rns_polymul  */
/* (asserted by the tests); it exists so that bench.py can also quote a CPU baseline that does  */
/* NOT pay for the reference's 256-bit containers -- the fairer number to hold a GPU against.   */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t mulmod64(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((unsigned __int128)a * b % q); }
static inline uint64_t shoup_mul64(uint64_t x, uint64_t w, uint64_t ws, uint64_t q) {      /* x*w mod q for x < 2q..., w < q, ws = floor(w 2^64 / q) */
    uint64_t hi = (uint64_t)(((unsigned __int128)x * ws) >> 64);
    uint64_t r = x * w - hi * q;
    return r >= q ? r - q : r;
}
int orc_rns_polymul_narrow(orc_plan *const *plans, uint32_t L, orc_u256 *r, const orc_u256 *a, const orc_u256 *b, uint32_t batch, int threads) {
    const uint32_t n = plans[0]->n;
    for (uint32_t l = 0; l < L; l++) if (plans[l]->q.limbs[1] | plans[l]->q.limbs[2] | plans[l]->q.limbs[3] | (plans[l]->q.limbs[0] >> 62)) return -1;
    /* per-limb word tables: plain twiddles + Shoup companions */
    uint64_t **tw = (uint64_t **)malloc(L * sizeof(*tw));
    for (uint32_t l = 0; l < L; l++) {
        const orc_plan *p = plans[l]; const uint64_t q = p->q.limbs[0];
        tw[l] = (uint64_t *)malloc((size_t)4 * n * sizeof(uint64_t));            /* w, ws, iw, iws */
        for (uint32_t k = 0; k < n; k++) {
            orc_u256 w, iw; from_mont(p, &w, &p->tw_m[k]); from_mont(p, &iw, &p->itw_m[k]);
            tw[l][k] = w.limbs[0]; tw[l][n + k] = (uint64_t)(((unsigned __int128)w.limbs[0] << 64) / q);
            tw[l][2 * n + k] = iw.limbs[0]; tw[l][3 * n + k] = (uint64_t)(((unsigned __int128)iw.limbs[0] << 64) / q);
        }
    }
    threads = clamp_threads(threads);
    const long total = (long)batch * L;
    int used = 1;
#pragma omp parallel num_threads(threads)
    {
#pragma omp single
        used = omp_get_num_threads();
        uint64_t *x = (uint64_t *)malloc((size_t)2 * n * sizeof(uint64_t)), *y = x + n;
#pragma omp for schedule(dynamic)
        for (long pi = 0; pi < total; pi++) {
            const uint32_t l = (uint32_t)(pi % L);
            const orc_plan *p = plans[l]; const uint64_t q = p->q.limbs[0];
            const uint64_t *w = tw[l], *ws = tw[l] + n, *iw = tw[l] + 2 * n, *iws = tw[l] + 3 * n;
            for (uint32_t j = 0; j < n; j++) { x[j] = a[(size_t)pi * n + j].limbs[0]; y[j] = b[(size_t)pi * n + j].limbs[0]; }
            for (int which = 0; which < 2; which++) {                            /* forward: natural in, bit-reversed out */
                uint64_t *v = which ? y : x; uint32_t t = n;
                for (uint32_t m = 1; m < n; m <<= 1) {
                    t >>= 1;
                    for (uint32_t i = 0; i < m; i++) {
                        const uint64_t W = w[m + i], WS = ws[m + i];
                        for (uint32_t j = 2 * i * t; j < 2 * i * t + t; j++) {
                            const uint64_t u = v[j], s = shoup_mul64(v[j + t], W, WS, q);
                            uint64_t e = u + s; v[j] = e >= q ? e - q : e;
                            v[j + t] = u >= s ? u - s : u + q - s;
                        }
                    }
                }
            }
            for (uint32_t j = 0; j < n; j++) x[j] = mulmod64(x[j], y[j], q);
            uint32_t t = 1;                                                      /* inverse: bit-reversed in, natural out */
            for (uint32_t m = n >> 1; m >= 1; m >>= 1) {
                for (uint32_t i = 0; i < m; i++) {
                    const uint64_t W = iw[m + i], WS = iws[m + i];
                    for (uint32_t j = 2 * i * t; j < 2 * i * t + t; j++) {
                        const uint64_t u = x[j], s = x[j + t];
                        uint64_t e = u + s; x[j] = e >= q ? e - q : e;
                        x[j + t] = shoup_mul64(u >= s ? u - s : u + q - s, W, WS, q);
                    }
                }
                t <<= 1;
            }
            orc_u256 ninv; from_mont(p, &ninv, &p->n_inv_m);
            for (uint32_t j = 0; j < n; j++) r[(size_t)pi * n + j] = u256_from(mulmod64(x[j], ninv.limbs[0], q));
        }
        free(x);
    }
    for (uint32_t l = 0; l < L; l++) free(tw[l]);
    free(tw);
    return used;
}
