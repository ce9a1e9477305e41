// ntt256.hip.h -- FHE_WIDTH_256 kernels: full-width (q < 2^255) negacyclic NTT and element-wise ops.
//
// The general path behind NTTEngine / RNS_NTTEngine for moduli that do not fit the word-sized fast
// paths (and for transform sizes outside 2^11..2^15).  It is built only from the reference's own
// primitives (u256_dev.h: add_mod / sub_mod / mul_mod_montgomery / ct_butterfly / gs_butterfly,
// include/bigint.cuh:27-140, include/ntt.cuh:147-167) with Montgomery-form twiddles, so data stays in
// plain form exactly as in the reference.
//
// Roofline note: one 256-bit Montgomery product is 128 v_mad_u64_u32 + 128 v_addc_co_u32 (hand-scheduled
// mont_mul_fips, u256_dev.h); at ~14 integer MADs per byte of compulsory traffic this path is bound by
// integer issue, not by HBM, so it runs as plain
// multi-pass radix-2^R register kernels over global memory (R <= 3 stages per launch, every access a
// whole 32-byte container, consecutive lanes on consecutive containers) without LDS staging.
#pragma once
#include "u256_dev.h"

namespace fhe_dev {

struct Limb256 {
    u256 q;
    u256 r2;          // R^2 mod q
    u256 ninv_m;      // n^-1 * R mod q
    uint64_t inv0;    // -q^-1 mod 2^64  (MontgomeryParams::inv.limbs[0], include/bigint.cuh:167-173)
    uint64_t _pad;
    const u256 *tw_m;   // [n] psi^bitrev(k) * R mod q
    const u256 *itw_m;  // [n] psi^-bitrev(k) * R mod q
};

// Forward pass: stages s0 .. s0+R-1 (stage s works on index bit b = log_n-1-s, m = 2^s twiddle groups).
// grid = (ceil(n / 2^R / 256), batch*L).
template <int R>
__global__ void __launch_bounds__(256)
ntt256_fwd_pass(u256 *__restrict__ data, const Limb256 *__restrict__ limbs, uint32_t L, uint32_t log_n, uint32_t s0) {
    const uint32_t n = 1u << log_n, u = blockIdx.x * 256 + threadIdx.x;
    if (u >= (n >> R)) return;
    const uint32_t p = blockIdx.y;
    const Limb256 &P = limbs[p % L];
    const u256 q = P.q; const uint64_t inv0 = P.inv0;
    const uint32_t b_last = log_n - s0 - R;                 // index bit of the pass's last stage
    const uint32_t t_last = 1u << b_last;
    const uint32_t i0 = ((u >> b_last) << (b_last + R)) | (u & (t_last - 1));
    u256 *poly = data + (size_t)p * n;
    u256 x[1 << R];
#pragma unroll
    for (int k = 0; k < (1 << R); k++) x[k] = load_u256(poly + i0 + ((uint32_t)k << b_last));
#pragma unroll
    for (int j = 0; j < R; j++) {
        const uint32_t b = b_last + (R - 1 - j), m = 1u << (s0 + j);
#pragma unroll
        for (int hh = 0; hh < (1 << (R - 1)); hh++) {
            const int pos = R - 1 - j;                                   // k-bit handled by this stage
            const int k = ((hh >> pos) << (pos + 1)) | (hh & ((1 << pos) - 1));
            const uint32_t i = i0 + ((uint32_t)k << b_last);
            const u256 w = load_u256(P.tw_m + m + (i >> (b + 1)));
            ct_butterfly_fast(x[k], x[k | (1 << pos)], w, q, (uint32_t)inv0);
        }
    }
#pragma unroll
    for (int k = 0; k < (1 << R); k++) store_u256(poly + i0 + ((uint32_t)k << b_last), x[k]);
}

// Inverse pass: index bits b0 .. b0+R-1 ascending (Gentleman-Sande); the pass that contains bit log_n-1
// also applies the n^-1 scaling (kernels/ntt_kernels.cu:117-120).
template <int R>
__global__ void __launch_bounds__(256)
ntt256_inv_pass(u256 *__restrict__ data, const Limb256 *__restrict__ limbs, uint32_t L, uint32_t log_n, uint32_t b0) {
    const uint32_t n = 1u << log_n, u = blockIdx.x * 256 + threadIdx.x;
    if (u >= (n >> R)) return;
    const uint32_t p = blockIdx.y;
    const Limb256 &P = limbs[p % L];
    const u256 q = P.q; const uint64_t inv0 = P.inv0;
    const uint32_t t0 = 1u << b0;
    const uint32_t i0 = ((u >> b0) << (b0 + R)) | (u & (t0 - 1));
    u256 *poly = data + (size_t)p * n;
    u256 x[1 << R];
#pragma unroll
    for (int k = 0; k < (1 << R); k++) x[k] = load_u256(poly + i0 + ((uint32_t)k << b0));
#pragma unroll
    for (int j = 0; j < R; j++) {
        const uint32_t b = b0 + j, m = n >> (b + 1);
#pragma unroll
        for (int hh = 0; hh < (1 << (R - 1)); hh++) {
            const int k = ((hh >> j) << (j + 1)) | (hh & ((1 << j) - 1));
            const uint32_t i = i0 + ((uint32_t)k << b0);
            const u256 w = load_u256(P.itw_m + m + (i >> (b + 1)));
            gs_butterfly_fast(x[k], x[k | (1 << j)], w, q, (uint32_t)inv0);
        }
    }
    if (b0 + R == log_n) {
        const u256 ninv = P.ninv_m;
#pragma unroll
        for (int k = 0; k < (1 << R); k++) x[k] = mont_mul_fips(x[k], ninv, q, (uint32_t)inv0);
    }
#pragma unroll
    for (int k = 0; k < (1 << R); k++) store_u256(poly + i0 + ((uint32_t)k << b0), x[k]);
}

// Element-wise over [batch][L][n] with per-limb moduli.  OP 0: plain product a*b mod q
// (= mont(mont(a,b), R^2)); 1: add_mod; 2: sub_mod; 3: literal mul_mod_montgomery(a, b) (rns_mul_kernel).
template <int OP>
__global__ void __launch_bounds__(256)
ew256_rns_kernel(u256 *r, const u256 *a, const u256 *b,              // no __restrict__: r may be a or b (in-place add / sub / product)
                 const Limb256 *__restrict__ limbs, uint32_t L, uint32_t log_n, size_t count) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const Limb256 &P = limbs[(uint32_t)((g >> log_n) % L)];
        u256 x = load_u256(a + g), y = load_u256(b + g), o;
        if (OP == 0) o = mont_mul_fips(mont_mul_fips(x, y, P.q, (uint32_t)P.inv0), P.r2, P.q, (uint32_t)P.inv0);
        else if (OP == 1) o = add_mod(x, y, P.q);
        else if (OP == 2) o = sub_mod(x, y, P.q);
        else o = mont_mul(x, y, P.q, P.inv0);          // 3: rns_mul_kernel, literal (src/rns.cu:160-181): carries R^-1
        store_u256(r + g, o);
    }
}

// Literal element-wise primitives with one modulus passed by value:
// batch_mod_add_kernel / batch_mod_sub_kernel / batch_mod_mul_kernel (src/bigint.cu:171-214),
// poly_add_kernel / poly_sub_kernel / poly_mul_scalar_kernel (src/polynomial.cu:70-111),
// ntt_pointwise_mul_kernel (kernels/ntt_kernels.cu:124-137).
// OP 0: mont(a,b); 1: add; 2: sub; 3: mont(a, scalar)
template <int OP>
__global__ void __launch_bounds__(256)
ew256_kernel(u256 *r, const u256 *a, const u256 *b,                  // no __restrict__: callers pass r == a (in-place mul_scalar, add_rns(acc, acc, tmp))
             const u256 q, const u256 scalar, uint64_t inv0, size_t count) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        u256 x = load_u256(a + g), o;
        if (OP == 3) o = mont_mul(x, scalar, q, inv0);
        else {
            u256 y = load_u256(b + g);
            if (OP == 0) o = mont_mul(x, y, q, inv0);
            else if (OP == 1) o = add_mod(x, y, q);
            else o = sub_mod(x, y, q);
        }
        store_u256(r + g, o);
    }
}

__global__ void __launch_bounds__(256)
check256_kernel(const u256 *__restrict__ a, const Limb256 *__restrict__ limbs, uint32_t L, uint32_t log_n,
                size_t count, uint32_t *__restrict__ flag) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t bad = 0;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const u256 q = limbs[(uint32_t)((g >> log_n) % L)].q;
        u256 x = load_u256(a + g), d;
        // x >= q  <=>  x - q does not borrow
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            u128_t t = (u128_t)x.l[i] - q.l[i] - borrow;
            d.l[i] = (uint64_t)t; borrow = (uint64_t)(t >> 64) & 1;
        }
        bad |= (borrow == 0);
    }
    if (bad) atomicOr(flag, 1u);
}

// ---- relinearisation building blocks, full-width path (same layouts as the word-sized kernels) -----------------------
__device__ __forceinline__ uint64_t extract_bits(const u256 &a, uint32_t lo, uint32_t w) {   // bits [lo, lo+w), w <= 64
    if (lo >= 256) return 0;
    const uint32_t limb = lo >> 6, sh = lo & 63;
    uint64_t v = a.l[limb] >> sh;
    if (sh && limb < 3) v |= a.l[limb + 1] << (64 - sh);
    return w >= 64 ? v : (v & ((1ull << w) - 1));
}
__global__ void __launch_bounds__(256)
digit_embed256_kernel(u256 *__restrict__ D, const u256 *__restrict__ c2, const Limb256 *__restrict__ limbs, uint32_t L,
                      uint32_t log_n, uint32_t K, uint32_t w, uint32_t batch) {
    const size_t per_poly = (size_t)1 << log_n, per_ct = per_poly * L, per_digit = per_ct * batch, total = per_digit * L * K;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint32_t jk = (uint32_t)(g / per_digit); const size_t rem = g - (size_t)jk * per_digit;
        const uint32_t b = (uint32_t)(rem / per_ct); const size_t r2 = rem - (size_t)b * per_ct;
        const uint32_t i = (uint32_t)(r2 >> log_n); const size_t x = r2 & (per_poly - 1);
        const uint32_t j = jk / K, k = jk % K;
        uint64_t d = extract_bits(load_u256(c2 + ((size_t)b * L + j) * per_poly + x), k * w, w);
        const u256 q = limbs[i].q;
        if (!(q.l[1] | q.l[2] | q.l[3])) d %= q.l[0];           // moduli above 2^64 exceed every digit
        u256 o; o.l[0] = d; o.l[1] = o.l[2] = o.l[3] = 0;
        store_u256(D + g, o);
    }
}
__global__ void __launch_bounds__(256)
relin_mac256_kernel(u256 *__restrict__ acc0, u256 *__restrict__ acc1, const u256 *__restrict__ D, const u256 *__restrict__ KB,
                    const u256 *__restrict__ KA, const Limb256 *__restrict__ limbs, uint32_t L, uint32_t log_n, uint32_t LK,
                    uint32_t batch) {
    const size_t per_poly = (size_t)1 << log_n, per_ct = per_poly * L, per_digit = per_ct * batch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < per_digit; g += stride) {
        const size_t kidx = g % per_ct;
        const Limb256 &P = limbs[(uint32_t)(kidx >> log_n)];
        u256 s0, s1;
        s0.l[0] = s0.l[1] = s0.l[2] = s0.l[3] = 0; s1 = s0;
        for (uint32_t jk = 0; jk < LK; jk++) {
            const uint32_t qi = (uint32_t)P.inv0;
            const u256 d = mont_mul_fips(load_u256(D + (size_t)jk * per_digit + g), P.r2, P.q, qi);      // d * R
            s0 = add_mod(s0, mont_mul_fips(d, load_u256(KB + (size_t)jk * per_ct + kidx), P.q, qi), P.q);
            s1 = add_mod(s1, mont_mul_fips(d, load_u256(KA + (size_t)jk * per_ct + kidx), P.q, qi), P.q);
        }
        store_u256(acc0 + g, s0);
        store_u256(acc1 + g, s1);
    }
}

// ---- RNS entry / exit (RNS_NTTEngine::to_rns / from_rns, include/ntt.cuh:114-117; src/rns.cu:93-141 are placeholders) ---
// Container-level operations, independent of the width class of the transforms.
struct CrtLimb {
    u256 q, r2;          // modulus, R^2 mod q
    u256 minv_m;         // ((Q/q)^-1 mod q) * R mod q
    u256 Mi_mQ;          // (Q/q) * R mod Q          (Montgomery form with respect to Q)
    uint64_t inv0, _pad;
};
struct CrtBig { u256 Q; uint64_t inv0, _pad; };
struct RescaleLimb { u256 qlast_inv_m; };   // (q_last^-1 mod q_l) * R mod q_l

// rns[b][l][x] = values[b][x] mod q_l : mont(mont(v, R^2), 1) is exact for ANY 256-bit v (the sum before the final
// subtraction is below 2q).  One lane per (b, x); the L residues are produced from one load of the value.
__global__ void __launch_bounds__(256)
to_rns_kernel(u256 *__restrict__ rns, const u256 *__restrict__ values, const CrtLimb *__restrict__ limbs, uint32_t L, uint32_t log_n, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    u256 one; one.l[0] = 1; one.l[1] = one.l[2] = one.l[3] = 0;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const u256 v = load_u256(values + g);
        const size_t b = g >> log_n, x = g & (n - 1);
        for (uint32_t l = 0; l < L; l++) {
            const CrtLimb &P = limbs[l];
            store_u256(rns + (b * L + l) * n + x, mont_mul(mont_mul(v, P.r2, P.q, P.inv0), one, P.q, P.inv0));
        }
    }
}
// values[b][x] = sum_l [r_l * (Q/q_l)^-1]_{q_l} * (Q/q_l) mod Q, accumulated with the 256-bit Montgomery primitives modulo Q.
__global__ void __launch_bounds__(256)
from_rns_kernel(u256 *__restrict__ values, const u256 *__restrict__ rns, const CrtLimb *__restrict__ limbs, const CrtBig big, uint32_t L,
                uint32_t log_n, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const size_t b = g >> log_n, x = g & (n - 1);
        u256 acc; acc.l[0] = acc.l[1] = acc.l[2] = acc.l[3] = 0;
        for (uint32_t l = 0; l < L; l++) {
            const CrtLimb &P = limbs[l];
            const u256 t = mont_mul(load_u256(rns + (b * L + l) * n + x), P.minv_m, P.q, P.inv0);
            acc = add_mod(acc, mont_mul(t, P.Mi_mQ, big.Q, big.inv0), big.Q);
        }
        store_u256(values + g, acc);
    }
}

// Modulus switching by dropping the last prime (rns_mod_switch_kernel, include/rns.cuh:128-136, undefined in the reference):
// out[b][l][x] = (c[b][l][x] - r) * q_last^-1 mod q_l with r the centred residue modulo q_last, i.e. round(C / q_last) limb-wise.
// One lane per (b, x): the last limb is read once and all L-1 outputs are produced from it.
__global__ void __launch_bounds__(256)
rescale_drop_last_kernel(u256 *__restrict__ out, const u256 *__restrict__ in, const CrtLimb *__restrict__ limbs,
                         const RescaleLimb *__restrict__ rs, uint32_t L, uint32_t log_n, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    u256 one; one.l[0] = 1; one.l[1] = one.l[2] = one.l[3] = 0;
    const u256 ql = limbs[L - 1].q;
    u256 half;                                                       // floor(q_last / 2)
#pragma unroll
    for (int i = 0; i < 4; i++) half.l[i] = (ql.l[i] >> 1) | (i < 3 ? ql.l[i + 1] << 63 : 0);
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const size_t b = g >> log_n, x = g & (n - 1);
        const u256 cl = load_u256(in + (b * L + (L - 1)) * n + x);
        u256 d; sub256(d, half, cl);                                 // borrow <=> cl > half
        bool neg = false;
#pragma unroll
        for (int i = 3; i >= 0; i--) { if (cl.l[i] != half.l[i]) { neg = cl.l[i] > half.l[i]; break; } }
        u256 mag;
        if (neg) sub256(mag, ql, cl); else mag = cl;
        for (uint32_t l = 0; l + 1 < L; l++) {
            const CrtLimb &P = limbs[l];
            u256 r = mont_mul(mont_mul(mag, P.r2, P.q, P.inv0), one, P.q, P.inv0);          // |r| mod q_l
            if (neg && (r.l[0] | r.l[1] | r.l[2] | r.l[3])) { u256 z; sub256(z, P.q, r); r = z; }
            const u256 diff = sub_mod(load_u256(in + (b * L + l) * n + x), r, P.q);
            store_u256(out + (b * (L - 1) + l) * n + x, mont_mul(diff, rs[l].qlast_inv_m, P.q, P.inv0));
        }
    }
}

// Fast base conversion (Bajard et al.; fast_base_conversion_kernel, include/rns.cuh:116-125, undefined in the reference):
// out[b][j][x] = sum_i [x_i * (Q/q_i)^-1]_{q_i} * (Q/q_i) mod p_j.  `mat` holds ((Q/q_i) mod p_j) * R_j, row-major [L][Lp].
// One lane per (b, x): the L scaled residues t_i are formed once and reused for every target prime.
constexpr int BASE_CONV_MAX_L = 16;
__global__ void __launch_bounds__(256)
fast_base_convert_kernel(u256 *__restrict__ out, const u256 *__restrict__ in, const CrtLimb *__restrict__ src, uint32_t L,
                         const CrtLimb *__restrict__ dst, uint32_t Lp, const u256 *__restrict__ mat, uint32_t log_n, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    u256 one; one.l[0] = 1; one.l[1] = one.l[2] = one.l[3] = 0;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const size_t b = g >> log_n, x = g & (n - 1);
        for (uint32_t j = 0; j < Lp; j++) {
            const CrtLimb &D = dst[j];
            u256 acc; acc.l[0] = acc.l[1] = acc.l[2] = acc.l[3] = 0;
            for (uint32_t i = 0; i < L; i++) {
                const CrtLimb &S = src[i];
                const u256 ti = mont_mul(load_u256(in + (b * L + i) * n + x), S.minv_m, S.q, S.inv0);      // [x_i * M_i^-1]_{q_i}
                const u256 ti_m = mont_mul(ti, D.r2, D.q, D.inv0);                                       // (t_i mod p_j) * R_j
                const u256 term = mont_mul(mont_mul(ti_m, load_u256(mat + (size_t)i * Lp + j), D.q, D.inv0), one, D.q, D.inv0);
                acc = add_mod(acc, term, D.q);
            }
            store_u256(out + (b * Lp + j) * n + x, acc);
        }
    }
}

// ---- the reference's transform kernels AS WRITTEN (L1 parity) ---------------------------------------------------------
// ntt_forward_optimized_kernel / ntt_inverse_optimized_kernel (kernels/ntt_kernels.cu:7-62, :65-121), launched by
// NTTEngine::forward / inverse as ONE block of n threads (src/ntt.cu:30-47): stage schedule with log_n = popc(n-1)+1 (sic),
// butterfly pairs (k*2m + j, k*2m + j + m) only where the second index < block size (= n), twiddle index j << (log_n-stage-1),
// caller-supplied tables (the reference fills them with placeholders, src/ntt.cu:86-97), literal mul_mod_montgomery /
// add_mod / sub_mod.  The n "threads" of the reference's block are walked by the lanes of one workgroup (within a stage every
// thread owns a private pair, so the order inside a stage is irrelevant); the data stay in device memory instead of the
// reference's dynamic shared memory (n * 32 bytes: beyond any LDS for n > 4096), which changes nothing observable.
// One workgroup per polynomial of a [batch][n] buffer.  bit_reverse_kernel is not applied (out-of-bounds accesses there make
// its result undefined); these kernels are what the reference's own source computes on the data it is given.
__global__ void __launch_bounds__(256)
ref_forward_literal_kernel(u256 *__restrict__ data, const u256 *__restrict__ tw, u256 q, uint64_t inv0, uint32_t n) {
    u256 *d = data + (size_t)blockIdx.x * n;
    const uint32_t log_n = (uint32_t)__popc(n - 1) + 1;
    for (uint32_t stage = 0; stage < log_n; stage++) {
        const uint32_t m = 1u << stage, m2 = m << 1;
        for (uint32_t tid = threadIdx.x; tid < n; tid += blockDim.x) {
            const uint32_t k = tid / m, j = tid % m;
            if ((uint64_t)k * m2 + j + m < n) {
                const uint32_t idx1 = k * m2 + j, idx2 = idx1 + m;
                const u256 u = load_u256(d + idx1);
                const u256 v = mont_mul(load_u256(d + idx2), load_u256(tw + (j << (log_n - stage - 1))), q, inv0);
                store_u256(d + idx1, add_mod(u, v, q));
                store_u256(d + idx2, sub_mod(u, v, q));
            }
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256)
ref_inverse_literal_kernel(u256 *__restrict__ data, const u256 *__restrict__ itw, u256 q, uint64_t inv0, u256 n_inv, uint32_t n) {
    u256 *d = data + (size_t)blockIdx.x * n;
    const uint32_t log_n = (uint32_t)__popc(n - 1) + 1;
    for (int stage = (int)log_n - 1; stage >= 0; stage--) {
        const uint32_t m = 1u << stage, m2 = m << 1;
        for (uint32_t tid = threadIdx.x; tid < n; tid += blockDim.x) {
            const uint32_t k = tid / m, j = tid % m;
            if ((uint64_t)k * m2 + j + m < n) {
                const uint32_t idx1 = k * m2 + j, idx2 = idx1 + m;
                const u256 u = load_u256(d + idx1), v = load_u256(d + idx2);
                store_u256(d + idx1, add_mod(u, v, q));
                store_u256(d + idx2, mont_mul(sub_mod(u, v, q), load_u256(itw + (j << (log_n - (uint32_t)stage - 1))), q, inv0));
            }
        }
        __syncthreads();
    }
    for (uint32_t tid = threadIdx.x; tid < n; tid += blockDim.x) store_u256(d + tid, mont_mul(load_u256(d + tid), n_inv, q, inv0));
}

// ntt_stockham_kernel (kernels/ntt_kernels.cu:213-243; never launched by the reference): ONE out-of-place butterfly stage,
// output[idx1] = input[idx1] + mont(input[idx2], tw[j * (n / 2m)]), output[idx2] = input[idx1] - ..., idx1 = k*2m + j, idx2 = idx1 + m.
// As written the kernel runs idx over [0, n) and indexes past the arrays for idx >= n/2 (undefined); this restatement runs the
// n/2 in-bounds butterflies, which are all of a stage.  One lane per butterfly, [batch][n] polynomials.
__global__ void __launch_bounds__(256)
ref_stockham_stage_kernel(u256 *__restrict__ output, const u256 *__restrict__ input, const u256 *__restrict__ tw, u256 q, uint64_t inv0,
                          uint32_t n, uint32_t stage, size_t count /* batch * n/2 */) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint32_t m = 1u << stage, m2 = m << 1, half = n >> 1;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const size_t b = g / half; const uint32_t idx = (uint32_t)(g % half);
        const uint32_t k = idx / m, j = idx % m, idx1 = k * m2 + j, idx2 = idx1 + m;
        const u256 *in = input + b * n; u256 *out = output + b * n;
        const u256 u = load_u256(in + idx1);
        const u256 v = mont_mul(load_u256(in + idx2), load_u256(tw + j * (n / m2)), q, inv0);
        store_u256(out + idx1, add_mod(u, v, q));
        store_u256(out + idx2, sub_mod(u, v, q));
    }
}

// bit_reverse_kernel's intent (kernels/ntt_kernels.cu:140-161): in-place bit-reversal permutation of each polynomial, swapping
// only where idx < rev(idx).  The reference reverses over popc(n-1)+1 = log2(n)+1 bits, which sends half of the indices past
// the array (undefined); this kernel reverses over log2(n) bits.  It converts between natural order and the order
// fhe_ntt_forward leaves its values in (X[k] sits at position bitrev(k)).
__global__ void __launch_bounds__(256)
bit_reverse_kernel(u256 *__restrict__ data, uint32_t log_n, size_t count /* batch * n */) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint32_t n = 1u << log_n;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const uint32_t idx = (uint32_t)(g & (n - 1));
        const uint32_t rev = __brev(idx) >> (32 - log_n);
        if (idx < rev) {
            u256 *d = data + (g - idx);
            const u256 a = load_u256(d + idx), b = load_u256(d + rev);
            store_u256(d + idx, b);
            store_u256(d + rev, a);
        }
    }
}

// (X^shift[b] - 1) * p on full-width containers (see monomial_mul_sub_kernel in ntt_lds.hip.h)
__global__ void __launch_bounds__(256)
monomial_mul_sub256_kernel(u256 *__restrict__ out, const u256 *__restrict__ in, const uint32_t *__restrict__ shifts,
                           const Limb256 *__restrict__ limbs, uint32_t L, uint32_t log_n, size_t count) {
    const uint32_t n = 1u << log_n;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const size_t poly = g >> log_n;
        const uint32_t x = (uint32_t)(g & (n - 1)), a = shifts[poly / L] & (2 * n - 1);
        uint32_t k = (x + 2 * n - a) & (2 * n - 1);
        const bool neg = k >= n; k &= n - 1;
        const u256 q = limbs[(uint32_t)(poly % L)].q;
        u256 v = load_u256(in + (poly << log_n) + k), zero;
        zero.l[0] = zero.l[1] = zero.l[2] = zero.l[3] = 0;
        if (neg) v = sub_mod(zero, v, q);
        store_u256(out + g, sub_mod(v, load_u256(in + g), q));
    }
}

}  // namespace fhe_dev
