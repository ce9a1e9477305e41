// ntt_wide.hip.h -- LDS-staged negacyclic NTT / polymul kernels for moduli wider than one machine word (FHE_WIDTH_256).
//
// Replaces ntt_forward_optimized_kernel / ntt_inverse_optimized_kernel / ntt_pointwise_mul_kernel (kernels/ntt_kernels.cu:7-137:
// shared-memory staging of 32-byte elements, one butterfly stage per barrier) and the multi-limb Montgomery product built on the
// PTX carry chains (include/bigint.cuh:76-140, kernels/ptx_bigint.cuh:34-117) for q up to 2^255.
//
//   * NL = number of 64-bit limbs the modulus needs: 2 (q < 2^127) or 4 (q < 2^255).  The Montgomery radix is R = 2^(64 NL);
//     twiddles are stored as w * R mod q, so mont(x, w R) on plain-form data is the exact product and every value stays the
//     canonical residue the reference's R = 2^256 primitives would produce (upper container words zero for NL = 2).
//   * Tile = 2^11 consecutive coefficients, one workgroup of 256 threads, 8 coefficients per thread (64 data VGPRs at NL = 4).
//     Eleven butterfly stages run per HBM round trip: three register groups of 3 stages + one of 2, exchanged through LDS.
//     The LDS image is limb-planar (plane w holds 32-bit word w of every coefficient) and XOR-swizzled so that all four
//     exchange patterns are bank-conflict-free for the 32-lane groups of ds_read_b32 / ds_write_b32 (see swz()).
//   * N = 2^11: one launch per transform, HBM traffic 2 S.  N > 2^11: the top log2(N) - 11 stages run first (forward) / last
//     (inverse) as a register-only radix-2^R pass over global memory (wide_pass_kernel, R <= 3 per launch), so N = 8192 costs
//     4 S per transform instead of the 10 S of five un-staged passes.
//   * Fused multiply (wide_tile_kernel<NL, TILE_MUL>): forward(a tile), forward(b tile), pointwise Montgomery product, inverse
//     stages, one store -- no operand copies, no NTT-domain round trip: 3 S at N = 2^11, 9 S at N = 8192 (was 37 S).
//   * The Montgomery product is finely integrated product scanning over 32-bit words on v_mad_u64_u32 with the carry-outs in
//     SGPR pairs (mac1 / mac2 / mac3 in u256_dev.h): 2 NW^2 + NW multiplies for NW = 2 NL words.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "u256_dev.h"

namespace fhe_dev {

typedef uint32_t v4u32w __attribute__((ext_vector_type(4)));

template <int NL> struct __attribute__((aligned(16))) wint { uint32_t w[2 * NL]; };   // 2 NL little-endian 32-bit words

#include "wide_asm.inc"

// Per-limb constants of the wide NTT kernels (device memory, one entry per RNS prime).
template <int NL>
struct WLimb {
    wint<NL> q;
    wint<NL> ninv_m;      // n^-1 * R mod q        (stand-alone inverse)
    wint<NL> ninv_r2;     // n^-1 * R^2 mod q      (fused multiply: absorbs the R^-1 of the pointwise product)
    wint<NL> r2;          // R^2 mod q
    uint32_t qinv32, _pad[3];   // -q^-1 mod 2^32
    const wint<NL> *tw;   // [n] psi^bitrev(k) * R mod q
    const wint<NL> *itw;  // [n] psi^-bitrev(k) * R mod q
};

// a * b * R^-1 mod q, canonical, for a, b < q (odd, < 2^(64 NL - 1)), R = 2^(64 NL); qinv32 = -q^-1 mod 2^32.  The value
// mul_mod_montgomery (include/bigint.cuh:76-140) returns with that radix.  Finely integrated product scanning over 32-bit
// words: column k sums a_i b_(k-i) and m_i q_(k-i) in one asm block (wide_asm.inc: macn), m_k = column * qinv clears its low
// word, and the words of the result are compared against q as they become final (macn_e), so the closing conditional
// subtraction is two more steps and a select (wsel).
template <int NL, int K>
__device__ __forceinline__ void wmont_column(uint64_t &lo, uint32_t &hi, const uint32_t (&a)[2 * NL], const uint32_t (&b)[2 * NL], const uint32_t (&q)[2 * NL],
                                             uint32_t (&m)[2 * NL], uint32_t (&t)[2 * NL], uint32_t (&tq)[2 * NL], uint64_t &bor, uint32_t qinv32) {
    constexpr int NW = 2 * NL;
    constexpr int CNT_AB = (K < NW ? K + 1 : 2 * NW - 1 - K), CNT_MQ = (K < NW ? K : 2 * NW - 1 - K), CNT = CNT_AB + CNT_MQ;
    uint32_t xs[16], ys[16];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) { const int j = K - i; if (j >= 0 && j < NW) { xs[cnt] = a[i]; ys[cnt] = b[j]; cnt++; } }
#pragma unroll
    for (int i = 0; i < NW; i++) { const int j = K - i; if (j >= 0 && j < NW && i < K) { xs[cnt] = m[i]; ys[cnt] = q[j]; cnt++; } }
    if constexpr (K == NW + 1) macn_e0<CNT>(lo, hi, bor, tq[0], t[0], q[0], xs, ys);
    else if constexpr (K > NW + 1) macn_e<CNT>(lo, hi, bor, tq[K - NW - 1], t[K - NW - 1], q[K - NW - 1], xs, ys);
    else macn<CNT>(lo, hi, xs, ys);
    if constexpr (K < NW) {
        m[K] = (uint32_t)lo * qinv32;
        macn_1(lo, hi, m[K], q[0]);                 // clears the low word of the column
    } else {
        t[K - NW] = (uint32_t)lo;
    }
    lo = (lo >> 32) | ((uint64_t)hi << 32);
    hi = 0;
    if constexpr (K + 1 < 2 * NW - 1) wmont_column<NL, K + 1>(lo, hi, a, b, q, m, t, tq, bor, qinv32);
}
template <int NL>
__device__ __forceinline__ wint<NL> wmont(const wint<NL> &a, const wint<NL> &b, const wint<NL> &q, uint32_t qinv32) {
    constexpr int NW = 2 * NL;
    wint<NL> t;
#ifndef FHE_WIDE_COLUMN_BLOCKS      // round 3: the whole product as ONE asm block (wide_asm.inc: wmontc); -DFHE_WIDE_COLUMN_BLOCKS restores one block per column
    if constexpr (NL == 4) wmontc_8(t.w, a.w, b.w, q.w, qinv32); else wmontc_4(t.w, a.w, b.w, q.w, qinv32);
    return t;
#endif
    uint32_t m[NW], tq[NW];
    uint64_t lo = 0, bor = 0; uint32_t hi = 0;
    wmont_column<NL, 0>(lo, hi, a.w, b.w, q.w, m, t.w, tq, bor, qinv32);
    t.w[NW - 1] = (uint32_t)lo;                      // the sum is below 2q < 2^(32 NW): nothing above this word
    if constexpr (NL == 4) wsel_8(t.w, tq, q.w, bor); else wsel_4(t.w, tq, q.w, bor);
    return t;
}
template <int NL> __device__ __forceinline__ void waddsub(wint<NL> &a, wint<NL> &t, const wint<NL> &q) {
    if constexpr (NL == 4) waddsub_8(a.w, t.w, q.w); else waddsub_4(a.w, t.w, q.w);
}

// ct_butterfly / gs_butterfly (include/ntt.cuh:147-167) on the radix-R product: (a, b) <- (a + b w, a - b w) / (a + b, (a - b) w)
template <int NL> __device__ __forceinline__ void wct(wint<NL> &a, wint<NL> &b, const wint<NL> &w, const wint<NL> &q, uint32_t qi) {
    b = wmont<NL>(b, w, q, qi);
    waddsub<NL>(a, b, q);
}
template <int NL> __device__ __forceinline__ void wgs(wint<NL> &a, wint<NL> &b, const wint<NL> &w, const wint<NL> &q, uint32_t qi) {
    waddsub<NL>(a, b, q);
    b = wmont<NL>(b, w, q, qi);
}
// ---- the lazy class (round 3): q < 2^(64 NL - 6) ----------------------------------------------------------------------------------------
// Six spare bits let a tile's forward transform run WITHOUT any reduction: with inputs below B q, T = b w R^-1 needs no closing subtraction
// (T < q (1 + B q / R) < 2q), and (a, b) <- (a + T, a - T + 2q) stays below (B + 2) q: canonical input (B = 1), eleven stages -> below 23 q
// < 2^(64 NL - 1).  Per forward butterfly that is 284 + 24 instructions instead of 302 + 48 (NL = 4).  The inverse (Gentleman-Sande) butterfly
// doubles its sum, so it keeps one conditional subtraction per output, of 2q (values below 2q; the same waddsub block with 2q for q), and the
// product without the closing subtraction.  Before anything leaves the kernel it is canonical again: the final scaling is a canonical Montgomery
// product (input below 2q), and where there is none a ladder of conditional subtractions (16q, 8q, 4q, 2q, q) follows.
template <int NL> __device__ __forceinline__ wint<NL> wmontl(const wint<NL> &a, const wint<NL> &b, const wint<NL> &q, uint32_t qinv32) {
    wint<NL> t;
    if constexpr (NL == 4) wmontl_8(t.w, a.w, b.w, q.w, qinv32); else wmontl_4(t.w, a.w, b.w, q.w, qinv32);
    return t;
}
template <int NL> __device__ __forceinline__ wint<NL> wshl(const wint<NL> &q, int s) {     // q << s, 0 <= s < 32 (the caller knows it fits)
    wint<NL> r;
#pragma unroll
    for (int i = 2 * NL - 1; i >= 0; i--) r.w[i] = s ? (q.w[i] << s) | (i ? q.w[i - 1] >> (32 - s) : 0u) : q.w[i];
    return r;
}
// x[k] < 2^STEPS q  ->  x[k] < q: conditional subtractions of 2^(STEPS-1) q ... 2^LAST q (LAST = 0: canonical; LAST = 1: below 2q)
template <int NL, int STEPS, int LAST = 0> __device__ __forceinline__ void wreduce(wint<NL> (&x)[8], const wint<NL> &q) {
#pragma unroll
    for (int s = STEPS - 1; s >= LAST; s--) {
        const wint<NL> c = wshl<NL>(q, s);
        if constexpr (NL == 4) { wcsub4_8(x[0].w, x[1].w, x[2].w, x[3].w, c.w); wcsub4_8(x[4].w, x[5].w, x[6].w, x[7].w, c.w); }
        else { wcsub4_4(x[0].w, x[1].w, x[2].w, x[3].w, c.w); wcsub4_4(x[4].w, x[5].w, x[6].w, x[7].w, c.w); }
    }
}
// forward: (a, b) <- (a + b w, a - b w + 2q), nothing reduced;  inverse: (a, b) <- (a + b mod 2q, (a - b mod 2q) w), inputs and outputs below 2q
template <int NL> __device__ __forceinline__ void wct_l(wint<NL> &a, wint<NL> &b, const wint<NL> &w, const wint<NL> &q, const wint<NL> &q2, uint32_t qi) {
    b = wmontl<NL>(b, w, q, qi);
    if constexpr (NL == 4) wfree_8(a.w, b.w, q2.w); else wfree_4(a.w, b.w, q2.w);
}
template <int NL> __device__ __forceinline__ void wgs_l(wint<NL> &a, wint<NL> &b, const wint<NL> &w, const wint<NL> &q, const wint<NL> &q2, uint32_t qi) {
    waddsub<NL>(a, b, q2);
    b = wmontl<NL>(b, w, q, qi);
}

// a + b mod q alone (tensor product): the butterfly tail on copies
template <int NL> __device__ __forceinline__ wint<NL> waddmod(wint<NL> a, wint<NL> b, const wint<NL> &q) { waddsub<NL>(a, b, q); return a; }

// 32-byte container <-> registers.  NL = 2 reads the low 16 bytes (the upper words of a canonical residue are zero) and
// writes them back as zeros.
template <int NL> __device__ __forceinline__ wint<NL> wload_c(const u256 *p) {
    const v4u32w *v = reinterpret_cast<const v4u32w *>(p);
    wint<NL> r;
    const v4u32w lo = v[0]; r.w[0] = lo.x; r.w[1] = lo.y; r.w[2] = lo.z; r.w[3] = lo.w;
    if constexpr (NL == 4) { const v4u32w hi = v[1]; r.w[4] = hi.x; r.w[5] = hi.y; r.w[6] = hi.z; r.w[7] = hi.w; }
    return r;
}
template <int NL> __device__ __forceinline__ void wstore_c(u256 *p, const wint<NL> &x) {
    v4u32w *v = reinterpret_cast<v4u32w *>(p);
    v4u32w lo = {x.w[0], x.w[1], x.w[2], x.w[3]};
    v[0] = lo;
    if constexpr (NL == 4) { v4u32w hi = {x.w[4], x.w[5], x.w[6], x.w[7]}; v[1] = hi; }
    else { v4u32w z = {0u, 0u, 0u, 0u}; v[1] = z; }
}
// Table entry (16-byte aligned).  The table pointers live in WLimb (device memory), so the compiler only knows them as generic pointers
// and would emit flat_load (lgkmcnt + vmcnt, 64-bit VGPR address, aperture check); they are global memory: say so (as load_global does
// for the word-sized fields, where the same change was worth +8 ... +45 % on the key-switch kernels of the 8-byte fields).
template <int NL> __device__ __forceinline__ wint<NL> wload_t(const wint<NL> *p) {
    const __attribute__((address_space(1))) v4u32w *v = (const __attribute__((address_space(1))) v4u32w *)p;
    wint<NL> r;
#pragma unroll
    for (int i = 0; i < NL / 2; i++) { const v4u32w t = v[i]; r.w[4 * i] = t.x; r.w[4 * i + 1] = t.y; r.w[4 * i + 2] = t.z; r.w[4 * i + 3] = t.w; }
    return r;
}

// ---- global-memory pass over the top stages ---------------------------------------------------------------------------
// FWD: stages s0 .. s0+R-1 (stage s works on index bit log_n-1-s).  INV: index bits b0 .. b0+R-1 ascending; `scale` != 0 multiplies
// the outputs by n^-1 R (1) or n^-1 R^2 (2) -- only the pass that contains bit log_n-1 is launched with it.
// src may differ from dst (the fused multiply transforms its operands into the workspace instead of copying them first).
// grid = (ceil(n / 2^R / 256), polys).
template <int NL, int R, bool FWD>
__global__ void __launch_bounds__(256)
wide_pass_kernel(u256 *dst, const u256 *src, const WLimb<NL> *__restrict__ limbs, uint32_t L, uint32_t log_n, uint32_t s0, uint32_t scale) {
    const uint32_t n = 1u << log_n, u = blockIdx.x * 256 + threadIdx.x;
    if (u >= (n >> R)) return;
    const uint32_t p = blockIdx.y;
    const WLimb<NL> &P = limbs[p % L];
    const wint<NL> q = P.q; const uint32_t qi = P.qinv32;
    const uint32_t b_lo = FWD ? log_n - s0 - R : s0;               // lowest index bit of the pass
    const uint32_t i0 = ((u >> b_lo) << (b_lo + R)) | (u & ((1u << b_lo) - 1));
    const u256 *in = src + (size_t)p * n; u256 *out = dst + (size_t)p * n;
    wint<NL> x[1 << R];
#pragma unroll
    for (int k = 0; k < (1 << R); k++) x[k] = wload_c<NL>(in + i0 + ((uint32_t)k << b_lo));
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int pos = FWD ? R - 1 - j : j;                        // k-bit handled by this stage
        const uint32_t b = b_lo + pos;
#pragma unroll
        for (int hh = 0; hh < (1 << (R - 1)); hh++) {
            const int k = ((hh >> pos) << (pos + 1)) | (hh & ((1 << pos) - 1));
            const uint32_t i = i0 + ((uint32_t)k << b_lo);
            const wint<NL> w = wload_t<NL>((FWD ? P.tw : P.itw) + (n >> (b + 1)) + (i >> (b + 1)));
            if (FWD) wct<NL>(x[k], x[k | (1 << pos)], w, q, qi);
            else wgs<NL>(x[k], x[k | (1 << pos)], w, q, qi);
        }
    }
    if (!FWD && scale) {
        const wint<NL> c = scale == 2 ? P.ninv_r2 : P.ninv_m;
#pragma unroll
        for (int k = 0; k < (1 << R); k++) x[k] = wmont<NL>(x[k], c, q, qi);
    }
#pragma unroll
    for (int k = 0; k < (1 << R); k++) wstore_c<NL>(out + i0 + ((uint32_t)k << b_lo), x[k]);
}

// ---- LDS tile: 2^11 coefficients, 256 threads x 8 ------------------------------------------------------------------------
constexpr int WT_LOG = 11, WT_N = 1 << WT_LOG, WT_T = WT_N / 8;

// XOR swizzle of the tile index (a permutation inside every aligned block of 32 slots, so the image needs no padding):
// index bits 5, 6, 7 are folded into the bank bits as 5 -> {2}, 6 -> {0, 3}, 7 -> {1, 4}.  For each exchange pattern the five
// index bits that vary over a 32-lane group then map to five independent bank bits:
//   pattern B0 >= 5 : bits 0..4                       -> identity
//   pattern B0 = 2  : bits 0, 1, 5, 6, 7              -> e0, e1, e2, e0+e3, e1+e4
//   pattern B0 = 0  : bits 3, 4, 5, 6, 7              -> e3, e4, e2, e0+e3, e1+e4
// (checked by SQ_LDS_BANK_CONFLICT = 0 in profiles/).  swz is linear over GF(2): swz(x ^ y) = swz(x) ^ swz(y).
__device__ __host__ constexpr uint32_t swz(uint32_t i) {
    return i ^ (((i >> 5) & 1u) << 2) ^ (((i >> 6) & 1u) * 9u) ^ (((i >> 7) & 1u) * 18u);
}
// pattern B0: register r <-> tile-index bits [B0, B0+3)
template <int B0> __device__ __forceinline__ uint32_t wt_base(uint32_t tid) { return ((tid >> B0) << (B0 + 3)) | (tid & ((1u << B0) - 1)); }

template <int NL, int B0>
__device__ __forceinline__ void wt_put(uint32_t *lds, uint32_t tid, const wint<NL> (&x)[8]) {
    const uint32_t pb = swz(wt_base<B0>(tid));
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint32_t *p = lds + (pb ^ swz((uint32_t)r << B0));
#pragma unroll
        for (int w = 0; w < 2 * NL; w++) p[w * WT_N] = x[r].w[w];
    }
}
template <int NL, int B0>
__device__ __forceinline__ void wt_get(const uint32_t *lds, uint32_t tid, wint<NL> (&x)[8]) {
    const uint32_t pb = swz(wt_base<B0>(tid));
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t *p = lds + (pb ^ swz((uint32_t)r << B0));
#pragma unroll
        for (int w = 0; w < 2 * NL; w++) x[r].w[w] = p[w * WT_N];
    }
}

// One register group: r-bits KHI .. KLO of pattern B0 (tile-index bits B0+KHI .. B0+KLO), forward order (descending).
// gbase = index of the tile's first coefficient inside its polynomial.
template <int NL, int B0, int K, bool LZ>
__device__ __forceinline__ void wt_fwd_stage(wint<NL> (&x)[8], uint32_t ibase, uint32_t log_n, const WLimb<NL> &P, const wint<NL> &q, const wint<NL> &q2) {
    constexpr int b = B0 + K;
    const wint<NL> *tw = P.tw + ((1u << (log_n - 1 - b)) + (ibase >> (b + 1)));
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (r & (1 << K)) continue;
        const wint<NL> w = wload_t<NL>(tw + (r >> (K + 1)));
        if constexpr (LZ) wct_l<NL>(x[r], x[r | (1 << K)], w, q, q2, P.qinv32);
        else wct<NL>(x[r], x[r | (1 << K)], w, q, P.qinv32);
    }
}
template <int NL, int B0, int K, bool LZ>
__device__ __forceinline__ void wt_inv_stage(wint<NL> (&x)[8], uint32_t ibase, uint32_t log_n, const WLimb<NL> &P, const wint<NL> &q, const wint<NL> &q2) {
    constexpr int b = B0 + K;
    const wint<NL> *tw = P.itw + ((1u << (log_n - 1 - b)) + (ibase >> (b + 1)));
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (r & (1 << K)) continue;
        const wint<NL> w = wload_t<NL>(tw + (r >> (K + 1)));
        if constexpr (LZ) wgs_l<NL>(x[r], x[r | (1 << K)], w, q, q2, P.qinv32);
        else wgs<NL>(x[r], x[r | (1 << K)], w, q, P.qinv32);
    }
}
template <int NL, int B0, int KHI, int KLO, bool LZ>
__device__ __forceinline__ void wt_fwd_group(wint<NL> (&x)[8], uint32_t tid, uint32_t gbase, uint32_t log_n, const WLimb<NL> &P, const wint<NL> &q, const wint<NL> &q2) {
    const uint32_t ibase = gbase + wt_base<B0>(tid);
    wt_fwd_stage<NL, B0, KHI, LZ>(x, ibase, log_n, P, q, q2);
    if constexpr (KHI - 1 >= KLO) wt_fwd_stage<NL, B0, KHI - 1, LZ>(x, ibase, log_n, P, q, q2);
    if constexpr (KHI - 2 >= KLO) wt_fwd_stage<NL, B0, KHI - 2, LZ>(x, ibase, log_n, P, q, q2);
}
template <int NL, int B0, int KLO, int KHI, bool LZ>
__device__ __forceinline__ void wt_inv_group(wint<NL> (&x)[8], uint32_t tid, uint32_t gbase, uint32_t log_n, const WLimb<NL> &P, const wint<NL> &q, const wint<NL> &q2) {
    const uint32_t ibase = gbase + wt_base<B0>(tid);
    wt_inv_stage<NL, B0, KLO, LZ>(x, ibase, log_n, P, q, q2);
    if constexpr (KLO + 1 <= KHI) wt_inv_stage<NL, B0, KLO + 1, LZ>(x, ibase, log_n, P, q, q2);
    if constexpr (KLO + 2 <= KHI) wt_inv_stage<NL, B0, KLO + 2, LZ>(x, ibase, log_n, P, q, q2);
}

// the low 11 stages of a forward transform: coefficients in pattern 8 (coalesced order) -> values in pattern 0 (8 consecutive per thread)
template <int NL, bool LZ>
__device__ __forceinline__ void wt_forward(wint<NL> (&x)[8], uint32_t *lds, uint32_t tid, uint32_t gbase, uint32_t log_n, const WLimb<NL> &P, const wint<NL> &q, const wint<NL> &q2) {
    wt_fwd_group<NL, 8, 2, 0, LZ>(x, tid, gbase, log_n, P, q, q2);          // tile bits 10, 9, 8
    wt_put<NL, 8>(lds, tid, x);
    __syncthreads();
    wt_get<NL, 5>(lds, tid, x);
    wt_fwd_group<NL, 5, 2, 0, LZ>(x, tid, gbase, log_n, P, q, q2);          // 7, 6, 5
    wt_put<NL, 5>(lds, tid, x);                                     // the slots this thread just read: no barrier needed before
    __syncthreads();
    wt_get<NL, 2>(lds, tid, x);
    wt_fwd_group<NL, 2, 2, 0, LZ>(x, tid, gbase, log_n, P, q, q2);          // 4, 3, 2
    wt_put<NL, 2>(lds, tid, x);
    __syncthreads();
    wt_get<NL, 0>(lds, tid, x);
    wt_fwd_group<NL, 0, 1, 0, LZ>(x, tid, gbase, log_n, P, q, q2);          // 1, 0
}
// the low 11 stages of an inverse transform: values in pattern 0 -> coefficients in pattern 8
template <int NL, bool LZ>
__device__ __forceinline__ void wt_inverse(wint<NL> (&x)[8], uint32_t *lds, uint32_t tid, uint32_t gbase, uint32_t log_n, const WLimb<NL> &P, const wint<NL> &q, const wint<NL> &q2) {
    wt_inv_group<NL, 0, 0, 1, LZ>(x, tid, gbase, log_n, P, q, q2);          // tile bits 0, 1
    wt_put<NL, 0>(lds, tid, x);
    __syncthreads();
    wt_get<NL, 2>(lds, tid, x);
    wt_inv_group<NL, 2, 0, 2, LZ>(x, tid, gbase, log_n, P, q, q2);          // 2, 3, 4
    wt_put<NL, 2>(lds, tid, x);
    __syncthreads();
    wt_get<NL, 5>(lds, tid, x);
    wt_inv_group<NL, 5, 0, 2, LZ>(x, tid, gbase, log_n, P, q, q2);          // 5, 6, 7
    wt_put<NL, 5>(lds, tid, x);
    __syncthreads();
    wt_get<NL, 8>(lds, tid, x);
    wt_inv_group<NL, 8, 0, 2, LZ>(x, tid, gbase, log_n, P, q, q2);          // 8, 9, 10
}

enum { TILE_FWD = 0, TILE_INV = 1, TILE_MUL = 2 };

// grid.x = polys * (n / 2^11): workgroup g handles tile g % tiles of polynomial g / tiles (limb = polynomial % L).
//   TILE_FWD : dst tile = low 11 forward stages of src tile (the top stages were done by wide_pass_kernel<.., true>)
//   TILE_INV : dst tile = low 11 inverse stages of src tile; scale as in wide_pass_kernel (applied when n = 2^11)
//   TILE_MUL : dst tile = inverse stages of (forward(src tile) .* forward(src2 tile)); the result carries R^-1 until the scaling
//              by n^-1 R^2 (here when n = 2^11, else in the inverse top pass)
// dst may alias src / src2 tile for tile: every workgroup loads its tiles completely before its first store.
// LZ: the lazy class (every q < 2^(64 NL - 6), see wct_l): same inputs, same canonical outputs, fewer instructions in between.
template <int NL, int MODE, bool LZ>
__global__ void __launch_bounds__(WT_T, 2)
wide_tile_kernel(u256 *dst, const u256 *src, const u256 *src2, const WLimb<NL> *__restrict__ limbs, uint32_t L, uint32_t log_n, uint32_t scale) {
    __shared__ uint32_t lds[2 * NL * WT_N];
    const uint32_t tid = threadIdx.x, tiles = 1u << (log_n - WT_LOG);
    const uint32_t p = blockIdx.x >> (log_n - WT_LOG), tile = blockIdx.x & (tiles - 1);
    const WLimb<NL> &P = limbs[p % L];
    const wint<NL> q = P.q;
    const wint<NL> q2 = LZ ? wshl<NL>(q, 1) : q;
    const uint32_t gbase = tile << WT_LOG;
    const size_t off = ((size_t)p << log_n) + gbase;
    wint<NL> x[8];
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = wload_c<NL>(src + off + tid + r * WT_T);
    if constexpr (MODE == TILE_FWD) {
        wt_forward<NL, LZ>(x, lds, tid, gbase, log_n, P, q, q2);
        if constexpr (LZ) wreduce<NL, 5>(x, q);                     // below 23 q -> canonical
        wt_put<NL, 0>(lds, tid, x);
        __syncthreads();
        wt_get<NL, 8>(lds, tid, x);                                 // back to the coalesced order for the store
    } else if constexpr (MODE == TILE_INV) {
        wt_put<NL, 8>(lds, tid, x);
        __syncthreads();
        wt_get<NL, 0>(lds, tid, x);
        wt_inverse<NL, LZ>(x, lds, tid, gbase, log_n, P, q, q2);
    } else {
        wint<NL> y[8];
#pragma unroll
        for (int r = 0; r < 8; r++) y[r] = wload_c<NL>(src2 + off + tid + r * WT_T);   // issued early: hides under a's butterflies
        wt_forward<NL, LZ>(x, lds, tid, gbase, log_n, P, q, q2);
        __syncthreads();                                            // a's last exchange reads are over before b's first put
        wt_forward<NL, LZ>(y, lds, tid, gbase, log_n, P, q, q2);
        if constexpr (LZ) {                                         // x < 23 q, y < 23 q -> y < 2q: the product is below q (1 + 46 q / R) < 2q
            wreduce<NL, 5, 1>(y, q);
#pragma unroll
            for (int r = 0; r < 8; r++) x[r] = wmontl<NL>(x[r], y[r], q, P.qinv32);
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) x[r] = wmont<NL>(x[r], y[r], q, P.qinv32);
        }
        wt_inverse<NL, LZ>(x, lds, tid, gbase, log_n, P, q, q2);    // starts in the pattern both transforms ended in: no exchange
    }
    if (MODE != TILE_FWD && scale) {                                // canonical product: input below 2q (lazy) or q
        const wint<NL> c = scale == 2 ? P.ninv_r2 : P.ninv_m;
#pragma unroll
        for (int r = 0; r < 8; r++) x[r] = wmont<NL>(x[r], c, q, P.qinv32);
    } else if (LZ && MODE != TILE_FWD) {
        wreduce<NL, 1>(x, q);                                       // below 2q -> canonical (the scaling happens in the top pass)
    }
#pragma unroll
    for (int r = 0; r < 8; r++) wstore_c<NL>(dst + off + tid + r * WT_T, x[r]);
}

// r = mont(a, b) per coefficient (NTT domain; carries R^-1 until the n^-1 R^2 scaling of the inverse transform that follows)
template <int NL>
__global__ void __launch_bounds__(256)
wide_pointwise_kernel(u256 *r, const u256 *a, const u256 *b, const WLimb<NL> *__restrict__ limbs, uint32_t L, uint32_t log_n, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const WLimb<NL> &P = limbs[(uint32_t)((g >> log_n) % L)];
        wstore_c<NL>(r + g, wmont<NL>(wload_c<NL>(a + g), wload_c<NL>(b + g), P.q, P.qinv32));
    }
}

// x *= n^-1 R (scale 1) or n^-1 R^2 (scale 2) per coefficient: the whole "inverse transform" of a degree-1 engine (an RNS base without
// a ring, fhe_rns_base_create), whose butterfly network is empty.
template <int NL>
__global__ void __launch_bounds__(256)
wide_scale_kernel(u256 *r, const u256 *a, const WLimb<NL> *__restrict__ limbs, uint32_t L, uint32_t log_n, uint32_t scale, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const WLimb<NL> &P = limbs[(uint32_t)((g >> log_n) % L)];
        wstore_c<NL>(r + g, wmont<NL>(wload_c<NL>(a + g), scale == 2 ? P.ninv_r2 : P.ninv_m, P.q, P.qinv32));
    }
}

// Tensor product in the NTT domain (FHEContext::multiply, src/fhe.cu:199-218) on four transformed operands, one pass:
// c0 = a0 b0, c1 = a0 b1 + a1 b0, c2 = a1 b1, every product a single Montgomery product (results carry R^-1, removed by the
// n^-1 R^2 scaling of the inverse transforms that follow).  One coefficient per lane.
template <int NL>
__global__ void __launch_bounds__(256)
wide_ct_pointwise_kernel(u256 *c0, u256 *c1, u256 *c2, const u256 *a0, const u256 *a1, const u256 *b0, const u256 *b1,
                         const WLimb<NL> *__restrict__ limbs, uint32_t L, uint32_t log_n, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const WLimb<NL> &P = limbs[(uint32_t)((g >> log_n) % L)];
        const wint<NL> q = P.q; const uint32_t qi = P.qinv32;
        const wint<NL> u0 = wload_c<NL>(a0 + g), u1 = wload_c<NL>(a1 + g), v0 = wload_c<NL>(b0 + g), v1 = wload_c<NL>(b1 + g);
        wstore_c<NL>(c0 + g, wmont<NL>(u0, v0, q, qi));
        wstore_c<NL>(c1 + g, waddmod<NL>(wmont<NL>(u0, v1, q, qi), wmont<NL>(u1, v0, q, qi), q));
        wstore_c<NL>(c2 + g, wmont<NL>(u1, v1, q, qi));
    }
}

}  // namespace fhe_dev
