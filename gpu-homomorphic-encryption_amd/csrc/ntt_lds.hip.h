// ntt_lds.hip.h -- LDS-resident negacyclic NTT / polymul kernels for word-sized RNS primes on gfx950.
//   F32 : q < 2^30, 32-bit residues, Harvey/Shoup integer butterflies          (FHE_WIDTH_32, N = 2^11 .. 2^15)
//   F52 : q < 2^43, residues held as exact integers in doubles, FMA butterflies (FHE_WIDTH_52, N = 2^11 .. 2^14)
//   F64 : q < 2^62, 64-bit residues, Harvey/Shoup integer butterflies           (FHE_WIDTH_64, N = 2^11 .. 2^14)
//
// Replaces ntt_forward_optimized_kernel / ntt_inverse_optimized_kernel / ntt_pointwise_mul_kernel /
// bit_reverse_kernel / ntt_forward_batch_kernel (kernels/ntt_kernels.cu:7-210) and the
// NTTEngine::multiply launch sequence (src/ntt.cu:49-75) for word-sized moduli.
//
// Why a narrow path is bit-exact: with reduced operands and a prime modulus every reference
// primitive returns the canonical residue (mont(x, w*R) = x*w mod q, add_mod, sub_mod), so any exact
// evaluation of the same butterfly network yields the same 256-bit containers (upper words zero).
//
// Shape (N = 2^LOGN coefficients, one workgroup of T = N/32 threads per (polynomial, limb)):
//   * HBM is touched exactly once per coefficient: a 4/8-byte load out of each 32-byte container
//     (the rest rides along in the same 128-byte lines) and one full 32-byte store.
//   * each thread owns 32 coefficients and runs 5 butterfly stages in registers (radix-32), then the
//     workgroup transposes through LDS; log2(N) = 5 + 5 + REM stages -> 3 register groups.
//   * LDS holds one residue per coefficient, padded by one element per 32 (phys = i + i/32) so that
//     all access patterns below are bank-conflict-free for 64-lane wavefronts.
//   * butterflies are Harvey lazy butterflies (twiddles: Montgomery form for F32, Shoup pairs for F64): values live in
//     [0,4q) (forward) / [0,2q) (inverse) and are made canonical once, before the store.
//   * twiddles of the first register group are wave-uniform (scalar loads); the rest are vector
//     loads from an L2-resident table shared by the whole batch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "u256_dev.h"

namespace fhe_dev {

typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));   // native vectors: accepted by the nontemporal builtins
typedef uint64_t v2u64 __attribute__((ext_vector_type(2)));

// ---- field traits ---------------------------------------------------------------------------------------
// A field supplies the residue type E, the twiddle record TW = (w, companion), the butterflies and the
// range bookkeeping.  "lazy" ranges: integer fields keep forward values in [0,4q) and inverse values in [0,2q);
// the floating-point field keeps signed values whose magnitude stays far below 2^53.
template <class E>
__device__ __forceinline__ E csub(E x, E c) {   // x - (x >= c ? c : 0) for unsigned E
    E d = x - c;
    return d < x ? d : x;                       // sub + unsigned min (d wraps above x exactly when x < c)
}

template <class Self, class E_, class TW_>
struct IntField {
    using E = E_;
    using TW = TW_;
    // Harvey lazy Cooley-Tukey butterfly, inputs and outputs in [0,4q)
    template <class L> __device__ static __forceinline__ void fwd_bfly(E &x0, E &x1, const TW &w, const L &P) {
        E X = csub<E>(x0, P.q2);
        E T = Self::tw_mul(x1, w, P);
        x0 = X + T;
        x1 = X - T + P.q2;
    }
    // Harvey lazy Gentleman-Sande butterfly, inputs and outputs in [0,2q)
    template <class L> __device__ static __forceinline__ void inv_bfly(E &x0, E &x1, const TW &w, const L &P) {
        E X = x0, Y = x1;
        x0 = csub<E>(X + Y, P.q2);
        x1 = Self::tw_mul(X - Y + P.q2, w, P);
    }
    // x * twiddle mod q, result in [0, 2q): Shoup form (w, floor(w*2^W/q)) unless the field overrides it
    template <class L> __device__ static __forceinline__ E tw_mul(E x, const TW &w, const L &P) { return Self::shoup_mul(x, w.x, w.y, P.q); }
    // last inverse stage with the n^-1 scaling folded in (outputs in [0,2q))
    __device__ static __forceinline__ void inv_last(E &x0, E &x1, E q, E q2, E ninv, E ninv_s, E ninvw, E ninvw_s) {
        E X = x0, Y = x1;
        x0 = Self::shoup_mul(X + Y, ninv, ninv_s, q);
        x1 = Self::shoup_mul(X - Y + q2, ninvw, ninvw_s, q);
    }
    __device__ static __forceinline__ void regroup(E (&)[32], E, E) {}                       // integer ranges never grow
    __device__ static __forceinline__ E regroup1(E x, E, E) { return x; }
    __device__ static __forceinline__ E canon_fwd(E x, E q, E q2, E) { return csub<E>(csub<E>(x, q2), q); }   // [0,4q) -> [0,q)
    __device__ static __forceinline__ E canon_inv(E x, E q) { return csub<E>(x, q); }       // [0,2q) -> [0,q)
    // NTT-domain product for the fused kernels: a canonical, b lazy (< 4q); result in [0,2q), carries 2^-W
    __device__ static __forceinline__ E pw_mul(E a, E b, E q, E qinv) { return Self::mont_mul(a, b, q, qinv); }
    // a0*b0 + a1*b1 in the NTT domain (a* canonical, b* lazy), result in [0,2q), carries 2^-W; F32 overrides it with one shared reduction
    __device__ static __forceinline__ E pw_mul2(E a0, E b0, E a1, E b1, E q, E q2, E qinv) {
        return csub<E>(Self::mont_mul(a0, b0, q, qinv) + Self::mont_mul(a1, b1, q, qinv), q2);
    }
    // [0,2q)+[0,2q) -> [0,2q)
    __device__ static __forceinline__ E pw_add(E a, E b, E, E q2) { return csub<E>(a + b, q2); }
    // element-wise canonical ops
    template <class L> __device__ static __forceinline__ E ew_mul(E x, E y, const L &P) {
        return csub<E>(Self::shoup_mul(Self::mont_mul(x, y, P.q, P.qinv), P.r1, P.r1_s, P.q), P.q);
    }
    __device__ static __forceinline__ E ew_add(E x, E y, E q) { return csub<E>(x + y, q); }
    __device__ static __forceinline__ E ew_sub(E x, E y, E q) { return csub<E>(x - y + q, q); }
    __device__ static __forceinline__ bool ge(uint64_t raw, E q) { return raw >= (uint64_t)q; }
    __device__ static __forceinline__ E from_u64(uint64_t d, E q) { return (E)(d % (uint64_t)q); }   // canonical residue of a small integer
    // bits [lo, lo+w) of a residue (lo < bit width of E)
    __device__ static __forceinline__ E digit(E x, uint32_t lo, uint32_t w) {
        E v = x >> lo;
        return w >= 8 * sizeof(E) ? v : (E)(v & (((E)1 << w) - 1));
    }
    // canonical x -> the operand form pw_mul expects on its canonical side so that the product comes out plain: x * 2^W mod q
    template <class L> __device__ static __forceinline__ E to_pw_operand(E x, const L &P) { return csub<E>(Self::shoup_mul(x, P.r1, P.r1_s, P.q), P.q); }
};

template <class Self, class TW_>
struct F32Base : IntField<Self, uint32_t, TW_> {
    using E = uint32_t;
    using V16 = v4u32;                  // one 16-byte half container
    static constexpr int MAX_LOGN = 15;
    static constexpr int MULT_MINW = 4; // waves per SIMD the fused multiply is compiled for (4 workgroups per CU at N = 8192)
    // x*w mod q for w < q with companion ws = floor(w*2^32/q); any x; result in [0, 2q).
    __device__ static __forceinline__ E shoup_mul(E x, E w, E ws, E q) { return x * w - __umulhi(x, ws) * q; }
    // a*b*2^-32 mod q for a*b < q*2^32; result in [0, 2q).  nqinv = -q^-1 mod 2^32 (Limb::qinv holds the NEGATED inverse on this
    // field): m = t * nqinv makes t + m*q divisible by 2^32, and the whole reduction is ONE v_mad_u64_u32 (multiply + 64-bit add)
    // instead of v_mul_hi + v_sub + v_add: 3 multiply-class instructions per product and no additions.  t + m*q < 2 q 2^32 < 2^63.
    __device__ static __forceinline__ E mont_mul(E a, E b, E q, E nqinv) {
        const uint64_t t = (uint64_t)a * b;
        const E m = (E)t * nqinv;
        return (E)(((uint64_t)m * q + t) >> 32);
    }
    // (a0*b0 + a1*b1) * 2^-32 mod q with ONE reduction: the second product rides on the first as the addend of its
    // v_mad_u64_u32.  a0, a1 < q (key entries), b0, b1 < 4q (lazy transform outputs), q < 2^30: the sum is < 8q^2 < 2^63 and
    // the result < 8q^2 / 2^32 + q < 3q; one conditional subtraction brings it to [0, 2q).  4 multiply-class instructions
    // for two products instead of 6, and one accumulation instead of two.
    __device__ static __forceinline__ E mont_mul2(E a0, E b0, E a1, E b1, E q, E q2, E nqinv) {
        const uint64_t t = (uint64_t)a1 * b1 + (uint64_t)a0 * b0;
        const E m = (E)t * nqinv;
        return csub<E>((E)(((uint64_t)m * q + t) >> 32), q2);
    }
    __device__ static __forceinline__ E pw_mul2(E a0, E b0, E a1, E b1, E q, E q2, E nqinv) { return mont_mul2(a0, b0, a1, b1, q, q2, nqinv); }
    __device__ static __forceinline__ E load_low(const void *container) { return __builtin_nontemporal_load((const E *)container); }
    __device__ static __forceinline__ V16 pack(E v) { V16 o = {v, 0u, 0u, 0u}; return o; }
    __device__ static __forceinline__ uint64_t low(const V16 &v) { return v.x; }
    __device__ static __forceinline__ bool upper_nonzero(const V16 &v) { return (v.y | v.z | v.w) != 0; }
    __device__ static __forceinline__ bool any_nonzero(const V16 &v) { return (v.x | v.y | v.z | v.w) != 0; }
};
// 32-bit field: twiddles in Montgomery form w*2^32 mod q -- 4 multiply-class instructions per butterfly instead
// of Shoup's 3, but 4 instead of 8 bytes per twiddle (registers, L2 traffic, vector-memory issue slots).  Interleaved A/B
// against Shoup-form twiddles on one MI355X (scratch/kbench.hip, batch 4096): 6.10 vs 6.02 TB/s on the fused multiply,
// bit-identical results.
struct F32 : F32Base<F32, uint32_t> {
    template <class L> __device__ static __forceinline__ E tw_mul(E x, const uint32_t &w, const L &P) { return mont_mul(x, w, P.q, P.qinv); }
};
struct F64 : IntField<F64, uint64_t, ulonglong2> {
    using V16 = v2u64;
    static constexpr int MAX_LOGN = 14; // 2^15 x 8 B does not fit the 160 KiB LDS
    static constexpr int MULT_MINW = 2;
    __device__ static __forceinline__ E shoup_mul(E x, E w, E ws, E q) { return x * w - __umul64hi(x, ws) * q; }
    __device__ static __forceinline__ E mont_mul(E a, E b, E q, E qinv) {
        E lo = a * b, hi = __umul64hi(a, b);
        E m = lo * qinv;
        return hi - __umul64hi(m, q) + q;
    }
    __device__ static __forceinline__ E load_low(const void *container) { return __builtin_nontemporal_load((const E *)container); }
    __device__ static __forceinline__ V16 pack(E v) { V16 o = {v, 0ull}; return o; }
    __device__ static __forceinline__ uint64_t low(const V16 &v) { return v.x; }
    __device__ static __forceinline__ bool upper_nonzero(const V16 &v) { return v.y != 0; }
    __device__ static __forceinline__ bool any_nonzero(const V16 &v) { return (v.x | v.y) != 0; }
};

// Full-range 64-bit field: ANY odd prime q < 2^64 (in practice 2^62 <= q < 2^64, the primes the lazy F64 ranges cannot hold: its
// [0, 4q) / [0, 2q) bookkeeping needs 4q < 2^64).  Values stay canonical in [0, q) through every butterfly; sums and differences go
// through the carry / borrow (addm / subm), products are canonical Montgomery products hi(a*b) - hi(m*q) (+ q on borrow), so
// twiddles are single words w * 2^64 mod q (8 bytes instead of F64's 16-byte Shoup pairs).  Limb<F64X> holds: qinv = q^-1 mod 2^64,
// r1 = 2^128 mod q, ninv / ninvw = n^-1 (* itw[1]) * 2^64, ninv_r / ninvw_r = the same * 2^128; every *_s slot holds qinv again
// (inv_last receives no qinv of its own, and there are no Shoup companions on this field).
struct F64X : IntField<F64X, uint64_t, uint64_t> {
    using V16 = v2u64;
    static constexpr int MAX_LOGN = 14;
    static constexpr int MULT_MINW = 2;
    __device__ static __forceinline__ E addm(E a, E b, E q) { E s = a + b; return (s < a || s >= q) ? s - q : s; }
    __device__ static __forceinline__ E subm(E a, E b, E q) { E d = a - b; return a < b ? d + q : d; }
    // a*b*2^-64 mod q, canonical, for a*b < q*2^64 (one factor below q, the other any 64-bit word)
    __device__ static __forceinline__ E mont_mul(E a, E b, E q, E qinv) {
        const E lo = a * b, hi = __umul64hi(a, b);
        const E mh = __umul64hi(lo * qinv, q);
        const E d = hi - mh;
        return hi < mh ? d + q : d;
    }
    template <class L> __device__ static __forceinline__ E tw_mul(E x, const TW &w, const L &P) { return mont_mul(x, w, P.q, P.qinv); }
    template <class L> __device__ static __forceinline__ void fwd_bfly(E &x0, E &x1, const TW &w, const L &P) {
        const E T = mont_mul(x1, w, P.q, P.qinv), X = x0;
        x0 = addm(X, T, P.q);
        x1 = subm(X, T, P.q);
    }
    template <class L> __device__ static __forceinline__ void inv_bfly(E &x0, E &x1, const TW &w, const L &P) {
        const E X = x0, Y = x1;
        x0 = addm(X, Y, P.q);
        x1 = mont_mul(subm(X, Y, P.q), w, P.q, P.qinv);
    }
    __device__ static __forceinline__ void inv_last(E &x0, E &x1, E q, E, E ninv, E qinv, E ninvw, E) {
        const E X = x0, Y = x1;
        x0 = mont_mul(addm(X, Y, q), ninv, q, qinv);
        x1 = mont_mul(subm(X, Y, q), ninvw, q, qinv);
    }
    __device__ static __forceinline__ E canon_fwd(E x, E, E, E) { return x; }
    __device__ static __forceinline__ E canon_inv(E x, E) { return x; }
    __device__ static __forceinline__ E pw_mul(E a, E b, E q, E qinv) { return mont_mul(a, b, q, qinv); }
    __device__ static __forceinline__ E pw_mul2(E a0, E b0, E a1, E b1, E q, E, E qinv) { return addm(mont_mul(a0, b0, q, qinv), mont_mul(a1, b1, q, qinv), q); }
    __device__ static __forceinline__ E pw_add(E a, E b, E q, E) { return addm(a, b, q); }
    template <class L> __device__ static __forceinline__ E ew_mul(E x, E y, const L &P) { return mont_mul(mont_mul(x, y, P.q, P.qinv), P.r1, P.q, P.qinv); }
    __device__ static __forceinline__ E ew_add(E x, E y, E q) { return addm(x, y, q); }
    __device__ static __forceinline__ E ew_sub(E x, E y, E q) { return subm(x, y, q); }
    template <class L> __device__ static __forceinline__ E to_pw_operand(E x, const L &P) { return mont_mul(x, P.r1, P.q, P.qinv); }
    __device__ static __forceinline__ E load_low(const void *container) { return __builtin_nontemporal_load((const E *)container); }
    __device__ static __forceinline__ V16 pack(E v) { V16 o = {v, 0ull}; return o; }
    __device__ static __forceinline__ uint64_t low(const V16 &v) { return v.x; }
    __device__ static __forceinline__ bool upper_nonzero(const V16 &v) { return v.y != 0; }
    __device__ static __forceinline__ bool any_nonzero(const V16 &v) { return (v.x | v.y) != 0; }
};

// Residues as exact integers in IEEE doubles (q < 2^43).  x*w mod q with the precomputed companion wq = fl(w/q):
//   h = fl(x*w), l = x*w - h (exact, one FMA), c = rint(fl(x*wq)), d = h - c*q (exact, one FMA), r = d + l.
// For |x| < 2^49: |fl(x*wq) - x*w/q| < 2^-2, so c is within 1 of the nearest integer and |r| < 0.76 q; h - c*q and l are
// integers below 2^53 in magnitude, hence every step is exact and r == x*w (mod q).  Six half-rate FP64 instructions
// replace the ~20 half-rate 32-bit integer multiplies of a 64-bit Shoup product (measured 4.5 vs 63+ cycles per wave).
// Butterfly outputs are not range-reduced: forward values grow by < 0.76 q per stage (<= 12 q after 14 stages), inverse
// values are brought back below 0.76 q once per 5-stage register group (regroup).
struct F52 {
    using E = double;
    using TW = double;                  // w only: the companion fl(w / q) is recomputed as fl(w * fl(1/q)) (one FP64 multiply
                                        // instead of 2 more VGPRs and 8 more L2 bytes per twiddle; |error| of c stays < 0.15)
    using V16 = v2u64;
    static constexpr int MAX_LOGN = 14;
    static constexpr int MULT_MINW = 2;
    __device__ static __forceinline__ E mulmod(E x, E w, E wq, E q) {
#pragma clang fp contract(off)
        E h = x * w;
        E l = __builtin_fma(x, w, -h);
        E c = __builtin_rint(x * wq);
        E d = __builtin_fma(-c, q, h);
        return d + l;
    }
    __device__ static __forceinline__ E reduce(E x, E q, E qinv) {       // |x| < 2^49 -> |r| < 0.76 q
#pragma clang fp contract(off)
        E c = __builtin_rint(x * qinv);
        return __builtin_fma(-c, q, x);
    }
    template <class L> __device__ static __forceinline__ void fwd_bfly(E &x0, E &x1, const TW &w, const L &P) {
#pragma clang fp contract(off)
        E T = mulmod(x1, w, w * P.qinv, P.q);
        x1 = x0 - T;
        x0 = x0 + T;
    }
    template <class L> __device__ static __forceinline__ void inv_bfly(E &x0, E &x1, const TW &w, const L &P) {
#pragma clang fp contract(off)
        E S = x0 + x1, D = x0 - x1;
        x0 = S;
        x1 = mulmod(D, w, w * P.qinv, P.q);
    }
    __device__ static __forceinline__ void inv_last(E &x0, E &x1, E q, E, E ninv, E ninv_s, E ninvw, E ninvw_s) {
        E S = x0 + x1, D = x0 - x1;
        x0 = mulmod(S, ninv, ninv_s, q);
        x1 = mulmod(D, ninvw, ninvw_s, q);
    }
    __device__ static __forceinline__ void regroup(E (&x)[32], E q, E qinv) {
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = reduce(x[r], q, qinv);
    }
    __device__ static __forceinline__ E regroup1(E x, E q, E qinv) { return reduce(x, q, qinv); }
    __device__ static __forceinline__ E canon_fwd(E x, E q, E, E qinv) { E r = reduce(x, q, qinv); return r < 0 ? r + q : r; }
    __device__ static __forceinline__ E canon_inv(E x, E q) { return x < 0 ? x + q : x; }   // |x| < 0.76 q
    __device__ static __forceinline__ E pw_mul(E a, E b, E q, E qinv) {                     // a in [0,q), |b| < 2^48
#pragma clang fp contract(off)
        E h = a * b;
        E l = __builtin_fma(a, b, -h);
        E c = __builtin_rint(h * qinv);
        E d = __builtin_fma(-c, q, h);
        return d + l;
    }
    __device__ static __forceinline__ E pw_add(E a, E b, E, E) { return a + b; }
    __device__ static __forceinline__ E pw_mul2(E a0, E b0, E a1, E b1, E q, E, E qinv) { return pw_mul(a0, b0, q, qinv) + pw_mul(a1, b1, q, qinv); }
    template <class L> __device__ static __forceinline__ E ew_mul(E x, E y, const L &P) { return canon_inv(pw_mul(x, y, P.q, P.qinv), P.q); }
    __device__ static __forceinline__ E ew_add(E x, E y, E q) { E t = x + y; return t >= q ? t - q : t; }
    __device__ static __forceinline__ E ew_sub(E x, E y, E q) { E t = x - y; return t < 0 ? t + q : t; }
    __device__ static __forceinline__ E load_low(const void *container) { return (E)__builtin_nontemporal_load((const uint64_t *)container); }
    __device__ static __forceinline__ V16 pack(E v) { V16 o = {(uint64_t)v, 0ull}; return o; }
    __device__ static __forceinline__ uint64_t low(const V16 &v) { return v.x; }
    __device__ static __forceinline__ bool upper_nonzero(const V16 &v) { return v.y != 0; }
    __device__ static __forceinline__ bool any_nonzero(const V16 &v) { return (v.x | v.y) != 0; }
    __device__ static __forceinline__ bool ge(uint64_t raw, E q) { return (E)raw >= q; }
    __device__ static __forceinline__ E from_u64(uint64_t d, E q) { return (E)(d % (uint64_t)q); }
    __device__ static __forceinline__ E digit(E x, uint32_t lo, uint32_t w) {
        uint64_t v = (uint64_t)x >> lo;
        return (E)(w >= 64 ? v : (v & ((1ull << w) - 1)));
    }
    template <class L> __device__ static __forceinline__ E to_pw_operand(E x, const L &) { return x; }
};

// Per-limb constants (device memory, one entry per RNS prime).  *_s = Shoup companion floor(x*2^W/q).
template <class F>
struct Limb {
    using E = typename F::E;
    E q, q2, qinv, _pad0;                 // qinv = q^-1 mod 2^W   (F32: -q^-1 mod 2^32; F52: fl(1/q))
    E r1, r1_s;                           // 2^W mod q             (undo the 2^-W of mont_mul in `pointwise`)
    E ninv, ninv_s;                       // n^-1
    E ninvw, ninvw_s;                     // n^-1 * itw[1]
    E ninv_r, ninv_r_s;                   // n^-1 * 2^W            (fused multiply: absorbs mont_mul's 2^-W)
    E ninvw_r, ninvw_r_s;                 // n^-1 * itw[1] * 2^W
    const typename F::TW *tw;             // [n] (psi^bitrev(k), shoup)
    const typename F::TW *itw;            // [n] (psi^-bitrev(k), shoup)
};
using Limb32 = Limb<F32>;
using Limb64 = Limb<F64>;
using Limb64X = Limb<F64X>;
using Limb52 = Limb<F52>;

template <int LOGN>
struct NttCfg {
    static_assert(LOGN >= 11 && LOGN <= 15, "LDS-resident path covers 2^11 .. 2^15");
    static constexpr int N = 1 << LOGN;
    static constexpr int LOGT = LOGN - 5;
    static constexpr int T = 1 << LOGT;            // threads per workgroup
    static constexpr int REM = LOGN - 10;          // stages in the third register group (1..5)
    static constexpr int LDS_ELEMS = N + N / 32;
};

// ---- index patterns: logical index = base(tid) | off(r), physical LDS slot = pbase(tid) + poff(r) ----
// A : r <-> index bits [LOGT, LOGN)      (coalesced global order: consecutive lanes, consecutive coefficients)
template <int LOGN> struct PatA {
    using C = NttCfg<LOGN>;
    static constexpr int BIT0 = C::LOGT;
    static constexpr bool TW_UNIFORM = true;      // every stage on these bits has i >> (b+1) independent of tid
    static constexpr uint32_t TW_STRIDE = 0;
    __device__ static uint32_t tw_thread(uint32_t) { return 0; }
    __device__ static uint32_t base(uint32_t tid) { return tid; }
    __device__ static uint32_t pbase(uint32_t tid) { return tid + (tid >> 5); }
    static constexpr uint32_t off(int r) { return (uint32_t)r << C::LOGT; }
    static constexpr uint32_t poff(int r) { return (uint32_t)r * (C::T + C::T / 32); }
};
// M : r <-> index bits [REM, REM+5)      (forward middle group)
template <int LOGN> struct PatM {
    using C = NttCfg<LOGN>;
    static constexpr int BIT0 = C::REM;
    static constexpr bool TW_UNIFORM = false;
    static constexpr uint32_t TW_STRIDE = 32;                                 // distinct twiddle-owning thread groups (T >> REM)
    __device__ static uint32_t tw_thread(uint32_t tid) { return tid >> C::REM; }
    __device__ static uint32_t base(uint32_t tid) { return ((tid >> C::REM) << (C::REM + 5)) | (tid & ((1u << C::REM) - 1)); }
    __device__ static uint32_t pbase(uint32_t tid) { uint32_t b = base(tid); return b + (b >> 5); }
    static constexpr uint32_t off(int r) { return (uint32_t)r << C::REM; }
    static constexpr uint32_t poff(int r) { return off(r) + (off(r) >> 5); }
};
// Z : r <-> index bits [0, 5)            (32 consecutive coefficients per thread)
template <int LOGN> struct PatZ {
    static constexpr int BIT0 = 0;
    static constexpr bool TW_UNIFORM = false;
    static constexpr uint32_t TW_STRIDE = NttCfg<LOGN>::T;
    __device__ static uint32_t tw_thread(uint32_t tid) { return tid; }
    __device__ static uint32_t base(uint32_t tid) { return tid << 5; }
    __device__ static uint32_t pbase(uint32_t tid) { return tid * 33; }
    static constexpr uint32_t off(int r) { return (uint32_t)r; }
    static constexpr uint32_t poff(int r) { return (uint32_t)r; }
};
// Y : r <-> index bits [5, 10)           (inverse middle group)
template <int LOGN> struct PatY {
    static constexpr int BIT0 = 5;
    static constexpr bool TW_UNIFORM = false;
    static constexpr uint32_t TW_STRIDE = NttCfg<LOGN>::T / 32;
    __device__ static uint32_t tw_thread(uint32_t tid) { return tid >> 5; }
    __device__ static uint32_t base(uint32_t tid) { return ((tid >> 5) << 10) | (tid & 31); }
    __device__ static uint32_t pbase(uint32_t tid) { return (tid >> 5) * 1056 + (tid & 31); }
    static constexpr uint32_t off(int r) { return (uint32_t)r << 5; }
    static constexpr uint32_t poff(int r) { return (uint32_t)r * 33; }
};

template <class Pat, class E>
__device__ __forceinline__ void lds_put(E *lds, uint32_t tid, const E (&x)[32]) {
    E *p = lds + Pat::pbase(tid);
#pragma unroll
    for (int r = 0; r < 32; r++) p[Pat::poff(r)] = x[r];
}
template <class Pat, class E>
__device__ __forceinline__ void lds_get(const E *lds, uint32_t tid, E (&x)[32]) {
    const E *p = lds + Pat::pbase(tid);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = p[Pat::poff(r)];
}

// ---- twiddle tables ----------------------------------------------------------------------------------------
// The stage on index bit b uses w[m + (i >> (b+1))], m = N >> (b+1), i = coefficient index.  In device memory the tables are
// in that natural order (a wave's strided walk re-uses the lines its first load brought into L1; a lane-consecutive order was
// measured 5-10 % SLOWER there).  Kernels that run many transforms under one modulus (key switching, external product) copy
// the table into LDS instead, PERMUTED within each stage range so that the lanes of a wave read consecutive words (no bank
// conflicts): for a stage that runs as r-bit k of pattern Pat, j = i >> (b+1) = tt << (4-k) | rh with tt = Pat::tw_thread(tid)
// and rh = r >> (k+1); the entry sits at slot m + rh * Pat::TW_STRIDE + tt.  Forward tables use patterns Z / M, inverse
// tables Z / Y; the stages of the uniform pattern A (m <= 16) always take scalar loads from device memory.
template <class Pat>
__device__ __host__ constexpr uint32_t tw_slot_off(int r, int k) { return (uint32_t)(r >> (k + 1)) * Pat::TW_STRIDE; }
// LDS slot of natural index idx in [1, n) of the forward (fwd = true) or inverse table for n = 2^log_n
__device__ __host__ inline uint32_t tw_slot(uint32_t log_n, bool fwd, uint32_t idx) {
    const uint32_t lg = 31 - (uint32_t)__builtin_clz(idx);
    const uint32_t m = 1u << lg, j = idx - m, b = log_n - 1 - lg, rem = log_n - 10, T = 1u << (log_n - 5);
    uint32_t k, stride;
    if (fwd) {
        if (b < rem) { k = b; stride = T; }                       // pattern Z
        else if (b < rem + 5) { k = b - rem; stride = 32; }       // pattern M
        else return idx;                                          // pattern A (never read from LDS)
    } else {
        if (b < 5) { k = b; stride = T; }                         // pattern Z
        else if (b < 10) { k = b - 5; stride = T / 32; }          // pattern Y
        else return idx;
    }
    const uint32_t sh = 4 - k, tt = j >> sh, rh = j & ((1u << sh) - 1);
    return m + rh * stride + tt;
}
// Copy a table into LDS in the permuted order (coalesced reads; the scattered LDS writes happen once per workgroup and table).
template <class F, int LOGN, bool FWD>
__device__ __forceinline__ void stage_twiddles(typename F::TW *twl, const typename F::TW *__restrict__ gtw, uint32_t tid) {
    using C = NttCfg<LOGN>;
#pragma unroll 8
    for (uint32_t idx = tid; idx < (uint32_t)C::N; idx += C::T)
        if (idx >= 1) twl[tw_slot(LOGN, FWD, idx)] = gtw[idx];     // entry 0 is unused; pattern-A entries keep their natural slot
}

// ---- register-resident butterfly stages -------------------------------------------------------------
// Forward (Cooley-Tukey, merged psi twiddles): stage on index bit b uses twiddle tw[m + (i >> (b+1))], m = N >> (b+1).
// Processes r-bits KHI down to KLO of pattern Pat.  Values stay in [0, 4q).
// TWL: `tw` is an LDS copy in the permuted order (non-uniform patterns only).
// SUB: the 2^LOGN coefficients are block number (pre - 2^k) of a larger transform of 2^(LOGN + k) coefficients whose top k stages
// ran elsewhere (word_pass_kernel); the stage on local bit b then uses the big table at (pre << (LOGN-1-b)) + (i >> (b+1)), which for
// pre = 1 is the whole-transform formula.
template <class F, int LOGN, class Pat, int KHI, int KLO, bool TWL = false, bool SUB = false>
__device__ __forceinline__ void fwd_stages(typename F::E (&x)[32], uint32_t tid, const typename F::TW *__restrict__ tw, const Limb<F> &P, uint32_t pre = 1) {
    static_assert(!(TWL && Pat::TW_UNIFORM), "uniform stages read device memory");
    static_assert(!(TWL && SUB), "sub-transforms read their twiddles from device memory");
    const uint32_t base = TWL ? Pat::tw_thread(tid) : Pat::TW_UNIFORM ? 0u : Pat::base(tid);   // uniform -> scalar twiddle loads
#pragma unroll
    for (int k = KHI; k >= KLO; k--) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + (((SUB ? pre : 1u) << (LOGN - 1 - b)) + (TWL ? base : (base >> (b + 1))));
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW w = p[TWL ? tw_slot_off<Pat>(r, k) : (Pat::off(r) >> (b + 1))];
            F::fwd_bfly(x[r], x[r | (1 << k)], w, P);
        }
    }
}
// Inverse (Gentleman-Sande): stage on index bit b uses itw[m + (i >> (b+1))].  Processes r-bits KLO up to KHI.
// Values stay in [0, 2q).
template <class F, int LOGN, class Pat, int KLO, int KHI, bool TWL = false, bool SUB = false>
__device__ __forceinline__ void inv_stages(typename F::E (&x)[32], uint32_t tid, const typename F::TW *__restrict__ itw, const Limb<F> &P, uint32_t pre = 1) {
    static_assert(!(TWL && Pat::TW_UNIFORM), "uniform stages read device memory");
    static_assert(!(TWL && SUB), "sub-transforms read their twiddles from device memory");
    const uint32_t base = TWL ? Pat::tw_thread(tid) : Pat::TW_UNIFORM ? 0u : Pat::base(tid);
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = itw + (((SUB ? pre : 1u) << (LOGN - 1 - b)) + (TWL ? base : (base >> (b + 1))));
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW w = p[TWL ? tw_slot_off<Pat>(r, k) : (Pat::off(r) >> (b + 1))];
            F::inv_bfly(x[r], x[r | (1 << k)], w, P);
        }
    }
}
// Last inverse stage (index bit LOGN-1, single twiddle itw[1]) with the n^-1 scaling folded in.
template <class F>
__device__ __forceinline__ void inv_last_stage(typename F::E (&x)[32], typename F::E q, typename F::E q2, typename F::E ninv,
                                               typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s) {
#pragma unroll
    for (int r = 0; r < 16; r++) F::inv_last(x[r], x[r | 16], q, q2, ninv, ninv_s, ninvw, ninvw_s);
}

// ---- global memory <-> registers ----------------------------------------------------------------------
// Coalesced gather of the low word(s) of each container, pattern A (lane stride = one 32-byte container).
template <class F, int LOGN>
__device__ __forceinline__ void load_A(const char *__restrict__ poly, uint32_t tid, typename F::E (&x)[32]) {
    const char *p = poly + (size_t)tid * 32;
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::load_low(p + (size_t)r * (NttCfg<LOGN>::T * 32));
}
// Compact polynomials (internal workspace only, never at the ABI): sizeof(E) bytes per coefficient, natural order.  The fused
// FHEContext::multiply (tensor product + relinearisation) hands c2 from the tensor-product kernel to the key-switch kernel in this
// form: c2 is written once as S/8 (S/4 for 8-byte residues) and each of the L limb workgroups that re-read it moves S/8 instead of S.
template <class F, int LOGN>
__device__ __forceinline__ void load_A_compact(const typename F::E *__restrict__ poly, uint32_t tid, typename F::E (&x)[32]) {
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = poly[tid + r * NttCfg<LOGN>::T];
}
template <class F, int LOGN>
__device__ __forceinline__ void store_A_compact(typename F::E *__restrict__ poly, uint32_t tid, const typename F::E (&x)[32]) {
#pragma unroll
    for (int r = 0; r < 32; r++) poly[tid + r * NttCfg<LOGN>::T] = x[r];
}
// source of a key-switch digit polynomial: 32-byte containers (the ABI's c2) or the compact workspace
template <class F, int LOGN, bool COMPACT>
__device__ __forceinline__ void load_src(const char *__restrict__ base, size_t poly_index, uint32_t tid, typename F::E (&x)[32]) {
    if constexpr (COMPACT) load_A_compact<F, LOGN>(reinterpret_cast<const typename F::E *>(base) + poly_index * NttCfg<LOGN>::N, tid, x);
    else load_A<F, LOGN>(base + poly_index * (NttCfg<LOGN>::N * 32), tid, x);
}

// 16-byte loads of table rows through a buffer descriptor: SGPR base + ONE shared VGPR offset (tid * 16) + a scalar offset per load
// (`buffer_load_dwordx4 v, voff, s[rsrc], soff offen`).  With flat pointers every chunk of a key row needs its own 64-bit VGPR address
// (the chunk stride exceeds the 12-bit immediate), the compiler hoists those out of the digit loops and, in the register-starved
// kernels, spills them: 34 address pairs in the three-array key switch.  Offsets are 32-bit: the host keeps packed tables below 4 GiB.
struct TableBuf {
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ explicit TableBuf(const void *base) : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xffffffffu, 0x00020000)) {}
    template <class VecE> __device__ __forceinline__ VecE load16(uint32_t voff, uint32_t soff) const {
        static_assert(sizeof(VecE) == 16, "one 16-byte lane load");
        return __builtin_bit_cast(VecE, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    }
    // one residue: the low word(s) of a container (an integer, as F::load_low reads it) or a compact slot (the field's own type)
    template <class F, bool COMPACT> __device__ __forceinline__ typename F::E load_residue(uint32_t voff, uint32_t soff) const {
        using E = typename F::E;
        constexpr int AUX = 0;                   // temporal: these loads serve operands that several workgroups re-read (measured: nt costs 2-7 % on relinearisation)
        if constexpr (sizeof(E) == 4) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
        else {
            const uint64_t raw = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
            if constexpr (COMPACT) return __builtin_bit_cast(E, raw);
            else return (E)raw;
        }
    }
};
// load_src through a descriptor based at the first limb polynomial of a ciphertext component (`poly` = limb index within it): the 32
// loads of a thread share one VGPR offset, the per-load strides are scalar
template <class F, int LOGN, bool COMPACT>
__device__ __forceinline__ void load_src_buf(const TableBuf &B, uint32_t poly, uint32_t tid, typename F::E (&x)[32]) {
    using C = NttCfg<LOGN>;
    constexpr uint32_t STRIDE = COMPACT ? sizeof(typename F::E) : 32;
    const uint32_t voff = tid * STRIDE, base = poly * (C::N * STRIDE);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = B.template load_residue<F, COMPACT>(voff, base + r * (C::T * STRIDE));
}
// one polynomial by its own base pointer (uniform per workgroup)
template <class F, int LOGN, bool COMPACT>
__device__ __forceinline__ void load_poly_buf(const void *poly, uint32_t tid, typename F::E (&x)[32]) {
    load_src_buf<F, LOGN, COMPACT>(TableBuf(poly), 0, tid, x);
}

// Every lane of a wave holds the residue `o` of one of 64 consecutive containers starting at half-container `dst`: lane pairs store
// the value half and the zero half of each container (two instructions of 1 KiB consecutive bytes per wave), so the arithmetic that
// produced `o` runs on all 64 lanes instead of on the even ones of a one-half-container-per-lane kernel.
template <class F>
__device__ __forceinline__ void store_wave_containers(typename F::V16 *dst, typename F::E o) {
    using E = typename F::E;
    const uint32_t lane = threadIdx.x & 63;
    const E lo = __shfl(o, (int)(lane >> 1), 64), hi = __shfl(o, (int)(32 + (lane >> 1)), 64);
    __builtin_nontemporal_store(F::pack((lane & 1) ? (E)0 : lo), dst + lane);
    __builtin_nontemporal_store(F::pack((lane & 1) ? (E)0 : hi), dst + 64 + lane);
}

// Store the whole polynomial from LDS as full containers: consecutive lanes write consecutive 16-byte
// halves (even lane: {value, 0...}; odd lane: zeros), i.e. 1 KiB contiguous per wave instruction.
template <class F, int LOGN>
__device__ __forceinline__ void store_from_lds(char *__restrict__ poly, const typename F::E *lds, uint32_t tid) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    const uint32_t half = tid & 1, c0 = tid >> 1;
    const E *p = lds + c0 + (c0 >> 5);
    typename F::V16 *dst = reinterpret_cast<typename F::V16 *>(poly) + tid;
#pragma unroll 16
    for (int s = 0; s < 64; s++) {
        E v = p[s * (C::T / 2 + C::T / 64)];
        typename F::V16 o = F::pack(half ? (E)0 : v);
        __builtin_nontemporal_store(o, dst + (size_t)s * C::T);
    }
}

// The same store for callers that hold a live register array across it: a ROLLED loop of eight containers per trip (the fully
// scheduled form above keeps up to 64 addresses and values in flight and spills around a live array).
template <class F, int LOGN>
__device__ __forceinline__ void store_from_lds_rolled(char *__restrict__ poly, const typename F::E *lds, uint32_t tid) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    const uint32_t half = tid & 1, c0 = tid >> 1;
    const E *p = lds + c0 + (c0 >> 5);
    typename F::V16 *dst = reinterpret_cast<typename F::V16 *>(poly) + tid;
#pragma unroll 1
    for (int s0 = 0; s0 < 64; s0 += 8) {
        E v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = p[(s0 + j) * (C::T / 2 + C::T / 64)];
#pragma unroll
        for (int j = 0; j < 8; j++) __builtin_nontemporal_store(F::pack(half ? (E)0 : v[j]), dst + (size_t)(s0 + j) * C::T);
    }
}

// ---- whole-transform building blocks (data in registers, lds = workgroup scratch) ----------------------
// natural-order coefficients in pattern A  ->  NTT values in pattern Z, in [0, 4q)
// TWL: the non-uniform stages take their twiddles from `twl`, an LDS copy made by stage_twiddles<F, LOGN, true>.
// PRESYNC: the barrier that protects the exchange buffer from the PREVIOUS transform's last reads sits here, after the
// register-only first group, instead of at the end of the caller's loop body: the latest legal place, where it coincides with
// the transform's own first barrier (a wave that is ahead keeps computing instead of waiting early).
template <class F, int LOGN, bool TWL = false, bool PRESYNC = false, bool SUB = false>
__device__ __forceinline__ void fwd_core(typename F::E (&x)[32], typename F::E *lds, uint32_t tid, const Limb<F> &P,
                                         const typename F::TW *twl = nullptr, uint32_t pre = 1) {
    using C = NttCfg<LOGN>;
    const typename F::TW *t2 = TWL ? twl : P.tw;
    fwd_stages<F, LOGN, PatA<LOGN>, 4, 0, false, SUB>(x, tid, P.tw, P, pre);
    if constexpr (PRESYNC) __syncthreads();
    lds_put<PatA<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatM<LOGN>>(lds, tid, x);
    fwd_stages<F, LOGN, PatM<LOGN>, 4, 0, TWL, SUB>(x, tid, t2, P, pre);
    lds_put<PatM<LOGN>>(lds, tid, x);          // same slots this thread just read: no barrier needed before
    __syncthreads();
    lds_get<PatZ<LOGN>>(lds, tid, x);
    fwd_stages<F, LOGN, PatZ<LOGN>, C::REM - 1, 0, TWL, SUB>(x, tid, t2, P, pre);
}
// ---- two forward transforms under ONE modulus at once ------------------------------------------------------------------
// The key-switch and external-product kernels transform many digit polynomials under the same modulus.  Doing two of them
// in lock step shares every twiddle load (one load feeds two butterflies), every barrier and every LDS wait between the two,
// and doubles the independent work in flight per wave -- for the same register budget as holding the undecomposed limb
// beside one digit polynomial.  x0 / x1 use their own exchange buffers lds0 / lds1.
template <class F, int LOGN, class Pat, int KHI, int KLO>
__device__ __forceinline__ void fwd_stages2(typename F::E (&x0)[32], typename F::E (&x1)[32], uint32_t tid, const typename F::TW *__restrict__ tw,
                                            const Limb<F> &P) {
    const uint32_t base = Pat::TW_UNIFORM ? 0u : Pat::base(tid);
#pragma unroll
    for (int k = KHI; k >= KLO; k--) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + ((1u << (LOGN - 1 - b)) + (base >> (b + 1)));
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW w = p[Pat::off(r) >> (b + 1)];
            F::fwd_bfly(x0[r], x0[r | (1 << k)], w, P);
            F::fwd_bfly(x1[r], x1[r | (1 << k)], w, P);
        }
    }
}
// PRESYNC as in fwd_core: the barrier that ends the previous transforms' use of the exchange buffers sits after the first group.
template <class F, int LOGN, bool PRESYNC = false>
__device__ __forceinline__ void fwd_core2(typename F::E (&x0)[32], typename F::E (&x1)[32], typename F::E *lds0, typename F::E *lds1, uint32_t tid,
                                          const Limb<F> &P) {
    using C = NttCfg<LOGN>;
    fwd_stages2<F, LOGN, PatA<LOGN>, 4, 0>(x0, x1, tid, P.tw, P);
    if constexpr (PRESYNC) __syncthreads();
    lds_put<PatA<LOGN>>(lds0, tid, x0);
    lds_put<PatA<LOGN>>(lds1, tid, x1);
    __syncthreads();
    lds_get<PatM<LOGN>>(lds0, tid, x0);
    lds_get<PatM<LOGN>>(lds1, tid, x1);
    fwd_stages2<F, LOGN, PatM<LOGN>, 4, 0>(x0, x1, tid, P.tw, P);
    lds_put<PatM<LOGN>>(lds0, tid, x0);
    lds_put<PatM<LOGN>>(lds1, tid, x1);
    __syncthreads();
    lds_get<PatZ<LOGN>>(lds0, tid, x0);
    lds_get<PatZ<LOGN>>(lds1, tid, x1);
    fwd_stages2<F, LOGN, PatZ<LOGN>, C::REM - 1, 0>(x0, x1, tid, P.tw, P);
}

// NTT values in pattern Z, in [0, 2q)  ->  coefficients in pattern A, in [0, 2q), scaled by the (ninv..) constants
template <class F, int LOGN, bool TWL = false, bool PRESYNC = false, bool SUB = false>
__device__ __forceinline__ void inv_core(typename F::E (&x)[32], typename F::E *lds, uint32_t tid, const Limb<F> &P,
                                         typename F::E ninv, typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s,
                                         const typename F::TW *twl = nullptr, uint32_t pre = 1) {
    using C = NttCfg<LOGN>;
    const typename F::TW *t2 = TWL ? twl : P.itw;
    inv_stages<F, LOGN, PatZ<LOGN>, 0, 4, TWL, SUB>(x, tid, t2, P, pre);
    F::regroup(x, P.q, P.qinv);
    if constexpr (PRESYNC) __syncthreads();
    lds_put<PatZ<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatY<LOGN>>(lds, tid, x);
    inv_stages<F, LOGN, PatY<LOGN>, 0, 4, TWL, SUB>(x, tid, t2, P, pre);
    F::regroup(x, P.q, P.qinv);
    lds_put<PatY<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatA<LOGN>>(lds, tid, x);
    if constexpr (SUB) {   // a block of a larger transform: bit LOGN-1 is an ordinary stage, the scaling belongs to the last pass
        inv_stages<F, LOGN, PatA<LOGN>, 5 - C::REM, 4, false, true>(x, tid, P.itw, P, pre);
        F::regroup(x, P.q, P.qinv);
    } else {
        // index bits [10, LOGN-1) <-> r-bits [5-REM, 4) ; bit LOGN-1 <-> r-bit 4 is the scaled last stage
        inv_stages<F, LOGN, PatA<LOGN>, 5 - C::REM, 3>(x, tid, P.itw, P);
        inv_last_stage<F>(x, P.q, P.q2, ninv, ninv_s, ninvw, ninvw_s);
    }
}

// two inverse transforms in lock step (see fwd_core2)
template <class F, int LOGN, class Pat, int KLO, int KHI>
__device__ __forceinline__ void inv_stages2(typename F::E (&x0)[32], typename F::E (&x1)[32], uint32_t tid, const typename F::TW *__restrict__ itw,
                                            const Limb<F> &P) {
    const uint32_t base = Pat::TW_UNIFORM ? 0u : Pat::base(tid);
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = itw + ((1u << (LOGN - 1 - b)) + (base >> (b + 1)));
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW w = p[Pat::off(r) >> (b + 1)];
            F::inv_bfly(x0[r], x0[r | (1 << k)], w, P);
            F::inv_bfly(x1[r], x1[r | (1 << k)], w, P);
        }
    }
}
template <class F, int LOGN, bool PRESYNC = false>
__device__ __forceinline__ void inv_core2(typename F::E (&x0)[32], typename F::E (&x1)[32], typename F::E *lds0, typename F::E *lds1, uint32_t tid,
                                          const Limb<F> &P, typename F::E ninv, typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s) {
    using C = NttCfg<LOGN>;
    inv_stages2<F, LOGN, PatZ<LOGN>, 0, 4>(x0, x1, tid, P.itw, P);
    F::regroup(x0, P.q, P.qinv); F::regroup(x1, P.q, P.qinv);
    if constexpr (PRESYNC) __syncthreads();
    lds_put<PatZ<LOGN>>(lds0, tid, x0);
    lds_put<PatZ<LOGN>>(lds1, tid, x1);
    __syncthreads();
    lds_get<PatY<LOGN>>(lds0, tid, x0);
    lds_get<PatY<LOGN>>(lds1, tid, x1);
    inv_stages2<F, LOGN, PatY<LOGN>, 0, 4>(x0, x1, tid, P.itw, P);
    F::regroup(x0, P.q, P.qinv); F::regroup(x1, P.q, P.qinv);
    lds_put<PatY<LOGN>>(lds0, tid, x0);
    lds_put<PatY<LOGN>>(lds1, tid, x1);
    __syncthreads();
    lds_get<PatA<LOGN>>(lds0, tid, x0);
    lds_get<PatA<LOGN>>(lds1, tid, x1);
    inv_stages2<F, LOGN, PatA<LOGN>, 5 - C::REM, 3>(x0, x1, tid, P.itw, P);
    inv_last_stage<F>(x0, P.q, P.q2, ninv, ninv_s, ninvw, ninvw_s);
    inv_last_stage<F>(x1, P.q, P.q2, ninv, ninv_s, ninvw, ninvw_s);
}

// =========================================================================================================
// Kernels.  grid.x = batch * L workgroups; workgroup p handles polynomial p (limb p % L).
// =========================================================================================================
template <class F, int LOGN>
__global__ void __launch_bounds__(NttCfg<LOGN>::T)
ntt_forward_kernel(char *__restrict__ data, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    char *poly = data + (size_t)p * (C::N * 32);
    E x[32];
    load_A<F, LOGN>(poly, tid, x);
    fwd_core<F, LOGN>(x, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    lds_put<PatZ<LOGN>>(lds, tid, x);      // the slots this thread read last: no barrier needed before
    __syncthreads();
    store_from_lds<F, LOGN>(poly, lds, tid);
}

template <class F, int LOGN>
__global__ void __launch_bounds__(NttCfg<LOGN>::T)
ntt_inverse_kernel(char *__restrict__ data, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    char *poly = data + (size_t)p * (C::N * 32);
    E x[32];
    load_A<F, LOGN>(poly, tid, x);
    lds_put<PatA<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatZ<LOGN>>(lds, tid, x);
    inv_core<F, LOGN>(x, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::canon_inv(x[r], P.q);
    lds_put<PatA<LOGN>>(lds, tid, x);
    __syncthreads();
    store_from_lds<F, LOGN>(poly, lds, tid);
}

// NTTEngine::multiply in one launch: r = INTT(NTT(a) .* NTT(b)); HBM traffic = read a + read b + write r.
// SQUARE: b is a (the host passes the flag when the operand pointers are equal): one load, one forward transform, 2*S of traffic.
// bcast != 0: b holds ONE RNS polynomial ([L][n]) that is multiplied into every element of the batch (served from L2 after its first
// use: 2*S of HBM traffic per product).
// COMPACT_OUT: the result goes to a compact workspace polynomial (load_A_compact) -- pieces of the fused multiply + relinearise.
template <class F, int LOGN, int MINW = 1, bool SQUARE = false, bool COMPACT_OUT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_multiply_kernel(char *res, const char *a, const char *b,      // no __restrict__: res may alias a and / or b (in-place product)
                    const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t bcast) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const uint32_t limb = p % L;
    const Limb<F> P = limbs[limb];
    const size_t off = (size_t)p * (C::N * 32);
    E x[32];
    load_A<F, LOGN>(a + off, tid, x);
    if constexpr (SQUARE) {
        fwd_core<F, LOGN>(x, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) { const E c = F::canon_fwd(x[r], P.q, P.q2, P.qinv); x[r] = F::pw_mul(c, c, P.q, P.qinv); }
    } else {
        E y[32];
        load_A<F, LOGN>(b + (size_t)(bcast ? limb : p) * (C::N * 32), tid, y);   // issued before a's butterflies: b's HBM latency hides under them
        fwd_core<F, LOGN>(x, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);   // canonical: keeps x*y < q*2^W
        __syncthreads();                   // all Z-pattern reads of a are done before b overwrites the slots
        fwd_core<F, LOGN>(y, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::pw_mul(x[r], y[r], P.q, P.qinv);   // [0,2q), carries 2^-W
    }
    inv_core<F, LOGN>(x, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::canon_inv(x[r], P.q);
    if constexpr (COMPACT_OUT) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(res) + (size_t)p * C::N, tid, x);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, x);
        __syncthreads();
        store_from_lds<F, LOGN>(res + off, lds, tid);
    }
}

// ---- transforms larger than the LDS range: N = 2^(LOGN + k), k = 1..3 --------------------------------------------------------------
// The top k stages (forward) / last k stages (inverse) run as one register-only pass over global memory (word_pass_kernel below);
// the 2^k blocks of 2^LOGN consecutive coefficients are then independent sub-transforms, each one workgroup of the LDS-resident
// machinery above with the big transform's twiddles (SUB = true, pre = 2^k + block).  HBM traffic: 4 S per transform, 9 S per fused
// multiply (top(a) + top(b) into the workspace 4 S, sub-multiply 3 S, last pass 2 S).
// grid.x = polys << k: workgroup g handles block g & (2^k - 1) of polynomial g >> k.
enum { SUB_FORWARD = 0, SUB_INVERSE = 1, SUB_MULTIPLY = 2 };
template <class F, int LOGN, int MODE, int MINW = 1>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_sub_kernel(char *res, const char *a, const char *b, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t k) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x >> k, blk = blockIdx.x & ((1u << k) - 1), pre = (1u << k) + blk;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)blockIdx.x * (C::N * 32);          // polynomial p starts at p << (LOGN + k) containers
    E x[32];
    load_A<F, LOGN>(a + off, tid, x);
    if constexpr (MODE == SUB_FORWARD) {
        fwd_core<F, LOGN, false, false, true>(x, lds, tid, P, nullptr, pre);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
        lds_put<PatZ<LOGN>>(lds, tid, x);
    } else {
        if constexpr (MODE == SUB_INVERSE) {
            lds_put<PatA<LOGN>>(lds, tid, x);
            __syncthreads();
            lds_get<PatZ<LOGN>>(lds, tid, x);
        } else {
            E y[32];
            load_A<F, LOGN>(b + off, tid, y);
            fwd_core<F, LOGN, false, false, true>(x, lds, tid, P, nullptr, pre);
#pragma unroll
            for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
            __syncthreads();
            fwd_core<F, LOGN, false, false, true>(y, lds, tid, P, nullptr, pre);
#pragma unroll
            for (int r = 0; r < 32; r++) x[r] = F::pw_mul(x[r], y[r], P.q, P.qinv);   // carries 2^-W until the last pass (ninv_r constants)
        }
        inv_core<F, LOGN, false, false, true>(x, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s, nullptr, pre);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::canon_inv(x[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, x);
    }
    __syncthreads();
    store_from_lds<F, LOGN>(res + off, lds, tid);
}

// One register-only pass over R <= 3 stages of a transform of 2^log_n coefficients on the field type: FWD: the top R stages (index
// bits log_n-1 .. log_n-R), natural-order canonical input; !FWD: the last R stages (the same bits, ascending) with the n^-1 scaling
// folded into the final butterfly (rconst: the constants that also absorb the 2^-W of a fused pointwise product).  Canonical
// residues in and out (full containers).  src may differ from dst.
// Lanes work in pairs: lanes 2c and 2c+1 both load the low word of container c (one request) and run the same butterflies; the even
// lane then stores the value half of each output container and the odd lane the zero half, so a wave instruction writes 1 KiB of
// consecutive bytes (like store_from_lds) instead of every other 16 bytes of 2 KiB: 5.5 instead of 4.1 TB/s on the pass, whose
// arithmetic is far from binding (12 butterflies per 8 containers; one column per lane with the outputs exchanged by shuffles was
// measured for the 8-byte fields and is no faster).  grid = (2^(log_n - R + 1) / 256, polys).
template <class F, int R, bool FWD>
__global__ void __launch_bounds__(256)
word_pass_kernel(typename F::V16 *dst, const typename F::V16 *src, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t log_n, uint32_t rconst) {
    using E = typename F::E;
    const uint32_t g = blockIdx.x * 256 + threadIdx.x, u = g >> 1, half = g & 1;    // u < 2^log_n >> R by construction of the grid
    const uint32_t p = blockIdx.y;
    const Limb<F> P = limbs[p % L];
    const uint32_t b_lo = log_n - R;
    const typename F::V16 *in = src + ((size_t)p << (log_n + 1)); typename F::V16 *out = dst + ((size_t)p << (log_n + 1));
    E x[1 << R];
#pragma unroll
    for (int k = 0; k < (1 << R); k++) x[k] = F::load_low(in + 2 * ((size_t)u + ((size_t)k << b_lo)));
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int pos = FWD ? R - 1 - j : j;                                  // k-bit of this stage; index bit b_lo + pos
        if (!FWD && pos == R - 1) {                                            // bit log_n-1: single twiddle itw[1], with the scaling
#pragma unroll
            for (int k = 0; k < (1 << (R - 1)); k++)
                F::inv_last(x[k], x[k | (1 << (R - 1))], P.q, P.q2, rconst ? P.ninv_r : P.ninv, rconst ? P.ninv_r_s : P.ninv_s,
                            rconst ? P.ninvw_r : P.ninvw, rconst ? P.ninvw_r_s : P.ninvw_s);
            continue;
        }
#pragma unroll
        for (int hh = 0; hh < (1 << (R - 1)); hh++) {
            const int k = ((hh >> pos) << (pos + 1)) | (hh & ((1 << pos) - 1));
            // index bit b = b_lo + pos: twiddle (n >> (b+1)) + (i >> (b+1)) = 2^(R-1-pos) + (k >> (pos+1)): independent of the lane
            const typename F::TW w = (FWD ? P.tw : P.itw)[(1u << (R - 1 - pos)) + (k >> (pos + 1))];
            if (FWD) F::fwd_bfly(x[k], x[k | (1 << pos)], w, P);
            else F::inv_bfly(x[k], x[k | (1 << pos)], w, P);
        }
    }
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
        const E v = FWD ? F::canon_fwd(x[k], P.q, P.q2, P.qinv) : F::canon_inv(F::regroup1(x[k], P.q, P.qinv), P.q);
        __builtin_nontemporal_store(F::pack(half ? (E)0 : v), out + 2 * ((size_t)u + ((size_t)k << b_lo)) + half);
    }
}

// r = a0 (*) b1 + a1 (*) b0 in one launch (the c1 term of the tensor product) for configurations whose four
// transformed operands do not fit the register file (8-byte residues at N = 2^14): at most three arrays are live.
// HBM traffic = read 4 polynomials + write 1.
template <class F, int LOGN, int MINW = 1, bool COMPACT_OUT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_mac2_kernel(char *__restrict__ res, const char *__restrict__ a0, const char *__restrict__ b1,
                const char *__restrict__ a1, const char *__restrict__ b0, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)p * (C::N * 32);
    E x[32], y[32], acc[32];
    load_A<F, LOGN>(a0 + off, tid, x);
    load_A<F, LOGN>(b1 + off, tid, y);
    fwd_core<F, LOGN>(x, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    __syncthreads();
    fwd_core<F, LOGN>(y, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 32; r++) acc[r] = F::pw_mul(x[r], y[r], P.q, P.qinv);
    load_A<F, LOGN>(a1 + off, tid, x);
    load_A<F, LOGN>(b0 + off, tid, y);
    __syncthreads();
    fwd_core<F, LOGN>(x, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    __syncthreads();
    fwd_core<F, LOGN>(y, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 32; r++) acc[r] = F::pw_add(acc[r], F::pw_mul(x[r], y[r], P.q, P.qinv), P.q, P.q2);
    inv_core<F, LOGN>(acc, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) acc[r] = F::canon_inv(acc[r], P.q);
    if constexpr (COMPACT_OUT) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(res) + (size_t)p * C::N, tid, acc);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, acc);
        __syncthreads();
        store_from_lds<F, LOGN>(res + off, lds, tid);
    }
}

// ---- tensor product in TWO launches for configurations whose four transformed operands do not fit the register file (8-byte
// residues at N = 2^14): 7 transforms instead of the 11 of multiply(c0) + multiply(c2) + mac2(c1).
// Launch 1 (ntt_forward_compact_kernel, grid (polys, 2)): NTT(b0), NTT(b1) into compact workspace polynomials, in REGISTER order
// (slot tid + r*T holds register r of thread tid after the last forward stage, still lazy) -- only ever multiplied pointwise against
// registers of the same thread of launch 2, so no ordering is needed, and 8 instead of 32 bytes per coefficient.
// Launch 2 (ntt_ct_a_kernel): NTT(a0); c0 = INTT(A0 . B0); t = A0 . B1; NTT(a1); c1 = INTT(t + A1 . B0); c2 = INTT(A1 . B1).  At most two
// arrays are live across any transform, three in the pointwise phases.  HBM traffic: 2 S + 2 S/4 written, then 2 S + 4 S/4 read and
// 3 S (or 3 S/4, compact outputs) written: 7.5 S (5.25 S) against 11 S (8.75 S).
template <class F, int LOGN, int MINW = 1>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_forward_compact_kernel(typename F::E *__restrict__ w0, typename F::E *__restrict__ w1, const char *__restrict__ b0,
                           const char *__restrict__ b1, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    E x[32];
    load_A<F, LOGN>((blockIdx.y ? b1 : b0) + (size_t)p * (C::N * 32), tid, x);
    fwd_core<F, LOGN>(x, lds, tid, P);
    store_A_compact<F, LOGN>((blockIdx.y ? w1 : w0) + (size_t)p * C::N, tid, x);
}
template <class F, int LOGN, bool COMPACT_OUT>
__device__ __forceinline__ void ct_store(char *__restrict__ c, size_t p, typename F::E *lds, uint32_t tid, const typename F::E (&x)[32]) {
    using C = NttCfg<LOGN>;
    if constexpr (COMPACT_OUT) {
        store_A_compact<F, LOGN>(reinterpret_cast<typename F::E *>(c) + p * C::N, tid, x);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, x);      // the slots this thread read last
        __syncthreads();
        store_from_lds_rolled<F, LOGN>(c + p * (C::N * 32), lds, tid);
    }
}
template <class F, int LOGN, int MINW = 1, bool COMPACT_OUT = false, bool EARLY = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_ct_a_kernel(char *__restrict__ c0, char *__restrict__ c1, char *__restrict__ c2, const char *__restrict__ a0, const char *__restrict__ a1,
                const typename F::E *__restrict__ w0, const typename F::E *__restrict__ w1, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)p * (C::N * 32);
    const E *B0 = w0 + (size_t)p * C::N, *B1 = w1 + (size_t)p * C::N;
    E X[32], Y[32], T[32];
    // CT_FENCE: the compiler may not move the next phase's 32 loads above the transform / store before it (they would be a third
    // live array across it: the kernel spills 600+ bytes per lane without the fences)
#define CT_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)
    load_poly_buf<F, LOGN, false>(a0 + off, tid, X);
    fwd_core<F, LOGN>(X, lds, tid, P);
    CT_FENCE();
    load_poly_buf<F, LOGN, true>(B0, tid, Y);
#pragma unroll
    for (int r = 0; r < 32; r++) { X[r] = F::canon_fwd(X[r], P.q, P.q2, P.qinv); T[r] = F::pw_mul(X[r], Y[r], P.q, P.qinv); }
    if constexpr (EARLY) load_poly_buf<F, LOGN, true>(B1, tid, Y);     // in flight under the inverse transform of c0 (a third live array)
    else CT_FENCE();
    inv_core<F, LOGN>(T, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) T[r] = F::canon_inv(T[r], P.q);
    ct_store<F, LOGN, COMPACT_OUT>(c0, p, lds, tid, T);
    CT_FENCE();
    if constexpr (!EARLY) load_poly_buf<F, LOGN, true>(B1, tid, Y);
#pragma unroll
    for (int r = 0; r < 32; r++) T[r] = F::pw_mul(X[r], Y[r], P.q, P.qinv);          // A0 . B1
    CT_FENCE();
    load_poly_buf<F, LOGN, false>(a1 + off, tid, X);
    __syncthreads();
    fwd_core<F, LOGN>(X, lds, tid, P);
    CT_FENCE();
    load_poly_buf<F, LOGN, true>(B0, tid, Y);
#pragma unroll
    for (int r = 0; r < 32; r++) { X[r] = F::canon_fwd(X[r], P.q, P.q2, P.qinv); T[r] = F::pw_add(T[r], F::pw_mul(X[r], Y[r], P.q, P.qinv), P.q, P.q2); }
    if constexpr (EARLY) load_poly_buf<F, LOGN, true>(B1, tid, Y);
    else CT_FENCE();
    inv_core<F, LOGN>(T, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) T[r] = F::canon_inv(T[r], P.q);
    ct_store<F, LOGN, COMPACT_OUT>(c1, p, lds, tid, T);
    CT_FENCE();
    if constexpr (!EARLY) load_poly_buf<F, LOGN, true>(B1, tid, Y);
#pragma unroll
    for (int r = 0; r < 32; r++) T[r] = F::pw_mul(X[r], Y[r], P.q, P.qinv);          // A1 . B1
    CT_FENCE();
    __syncthreads();
    inv_core<F, LOGN>(T, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) T[r] = F::canon_inv(T[r], P.q);
    ct_store<F, LOGN, COMPACT_OUT>(c2, p, lds, tid, T);
#undef CT_FENCE
}

// FHEContext::multiply tensor product in one launch (src/fhe.cu:199-218): 4 forward + 3 inverse transforms,
// HBM traffic = read 4 polynomials + write 3.
// SQUARE: (b0, b1) is (a0, a1) (the host passes the flag when the operand pointers are equal -- squaring a ciphertext): two loads and
// two forward transforms instead of four, c1 = 2 a0 a1; 5*S of traffic instead of 7*S.
// COMPACT_C2: all three outputs go to compact workspace polynomials (see load_A_compact) instead of container buffers: the fused
// multiply + relinearise, whose key-switch kernel reads them back as digit source (c2) and addends (c0, c1).
template <class F, int LOGN, bool SQUARE = false, bool COMPACT_C2 = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, (sizeof(typename F::E) == 8 && NttCfg<LOGN>::T <= 256) ? 2 : 1)   // 8-byte residues: two workgroups per CU
ntt_ct_multiply_kernel(char *__restrict__ c0, char *__restrict__ c1, char *__restrict__ c2,
                       const char *__restrict__ a0, const char *__restrict__ a1,
                       const char *__restrict__ b0, const char *__restrict__ b1,
                       const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)p * (C::N * 32);
    E A0[32], A1[32], B0[32];
    load_A<F, LOGN>(a0 + off, tid, A0);
    load_A<F, LOGN>(a1 + off, tid, A1);
    fwd_core<F, LOGN>(A0, lds, tid, P);
    if constexpr (SQUARE) {
        __syncthreads();
        fwd_core<F, LOGN>(A1, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const E u0 = F::canon_fwd(A0[r], P.q, P.q2, P.qinv), u1 = F::canon_fwd(A1[r], P.q, P.q2, P.qinv);
            A0[r] = F::pw_mul(u0, u0, P.q, P.qinv);
            A1[r] = F::pw_mul2(u0, u1, u1, u0, P.q, P.q2, P.qinv);   // 2 a0 a1
            B0[r] = F::pw_mul(u1, u1, P.q, P.qinv);
        }
    } else {
        E B1[32];
        load_A<F, LOGN>(b0 + off, tid, B0);
        __syncthreads();
        fwd_core<F, LOGN>(A1, lds, tid, P);
        load_A<F, LOGN>(b1 + off, tid, B1);
        __syncthreads();
        fwd_core<F, LOGN>(B0, lds, tid, P);
        __syncthreads();
        fwd_core<F, LOGN>(B1, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) {
            E u0 = F::canon_fwd(A0[r], P.q, P.q2, P.qinv), u1 = F::canon_fwd(A1[r], P.q, P.q2, P.qinv);   // canonical a-side
            E v0 = B0[r], v1 = B1[r];                                                             // lazy b-side (< 4q)
            A0[r] = F::pw_mul(u0, v0, P.q, P.qinv);                  // [0,2q)
            A1[r] = F::pw_mul2(u0, v1, u1, v0, P.q, P.q2, P.qinv);   // a0*b1 + a1*b0 with one shared reduction on the 32-bit field
            B0[r] = F::pw_mul(u1, v1, P.q, P.qinv);
        }
    }
    inv_core<F, LOGN>(A0, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) A0[r] = F::canon_inv(A0[r], P.q);
    if constexpr (COMPACT_C2) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(c0) + (size_t)p * C::N, tid, A0);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, A0);
        __syncthreads();
        store_from_lds<F, LOGN>(c0 + off, lds, tid);
    }
    __syncthreads();
    inv_core<F, LOGN>(A1, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) A1[r] = F::canon_inv(A1[r], P.q);
    if constexpr (COMPACT_C2) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(c1) + (size_t)p * C::N, tid, A1);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, A1);
        __syncthreads();
        store_from_lds<F, LOGN>(c1 + off, lds, tid);
    }
    __syncthreads();
    inv_core<F, LOGN>(B0, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) B0[r] = F::canon_inv(B0[r], P.q);
    if constexpr (COMPACT_C2) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(c2) + (size_t)p * C::N, tid, B0);   // pattern A: consecutive lanes, consecutive words
    } else {
        lds_put<PatA<LOGN>>(lds, tid, B0);
        __syncthreads();
        store_from_lds<F, LOGN>(c2 + off, lds, tid);
    }
}

// Element-wise kernels over [batch][L][n] containers of word-sized residues: one 16-byte half-container per lane.
// op 0: r = a*b mod q (plain product in the NTT domain); 1: a+b; 2: a-b.
template <class F, int OP>
__global__ void __launch_bounds__(256)
ew_kernel(typename F::V16 *r, const typename F::V16 *a, const typename F::V16 *b,      // no __restrict__: r may be a or b (in-place add / sub / product)
          const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t log_n, size_t halves) {
    using E = typename F::E;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < halves; g += stride) {
        E o = 0;
        if (!(g & 1)) {
            const Limb<F> &P = limbs[(uint32_t)((g >> (log_n + 1)) % L)];
            E x = F::load_low(a + g), y = F::load_low(b + g), q = P.q;
            if (OP == 0) o = F::ew_mul(x, y, P);
            else if (OP == 1) o = F::ew_add(x, y, q);
            else o = F::ew_sub(x, y, q);
        }
        __builtin_nontemporal_store(F::pack(o), r + g);
    }
}

// Canonical-input scan: flags any container whose value is >= q or whose upper words are not zero.
template <class F>
__global__ void __launch_bounds__(256)
check_kernel(const typename F::V16 *__restrict__ a, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t log_n,
             size_t halves, uint32_t *__restrict__ flag) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < halves; g += stride) {
        typename F::V16 v = a[g];
        if (g & 1) bad |= F::any_nonzero(v);
        else bad |= F::upper_nonzero(v) || F::ge(F::low(v), limbs[(uint32_t)((g >> (log_n + 1)) % L)].q);
    }
    if (bad) atomicOr(flag, 1u);
}

// ---- fused key switching (relinearisation) -----------------------------------------------------------------------------
// Key tables in the kernel's own register order ("packed"): for level jk and limb i, element (chunk c, thread tid, e) holds
// KEY_ntt[i][tid*32 + c*VPL + e] * 2^W mod q_i, VPL = 16 / sizeof(E) values per 16-byte lane load, so a wave instruction
// reads 1 KiB contiguous and pw_mul's 2^-W cancels.  Tables for one engine total 2 * L*K * L * N * sizeof(E) bytes
// (2 MiB at N = 8192, L = 4, K = 2, F32) and stay L2-resident across the batch.
template <class F>
__global__ void __launch_bounds__(256)
pack_keys_kernel(typename F::E *__restrict__ packed, const typename F::V16 *__restrict__ keys_ntt, const Limb<F> *__restrict__ limbs,
                 uint32_t L, uint32_t log_n, uint32_t num_keys) {
    using E = typename F::E;
    constexpr uint32_t VPL = 16 / sizeof(E);
    const uint32_t n = 1u << log_n, T = n >> 5;
    const size_t total = (size_t)num_keys * L * n, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint32_t x = (uint32_t)(g & (n - 1));                  // NTT-domain index = tid*32 + r
        const size_t poly = g >> log_n;                              // jk * L + i
        const Limb<F> &P = limbs[(uint32_t)(poly % L)];
        const uint32_t tid = x >> 5, r = x & 31, c = r / VPL, e = r % VPL;
        E v = F::load_low(keys_ntt + g * 2);
        packed[poly * n + ((size_t)c * T + tid) * VPL + e] = F::to_pw_operand(v, P);
    }
}

// One workgroup per (ciphertext b, limb i): for every limb j of c2 and every digit k, the digit polynomial is formed in
// registers, transformed under q_i, multiplied with both key halves and accumulated in the NTT domain; two inverse
// transforms and the additions to c0, c1 finish the job.  HBM traffic per ciphertext: L reads of c2 + read/write of c0, c1
// = (L + 4) * S bytes (re-reads of c2 by the L workgroups of one ciphertext mostly hit the Infinity Cache).
// SPLIT = false: one workgroup per (ciphertext b, limb i) accumulates both key halves (4 live arrays per thread; 4-byte residues).
// SPLIT = true : two workgroups per (b, i), one per key half (3 live arrays; 8-byte residues, whose four arrays would not fit the
//                register file); the digit transforms are then computed twice, which is still far cheaper than the general
//                composition's container-sized digit workspace.
// TWL = true: the forward (then the inverse) twiddle table of limb i is copied into LDS once per workgroup; the 2*L*K + 2
// transforms then take their non-uniform twiddles from LDS (~100 cycles, conflict-free in the permuted order) instead of L2
// (500+ cycles), which these register-starved kernels (2 waves per SIMD) cannot hide.  N * sizeof(TW) more LDS per workgroup.
template <class F, int LOGN, int MINW = 1, bool SPLIT = false, bool TWL = false, bool COMPACT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_keyswitch_kernel(char *c0, char *c1, const char *__restrict__ c2, const char *add0, const char *add1,   // add* = c* (in place) unless COMPACT
                     const typename F::E *__restrict__ kb, const typename F::E *__restrict__ ka,
                     const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    static_assert(!(TWL && SPLIT), "LDS twiddles are wired into the one-workgroup-per-limb form only");
    __shared__ E lds[C::LDS_ELEMS];
    __shared__ typename F::TW twl[TWL ? C::N : 1];
    // XCD-aware workgroup -> (ciphertext, limb[, half]) map: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8
    // names the L2 a workgroup shares), and the workgroups of one ciphertext all re-read the same c2, so they are given block
    // indices that are congruent mod 8 and close together: the re-reads then hit one XCD's L2 instead of crossing the
    // fabric.  Placement only changes speed, never results.
    constexpr uint32_t H = SPLIT ? 2 : 1;
    const uint32_t LH = L * H;
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * LH)) * (8 * LH);
    uint32_t b, u;
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / LH); u = s % LH; }
    else { b = bid / LH; u = bid % LH; }
    const uint32_t i = u / H, half = u % H;
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    if constexpr (!SPLIT) {
        E acc0[32], acc1[32], x[32], d[32];
#pragma unroll
        for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
        if constexpr (TWL) { stage_twiddles<F, LOGN, true>(twl, P.tw, tid); __syncthreads(); }
        for (uint32_t j = 0; j < L; j++) {
            load_src<F, LOGN, COMPACT>(c2, (size_t)b * L + j, tid, x);
            for (uint32_t k = 0; k < K; k++) {
#pragma unroll
                for (int r = 0; r < 32; r++) d[r] = F::digit(x[r], k * w, w);
                fwd_core<F, LOGN, TWL, true>(d, lds, tid, P, twl);   // PRESYNC: the previous digit's Z-pattern reads are done before this exchange
                const size_t tbl = ((size_t)(j * K + k) * L + i) * C::N;
                const VecE *pb = reinterpret_cast<const VecE *>(kb + tbl) + tid, *pa = reinterpret_cast<const VecE *>(ka + tbl) + tid;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const VecE vb = pb[c * C::T], va = pa[c * C::T];
#pragma unroll
                    for (int e = 0; e < VPL; e++) {
                        const int r = c * VPL + e;
                        acc0[r] = F::pw_add(acc0[r], F::pw_mul(vb[e], d[r], P.q, P.qinv), P.q, P.q2);
                        acc1[r] = F::pw_add(acc1[r], F::pw_mul(va[e], d[r], P.q, P.qinv), P.q, P.q2);
                    }
                }
            }
        }
        // acc0 -> coefficient domain, + c0.  With LDS twiddles every forward-table read must be over before the table is swapped;
        // otherwise the barrier before the first exchange is inv_core's PRESYNC.
        if constexpr (TWL) { __syncthreads(); stage_twiddles<F, LOGN, false>(twl, P.itw, tid); __syncthreads(); }
        inv_core<F, LOGN, TWL, !TWL>(acc0, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s, twl);
        load_src<F, LOGN, COMPACT>(add0, p, tid, x);
#pragma unroll
        for (int r = 0; r < 32; r++) acc0[r] = F::ew_add(F::canon_inv(acc0[r], P.q), x[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc0);
        __syncthreads();
        store_from_lds<F, LOGN>(c0 + (size_t)p * (C::N * 32), lds, tid);
        __syncthreads();
        inv_core<F, LOGN, TWL>(acc1, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s, twl);
        load_src<F, LOGN, COMPACT>(add1, p, tid, x);
#pragma unroll
        for (int r = 0; r < 32; r++) acc1[r] = F::ew_add(F::canon_inv(acc1[r], P.q), x[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc1);
        __syncthreads();
        store_from_lds<F, LOGN>(c1 + (size_t)p * (C::N * 32), lds, tid);
    } else {
        const E *keys = half ? ka : kb;
        char *dst = half ? c1 : c0;
        E acc[32], d[32];
#pragma unroll
        for (int r = 0; r < 32; r++) acc[r] = 0;
        for (uint32_t j = 0; j < L; j++) {
            for (uint32_t k = 0; k < K; k++) {
                // the c2 limb is re-read per digit (L2 / Infinity-Cache hits after the first) rather than held in 64 more VGPRs
                load_src<F, LOGN, COMPACT>(c2, (size_t)b * L + j, tid, d);
#pragma unroll
                for (int r = 0; r < 32; r++) d[r] = F::digit(d[r], k * w, w);
                fwd_core<F, LOGN, false, true>(d, lds, tid, P);
                __builtin_amdgcn_sched_barrier(0);   // keep the 16 key loads (64 VGPRs) from being hoisted into the transform
                const VecE *pk = reinterpret_cast<const VecE *>(keys + ((size_t)(j * K + k) * L + i) * C::N) + tid;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const VecE v = pk[c * C::T];
#pragma unroll
                    for (int e = 0; e < VPL; e++) {
                        const int r = c * VPL + e;
                        acc[r] = F::pw_add(acc[r], F::pw_mul(v[e], d[r], P.q, P.qinv), P.q, P.q2);
                    }
                }
            }
        }
        F::regroup(acc, P.q, P.qinv);              // floating-point sums of L*K products: back below q before the inverse
        inv_core<F, LOGN, false, true>(acc, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
        load_src<F, LOGN, COMPACT>(half ? add1 : add0, p, tid, d);
#pragma unroll
        for (int r = 0; r < 32; r++) acc[r] = F::ew_add(F::canon_inv(acc[r], P.q), d[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc);
        __syncthreads();
        store_from_lds<F, LOGN>(dst + (size_t)p * (C::N * 32), lds, tid);
    }
}

// Key switching for the 8-byte residues in ONE workgroup per (ciphertext, limb) with THREE live arrays: both accumulators and the digit
// polynomial; the c2 limb is re-read for every digit (compact workspace: N * 8 bytes from L2) instead of being held.  Against the SPLIT
// form of ntt_keyswitch_kernel (two workgroups per limb, one per key half, every digit transform computed twice) this halves the
// transforms; the register file is exceeded by what the compiler parks in scratch around the transforms.
template <class F, int LOGN, int MINW = 1, bool COMPACT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_keyswitch3_kernel(char *c0, char *c1, const char *__restrict__ c2, const char *add0, const char *add1,
                      const typename F::E *__restrict__ kb, const typename F::E *__restrict__ ka,
                      const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * L)) * (8 * L);
    uint32_t b, i;
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / L); i = s % L; }
    else { b = bid / L; i = bid % L; }
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    E acc0[32], acc1[32], d[32];
#pragma unroll
    for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
    const TableBuf KB(kb), KA(ka), C2(c2 + (size_t)b * L * (C::N * (COMPACT ? sizeof(E) : 32)));   // c2 of this ciphertext
    const uint32_t voff = tid * 16;
    for (uint32_t j = 0; j < L; j++) {
        for (uint32_t k = 0; k < K; k++) {
            load_src_buf<F, LOGN, COMPACT>(C2, j, tid, d);
#pragma unroll
            for (int r = 0; r < 32; r++) d[r] = F::digit(d[r], k * w, w);
            fwd_core<F, LOGN, false, true>(d, lds, tid, P);
            __builtin_amdgcn_sched_barrier(0);   // keep the key loads out of the transform
            const uint32_t tbl = (uint32_t)((((size_t)(j * K + k) * L + i) * C::N) * sizeof(E));   // byte offset of the row (< 4 GiB: host)
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const VecE vb = KB.template load16<VecE>(voff, tbl + c * C::T * 16), va = KA.template load16<VecE>(voff, tbl + c * C::T * 16);
#pragma unroll
                for (int e = 0; e < VPL; e++) {
                    const int r = c * VPL + e;
                    acc0[r] = F::pw_add(acc0[r], F::pw_mul(vb[e], d[r], P.q, P.qinv), P.q, P.q2);
                    acc1[r] = F::pw_add(acc1[r], F::pw_mul(va[e], d[r], P.q, P.qinv), P.q, P.q2);
                }
                if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four chunks of key loads in flight at a time
            }
        }
    }
    F::regroup(acc0, P.q, P.qinv);
    inv_core<F, LOGN, false, true>(acc0, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    __builtin_amdgcn_sched_barrier(0);
    load_poly_buf<F, LOGN, COMPACT>(add0 + (size_t)p * (C::N * (COMPACT ? sizeof(E) : 32)), tid, d);
#pragma unroll
    for (int r = 0; r < 32; r++) acc0[r] = F::ew_add(F::canon_inv(acc0[r], P.q), d[r], P.q);
    lds_put<PatA<LOGN>>(lds, tid, acc0);
    __syncthreads();
    store_from_lds_rolled<F, LOGN>(c0 + (size_t)p * (C::N * 32), lds, tid);
    __builtin_amdgcn_sched_barrier(0);
    F::regroup(acc1, P.q, P.qinv);
    __syncthreads();
    inv_core<F, LOGN>(acc1, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    load_poly_buf<F, LOGN, COMPACT>(add1 + (size_t)p * (C::N * (COMPACT ? sizeof(E) : 32)), tid, d);
#pragma unroll
    for (int r = 0; r < 32; r++) acc1[r] = F::ew_add(F::canon_inv(acc1[r], P.q), d[r], P.q);
    lds_put<PatA<LOGN>>(lds, tid, acc1);
    __syncthreads();
    store_from_lds<F, LOGN>(c1 + (size_t)p * (C::N * 32), lds, tid);
}

// ---- fused blind-rotation step (external product) ------------------------------------------------------------------------
// out = in + ExtProd((X^a - 1) * in, RGSW)  for the RLWE pair in = (in0, in1), OUT OF PLACE (workgroup (b, i) reads every limb of
// in and writes limb i of out, so out must not alias in).  (FHEContext::blind_rotate is declared only, include/fhe.cuh:139.)
//   out0[i] = in0[i] + INTT_i( sum_{c in {0,1}} sum_{j,k} NTT_i(digit_k(((X^a - 1) * in_c)[j])) .* KB_c[jk][i] ),  out1 likewise with KA_c
// The monomial factor is applied while loading: coefficient x of (X^a - 1) * p is +-p[(x - a) mod n] - p[x]; the limb is loaded once
// (coalesced, pattern A), parked in the exchange buffer and read back rotated, so HBM and the texture path see one read per limb.
// 2*L*K forward + 2 inverse transforms per workgroup; HBM traffic per ciphertext: read in0, in1 (re-reads by the L workgroups of a
// ciphertext are XCD-L2 / Infinity-Cache hits, same block map as the key-switch kernel) + write out0, out1 = 4 * S.
// SPLIT as for the key-switch kernel: two workgroups per (b, i), one per output component.
template <class F, int LOGN>
__device__ __forceinline__ void rotate_through_lds(typename F::E *lds, uint32_t tid, uint32_t a, typename F::E qj, typename F::E (&x)[32]);
template <class F, int LOGN, bool COMPACT = false>      // the same through a descriptor based at the accumulator's first limb (load_src_buf)
__device__ __forceinline__ void load_monomial_A_buf(const TableBuf &B, uint32_t limb, typename F::E *lds, uint32_t tid, uint32_t a,
                                                    typename F::E qj, typename F::E (&x)[32]) {
    load_src_buf<F, LOGN, COMPACT>(B, limb, tid, x);
    rotate_through_lds<F, LOGN>(lds, tid, a, qj, x);
}
template <class F, int LOGN, bool COMPACT = false>
__device__ __forceinline__ void load_monomial_A(const char *__restrict__ base, size_t poly_index, typename F::E *lds, uint32_t tid, uint32_t a,
                                                typename F::E qj, typename F::E (&x)[32]) {
    load_src<F, LOGN, COMPACT>(base, poly_index, tid, x);   // p[i], i = tid + r*T
    rotate_through_lds<F, LOGN>(lds, tid, a, qj, x);
}
template <class F, int LOGN>
__device__ __forceinline__ void rotate_through_lds(typename F::E *lds, uint32_t tid, uint32_t a, typename F::E qj, typename F::E (&x)[32]) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __syncthreads();                                // the previous transform's last reads of the exchange buffer are over
    lds_put<PatA<LOGN>>(lds, tid, x);
    __syncthreads();
    const uint32_t k0 = tid + 2 * C::N - a;
#pragma unroll
    for (int r = 0; r < 32; r++) {
        uint32_t k = (k0 + (uint32_t)r * C::T) & (2 * C::N - 1);      // (i - a) mod 2n
        const bool neg = k >= (uint32_t)C::N;                         // X^n = -1
        k &= C::N - 1;
        E v = lds[k + (k >> 5)];                                      // consecutive lanes, consecutive slots: conflict-free like pattern A
        if (neg) v = F::ew_sub((E)0, v, qj);
        x[r] = F::ew_sub(v, x[r], qj);
    }
    __syncthreads();                                // rotated reads are done before the first transform reuses the buffer
}

template <class F, int LOGN, int MINW = 1, bool SPLIT = false, bool TWL = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_extprod_kernel(char *__restrict__ out0, char *__restrict__ out1, const char *__restrict__ in0, const char *__restrict__ in1,
                   const uint32_t *__restrict__ shifts,
                   const typename F::E *__restrict__ kb0, const typename F::E *__restrict__ ka0,      // rows for component 0
                   const typename F::E *__restrict__ kb1, const typename F::E *__restrict__ ka1,      // rows for component 1
                   const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    static_assert(!(TWL && SPLIT), "LDS twiddles are wired into the one-workgroup-per-limb form only");
    __shared__ E lds[C::LDS_ELEMS];
    __shared__ typename F::TW twl[TWL ? C::N : 1];
    constexpr uint32_t H = SPLIT ? 2 : 1;
    const uint32_t LH = L * H;
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * LH)) * (8 * LH);
    uint32_t b, u;
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / LH); u = s % LH; }
    else { b = bid / LH; u = bid % LH; }
    const uint32_t i = u / H, half = u % H;
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    const uint32_t a = shifts[b] & (2 * C::N - 1);
    if constexpr (!SPLIT) {
        E acc0[32], acc1[32], x[32], d[32];
#pragma unroll
        for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
        if constexpr (TWL) stage_twiddles<F, LOGN, true>(twl, P.tw, tid);     // visible after the barriers inside load_monomial_A
        for (uint32_t c = 0; c < 2; c++) {
            const char *src = c ? in1 : in0;
            const E *kb = c ? kb1 : kb0, *ka = c ? ka1 : ka0;
            for (uint32_t j = 0; j < L; j++) {
                load_monomial_A<F, LOGN>(src, (size_t)b * L + j, lds, tid, a, limbs[j].q, x);
                for (uint32_t k = 0; k < K; k++) {
#pragma unroll
                    for (int r = 0; r < 32; r++) d[r] = F::digit(x[r], k * w, w);
                    fwd_core<F, LOGN, TWL, true>(d, lds, tid, P, twl);
                    const size_t tbl = ((size_t)(j * K + k) * L + i) * C::N;
                    const VecE *pb = reinterpret_cast<const VecE *>(kb + tbl) + tid, *pa = reinterpret_cast<const VecE *>(ka + tbl) + tid;
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++) {
                        const VecE vb = pb[ch * C::T], va = pa[ch * C::T];
#pragma unroll
                        for (int e = 0; e < VPL; e++) {
                            const int r = ch * VPL + e;
                            acc0[r] = F::pw_add(acc0[r], F::pw_mul(vb[e], d[r], P.q, P.qinv), P.q, P.q2);
                            acc1[r] = F::pw_add(acc1[r], F::pw_mul(va[e], d[r], P.q, P.qinv), P.q, P.q2);
                        }
                    }
                }
            }
        }
        if constexpr (TWL) { __syncthreads(); stage_twiddles<F, LOGN, false>(twl, P.itw, tid); __syncthreads(); }
        inv_core<F, LOGN, TWL, !TWL>(acc0, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s, twl);
        load_A<F, LOGN>(in0 + (size_t)p * (C::N * 32), tid, x);
#pragma unroll
        for (int r = 0; r < 32; r++) acc0[r] = F::ew_add(F::canon_inv(acc0[r], P.q), x[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc0);
        __syncthreads();
        store_from_lds<F, LOGN>(out0 + (size_t)p * (C::N * 32), lds, tid);
        __syncthreads();
        inv_core<F, LOGN, TWL>(acc1, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s, twl);
        load_A<F, LOGN>(in1 + (size_t)p * (C::N * 32), tid, x);
#pragma unroll
        for (int r = 0; r < 32; r++) acc1[r] = F::ew_add(F::canon_inv(acc1[r], P.q), x[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc1);
        __syncthreads();
        store_from_lds<F, LOGN>(out1 + (size_t)p * (C::N * 32), lds, tid);
    } else {
        E acc[32], d[32];
#pragma unroll
        for (int r = 0; r < 32; r++) acc[r] = 0;
        for (uint32_t c = 0; c < 2; c++) {
            const char *src = c ? in1 : in0;
            const E *keys = c ? (half ? ka1 : kb1) : (half ? ka0 : kb0);
            for (uint32_t j = 0; j < L; j++) {
                const E qj = limbs[j].q;
                for (uint32_t k = 0; k < K; k++) {
                    load_monomial_A<F, LOGN>(src, (size_t)b * L + j, lds, tid, a, qj, d);
#pragma unroll
                    for (int r = 0; r < 32; r++) d[r] = F::digit(d[r], k * w, w);
                    fwd_core<F, LOGN, false, true>(d, lds, tid, P);
                    __builtin_amdgcn_sched_barrier(0);
                    const VecE *pk = reinterpret_cast<const VecE *>(keys + ((size_t)(j * K + k) * L + i) * C::N) + tid;
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++) {
                        const VecE v = pk[ch * C::T];
#pragma unroll
                        for (int e = 0; e < VPL; e++) {
                            const int r = ch * VPL + e;
                            acc[r] = F::pw_add(acc[r], F::pw_mul(v[e], d[r], P.q, P.qinv), P.q, P.q2);
                        }
                    }
                }
            }
            F::regroup(acc, P.q, P.qinv);          // floating-point sums of L*K products: back below q (no-op for the integer fields)
        }
        inv_core<F, LOGN, false, true>(acc, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
        load_A<F, LOGN>((half ? in1 : in0) + (size_t)p * (C::N * 32), tid, d);
#pragma unroll
        for (int r = 0; r < 32; r++) acc[r] = F::ew_add(F::canon_inv(acc[r], P.q), d[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc);
        __syncthreads();
        store_from_lds<F, LOGN>((half ? out1 : out0) + (size_t)p * (C::N * 32), lds, tid);
    }
}

// External product for the 8-byte residues at N = 2^14 in ONE workgroup per (accumulator, limb) with three live arrays (both output
// accumulators and the digit polynomial; the input limb is re-read, rotated, for every digit) -- the counterpart of
// ntt_keyswitch3_kernel: half the transforms of the SPLIT form above.
template <class F, int LOGN, int MINW = 1>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_extprod3_kernel(char *__restrict__ out0, char *__restrict__ out1, const char *__restrict__ in0, const char *__restrict__ in1,
                    const uint32_t *__restrict__ shifts,
                    const typename F::E *__restrict__ kb0, const typename F::E *__restrict__ ka0,
                    const typename F::E *__restrict__ kb1, const typename F::E *__restrict__ ka1,
                    const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * L)) * (8 * L);
    uint32_t b, i;
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / L); i = s % L; }
    else { b = bid / L; i = bid % L; }
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    const uint32_t a = shifts[b] & (2 * C::N - 1);
    E acc0[32], acc1[32], d[32];
#pragma unroll
    for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
    const uint32_t voff = tid * 16;
    for (uint32_t c = 0; c < 2; c++) {
        const TableBuf KB(c ? kb1 : kb0), KA(c ? ka1 : ka0), SRC((c ? in1 : in0) + (size_t)b * L * (C::N * 32));   // this accumulator's component c
        for (uint32_t j = 0; j < L; j++) {
            const E qj = limbs[j].q;
            for (uint32_t k = 0; k < K; k++) {
                if constexpr (__is_same(typename F::E, double)) load_monomial_A<F, LOGN>(c ? in1 : in0, (size_t)b * L + j, lds, tid, a, qj, d);
                else load_monomial_A_buf<F, LOGN>(SRC, j, lds, tid, a, qj, d);
#pragma unroll
                for (int r = 0; r < 32; r++) d[r] = F::digit(d[r], k * w, w);
                fwd_core<F, LOGN, false, true>(d, lds, tid, P);
                __builtin_amdgcn_sched_barrier(0);   // keep the key loads out of the transform
                const uint32_t tbl = (uint32_t)((((size_t)(j * K + k) * L + i) * C::N) * sizeof(E));
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) {
                    const VecE vb = KB.template load16<VecE>(voff, tbl + ch * C::T * 16), va = KA.template load16<VecE>(voff, tbl + ch * C::T * 16);
#pragma unroll
                    for (int e = 0; e < VPL; e++) {
                        const int r = ch * VPL + e;
                        acc0[r] = F::pw_add(acc0[r], F::pw_mul(vb[e], d[r], P.q, P.qinv), P.q, P.q2);
                        acc1[r] = F::pw_add(acc1[r], F::pw_mul(va[e], d[r], P.q, P.qinv), P.q, P.q2);
                    }
                    if ((ch & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        F::regroup(acc0, P.q, P.qinv);             // floating-point sums of L*K products: back below q (no-op for the integer fields)
        F::regroup(acc1, P.q, P.qinv);
    }
    inv_core<F, LOGN, false, true>(acc0, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    __builtin_amdgcn_sched_barrier(0);
    load_poly_buf<F, LOGN, false>(in0 + (size_t)p * (C::N * 32), tid, d);
#pragma unroll
    for (int r = 0; r < 32; r++) acc0[r] = F::ew_add(F::canon_inv(acc0[r], P.q), d[r], P.q);
    lds_put<PatA<LOGN>>(lds, tid, acc0);
    __syncthreads();
    store_from_lds_rolled<F, LOGN>(out0 + (size_t)p * (C::N * 32), lds, tid);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    inv_core<F, LOGN>(acc1, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    load_poly_buf<F, LOGN, false>(in1 + (size_t)p * (C::N * 32), tid, d);
#pragma unroll
    for (int r = 0; r < 32; r++) acc1[r] = F::ew_add(F::canon_inv(acc1[r], P.q), d[r], P.q);
    lds_put<PatA<LOGN>>(lds, tid, acc1);
    __syncthreads();
    store_from_lds<F, LOGN>(out1 + (size_t)p * (C::N * 32), lds, tid);
}

// ---- paired forms of the key-switch and external-product kernels (4-byte residues) --------------------------------------
// Same results as ntt_keyswitch_kernel / ntt_extprod_kernel (SPLIT = false); the digit polynomials are transformed two at a
// time (fwd_core2).  Registers: acc0, acc1, d0, d1 (the undecomposed limb is not kept: both digits of a pair are cut from the
// freshly loaded values).  LDS: two exchange buffers (66 KiB at N = 8192: two workgroups per CU; 132 KiB at N = 16384: one).
template <class F>
__device__ __forceinline__ void mac_keys(typename F::E (&acc0)[32], typename F::E (&acc1)[32], const typename F::E (&d)[32],
                                         const typename F::E *__restrict__ kb, const typename F::E *__restrict__ ka, size_t tbl, uint32_t tid,
                                         uint32_t T, const Limb<F> &P) {
    using E = typename F::E;
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    const TableBuf KB(kb), KA(ka);                          // SGPR descriptors + one shared VGPR offset: no per-chunk 64-bit addresses
    const uint32_t voff = tid * 16, row = (uint32_t)(tbl * sizeof(E));
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const VecE vb = KB.template load16<VecE>(voff, row + c * T * 16), va = KA.template load16<VecE>(voff, row + c * T * 16);
#pragma unroll
        for (int e = 0; e < VPL; e++) {
            const int r = c * VPL + e;
            acc0[r] = F::pw_add(acc0[r], F::pw_mul(vb[e], d[r], P.q, P.qinv), P.q, P.q2);
            acc1[r] = F::pw_add(acc1[r], F::pw_mul(va[e], d[r], P.q, P.qinv), P.q, P.q2);
        }
    }
}
// acc0 += kb[t0] .* d0 + kb[t1] .* d1, acc1 likewise with ka: the two products of a pair share one Montgomery reduction (mont_mul2)
template <class F>
__device__ __forceinline__ void mac_keys2(typename F::E (&acc0)[32], typename F::E (&acc1)[32], const typename F::E (&d0)[32], const typename F::E (&d1)[32],
                                          const typename F::E *__restrict__ kb0, const typename F::E *__restrict__ ka0, size_t tbl0,
                                          const typename F::E *__restrict__ kb1, const typename F::E *__restrict__ ka1, size_t tbl1,
                                          uint32_t tid, uint32_t T, const Limb<F> &P) {
    using E = typename F::E;
    static_assert(sizeof(E) == 4, "mont_mul2 is a 32-bit field operation");
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    const TableBuf KB0(kb0), KA0(ka0), KB1(kb1), KA1(ka1);
    const uint32_t voff = tid * 16, row0 = (uint32_t)(tbl0 * sizeof(E)), row1 = (uint32_t)(tbl1 * sizeof(E));
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const VecE vb0 = KB0.template load16<VecE>(voff, row0 + c * T * 16), va0 = KA0.template load16<VecE>(voff, row0 + c * T * 16);
        const VecE vb1 = KB1.template load16<VecE>(voff, row1 + c * T * 16), va1 = KA1.template load16<VecE>(voff, row1 + c * T * 16);
#pragma unroll
        for (int e = 0; e < VPL; e++) {
            const int r = c * VPL + e;
            acc0[r] = F::pw_add(acc0[r], F::mont_mul2(vb0[e], d0[r], vb1[e], d1[r], P.q, P.q2, P.qinv), P.q, P.q2);
            acc1[r] = F::pw_add(acc1[r], F::mont_mul2(va0[e], d0[r], va1[e], d1[r], P.q, P.q2, P.qinv), P.q, P.q2);
        }
    }
}

// both accumulators back to the coefficient domain in lock step, + the addend polynomials, store  (tail of the paired kernels)
// add0 / add1: base pointers of the addend buffers (containers, or compact polynomials when COMPACT), p = polynomial index
// dst0 / dst1: base pointers of the output buffers (containers, or compact polynomials when COMPACT_OUT)
template <class F, int LOGN, bool COMPACT = false, bool COMPACT_OUT = false>
__device__ __forceinline__ void finish_pair(typename F::E (&acc0)[32], typename F::E (&acc1)[32], typename F::E (&t0)[32], typename F::E (&t1)[32],
                                            typename F::E *lds0, typename F::E *lds1, uint32_t tid, const Limb<F> &P,
                                            const char *add0, const char *add1, size_t p, char *dst0, char *dst1) {
    load_src<F, LOGN, COMPACT>(add0, p, tid, t0);   // issued first: the HBM latency hides under the inverse transforms
    load_src<F, LOGN, COMPACT>(add1, p, tid, t1);
    inv_core2<F, LOGN, true>(acc0, acc1, lds0, lds1, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
#pragma unroll
    for (int r = 0; r < 32; r++) {
        acc0[r] = F::ew_add(F::canon_inv(acc0[r], P.q), t0[r], P.q);
        acc1[r] = F::ew_add(F::canon_inv(acc1[r], P.q), t1[r], P.q);
    }
    if constexpr (COMPACT_OUT) {
        store_A_compact<F, LOGN>(reinterpret_cast<typename F::E *>(dst0) + p * NttCfg<LOGN>::N, tid, acc0);
        store_A_compact<F, LOGN>(reinterpret_cast<typename F::E *>(dst1) + p * NttCfg<LOGN>::N, tid, acc1);
    } else {
        lds_put<PatA<LOGN>>(lds0, tid, acc0);
        lds_put<PatA<LOGN>>(lds1, tid, acc1);
        __syncthreads();
        store_from_lds<F, LOGN>(dst0 + p * (NttCfg<LOGN>::N * 32), lds0, tid);
        store_from_lds<F, LOGN>(dst1 + p * (NttCfg<LOGN>::N * 32), lds1, tid);
    }
}

template <class F, int LOGN, int MINW = 1, bool COMPACT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_keyswitch2_kernel(char *c0, char *c1, const char *__restrict__ c2, const char *add0, const char *add1,   // add* = c* (in place) unless COMPACT
                      const typename F::E *__restrict__ kb, const typename F::E *__restrict__ ka,
                      const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[2 * C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * L)) * (8 * L);
    uint32_t b, i;                                        // XCD-aware map, as in ntt_keyswitch_kernel
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / L); i = s % L; }
    else { b = bid / L; i = bid % L; }
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    const TableBuf C2(c2 + (size_t)b * L * (C::N * (COMPACT ? sizeof(E) : 32)));   // descriptor based at the first limb polynomial of this ciphertext's c2
    E acc0[32], acc1[32], d0[32], d1[32];
#pragma unroll
    for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
    const uint32_t LK = L * K;
    uint32_t jk = 0;
    for (; jk + 1 < LK; jk += 2) {
        const uint32_t j0 = jk / K, k0 = jk % K, j1 = (jk + 1) / K, k1 = (jk + 1) % K;
        load_src_buf<F, LOGN, COMPACT>(C2, j1, tid, d1);
        if (j0 == j1) {
#pragma unroll
            for (int r = 0; r < 32; r++) d0[r] = F::digit(d1[r], k0 * w, w);
        } else {
            load_src_buf<F, LOGN, COMPACT>(C2, j0, tid, d0);
#pragma unroll
            for (int r = 0; r < 32; r++) d0[r] = F::digit(d0[r], k0 * w, w);
        }
#pragma unroll
        for (int r = 0; r < 32; r++) d1[r] = F::digit(d1[r], k1 * w, w);
        fwd_core2<F, LOGN, true>(d0, d1, lds, lds + C::LDS_ELEMS, tid, P);
        mac_keys2<F>(acc0, acc1, d0, d1, kb, ka, ((size_t)jk * L + i) * C::N, kb, ka, ((size_t)(jk + 1) * L + i) * C::N, tid, C::T, P);
    }
    if (jk < LK) {                                        // odd number of digit polynomials: the last one alone
        const uint32_t j0 = jk / K, k0 = jk % K;
        load_src_buf<F, LOGN, COMPACT>(C2, j0, tid, d0);
#pragma unroll
        for (int r = 0; r < 32; r++) d0[r] = F::digit(d0[r], k0 * w, w);
        fwd_core<F, LOGN, false, true>(d0, lds, tid, P);
        mac_keys<F>(acc0, acc1, d0, kb, ka, ((size_t)jk * L + i) * C::N, tid, C::T, P);
    }
    finish_pair<F, LOGN, COMPACT>(acc0, acc1, d0, d1, lds, lds + C::LDS_ELEMS, tid, P, add0, add1, p, c0, c1);
}

// IN_COMPACT / OUT_COMPACT: the accumulator pair is read from / written to compact polynomials (load_A_compact): inside fhe_blind_rotate
// the accumulators stay in that form between the first and the last step of a loop, so a step moves 2 S/8 per limb workgroup in and
// 2 S/8 out instead of 2 S each way (the L workgroups of an accumulator each read ALL its limbs).
template <class F, int LOGN, int MINW = 1, bool IN_COMPACT = false, bool OUT_COMPACT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_extprod2_kernel(char *__restrict__ out0, char *__restrict__ out1, const char *__restrict__ in0, const char *__restrict__ in1,
                    const uint32_t *__restrict__ shifts,
                    const typename F::E *__restrict__ kb0, const typename F::E *__restrict__ ka0,
                    const typename F::E *__restrict__ kb1, const typename F::E *__restrict__ ka1,
                    const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[2 * C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * L)) * (8 * L);
    uint32_t b, i;
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / L); i = s % L; }
    else { b = bid / L; i = bid % L; }
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    const uint32_t a = shifts[b] & (2 * C::N - 1);
    const size_t cbytes = (size_t)b * L * (C::N * (IN_COMPACT ? sizeof(E) : 32));
    const TableBuf IN0(in0 + cbytes), IN1(in1 + cbytes);   // descriptors based at the first limb polynomial of this accumulator's two components
    E acc0[32], acc1[32], d0[32], d1[32];
#pragma unroll
    for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
    const uint32_t LK = L * K, G = 2 * LK;                // digit polynomials of component 0, then of component 1: always an even number
    for (uint32_t g = 0; g < G; g += 2) {
        const uint32_t c0i = g / LK, jk0 = g % LK, j0 = jk0 / K, k0 = jk0 % K;
        const uint32_t c1i = (g + 1) / LK, jk1 = (g + 1) % LK, j1 = jk1 / K, k1 = jk1 % K;
        load_monomial_A_buf<F, LOGN, IN_COMPACT>(c1i ? IN1 : IN0, j1, lds, tid, a, limbs[j1].q, d1);
        if (c0i == c1i && j0 == j1) {
#pragma unroll
            for (int r = 0; r < 32; r++) d0[r] = F::digit(d1[r], k0 * w, w);
        } else {
            load_monomial_A_buf<F, LOGN, IN_COMPACT>(c0i ? IN1 : IN0, j0, lds, tid, a, limbs[j0].q, d0);
#pragma unroll
            for (int r = 0; r < 32; r++) d0[r] = F::digit(d0[r], k0 * w, w);
        }
#pragma unroll
        for (int r = 0; r < 32; r++) d1[r] = F::digit(d1[r], k1 * w, w);
        fwd_core2<F, LOGN, true>(d0, d1, lds, lds + C::LDS_ELEMS, tid, P);
        mac_keys2<F>(acc0, acc1, d0, d1, c0i ? kb1 : kb0, c0i ? ka1 : ka0, ((size_t)jk0 * L + i) * C::N, c1i ? kb1 : kb0, c1i ? ka1 : ka0,
                     ((size_t)jk1 * L + i) * C::N, tid, C::T, P);
    }
    finish_pair<F, LOGN, IN_COMPACT, OUT_COMPACT>(acc0, acc1, d0, d1, lds, lds + C::LDS_ELEMS, tid, P, in0, in1, p, out0, out1);
}

// ---- relinearisation building blocks (general path; the fused key-switch kernels are above) --------------------
// Digit polynomials of c2 embedded in every limb: D[jk][b][i][x] = ((c2[b][j][x] >> (k*w)) & (2^w - 1)) mod q_i,
// jk = j*K + k.  One 16-byte half container per lane.
template <class F>
__global__ void __launch_bounds__(256)
digit_embed_kernel(typename F::V16 *__restrict__ D, const typename F::V16 *__restrict__ c2, const Limb<F> *__restrict__ limbs,
                   uint32_t L, uint32_t log_n, uint32_t K, uint32_t w, uint32_t batch) {
    using E = typename F::E;
    const size_t per_poly = (size_t)2 << log_n, per_ct = per_poly * L, per_digit = per_ct * batch, total = per_digit * L * K;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        E o = 0;
        if (!(g & 1)) {
            const uint32_t jk = (uint32_t)(g / per_digit); const size_t rem = g - (size_t)jk * per_digit;
            const uint32_t b = (uint32_t)(rem / per_ct); const size_t r2 = rem - (size_t)b * per_ct;
            const uint32_t i = (uint32_t)(r2 >> (log_n + 1)); const size_t x2 = r2 & (per_poly - 1);
            const uint32_t j = jk / K, k = jk % K;
            const uint64_t v = F::low(c2[((size_t)b * L + j) * per_poly + x2]);
            const uint32_t sh = k * w;
            uint64_t d = sh >= 64 ? 0 : (v >> sh);
            if (w < 64) d &= (1ull << w) - 1;
            o = F::from_u64(d, limbs[i].q);
        }
        __builtin_nontemporal_store(F::pack(o), D + g);
    }
}
// acc0[b][i][x] = sum_jk D[jk][b][i][x] * KB[jk][i][x],  acc1 likewise with KA  (all NTT-domain, canonical)
template <class F>
__global__ void __launch_bounds__(256)
relin_mac_kernel(typename F::V16 *__restrict__ acc0, typename F::V16 *__restrict__ acc1, const typename F::V16 *__restrict__ D,
                 const typename F::V16 *__restrict__ KB, const typename F::V16 *__restrict__ KA, const Limb<F> *__restrict__ limbs,
                 uint32_t L, uint32_t log_n, uint32_t LK, uint32_t batch) {
    using E = typename F::E;
    const size_t per_poly = (size_t)2 << log_n, per_ct = per_poly * L, per_digit = per_ct * batch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < per_digit; g += stride) {
        E s0 = 0, s1 = 0;
        if (!(g & 1)) {
            const size_t kidx = g % per_ct;
            const Limb<F> &P = limbs[(uint32_t)(kidx >> (log_n + 1))];
            for (uint32_t jk = 0; jk < LK; jk++) {
                const E d = F::load_low(D + (size_t)jk * per_digit + g);
                s0 = F::ew_add(s0, F::ew_mul(d, F::load_low(KB + (size_t)jk * per_ct + kidx), P), P.q);
                s1 = F::ew_add(s1, F::ew_mul(d, F::load_low(KA + (size_t)jk * per_ct + kidx), P), P.q);
            }
        }
        __builtin_nontemporal_store(F::pack(s0), acc0 + g);
        __builtin_nontemporal_store(F::pack(s1), acc1 + g);
    }
}

// ---- RNS conversions on word-sized residues (row N2): rounded drop of the last prime, Bajard fast base conversion -----------
// The container-level kernels in ntt256.hip.h do these through 256-bit Montgomery products for every width class; for word-sized
// classes the same arithmetic fits the field type and the kernels are streaming kernels.  Constants are "pw operands"
// (c * 2^W mod q for the integer fields, c for F52) so that canon(pw_mul(constant, x)) is the plain product c * x mod q.
template <class F>
__device__ __forceinline__ typename F::E mul_const(typename F::E cop, typename F::E x, const Limb<F> &P) {
    return F::canon_inv(F::pw_mul(cop, x, P.q, P.qinv), P.q);
}
// out[b][l][x] = (in[b][l][x] - [c]_{q_l}) * q_last^-1 mod q_l, c = the CENTRED residue of in[b][L-1][x] modulo q_last
// (RNSContext::mod_switch_rns, include/rns.cuh:44, declared only).  (x_l - c) * inv = x_l * inv - c * inv with c = +-mag, mag < q_last:
// a residue of ANOTHER prime of the class is a valid second operand of the constant product as it stands (no division to reduce it).
// One lane per (b, x): the last limb is loaded once and every remaining limb produced from it, full 32-byte containers stored as two
// 16-byte halves by the same lane (+22 % at N = 8192 over one lane per output half-container, which re-read the last limb per limb).
template <class F>
__global__ void __launch_bounds__(256)
rescale_word_kernel(typename F::V16 *__restrict__ out, const typename F::V16 *__restrict__ in, const Limb<F> *__restrict__ limbs,
                            const typename F::E *__restrict__ inv_ops, uint32_t L, uint32_t log_n, size_t count /* batch * n */) {
    using E = typename F::E;
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    const E q_last = limbs[L - 1].q, half = (E)(((uint64_t)q_last - 1) >> 1);
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const size_t b = g >> log_n, x = g & (n - 1);
        const E cl = F::load_low(in + (((b * L + (L - 1)) << log_n) + x) * 2);
        const bool neg = cl > half;
        const E mag = neg ? q_last - cl : cl;
        for (uint32_t l = 0; l + 1 < L; l++) {
            const Limb<F> &P = limbs[l];
            const E xl = F::load_low(in + (((b * L + l) << log_n) + x) * 2);
            const E a = mul_const<F>(inv_ops[l], xl, P), m = mul_const<F>(inv_ops[l], mag, P);
            const E o = neg ? F::ew_add(a, m, P.q) : F::ew_sub(a, m, P.q);
            typename F::V16 *dst = out + (((b * (L - 1) + l) << log_n) + x) * 2;
            __builtin_nontemporal_store(F::pack(o), dst);
            __builtin_nontemporal_store(F::pack((E)0), dst + 1);
        }
    }
}

// out[b][j][x] = sum_i ([x_i * (Q/q_i)^-1]_{q_i} mod p_j) * ((Q/q_i) mod p_j) mod p_j   (RNSContext::base_extend, include/rns.cuh:47-48, declared only).
// minv_ops[i] is an operand of source limb i, mat_ops[i * Lp + j] an operand of target limb j.  One half container of the output per lane.
// ALL_LANES: one output container per lane, stored by lane pairs (store_wave_containers): +8..10 % on the 64-bit integer fields, whose
// constant products are the cost; the 4-byte and FP64 fields are bandwidth-bound either way and keep one half container per lane
// (measured 5.0 vs 4.4 and 4.7 vs 4.5 TB/s).
template <class F, bool ALL_LANES>
__global__ void __launch_bounds__(256)
base_convert_word_kernel(typename F::V16 *__restrict__ out, const typename F::V16 *__restrict__ in, const Limb<F> *__restrict__ src, uint32_t L,
                         const Limb<F> *__restrict__ dst, uint32_t Lp, const typename F::E *__restrict__ minv_ops,
                         const typename F::E *__restrict__ mat_ops, uint32_t log_n, size_t work /* containers if ALL_LANES, else half containers */) {
    using E = typename F::E;
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < work; g += stride) {   // n is a multiple of 256: whole waves
        E o = 0;
        if (ALL_LANES || !(g & 1)) {
            const size_t c = ALL_LANES ? g : g >> 1, x = c & (n - 1), pl = c >> log_n, b = pl / Lp;
            const uint32_t j = (uint32_t)(pl % Lp);
            const Limb<F> &D = dst[j];
            for (uint32_t i = 0; i < L; i++) {
                const Limb<F> &S = src[i];
                const E ti = mul_const<F>(minv_ops[i], F::load_low(in + (((b * L + i) << log_n) + x) * 2), S);
                o = F::ew_add(o, mul_const<F>(mat_ops[(size_t)i * Lp + j], ti, D), D.q);   // t_i < q_i: a valid operand modulo p_j as it stands
            }
        }
        if constexpr (ALL_LANES) store_wave_containers<F>(out + 2 * (g - (threadIdx.x & 63)), o);
        else __builtin_nontemporal_store(F::pack(o), out + g);
    }
}

// rns[b][l][x] = values[b][x] mod q_l for ANY 256-bit value (RNS_NTTEngine::to_rns, include/ntt.cuh:114-115, declared only): the value is
// read as 256 / W words of W bits and reduced as sum_k word_k * (2^(W k) mod q_l); pow_ops[l * NW + k] is the pw operand of
// 2^(W k) mod q_l, and a word needs no reduction of its own (integer fields: operand < q, word < 2^W, product < q 2^W; FP64 field:
// 32-bit words, far below its 2^48 operand bound).
template <class F, class WT>     // WT: the word type the value is cut into (uint32_t for F32 and F52, uint64_t for F64)
__global__ void __launch_bounds__(256)
to_rns_word_kernel(typename F::V16 *__restrict__ rns, const typename F::V16 *__restrict__ values, const Limb<F> *__restrict__ limbs,
                   const typename F::E *__restrict__ pow_ops, uint32_t L, uint32_t log_n, size_t out_containers) {
    using E = typename F::E;
    constexpr int NW = 32 / sizeof(WT), HW = NW / 2;                   // words per container / per 16-byte half
    typedef WT VecW __attribute__((ext_vector_type(HW)));
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < out_containers; c += stride) {   // n is a multiple of 256: whole waves
        const size_t x = c & (n - 1), pl = c >> log_n, b = pl / L;
        const uint32_t l = (uint32_t)(pl % L);
        const Limb<F> &P = limbs[l];
        const VecW *v = reinterpret_cast<const VecW *>(values + ((b << log_n) + x) * 2);
        const VecW lo = v[0], hi = v[1];
        const E *ops = pow_ops + (size_t)l * NW;
        E o = 0;
#pragma unroll
        for (int k = 0; k < HW; k++) {
            o = F::ew_add(o, mul_const<F>(ops[k], (E)lo[k], P), P.q);
            o = F::ew_add(o, mul_const<F>(ops[HW + k], (E)hi[k], P), P.q);
        }
        store_wave_containers<F>(rns + 2 * (c - (threadIdx.x & 63)), o);
    }
}

// values[b][x] = CRT of the L residues, in [0, Q)  (RNS_NTTEngine::from_rns, include/ntt.cuh:116-117, declared only), word-sized
// classes: sum_l [x_l * (Q/q_l)^-1]_{q_l} * (Q/q_l) is accumulated as word x 256-bit products in a 320-bit register array (the sum is
// below L * Q) and brought into [0, Q) by at most L - 1 subtractions.  One lane per value; Mi[l] = Q / q_l as a plain integer.
template <class F>
__global__ void __launch_bounds__(256)
from_rns_word_kernel(u256 *__restrict__ values, const typename F::V16 *__restrict__ rns, const Limb<F> *__restrict__ limbs,
                     const typename F::E *__restrict__ minv_ops, const u256 *__restrict__ Mi, u256 Q, uint32_t L, uint32_t log_n, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const size_t b = g >> log_n, x = g & (n - 1);
        uint64_t acc[5] = {0, 0, 0, 0, 0};
        for (uint32_t l = 0; l < L; l++) {
            const uint64_t t = (uint64_t)mul_const<F>(minv_ops[l], F::load_low(rns + (((b * L + l) << log_n) + x) * 2), limbs[l]);
            const u256 M = Mi[l];
            u128_t c = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { c += (u128_t)t * M.l[i] + acc[i]; acc[i] = (uint64_t)c; c >>= 64; }
            acc[4] += (uint64_t)c;
        }
        for (uint32_t it = 0; it < L; it++) {                          // acc < L * Q
            bool ge = acc[4] != 0;
            if (!ge) {
                ge = true;
#pragma unroll
                for (int i = 3; i >= 0; i--) { if (acc[i] != Q.l[i]) { ge = acc[i] > Q.l[i]; break; } }
            }
            if (!ge) break;
            uint64_t borrow = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { u128_t d = (u128_t)acc[i] - Q.l[i] - borrow; acc[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1; }
            acc[4] -= borrow;
        }
        u256 r; r.l[0] = acc[0]; r.l[1] = acc[1]; r.l[2] = acc[2]; r.l[3] = acc[3];
        store_u256(values + g, r);
    }
}

// ---- blind-rotation building block: out[b][l][x] = ((X^shift[b] - 1) * in[b][l])[x] over Z_q[x]/(x^n + 1), shift in [0, 2n) ----
// (FHEContext::blind_rotate is only declared in the reference, include/fhe.cuh:139.)  One 16-byte half container per lane;
// the rotated read is a shifted contiguous run, so it stays coalesced.
template <class F>
__global__ void __launch_bounds__(256)
monomial_mul_sub_kernel(typename F::V16 *__restrict__ out, const typename F::V16 *__restrict__ in, const uint32_t *__restrict__ shifts,
                        const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t log_n, size_t halves) {
    using E = typename F::E;
    const uint32_t n = 1u << log_n;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < halves; g += stride) {
        E o = 0;
        if (!(g & 1)) {
            const size_t c = g >> 1, poly = c >> log_n;                   // container index, polynomial index b*L + l
            const uint32_t x = (uint32_t)(c & (n - 1));
            const uint32_t a = shifts[poly / L] & (2 * n - 1);
            uint32_t k = (x + 2 * n - a) & (2 * n - 1);
            const bool neg = k >= n; k &= n - 1;
            const E q = limbs[(uint32_t)(poly % L)].q;
            E v = F::load_low(in + ((poly << log_n) + k) * 2);
            if (neg) v = F::ew_sub((E)0, v, q);
            o = F::ew_sub(v, F::load_low(in + g), q);
        }
        __builtin_nontemporal_store(F::pack(o), out + g);
    }
}

}  // namespace fhe_dev
