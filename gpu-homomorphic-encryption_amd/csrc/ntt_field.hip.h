// ntt_field.hip.h -- field traits of the word-sized RNS primes on gfx950 (residue type, butterflies, products, range bookkeeping),
// the per-limb constant record and the two memory helpers every kernel family shares (buffer-descriptor loads, lane-pair container
// stores).  Included by ntt_lds.hip.h (LDS-resident transforms: one object per (field, log2 n)) and ntt_word.hip.h (streaming kernels,
// compiled into fhe_hip.o), so that an edit to the transform kernels does not rebuild the host translation unit.
//   F32 : q < 2^30, 32-bit residues, Harvey lazy butterflies on Montgomery-form twiddles   (FHE_WIDTH_32)
//   F52 : q < 2^43, residues held as exact integers in doubles, FMA butterflies            (FHE_WIDTH_52)
//   F64 : q < 2^62, 64-bit residues, Harvey/Shoup integer butterflies                      (FHE_WIDTH_64)
//   F64X: q < 2^64, canonical residues through every butterfly                             (FHE_WIDTH_64X)
// Why a narrow path is bit-exact against include/bigint.cuh:27-140: with reduced operands and a prime modulus every reference
// primitive returns the canonical residue (mont(x, w*R) = x*w mod q, add_mod, sub_mod), so any exact evaluation of the same
// butterfly network yields the same 256-bit containers (upper words zero).
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

}  // namespace fhe_dev
