// u256_dev.h -- 256-bit modular arithmetic for gfx950 (full-width path, FHE_WIDTH_256).
//
// Semantics are the reference's, bit for bit (include/bigint.cuh:27-140), including the top-limb
// borrow test and the lost carry out of the 512-bit accumulator, so the element-wise
// kernels reproduce the reference primitives even on unreduced operands.
//
// Where the reference chains PTX add.cc / addc / madc through the implicit carry flag
// (kernels/ptx_bigint.cuh:34-117), this code keeps every limb as 2 x 32-bit VGPRs and lets the
// 64x64->128 products lower to v_mad_u64_u32 with v_add_co_u32 / v_addc_co_u32 carry chains
// (checked with `hipcc -S`: see DESIGN.md "ISA check").  No MFMA: integer modular arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fhe_dev {

struct __attribute__((aligned(16))) u256 {
    uint64_t l[4];
};

typedef unsigned __int128 u128_t;

// Full 64x64 -> 128 multiply-accumulate: (hi, lo) = a*b + c + d, cannot overflow 128 bits.
__device__ __forceinline__ void mac64(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t &lo, uint64_t &hi) {
    u128_t p = (u128_t)a * b + c + d;
    lo = (uint64_t)p;
    hi = (uint64_t)(p >> 64);
}

__device__ __forceinline__ void add256(u256 &r, const u256 &a, const u256 &b) {
    u128_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { c += (u128_t)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
}
__device__ __forceinline__ void sub256(u256 &r, const u256 &a, const u256 &b) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u128_t d = (u128_t)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}

// include/bigint.cuh:27-48
__device__ __forceinline__ u256 add_mod(const u256 &a, const u256 &b, const u256 &q) {
    u256 s, t;
    add256(s, a, b);
    sub256(t, s, q);
    bool underflow = t.l[3] > s.l[3];
    u256 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = underflow ? s.l[i] : t.l[i];
    return r;
}

// include/bigint.cuh:50-73
__device__ __forceinline__ u256 sub_mod(const u256 &a, const u256 &b, const u256 &q) {
    u256 d, t;
    sub256(d, a, b);
    bool borrow = d.l[3] > a.l[3];
    add256(t, d, q);
    u256 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = borrow ? t.l[i] : d.l[i];
    return r;
}

// include/bigint.cuh:76-140.  Same value as the reference's separated product + 4 reduction rounds:
// the rows are interleaved (product row i, then reduction round i) so only 6 limbs are live; the
// multipliers m_i are identical because round i only ever reads limb i of the running sum, which is
// final once product rows 0..i and rounds 0..i-1 have been added.  The carry that the reference loses
// beyond limb 7 is the bit dropped when `hi` below is truncated to 64 bits at the end.
__device__ __forceinline__ u256 mont_mul(const u256 &a, const u256 &b, const u256 &q, uint64_t inv0) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;   // running sum / 2^(64 i), low 5 limbs
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t c, bi = b.l[i];
        mac64(a.l[0], bi, t0, 0, t0, c);
        mac64(a.l[1], bi, t1, c, t1, c);
        mac64(a.l[2], bi, t2, c, t2, c);
        mac64(a.l[3], bi, t3, c, t3, c);
        u128_t top = (u128_t)t4 + c;
        t4 = (uint64_t)top;
        uint64_t t5 = (uint64_t)(top >> 64);
        uint64_t m = t0 * inv0;
        uint64_t dummy;
        mac64(m, q.l[0], t0, 0, dummy, c);
        mac64(m, q.l[1], t1, c, t0, c);
        mac64(m, q.l[2], t2, c, t1, c);
        mac64(m, q.l[3], t3, c, t2, c);
        top = (u128_t)t4 + c;
        t3 = (uint64_t)top;
        t4 = t5 + (uint64_t)(top >> 64);               // bit 512 of the exact sum lives in t4; dropped below
    }
    u256 u, d, r;
    u.l[0] = t0; u.l[1] = t1; u.l[2] = t2; u.l[3] = t3;
    sub256(d, u, q);
    bool underflow = d.l[3] > u.l[3];
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = underflow ? u.l[i] : d.l[i];
    return r;
}

// include/ntt.cuh:147-155
__device__ __forceinline__ void ct_butterfly(u256 &a, u256 &b, const u256 &w, const u256 &q, uint64_t inv0) {
    u256 t = mont_mul(b, w, q, inv0);
    b = sub_mod(a, t, q);
    a = add_mod(a, t, q);
}
// include/ntt.cuh:158-167
__device__ __forceinline__ void gs_butterfly(u256 &a, u256 &b, const u256 &w, const u256 &q, uint64_t inv0) {
    u256 s = add_mod(a, b, q);
    b = mont_mul(sub_mod(a, b, q), w, q, inv0);
    a = s;
}

// 32-byte container <-> registers: two 16-byte accesses per lane.
__device__ __forceinline__ u256 load_u256(const u256 *p) {
    const ulonglong2 *v = reinterpret_cast<const ulonglong2 *>(p);
    ulonglong2 lo = v[0], hi = v[1];
    u256 r; r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = hi.x; r.l[3] = hi.y;
    return r;
}
__device__ __forceinline__ void store_u256(u256 *p, const u256 &x) {
    ulonglong2 *v = reinterpret_cast<ulonglong2 *>(p);
    v[0] = make_ulonglong2(x.l[0], x.l[1]);
    v[1] = make_ulonglong2(x.l[2], x.l[3]);
}

}  // namespace fhe_dev
