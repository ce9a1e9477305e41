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


// ---------------------------------------------------------------------------------------------------------------------
// Hand-scheduled Montgomery product for the NTT passes (odd q, exact inv0): finely integrated product scanning over
// eight 32-bit words.  Every 32x32 product is one v_mad_u64_u32 into a 64-bit column accumulator whose carry-out goes
// to its own SGPR pair and is folded into a third accumulator word by v_addc_co_u32.  On gfx950 a VALU read of an SGPR
// (incl. VCC) written by the previous VALU instruction needs two wait states -- the compiler pads its own carry chains
// with s_nop (156 of the 818 instructions of the compiled 64-bit-limb product).  Here three independent carries are in
// flight at once (s[20:25]), so the multiplies themselves are the padding: 128 mads + 128 addc + 8 v_mul_lo_u32.
// The value is the same canonical a*b*R^-1 mod q as mul_mod_montgomery (include/bigint.cuh:76-140) for a, b < q.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mac1(uint64_t &lo, uint32_t &hi, uint32_t a0, uint32_t b0) {
    asm("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]"
        : "+v"(lo), "+v"(hi) : "v"(a0), "v"(b0) : "s20", "s21");
}
__device__ __forceinline__ void mac2(uint64_t &lo, uint32_t &hi, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
    asm("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\t"
        "v_mad_u64_u32 %0, s[22:23], %4, %5, %0\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\t"
        "v_addc_co_u32_e64 %1, s[22:23], 0, %1, s[22:23]"
        : "+v"(lo), "+v"(hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "s20", "s21", "s22", "s23");
}
__device__ __forceinline__ void mac3(uint64_t &lo, uint32_t &hi, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2) {
    asm("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\t"
        "v_mad_u64_u32 %0, s[22:23], %4, %5, %0\n\t"
        "v_mad_u64_u32 %0, s[24:25], %6, %7, %0\n\t"
        "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\t"
        "v_addc_co_u32_e64 %1, s[22:23], 0, %1, s[22:23]\n\t"
        "v_addc_co_u32_e64 %1, s[24:25], 0, %1, s[24:25]"
        : "+v"(lo), "+v"(hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2) : "s20", "s21", "s22", "s23", "s24", "s25");
}

struct w256 { uint32_t w[8]; };   // eight 32-bit words, little-endian (the same bytes as u256)
__device__ __forceinline__ w256 to_words(const u256 &a) {
    w256 r;
#pragma unroll
    for (int i = 0; i < 4; i++) { r.w[2 * i] = (uint32_t)a.l[i]; r.w[2 * i + 1] = (uint32_t)(a.l[i] >> 32); }
    return r;
}
__device__ __forceinline__ u256 from_words(const w256 &a) {
    u256 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)a.w[2 * i] | ((uint64_t)a.w[2 * i + 1] << 32);
    return r;
}

// a, b < q (odd, < 2^255); qinv32 = -q^-1 mod 2^32 (= low word of MontgomeryParams::inv)
__device__ __forceinline__ u256 mont_mul_fips(const u256 &a_, const u256 &b_, const u256 &q_, uint32_t qinv32) {
    const w256 a = to_words(a_), b = to_words(b_), q = to_words(q_);
    uint32_t m[8], t[8];
    uint64_t lo = 0; uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
        // operand list of column k: a_i*b_(k-i) for every valid i, m_i*q_(k-i) for every already known m_i
        uint32_t xs[16], ys[16]; int cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { const int j = k - i; if (j >= 0 && j < 8) { xs[cnt] = a.w[i]; ys[cnt] = b.w[j]; cnt++; } }
#pragma unroll
        for (int i = 0; i < 8; i++) { const int j = k - i; if (j >= 0 && j < 8 && i < k && !(k < 8 && i == k)) { xs[cnt] = m[i]; ys[cnt] = q.w[j]; cnt++; } }
        int c = 0;
#pragma unroll
        for (int g = 0; g < 6; g++) {
            if (cnt - c >= 3) { mac3(lo, hi, xs[c], ys[c], xs[c + 1], ys[c + 1], xs[c + 2], ys[c + 2]); c += 3; }
        }
        if (cnt - c == 2) mac2(lo, hi, xs[c], ys[c], xs[c + 1], ys[c + 1]);
        else if (cnt - c == 1) mac1(lo, hi, xs[c], ys[c]);
        if (k < 8) {
            m[k] = (uint32_t)lo * qinv32;
            mac1(lo, hi, m[k], q.w[0]);            // clears the low word of the column
        } else {
            t[k - 8] = (uint32_t)lo;
        }
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    t[7] = (uint32_t)lo;                             // bit 256 of the sum (lo >> 32) is zero for reduced operands
    w256 tw; 
#pragma unroll
    for (int i = 0; i < 8; i++) tw.w[i] = t[i];
    const u256 u = from_words(tw);
    u256 d, r;
    sub256(d, u, q_);
    const bool underflow = d.l[3] > u.l[3];
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

// the same butterflies on the hand-scheduled product (NTT passes; q odd prime, inv0 exact)
__device__ __forceinline__ void ct_butterfly_fast(u256 &a, u256 &b, const u256 &w, const u256 &q, uint32_t qinv32) {
    u256 t = mont_mul_fips(b, w, q, qinv32);
    b = sub_mod(a, t, q);
    a = add_mod(a, t, q);
}
__device__ __forceinline__ void gs_butterfly_fast(u256 &a, u256 &b, const u256 &w, const u256 &q, uint32_t qinv32) {
    u256 s = add_mod(a, b, q);
    b = mont_mul_fips(sub_mod(a, b, q), w, q, qinv32);
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
