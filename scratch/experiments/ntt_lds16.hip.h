// ntt_lds16.hip.h -- 16-coefficients-per-thread form of the LDS-resident NTT (T = N/16 threads per workgroup).
// Used by the fused key-switch kernel, whose four live per-thread arrays make the 32-per-thread form register-bound;
// written for any field.  (Measured as a general replacement for the 8-byte fields: scratch/experiments/README.md.)
//
// Same mathematics, tables and HBM access shapes as ntt_lds.hip.h (whose 32-coefficients-per-thread form is kept for the
// 4-byte field F32).  With 8-byte residues that form needs 2 x 64 VGPRs for the operands of a fused multiply plus 64 for
// twiddles, and N x 8 bytes of LDS per workgroup of only N/32 threads: 2 waves per SIMD at N = 2^13, 2^14.  Halving the
// per-thread tile doubles the threads (T = N/16) on the same LDS footprint, i.e. twice the resident waves:
//   * log2 N = 4 + 4 + 4 (+ REM) stages in 3 or 4 register groups (radix-16) instead of 5 + 5 + REM;
//   * LDS is not padded but XOR-swizzled, slot(i) = i ^ ((i >> 4) & 31): the map lane -> bank is GF(2)-linear and
//     invertible for every access pattern used here (consecutive lanes, strided middle groups, 16-consecutive tiles), so
//     all exchanges are conflict-free for the 32-lane groups of ds_read_b64 / 16-lane groups of ds_write_b64, and
//     because XOR distributes over the disjoint bit-fields of an index, slot = slotbase(tid) ^ constant(r).
#pragma once
#include "ntt_lds.hip.h"

namespace fhe_dev {

template <int LOGN>
struct Cfg16 {
    static_assert(LOGN >= 11 && LOGN <= 14, "16-per-thread LDS path covers 2^11 .. 2^14");
    static constexpr int N = 1 << LOGN;
    static constexpr int LOGT = LOGN - 4;
    static constexpr int T = 1 << LOGT;
    static constexpr int NG = (LOGN + 3) / 4;           // register groups: 3 (2^11, 2^12) or 4 (2^13, 2^14)
    static constexpr int REM = LOGN - 4 * (NG - 1);     // stages of the last forward / first... group (1..4)
};

__device__ __forceinline__ constexpr uint32_t swz(uint32_t i) { return i ^ ((i >> 4) & 31u); }

// patterns: logical index = base(tid) | off(r); LDS slot = swz(base(tid)) ^ swz(off(r))  (disjoint bit-fields)
template <int LOGN> struct P16A {                       // r <-> index bits [LOGT, LOGN)
    using C = Cfg16<LOGN>;
    static constexpr int BIT0 = C::LOGT;
    static constexpr bool TW_UNIFORM = true;
    __device__ static uint32_t base(uint32_t tid) { return tid; }
    static constexpr uint32_t off(int r) { return (uint32_t)r << C::LOGT; }
};
template <int LOGN, int LO> struct P16Mid {             // r <-> index bits [LO, LO+4)
    static constexpr int BIT0 = LO;
    static constexpr bool TW_UNIFORM = false;
    __device__ static uint32_t base(uint32_t tid) { return ((tid >> LO) << (LO + 4)) | (tid & ((1u << LO) - 1)); }
    static constexpr uint32_t off(int r) { return (uint32_t)r << LO; }
};
template <int LOGN> struct P16Z {                       // r <-> index bits [0, 4): 16 consecutive coefficients per thread
    static constexpr int BIT0 = 0;
    static constexpr bool TW_UNIFORM = false;
    __device__ static uint32_t base(uint32_t tid) { return tid << 4; }
    static constexpr uint32_t off(int r) { return (uint32_t)r; }
};

template <class Pat, class E>
__device__ __forceinline__ void put16(E *lds, uint32_t tid, const E (&x)[16]) {
    const uint32_t pb = swz(Pat::base(tid));
#pragma unroll
    for (int r = 0; r < 16; r++) lds[pb ^ swz(Pat::off(r))] = x[r];
}
template <class Pat, class E>
__device__ __forceinline__ void get16(const E *lds, uint32_t tid, E (&x)[16]) {
    const uint32_t pb = swz(Pat::base(tid));
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = lds[pb ^ swz(Pat::off(r))];
}

template <class F, int LOGN, class Pat, int KHI, int KLO>
__device__ __forceinline__ void fwd16(typename F::E (&x)[16], uint32_t tid, const typename F::TW *__restrict__ tw, const Limb<F> &P) {
    const uint32_t base = Pat::TW_UNIFORM ? 0u : Pat::base(tid);
#pragma unroll
    for (int k = KHI; k >= KLO; k--) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + ((1u << (LOGN - 1 - b)) + (base >> (b + 1)));
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW w = p[Pat::off(r) >> (b + 1)];
            F::fwd_bfly(x[r], x[r | (1 << k)], w, P);
        }
    }
}
template <class F, int LOGN, class Pat, int KLO, int KHI>
__device__ __forceinline__ void inv16(typename F::E (&x)[16], uint32_t tid, const typename F::TW *__restrict__ itw, const Limb<F> &P) {
    const uint32_t base = Pat::TW_UNIFORM ? 0u : Pat::base(tid);
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = itw + ((1u << (LOGN - 1 - b)) + (base >> (b + 1)));
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW w = p[Pat::off(r) >> (b + 1)];
            F::inv_bfly(x[r], x[r | (1 << k)], w, P);
        }
    }
}
template <class F>
__device__ __forceinline__ void regroup16(typename F::E (&x)[16], typename F::E q, typename F::E qinv) {
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::regroup1(x[r], q, qinv);      // no-op for the integer fields
}

template <class F, int LOGN>
__device__ __forceinline__ void load16(const char *__restrict__ poly, uint32_t tid, typename F::E (&x)[16]) {
    const char *p = poly + (size_t)tid * 32;
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::load_low(p + (size_t)r * (Cfg16<LOGN>::T * 32));
}
template <class F, int LOGN>
__device__ __forceinline__ void store16(char *__restrict__ poly, const typename F::E *lds, uint32_t tid) {
    using C = Cfg16<LOGN>;
    using E = typename F::E;
    const uint32_t half = tid & 1, pb = swz(tid >> 1);
    typename F::V16 *dst = reinterpret_cast<typename F::V16 *>(poly) + tid;
#pragma unroll 16
    for (int s = 0; s < 32; s++) {
        E v = lds[pb ^ swz((uint32_t)s * (C::T / 2))];
        __builtin_nontemporal_store(F::pack(half ? (E)0 : v), dst + (size_t)s * C::T);
    }
}

// natural-order coefficients in pattern A -> NTT values in pattern Z (lazy range)
template <class F, int LOGN>
__device__ __forceinline__ void fwd_core16(typename F::E (&x)[16], typename F::E *lds, uint32_t tid, const Limb<F> &P) {
    using C = Cfg16<LOGN>;
    fwd16<F, LOGN, P16A<LOGN>, 3, 0>(x, tid, P.tw, P);
    put16<P16A<LOGN>>(lds, tid, x);
    __syncthreads();
    using M1 = P16Mid<LOGN, LOGN - 8>;
    get16<M1>(lds, tid, x);
    fwd16<F, LOGN, M1, 3, 0>(x, tid, P.tw, P);
    put16<M1>(lds, tid, x);                          // the slots this thread just read
    __syncthreads();
    if constexpr (C::NG == 4) {
        using M2 = P16Mid<LOGN, LOGN - 12>;
        get16<M2>(lds, tid, x);
        fwd16<F, LOGN, M2, 3, 0>(x, tid, P.tw, P);
        put16<M2>(lds, tid, x);
        __syncthreads();
    }
    get16<P16Z<LOGN>>(lds, tid, x);
    fwd16<F, LOGN, P16Z<LOGN>, C::REM - 1, 0>(x, tid, P.tw, P);
}
// NTT values in pattern Z -> coefficients in pattern A, scaled by the (ninv..) constants
template <class F, int LOGN>
__device__ __forceinline__ void inv_core16(typename F::E (&x)[16], typename F::E *lds, uint32_t tid, const Limb<F> &P,
                                           typename F::E ninv, typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s) {
    using C = Cfg16<LOGN>;
    inv16<F, LOGN, P16Z<LOGN>, 0, 3>(x, tid, P.itw, P);
    regroup16<F>(x, P.q, P.qinv);
    put16<P16Z<LOGN>>(lds, tid, x);
    __syncthreads();
    using Y1 = P16Mid<LOGN, 4>;
    get16<Y1>(lds, tid, x);
    inv16<F, LOGN, Y1, 0, 3>(x, tid, P.itw, P);
    regroup16<F>(x, P.q, P.qinv);
    put16<Y1>(lds, tid, x);
    __syncthreads();
    if constexpr (C::NG == 4) {
        using Y2 = P16Mid<LOGN, 8>;
        get16<Y2>(lds, tid, x);
        inv16<F, LOGN, Y2, 0, 3>(x, tid, P.itw, P);
        regroup16<F>(x, P.q, P.qinv);
        put16<Y2>(lds, tid, x);
        __syncthreads();
    }
    get16<P16A<LOGN>>(lds, tid, x);
    // index bits [4*(NG-1), LOGN-1) <-> r-bits [4-REM, 3) ; bit LOGN-1 <-> r-bit 3 is the scaled last stage
    inv16<F, LOGN, P16A<LOGN>, 4 - C::REM, 2>(x, tid, P.itw, P);
#pragma unroll
    for (int r = 0; r < 8; r++) F::inv_last(x[r], x[r | 8], P.q, P.q2, ninv, ninv_s, ninvw, ninvw_s);
}

template <class F, int LOGN, int MINW = 1>
__global__ void __launch_bounds__(Cfg16<LOGN>::T, MINW)
ntt16_forward_kernel(char *__restrict__ data, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = Cfg16<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::N];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    char *poly = data + (size_t)p * (C::N * 32);
    E x[16];
    load16<F, LOGN>(poly, tid, x);
    fwd_core16<F, LOGN>(x, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    put16<P16Z<LOGN>>(lds, tid, x);
    __syncthreads();
    store16<F, LOGN>(poly, lds, tid);
}

template <class F, int LOGN, int MINW = 1>
__global__ void __launch_bounds__(Cfg16<LOGN>::T, MINW)
ntt16_inverse_kernel(char *__restrict__ data, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = Cfg16<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::N];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    char *poly = data + (size_t)p * (C::N * 32);
    E x[16];
    load16<F, LOGN>(poly, tid, x);
    put16<P16A<LOGN>>(lds, tid, x);
    __syncthreads();
    get16<P16Z<LOGN>>(lds, tid, x);
    inv_core16<F, LOGN>(x, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::canon_inv(x[r], P.q);
    put16<P16A<LOGN>>(lds, tid, x);
    __syncthreads();
    store16<F, LOGN>(poly, lds, tid);
}

// PREFETCH_B: issue b's loads before a's transform (hides their HBM latency, costs 32 VGPRs for the whole transform)
template <class F, int LOGN, int MINW = 1, bool PREFETCH_B = false>
__global__ void __launch_bounds__(Cfg16<LOGN>::T, MINW)
ntt16_multiply_kernel(char *__restrict__ res, const char *__restrict__ a, const char *__restrict__ b,
                      const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = Cfg16<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::N];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)p * (C::N * 32);
    E x[16], y[16];
    load16<F, LOGN>(a + off, tid, x);
    if (PREFETCH_B) load16<F, LOGN>(b + off, tid, y);
    fwd_core16<F, LOGN>(x, lds, tid, P);
    if (!PREFETCH_B) load16<F, LOGN>(b + off, tid, y);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    __syncthreads();
    fwd_core16<F, LOGN>(y, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::pw_mul(x[r], y[r], P.q, P.qinv);
    inv_core16<F, LOGN>(x, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::canon_inv(x[r], P.q);
    put16<P16A<LOGN>>(lds, tid, x);
    __syncthreads();
    store16<F, LOGN>(res + off, lds, tid);
}

// r = a0 (*) b1 + a1 (*) b0 (the c1 term of the tensor product)
template <class F, int LOGN, int MINW = 1>
__global__ void __launch_bounds__(Cfg16<LOGN>::T, MINW)
ntt16_mac2_kernel(char *__restrict__ res, const char *__restrict__ a0, const char *__restrict__ b1,
                  const char *__restrict__ a1, const char *__restrict__ b0, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = Cfg16<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::N];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)p * (C::N * 32);
    E x[16], y[16], acc[16];
    load16<F, LOGN>(a0 + off, tid, x);
    fwd_core16<F, LOGN>(x, lds, tid, P);
    load16<F, LOGN>(b1 + off, tid, y);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    __syncthreads();
    fwd_core16<F, LOGN>(y, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = F::pw_mul(x[r], y[r], P.q, P.qinv);
    load16<F, LOGN>(a1 + off, tid, x);
    __syncthreads();
    fwd_core16<F, LOGN>(x, lds, tid, P);
    load16<F, LOGN>(b0 + off, tid, y);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    __syncthreads();
    fwd_core16<F, LOGN>(y, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = F::pw_add(acc[r], F::pw_mul(x[r], y[r], P.q, P.qinv), P.q, P.q2);
    inv_core16<F, LOGN>(acc, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = F::canon_inv(acc[r], P.q);
    put16<P16A<LOGN>>(lds, tid, acc);
    __syncthreads();
    store16<F, LOGN>(res + off, lds, tid);
}

// Packed key tables for the 16-per-thread key-switch kernel: element (chunk c, thread tid, e) = KEY_ntt[tid*16 + c*VPL + e] * 2^W
template <class F>
__global__ void __launch_bounds__(256)
pack_keys16_kernel(typename F::E *__restrict__ packed, const typename F::V16 *__restrict__ keys_ntt, const Limb<F> *__restrict__ limbs,
                   uint32_t L, uint32_t log_n, uint32_t num_keys) {
    using E = typename F::E;
    constexpr uint32_t VPL = 16 / sizeof(E);
    const uint32_t n = 1u << log_n, T = n >> 4;
    const size_t total = (size_t)num_keys * L * n, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint32_t x = (uint32_t)(g & (n - 1));
        const size_t poly = g >> log_n;
        const Limb<F> &P = limbs[(uint32_t)(poly % L)];
        const uint32_t tid = x >> 4, r = x & 15, c = r / VPL, e = r % VPL;
        packed[poly * n + ((size_t)c * T + tid) * VPL + e] = F::to_pw_operand(F::load_low(keys_ntt + g * 2), P);
    }
}

// Fused key switching, one workgroup of N/16 threads per (ciphertext, limb): see ntt_keyswitch_kernel in ntt_lds.hip.h for the
// algorithm; here every per-thread array has 16 entries, so acc0, acc1, the raw c2 limb and the digit fit 4 waves per SIMD.
template <class F, int LOGN, int MINW = 1>
__global__ void __launch_bounds__(Cfg16<LOGN>::T, MINW)
ntt16_keyswitch_kernel(char *__restrict__ c0, char *__restrict__ c1, const char *__restrict__ c2,
                       const typename F::E *__restrict__ kb, const typename F::E *__restrict__ ka,
                       const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = Cfg16<LOGN>;
    using E = typename F::E;
    constexpr int VPL = 16 / sizeof(E), NCH = 16 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    __shared__ E lds[C::N];
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * L)) * (8 * L);
    uint32_t b, i;
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / L); i = s % L; }      // same-XCD placement of a ciphertext's limbs
    else { b = bid / L; i = bid % L; }
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    E acc0[16], acc1[16], x[16], d[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }
    for (uint32_t j = 0; j < L; j++) {
        load16<F, LOGN>(c2 + ((size_t)b * L + j) * (C::N * 32), tid, x);
        for (uint32_t k = 0; k < K; k++) {
#pragma unroll
            for (int r = 0; r < 16; r++) d[r] = F::digit(x[r], k * w, w);
            fwd_core16<F, LOGN>(d, lds, tid, P);
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
            __syncthreads();
        }
    }
    inv_core16<F, LOGN>(acc0, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    load16<F, LOGN>(c0 + (size_t)p * (C::N * 32), tid, x);
#pragma unroll
    for (int r = 0; r < 16; r++) acc0[r] = F::ew_add(F::canon_inv(acc0[r], P.q), x[r], P.q);
    put16<P16A<LOGN>>(lds, tid, acc0);
    __syncthreads();
    store16<F, LOGN>(c0 + (size_t)p * (C::N * 32), lds, tid);
    __syncthreads();
    inv_core16<F, LOGN>(acc1, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    load16<F, LOGN>(c1 + (size_t)p * (C::N * 32), tid, x);
#pragma unroll
    for (int r = 0; r < 16; r++) acc1[r] = F::ew_add(F::canon_inv(acc1[r], P.q), x[r], P.q);
    put16<P16A<LOGN>>(lds, tid, acc1);
    __syncthreads();
    store16<F, LOGN>(c1 + (size_t)p * (C::N * 32), lds, tid);
}

}  // namespace fhe_dev
