// ntt_lds16.hip.h -- LDS-resident negacyclic NTT / polymul for the 8-byte residue fields (F52, F64), 16 coefficients
// per thread.
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
__device__ __forceinline__ void fwd16(typename F::E (&x)[16], uint32_t tid, const typename F::TW *__restrict__ tw, typename F::E q, typename F::E q2) {
    const uint32_t base = Pat::TW_UNIFORM ? 0u : Pat::base(tid);
#pragma unroll
    for (int k = KHI; k >= KLO; k--) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + ((1u << (LOGN - 1 - b)) + (base >> (b + 1)));
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW w = p[Pat::off(r) >> (b + 1)];
            F::fwd_bfly(x[r], x[r | (1 << k)], w, q, q2);
        }
    }
}
template <class F, int LOGN, class Pat, int KLO, int KHI>
__device__ __forceinline__ void inv16(typename F::E (&x)[16], uint32_t tid, const typename F::TW *__restrict__ itw, typename F::E q, typename F::E q2) {
    const uint32_t base = Pat::TW_UNIFORM ? 0u : Pat::base(tid);
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = itw + ((1u << (LOGN - 1 - b)) + (base >> (b + 1)));
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW w = p[Pat::off(r) >> (b + 1)];
            F::inv_bfly(x[r], x[r | (1 << k)], w, q, q2);
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
    fwd16<F, LOGN, P16A<LOGN>, 3, 0>(x, tid, P.tw, P.q, P.q2);
    put16<P16A<LOGN>>(lds, tid, x);
    __syncthreads();
    using M1 = P16Mid<LOGN, LOGN - 8>;
    get16<M1>(lds, tid, x);
    fwd16<F, LOGN, M1, 3, 0>(x, tid, P.tw, P.q, P.q2);
    put16<M1>(lds, tid, x);                          // the slots this thread just read
    __syncthreads();
    if constexpr (C::NG == 4) {
        using M2 = P16Mid<LOGN, LOGN - 12>;
        get16<M2>(lds, tid, x);
        fwd16<F, LOGN, M2, 3, 0>(x, tid, P.tw, P.q, P.q2);
        put16<M2>(lds, tid, x);
        __syncthreads();
    }
    get16<P16Z<LOGN>>(lds, tid, x);
    fwd16<F, LOGN, P16Z<LOGN>, C::REM - 1, 0>(x, tid, P.tw, P.q, P.q2);
}
// NTT values in pattern Z -> coefficients in pattern A, scaled by the (ninv..) constants
template <class F, int LOGN>
__device__ __forceinline__ void inv_core16(typename F::E (&x)[16], typename F::E *lds, uint32_t tid, const Limb<F> &P,
                                           typename F::E ninv, typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s) {
    using C = Cfg16<LOGN>;
    inv16<F, LOGN, P16Z<LOGN>, 0, 3>(x, tid, P.itw, P.q, P.q2);
    regroup16<F>(x, P.q, P.qinv);
    put16<P16Z<LOGN>>(lds, tid, x);
    __syncthreads();
    using Y1 = P16Mid<LOGN, 4>;
    get16<Y1>(lds, tid, x);
    inv16<F, LOGN, Y1, 0, 3>(x, tid, P.itw, P.q, P.q2);
    regroup16<F>(x, P.q, P.qinv);
    put16<Y1>(lds, tid, x);
    __syncthreads();
    if constexpr (C::NG == 4) {
        using Y2 = P16Mid<LOGN, 8>;
        get16<Y2>(lds, tid, x);
        inv16<F, LOGN, Y2, 0, 3>(x, tid, P.itw, P.q, P.q2);
        regroup16<F>(x, P.q, P.qinv);
        put16<Y2>(lds, tid, x);
        __syncthreads();
    }
    get16<P16A<LOGN>>(lds, tid, x);
    // index bits [4*(NG-1), LOGN-1) <-> r-bits [4-REM, 3) ; bit LOGN-1 <-> r-bit 3 is the scaled last stage
    inv16<F, LOGN, P16A<LOGN>, 4 - C::REM, 2>(x, tid, P.itw, P.q, P.q2);
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

}  // namespace fhe_dev
