// ntt_lds_small.hip.h -- the fused polynomial product for SMALL batches: 16 coefficients per thread (T = N/16 threads per workgroup).
//
// NTTEngine::multiply is called one polynomial at a time by the reference's own callers (include/ntt.cuh:78-84, tests/test_fhe.cu:65-124).
// With fewer (polynomial, limb) pairs than CUs the 32-per-thread kernel of ntt_lds.hip.h is bound by ONE workgroup's latency, not by
// throughput: at N = 8192 its 256 threads are one wave per SIMD, an in-order wave issues a VALU instruction every ~4.5 cycles (5 600 of them
// for the three dependent transforms), and every cold twiddle load exposes an HBM round trip: 22-24 us per product at batch 1.  Here
//   * the same polynomial is spread over twice the waves (two per SIMD at N = 8192): half the instructions per wave;
//   * log2 N = 4 + 4 + 4 (+ REM) stages in 3 or 4 register groups (radix 16), exchanged through an XOR-swizzled LDS image
//     (slot(i) = i ^ ((i >> 4) & 31): the lane -> bank map is GF(2)-linear and invertible for every access pattern used, so the exchanges are
//     conflict-free for the 32-lane groups of a 4-byte LDS access and the 16 / 32-lane groups of an 8-byte one; slot = slotbase(tid) ^ const(r));
//   * EVERY twiddle the three transforms need (15 per non-uniform register group) is loaded into registers at the top of the kernel
//     together with both operands: one memory round trip in all (the workgroup has 256 VGPRs per lane to itself).
// Same tables, same HBM access shapes, same lazy butterflies and bit-identical results as ntt_multiply_kernel; the host uses it while
// batch x limbs stays below the number of CUs (lds_small_batch in lds_launch.h).  16-per-thread forms of the throughput kernels were
// measured in round 1 (scratch/experiments/ntt_lds16.hip.h: the extra exchange costs what the occupancy buys) -- this is the latency case.
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
    static constexpr int REM = LOGN - 4 * (NG - 1);     // stages of the last forward group / of the uniform last inverse group (1..4)
};

__device__ __forceinline__ constexpr uint32_t swz16(uint32_t i) { return i ^ ((i >> 4) & 31u); }

// patterns: logical index = base(tid) | off(r); LDS slot = swz16(base(tid)) ^ swz16(off(r))  (disjoint bit-fields)
template <int LOGN> struct P16A {                       // r <-> index bits [LOGT, LOGN): uniform twiddles
    using C = Cfg16<LOGN>;
    static constexpr int BIT0 = C::LOGT;
    __device__ static uint32_t base(uint32_t tid) { return tid; }
    static constexpr uint32_t off(int r) { return (uint32_t)r << C::LOGT; }
};
template <int LOGN, int LO> struct P16Mid {             // r <-> index bits [LO, LO+4)
    static constexpr int BIT0 = LO;
    __device__ static uint32_t base(uint32_t tid) { return ((tid >> LO) << (LO + 4)) | (tid & ((1u << LO) - 1)); }
    static constexpr uint32_t off(int r) { return (uint32_t)r << LO; }
};
template <int LOGN> struct P16Z {                       // r <-> index bits [0, 4): 16 consecutive coefficients per thread
    static constexpr int BIT0 = 0;
    __device__ static uint32_t base(uint32_t tid) { return tid << 4; }
    static constexpr uint32_t off(int r) { return (uint32_t)r; }
};

template <class Pat, class E>
__device__ __forceinline__ void put16(E *lds, uint32_t tid, const E (&x)[16]) {
    const uint32_t pb = swz16(Pat::base(tid));
#pragma unroll
    for (int r = 0; r < 16; r++) lds[pb ^ swz16(Pat::off(r))] = x[r];
}
template <class Pat, class E>
__device__ __forceinline__ void get16(const E *lds, uint32_t tid, E (&x)[16]) {
    const uint32_t pb = swz16(Pat::base(tid));
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = lds[pb ^ swz16(Pat::off(r))];
}

// the 2^(KHI+1) - 2^KLO per-lane twiddles of the stages on r-bits KLO..KHI of pattern Pat; stage k, twiddle j = r >> (k+1): slot (8 >> k) - 1 + j
template <class F, int LOGN, class Pat, int KHI, int KLO>
__device__ __forceinline__ void preload16(typename F::TW (&w)[15], uint32_t tid, const typename F::TW *__restrict__ tw) {
    const uint32_t base = Pat::base(tid);
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + ((1u << (LOGN - 1 - b)) + (base >> (b + 1)));
#pragma unroll
        for (int j = 0; j < (8 >> k); j++) w[(8 >> k) - 1 + j] = load_global(p + j);
    }
}
// the wave-uniform twiddles of the pattern-A group (stage k, twiddle j = r >> (k+1), same slots): scalar loads, but issued at the top of the
// kernel like the others -- left where they are used, the first transform waited for four dependent scalar-cache misses (~9 K cycles,
// scripts/small_batch_timeline.py)
template <class F, int LOGN, int KHI, int KLO>
__device__ __forceinline__ void preload16_uniform(typename F::TW (&w)[15], const typename F::TW *__restrict__ tw) {
    using Pat = P16A<LOGN>;
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + (1u << (LOGN - 1 - b));
#pragma unroll
        for (int j = 0; j < (8 >> k); j++) w[(8 >> k) - 1 + j] = load_global(p + j);
    }
}
template <class F, int KHI, int KLO>
__device__ __forceinline__ void fwd16_pre(typename F::E (&x)[16], const typename F::TW (&w)[15], const Limb<F> &P) {
#pragma unroll
    for (int k = KHI; k >= KLO; k--) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r & (1 << k)) continue;
            F::fwd_bfly(x[r], x[r | (1 << k)], w[(8 >> k) - 1 + (r >> (k + 1))], P);
        }
    }
}
template <class F, int KLO, int KHI>
__device__ __forceinline__ void inv16_pre(typename F::E (&x)[16], const typename F::TW (&w)[15], const Limb<F> &P) {
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r & (1 << k)) continue;
            F::inv_bfly(x[r], x[r | (1 << k)], w[(8 >> k) - 1 + (r >> (k + 1))], P);
        }
    }
}
// stages with wave-uniform twiddles (pattern A): scalar loads where they are used
template <class F, int LOGN, int KHI, int KLO>
__device__ __forceinline__ void fwd16_uniform(typename F::E (&x)[16], const typename F::TW *__restrict__ tw, const Limb<F> &P) {
    using Pat = P16A<LOGN>;
#pragma unroll
    for (int k = KHI; k >= KLO; k--) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + (1u << (LOGN - 1 - b));
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r & (1 << k)) continue;
            F::fwd_bfly(x[r], x[r | (1 << k)], load_global(p + (Pat::off(r) >> (b + 1))), P);
        }
    }
}
template <class F, int LOGN, int KLO, int KHI>
__device__ __forceinline__ void inv16_uniform(typename F::E (&x)[16], const typename F::TW *__restrict__ itw, const Limb<F> &P) {
    using Pat = P16A<LOGN>;
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = itw + (1u << (LOGN - 1 - b));
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r & (1 << k)) continue;
            F::inv_bfly(x[r], x[r | (1 << k)], load_global(p + (Pat::off(r) >> (b + 1))), P);
        }
    }
}
template <class F>
__device__ __forceinline__ void regroup16(typename F::E (&x)[16], typename F::E q, typename F::E qinv) {
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::regroup1(x[r], q, qinv);      // no-op for the integer fields
}

// every per-lane twiddle of a forward + inverse transform, in registers (NG - 1 non-uniform groups each way)
template <class F, int LOGN>
struct Twiddles16 {
    using C = Cfg16<LOGN>;
    typename F::TW fa[15], f1[15], f2[15], fz[15], iz[15], i1[15], i2[15], ia[15];     // fa / ia: the uniform pattern-A groups (SGPRs)
    __device__ __forceinline__ void load(uint32_t tid, const Limb<F> &P) {
        preload16_uniform<F, LOGN, 3, 0>(fa, P.tw);
        if constexpr (C::REM > 1) preload16_uniform<F, LOGN, 2, 4 - C::REM>(ia, P.itw);
        preload16<F, LOGN, P16Mid<LOGN, LOGN - 8>, 3, 0>(f1, tid, P.tw);
        if constexpr (C::NG == 4) preload16<F, LOGN, P16Mid<LOGN, LOGN - 12>, 3, 0>(f2, tid, P.tw);
        preload16<F, LOGN, P16Z<LOGN>, C::REM - 1, 0>(fz, tid, P.tw);
        preload16<F, LOGN, P16Z<LOGN>, 3, 0>(iz, tid, P.itw);
        preload16<F, LOGN, P16Mid<LOGN, 4>, 3, 0>(i1, tid, P.itw);
        if constexpr (C::NG == 4) preload16<F, LOGN, P16Mid<LOGN, 8>, 3, 0>(i2, tid, P.itw);
    }
};

// natural-order coefficients in pattern A -> NTT values in pattern Z (lazy range)
template <class F, int LOGN>
__device__ __forceinline__ void fwd_core16(typename F::E (&x)[16], typename F::E *lds, uint32_t tid, const Limb<F> &P, const Twiddles16<F, LOGN> &W) {
    using C = Cfg16<LOGN>;
    fwd16_pre<F, 3, 0>(x, W.fa, P);
    put16<P16A<LOGN>>(lds, tid, x);
    __syncthreads();
    using M1 = P16Mid<LOGN, LOGN - 8>;
    get16<M1>(lds, tid, x);
    fwd16_pre<F, 3, 0>(x, W.f1, P);
    put16<M1>(lds, tid, x);                          // the slots this thread just read
    __syncthreads();
    if constexpr (C::NG == 4) {
        using M2 = P16Mid<LOGN, LOGN - 12>;
        get16<M2>(lds, tid, x);
        fwd16_pre<F, 3, 0>(x, W.f2, P);
        put16<M2>(lds, tid, x);
        __syncthreads();
    }
    get16<P16Z<LOGN>>(lds, tid, x);
    fwd16_pre<F, C::REM - 1, 0>(x, W.fz, P);
}
// NTT values in pattern Z -> coefficients in pattern A, scaled by the (ninv..) constants
template <class F, int LOGN>
__device__ __forceinline__ void inv_core16(typename F::E (&x)[16], typename F::E *lds, uint32_t tid, const Limb<F> &P, const Twiddles16<F, LOGN> &W,
                                           typename F::E ninv, typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s) {
    using C = Cfg16<LOGN>;
    inv16_pre<F, 0, 3>(x, W.iz, P);
    regroup16<F>(x, P.q, P.qinv);
    put16<P16Z<LOGN>>(lds, tid, x);
    __syncthreads();
    using Y1 = P16Mid<LOGN, 4>;
    get16<Y1>(lds, tid, x);
    inv16_pre<F, 0, 3>(x, W.i1, P);
    regroup16<F>(x, P.q, P.qinv);
    put16<Y1>(lds, tid, x);
    __syncthreads();
    if constexpr (C::NG == 4) {
        using Y2 = P16Mid<LOGN, 8>;
        get16<Y2>(lds, tid, x);
        inv16_pre<F, 0, 3>(x, W.i2, P);
        regroup16<F>(x, P.q, P.qinv);
        put16<Y2>(lds, tid, x);
        __syncthreads();
    }
    get16<P16A<LOGN>>(lds, tid, x);
    // index bits [4*(NG-1), LOGN-1) <-> r-bits [4-REM, 3) ; bit LOGN-1 <-> r-bit 3 is the scaled last stage
    if constexpr (C::REM > 1) inv16_pre<F, 4 - C::REM, 2>(x, W.ia, P);
#pragma unroll
    for (int r = 0; r < 8; r++) F::inv_last(x[r], x[r | 8], P.q, P.q2, ninv, ninv_s, ninvw, ninvw_s);
}

template <class F, int LOGN>
__device__ __forceinline__ void load16(const char *__restrict__ poly, uint32_t tid, typename F::E (&x)[16]) {
    const char *p = poly + (size_t)tid * 32;
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::load_low(p + (size_t)r * (Cfg16<LOGN>::T * 32));
}
// whole polynomial from the swizzled image as full containers: consecutive lanes write consecutive 16-byte halves
template <class F, int LOGN>
__device__ __forceinline__ void store16(char *__restrict__ poly, const typename F::E *lds, uint32_t tid) {
    using C = Cfg16<LOGN>;
    using E = typename F::E;
    const uint32_t half = tid & 1, pb = swz16(tid >> 1);
    typename F::V16 *dst = reinterpret_cast<typename F::V16 *>(poly) + tid;
#pragma unroll 16
    for (int s = 0; s < 32; s++) {
        const E v = lds[pb ^ swz16((uint32_t)s * (C::T / 2))];
        __builtin_nontemporal_store(F::pack(half ? (E)0 : v), dst + (size_t)s * C::T);
    }
}

#ifdef FHE_STAMPS      // diagnostic build only (scripts/small_batch_timeline.py): s_memtime at the phase boundaries of workgroup 0, wave 0
__device__ unsigned long long g_stamps16[16];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps16[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// r = INTT(NTT(a) .* NTT(b)) for few polynomials: grid.x = batch * L, workgroup p handles polynomial p (limb p % L).
// res may alias a and / or b (both operands are in registers before the first store); bcast != 0: b holds ONE RNS polynomial.
template <class F, int LOGN>
__global__ void __launch_bounds__(Cfg16<LOGN>::T)
ntt16_multiply_kernel(char *res, const char *a, const char *b, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t bcast) {
    using C = Cfg16<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::N];
    const uint32_t tid = threadIdx.x, p = blockIdx.x, limb = p % L;
    const Limb<F> P = limbs[limb];
    const size_t off = (size_t)p * (C::N * 32);
    E x[16], y[16];
    STAMP(0);
    // Twiddles first: the operand loads (512 KiB of container lines through ONE CU's 64 B/clk vector-memory path: ~9 K cycles, during
    // which the issuing waves are blocked on the full address FIFO) would otherwise delay them by the same ~9 K cycles (timeline of
    // scripts/small_batch_timeline.py: first forward transform 18.5 K cycles against 9.1 K for the second)
    Twiddles16<F, LOGN> W;
    W.load(tid, P);
    load16<F, LOGN>(a + off, tid, x);
    load16<F, LOGN>(b + (size_t)(bcast ? limb : p) * (C::N * 32), tid, y);
    STAMP(1);
#ifdef FHE_STAMPS
    { E t = 0;
#pragma unroll
      for (int r = 0; r < 16; r++) t ^= x[r] ^ y[r];
      asm volatile("" :: "v"(t) : "memory"); }          // force the operands to have arrived
    STAMP(2);
#endif
    fwd_core16<F, LOGN>(x, lds, tid, P, W);
    STAMP(3);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    __syncthreads();                                  // all pattern-Z reads of a are done before b's first exchange
    fwd_core16<F, LOGN>(y, lds, tid, P, W);
    STAMP(4);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::pw_mul(x[r], y[r], P.q, P.qinv);
    inv_core16<F, LOGN>(x, lds, tid, P, W, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::canon_inv(x[r], P.q);
    STAMP(5);
    put16<P16A<LOGN>>(lds, tid, x);                   // the slots this thread read last
    __syncthreads();
    store16<F, LOGN>(res + off, lds, tid);
    STAMP(6);
}

}  // namespace fhe_dev
