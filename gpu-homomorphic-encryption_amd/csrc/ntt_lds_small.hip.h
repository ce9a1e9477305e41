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
__device__ __forceinline__ void preload16(typename F::TW (&w)[15], uint32_t tid, const typename F::TW *__restrict__ tw, uint32_t pre = 1) {
    const uint32_t base = Pat::base(tid);
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + ((pre << (LOGN - 1 - b)) + (base >> (b + 1)));     // pre != 1: block pre - 2^k of a larger transform (see fwd_stages, SUB)
#pragma unroll
        for (int j = 0; j < (8 >> k); j++) w[(8 >> k) - 1 + j] = load_global(p + j);
    }
}
// the wave-uniform twiddles of the pattern-A group (stage k, twiddle j = r >> (k+1), same slots): scalar loads, but issued at the top of the
// kernel like the others -- left where they are used, the first transform waited for four dependent scalar-cache misses (~9 K cycles,
// scripts/small_batch_timeline.py)
template <class F, int LOGN, int KHI, int KLO>
__device__ __forceinline__ void preload16_uniform(typename F::TW (&w)[15], const typename F::TW *__restrict__ tw, uint32_t pre = 1) {
    using Pat = P16A<LOGN>;
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + (pre << (LOGN - 1 - b));
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
    // SUB: the 2^LOGN coefficients are block pre - 2^k of a transform of 2^(LOGN + k) coefficients (twiddles from the big table; the last inverse group
    // is made of ordinary stages, the scaling belongs to the pass over the top stages)
    // only what fwd_core16 reads
    __device__ __forceinline__ void load_forward(uint32_t tid, const Limb<F> &P) {
        preload16_uniform<F, LOGN, 3, 0>(fa, P.tw);
        preload16<F, LOGN, P16Mid<LOGN, LOGN - 8>, 3, 0>(f1, tid, P.tw);
        if constexpr (C::NG == 4) preload16<F, LOGN, P16Mid<LOGN, LOGN - 12>, 3, 0>(f2, tid, P.tw);
        preload16<F, LOGN, P16Z<LOGN>, C::REM - 1, 0>(fz, tid, P.tw);
    }
    // only what inv_core16 (not SUB) reads
    __device__ __forceinline__ void load_inverse(uint32_t tid, const Limb<F> &P) {
        if constexpr (C::REM > 1) preload16_uniform<F, LOGN, 2, 4 - C::REM>(ia, P.itw);
        preload16<F, LOGN, P16Z<LOGN>, 3, 0>(iz, tid, P.itw);
        preload16<F, LOGN, P16Mid<LOGN, 4>, 3, 0>(i1, tid, P.itw);
        if constexpr (C::NG == 4) preload16<F, LOGN, P16Mid<LOGN, 8>, 3, 0>(i2, tid, P.itw);
    }
    template <bool SUB = false>
    __device__ __forceinline__ void load(uint32_t tid, const Limb<F> &P, uint32_t pre = 1) {
        preload16_uniform<F, LOGN, 3, 0>(fa, P.tw, pre);
        if constexpr (SUB) preload16_uniform<F, LOGN, 3, 4 - C::REM>(ia, P.itw, pre);
        else if constexpr (C::REM > 1) preload16_uniform<F, LOGN, 2, 4 - C::REM>(ia, P.itw);
        preload16<F, LOGN, P16Mid<LOGN, LOGN - 8>, 3, 0>(f1, tid, P.tw, pre);
        if constexpr (C::NG == 4) preload16<F, LOGN, P16Mid<LOGN, LOGN - 12>, 3, 0>(f2, tid, P.tw, pre);
        preload16<F, LOGN, P16Z<LOGN>, C::REM - 1, 0>(fz, tid, P.tw, pre);
        preload16<F, LOGN, P16Z<LOGN>, 3, 0>(iz, tid, P.itw, pre);
        preload16<F, LOGN, P16Mid<LOGN, 4>, 3, 0>(i1, tid, P.itw, pre);
        if constexpr (C::NG == 4) preload16<F, LOGN, P16Mid<LOGN, 8>, 3, 0>(i2, tid, P.itw, pre);
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
template <class F, int LOGN, bool SUB = false>
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
    if constexpr (SUB) {        // a block of a larger transform: every stage of the group is an ordinary one
        inv16_pre<F, 4 - C::REM, 3>(x, W.ia, P);
        regroup16<F>(x, P.q, P.qinv);
    } else {
        if constexpr (C::REM > 1) inv16_pre<F, 4 - C::REM, 2>(x, W.ia, P);
#pragma unroll
        for (int r = 0; r < 8; r++) F::inv_last(x[r], x[r | 8], P.q, P.q2, ninv, ninv_s, ninvw, ninvw_s);
    }
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
    W.template load<false>(tid, P);
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

// ---- ONE polynomial over FOUR workgroups: the fused product for the smallest batches -----------------------------------------------------
// The timeline of ntt16_multiply_kernel (scripts/small_batch_timeline.py) shows what bounds a lone workgroup: one CU needs ~19 K cycles to ingest
// the 512 KiB of container lines of an operand pair and ~4.5 K to emit the result -- 10 of its 22 us are the container format moving through a
// single CU.  Here every (polynomial, limb) is shared by Q = 4 workgroups (4 CUs) in three phases, each its own launch (the stream order is the
// barrier between them; what crosses is a compact polynomial in a library workspace):
//   1. workgroup k loads a quarter of the COLUMNS of a and b (column c = coefficients c, c + N/4, c + N/2, c + 3N/4), runs the top two forward
//      stages on them in registers (radix 4, wave-uniform twiddles tw[1..3]) and writes the four results of each column to the workspace;
//   2. workgroup q runs block q (N/4 consecutive coefficients) of both operands: 16-per-thread sub-transforms with the big transform's twiddles
//      (SUB, pre = 4 + q), pointwise product, inverse sub-transform, compact result;
//   3. workgroup k runs the last two inverse stages on its columns (n^-1 folded into the final butterfly) and stores its quarter of the containers.
// Each CU ingests a quarter of the lines.  (The same three phases as ONE kernel with atomic-counter barriers between the four workgroups of a
// polynomial were measured first: 16.5 us of kernel time either way, but hipLaunchCooperativeKernel -- the only launch that guarantees the
// co-residency such barriers need -- costs ~20 us per call on this stack, and an ordinary launch of spinning workgroups is not something a
// library should do.  Three dependent launches need no guarantee and can be captured into a graph.)
template <class F, int LOGN>
struct Coop4 {
    static_assert(LOGN == 13 || LOGN == 14, "four workgroups per polynomial: N = 2^13, 2^14");
    static constexpr int LOGS = LOGN - 2;                 // log2 of a block
    using S = Cfg16<LOGS>;
    static constexpr int T = S::T;                        // threads per workgroup (128 / 256)
    static constexpr int NS = 1 << LOGS;                  // coefficients per block = columns of the top stages
#ifndef FHE_COOP_COLUMNS_PER_THREAD
#define FHE_COOP_COLUMNS_PER_THREAD 1
#endif
    // phases 1 and 3 work on COLUMNS (coefficients c, c + N/4, c + N/2, c + 3N/4) and can be cut anywhere: one column per thread = 16 workgroups per limb
    // polynomial, so that a CU ingests 32 KiB of operand containers instead of 128 KiB (-DFHE_COOP_COLUMNS_PER_THREAD=4: the first form, four workgroups)
    static constexpr int CPT = FHE_COOP_COLUMNS_PER_THREAD;
    static constexpr int CWG = NS / (T * CPT);            // column workgroups per limb polynomial (16 or 4)
    static_assert(CWG * T * CPT == NS, "columns per thread must divide the columns of a workgroup");
};
// ws: per polynomial three compact polynomials (top(a), top(b), block results)
template <class F, int LOGN>
__global__ void __launch_bounds__(1 << (LOGN - 6))      // = Coop4<F, LOGN>::T (a comma inside the macro argument would split it)
ntt_multiply4_top_kernel(const char *a, const char *b, typename F::E *ws, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t bcast) {
    using K = Coop4<F, LOGN>;
    using E = typename F::E;
    constexpr int NS = K::NS, T = K::T, N = 1 << LOGN;
    const uint32_t tid = threadIdx.x, p = blockIdx.x / K::CWG, k = blockIdx.x % K::CWG, limb = p % L;
    const Limb<F> P = limbs[limb];
    E *wa = ws + (size_t)p * (3 * N), *wb = wa + N;
    const typename F::TW t1 = load_global(P.tw + 1), t2 = load_global(P.tw + 2), t3 = load_global(P.tw + 3);
    const char *pa = a + (size_t)p * (N * 32), *pb = b + (size_t)(bcast ? limb : p) * (N * 32);
    E xa[K::CPT][4], xb[K::CPT][4];
#pragma unroll
    for (int m = 0; m < K::CPT; m++) {
        const uint32_t c = k * (K::CPT * T) + tid + m * T;
#pragma unroll
        for (int q = 0; q < 4; q++) { xa[m][q] = F::load_low(pa + (size_t)(q * NS + c) * 32); xb[m][q] = F::load_low(pb + (size_t)(q * NS + c) * 32); }
    }
#pragma unroll
    for (int m = 0; m < K::CPT; m++) {
        const uint32_t c = k * (K::CPT * T) + tid + m * T;
        F::fwd_bfly(xa[m][0], xa[m][2], t1, P); F::fwd_bfly(xa[m][1], xa[m][3], t1, P);      // index bit LOGN-1
        F::fwd_bfly(xa[m][0], xa[m][1], t2, P); F::fwd_bfly(xa[m][2], xa[m][3], t3, P);      // index bit LOGN-2
        F::fwd_bfly(xb[m][0], xb[m][2], t1, P); F::fwd_bfly(xb[m][1], xb[m][3], t1, P);
        F::fwd_bfly(xb[m][0], xb[m][1], t2, P); F::fwd_bfly(xb[m][2], xb[m][3], t3, P);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            wa[q * NS + c] = F::canon_fwd(xa[m][q], P.q, P.q2, P.qinv);
            wb[q * NS + c] = F::canon_fwd(xb[m][q], P.q, P.q2, P.qinv);
        }
    }
}
// Phase 2 with the two forward transforms SIDE BY SIDE: the workgroup is two groups of T threads (group 0: the a block, group 1: the b block, each with
// its own LDS image); NTT(b) crosses to group 0 through group 1's image in register order, group 0 multiplies and runs the inverse.  The critical path is
// two transforms instead of three.  Group 1 walks through the inverse's barriers on its stale registers (its waves sit on other SIMDs) and stores nothing.
// Same arithmetic per coefficient as the one-group form (-DFHE_COOP_ONE_GROUP): a-side canonical, b-side lazy, pw_mul.
#ifndef FHE_COOP_ONE_GROUP
template <class F, int LOGN>
__global__ void __launch_bounds__(2 << (LOGN - 6))
ntt_multiply4_block_kernel(typename F::E *ws, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using K = Coop4<F, LOGN>;
    using E = typename F::E;
    constexpr int LOGS = K::LOGS, NS = K::NS, T = K::T, N = 1 << LOGN;
    __shared__ E lds[2][NS];
    const uint32_t g = threadIdx.x / T, tid = threadIdx.x % T, p = blockIdx.x >> 2, k = blockIdx.x & 3;      // g is wave-uniform (T is a multiple of 64)
    const Limb<F> P = limbs[p % L];
    E *wa = ws + (size_t)p * (3 * N), *wb = wa + N, *wr = wb + N;
    Twiddles16<F, LOGS> W;
    W.template load<true>(tid, P, 4 + k);
    E x[16], y[16];
    const E *src = g ? wb : wa;
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = src[k * NS + tid + r * T];
    fwd_core16<F, LOGS>(x, lds[g], tid, P, W);
    if (g) put16<P16Z<LOGS>>(lds[1], tid, x);         // the slots this thread read last
    __syncthreads();
    if (!g) {
        get16<P16Z<LOGS>>(lds[1], tid, y);
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = F::pw_mul(F::canon_fwd(x[r], P.q, P.q2, P.qinv), y[r], P.q, P.qinv);   // carries 2^-W until the last stage (ninv_r constants)
    }
    inv_core16<F, LOGS, true>(x, lds[g], tid, P, W, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    if (!g) {
#pragma unroll
        for (int r = 0; r < 16; r++) wr[k * NS + tid + r * T] = F::canon_inv(x[r], P.q);
    }
}
#else
template <class F, int LOGN>
__global__ void __launch_bounds__(1 << (LOGN - 6))
ntt_multiply4_block_kernel(typename F::E *ws, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using K = Coop4<F, LOGN>;
    using E = typename F::E;
    constexpr int LOGS = K::LOGS, NS = K::NS, T = K::T, N = 1 << LOGN;
    __shared__ E lds[NS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x >> 2, k = blockIdx.x & 3;
    const Limb<F> P = limbs[p % L];
    E *wa = ws + (size_t)p * (3 * N), *wb = wa + N, *wr = wb + N;
    Twiddles16<F, LOGS> W;
    W.template load<true>(tid, P, 4 + k);
    E x[16], y[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { x[r] = wa[k * NS + tid + r * T]; y[r] = wb[k * NS + tid + r * T]; }
    fwd_core16<F, LOGS>(x, lds, tid, P, W);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    __syncthreads();
    fwd_core16<F, LOGS>(y, lds, tid, P, W);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::pw_mul(x[r], y[r], P.q, P.qinv);               // carries 2^-W until the last stage (ninv_r constants)
    inv_core16<F, LOGS, true>(x, lds, tid, P, W, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
#pragma unroll
    for (int r = 0; r < 16; r++) wr[k * NS + tid + r * T] = F::canon_inv(x[r], P.q);
}
#endif
template <class F, int LOGN>
__global__ void __launch_bounds__(1 << (LOGN - 6))
ntt_multiply4_last_kernel(char *res, const typename F::E *ws, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using K = Coop4<F, LOGN>;
    using E = typename F::E;
    constexpr int NS = K::NS, T = K::T, N = 1 << LOGN;
    const uint32_t tid = threadIdx.x, p = blockIdx.x / K::CWG, k = blockIdx.x % K::CWG;
    const Limb<F> P = limbs[p % L];
    const E *wr = ws + (size_t)p * (3 * N) + 2 * N;
    const typename F::TW i2 = load_global(P.itw + 2), i3 = load_global(P.itw + 3);
    typename F::V16 *out = reinterpret_cast<typename F::V16 *>(res + (size_t)p * (N * 32));
#pragma unroll
    for (int m = 0; m < K::CPT; m++) {
        const uint32_t c = k * (K::CPT * T) + tid + m * T;
        E v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = wr[q * NS + c];
        F::inv_bfly(v[0], v[1], i2, P); F::inv_bfly(v[2], v[3], i3, P);                     // index bit LOGN-2
        F::inv_last(v[0], v[2], P.q, P.q2, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);   // index bit LOGN-1, scaled
        F::inv_last(v[1], v[3], P.q, P.q2, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            typename F::V16 *d = out + 2 * (size_t)(q * NS + c);
            __builtin_nontemporal_store(F::pack(F::canon_inv(F::regroup1(v[q], P.q, P.qinv), P.q)), d);
            __builtin_nontemporal_store(F::pack((E)0), d + 1);
        }
    }
}

// ---- the tensor product of few ciphertexts the same way: four workgroups per (ciphertext, limb), three launches, COMPACT outputs -------------
// (the first half of the one-call multiply + relinearise; ws: per limb polynomial 4 operand + 3 result compact polynomials)
template <class F, int LOGN>
__global__ void __launch_bounds__(1 << (LOGN - 6))
ntt_ct4_top_kernel(const char *a0, const char *a1, const char *b0, const char *b1, typename F::E *ws, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using K = Coop4<F, LOGN>;
    using E = typename F::E;
    constexpr int NS = K::NS, T = K::T, N = 1 << LOGN;
    const uint32_t tid = threadIdx.x, p = blockIdx.x / K::CWG, k = blockIdx.x % K::CWG;
    const uint32_t which = blockIdx.y;                    // operand 0..3 (grid.y = 4)
    const Limb<F> P = limbs[p % L];
    const char *src = (which == 0 ? a0 : which == 1 ? a1 : which == 2 ? b0 : b1) + (size_t)p * (N * 32);
    E *w = ws + (size_t)p * (7 * N) + (size_t)which * N;
    const typename F::TW t1 = load_global(P.tw + 1), t2 = load_global(P.tw + 2), t3 = load_global(P.tw + 3);
    E x[K::CPT][4];
#pragma unroll
    for (int m = 0; m < K::CPT; m++) {
        const uint32_t c = k * (K::CPT * T) + tid + m * T;
#pragma unroll
        for (int q = 0; q < 4; q++) x[m][q] = F::load_low(src + (size_t)(q * NS + c) * 32);
    }
#pragma unroll
    for (int m = 0; m < K::CPT; m++) {
        const uint32_t c = k * (K::CPT * T) + tid + m * T;
        F::fwd_bfly(x[m][0], x[m][2], t1, P); F::fwd_bfly(x[m][1], x[m][3], t1, P);
        F::fwd_bfly(x[m][0], x[m][1], t2, P); F::fwd_bfly(x[m][2], x[m][3], t3, P);
#pragma unroll
        for (int q = 0; q < 4; q++) w[q * NS + c] = F::canon_fwd(x[m][q], P.q, P.q2, P.qinv);
    }
}
template <class F, int LOGN>
__global__ void __launch_bounds__(1 << (LOGN - 6))
ntt_ct4_block1_kernel(typename F::E *ws, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using K = Coop4<F, LOGN>;
    using E = typename F::E;
    constexpr int LOGS = K::LOGS, NS = K::NS, T = K::T, N = 1 << LOGN;
    __shared__ E lds[NS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x >> 2, k = blockIdx.x & 3;
    const Limb<F> P = limbs[p % L];
    E *w = ws + (size_t)p * (7 * N);
    Twiddles16<F, LOGS> W;
    W.template load<true>(tid, P, 4 + k);
    E A0[16], A1[16], B0[16], B1[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t i = k * NS + tid + r * T;
        A0[r] = w[i]; A1[r] = w[N + i]; B0[r] = w[2 * N + i]; B1[r] = w[3 * N + i];
    }
    fwd_core16<F, LOGS>(A0, lds, tid, P, W);
    __syncthreads();
    fwd_core16<F, LOGS>(A1, lds, tid, P, W);
    __syncthreads();
    fwd_core16<F, LOGS>(B0, lds, tid, P, W);
    __syncthreads();
    fwd_core16<F, LOGS>(B1, lds, tid, P, W);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const E u0 = F::canon_fwd(A0[r], P.q, P.q2, P.qinv), u1 = F::canon_fwd(A1[r], P.q, P.q2, P.qinv);
        const E v0 = B0[r], v1 = B1[r];
        A0[r] = F::pw_mul(u0, v0, P.q, P.qinv);
        A1[r] = F::pw_mul2(u0, v1, u1, v0, P.q, P.q2, P.qinv);
        B0[r] = F::pw_mul(u1, v1, P.q, P.qinv);
    }
    inv_core16<F, LOGS, true>(A0, lds, tid, P, W, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
#pragma unroll
    for (int r = 0; r < 16; r++) w[4 * N + k * NS + tid + r * T] = F::canon_inv(A0[r], P.q);
    __syncthreads();
    inv_core16<F, LOGS, true>(A1, lds, tid, P, W, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
#pragma unroll
    for (int r = 0; r < 16; r++) w[5 * N + k * NS + tid + r * T] = F::canon_inv(A1[r], P.q);
    __syncthreads();
    inv_core16<F, LOGS, true>(B0, lds, tid, P, W, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
#pragma unroll
    for (int r = 0; r < 16; r++) w[6 * N + k * NS + tid + r * T] = F::canon_inv(B0[r], P.q);
}
// Phase 2 of the tensor product with its four forward transforms SIDE BY SIDE (N = 2^13: four groups of 128 threads, one operand block and one LDS image
// each; at N = 2^14 four groups would be 1024 threads at 128 VGPRs, so that size keeps the one-group form, ntt_ct4_block1_kernel).  The transformed blocks
// cross through the images in register order; groups 0, 1, 2 form c0 = a0 b0, c1 = a0 b1 + a1 b0, c2 = a1 b1 and run one inverse each (group 3 walks the
// barriers).  Critical path: two transforms instead of seven.  Same arithmetic per coefficient: a-side canonical, b-side lazy, pw_mul / pw_mul2.
template <class F, int LOGN>
__global__ void __launch_bounds__(4 << (LOGN - 6))
ntt_ct4_block_kernel(typename F::E *ws, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using K = Coop4<F, LOGN>;
    using E = typename F::E;
    constexpr int LOGS = K::LOGS, NS = K::NS, T = K::T, N = 1 << LOGN;
    __shared__ E lds[4][NS];
    const uint32_t g = threadIdx.x / T, tid = threadIdx.x % T, p = blockIdx.x >> 2, k = blockIdx.x & 3;      // g is wave-uniform
    const Limb<F> P = limbs[p % L];
    E *w = ws + (size_t)p * (7 * N);
    Twiddles16<F, LOGS> W;
    W.template load<true>(tid, P, 4 + k);
    E x[16];
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = w[(size_t)g * N + k * NS + tid + r * T];      // g = 0..3: a0, a1, b0, b1
    fwd_core16<F, LOGS>(x, lds[g], tid, P, W);
    if (g < 2) {
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);    // canonical a-side, lazy b-side
    }
    put16<P16Z<LOGS>>(lds[g], tid, x);                    // the slots this thread read last
    __syncthreads();
    if (g == 0) {                                         // c0 = a0 b0
        E v0[16];
        get16<P16Z<LOGS>>(lds[2], tid, v0);
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = F::pw_mul(x[r], v0[r], P.q, P.qinv);
    } else if (g == 1) {                                  // c1 = a0 b1 + a1 b0 (x = a1)
        E u0[16], v0[16], v1[16];
        get16<P16Z<LOGS>>(lds[0], tid, u0);
        get16<P16Z<LOGS>>(lds[2], tid, v0);
        get16<P16Z<LOGS>>(lds[3], tid, v1);
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = F::pw_mul2(u0[r], v1[r], x[r], v0[r], P.q, P.q2, P.qinv);
    } else if (g == 2) {                                  // c2 = a1 b1
        E u1[16], v1[16];
        get16<P16Z<LOGS>>(lds[1], tid, u1);
        get16<P16Z<LOGS>>(lds[3], tid, v1);
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = F::pw_mul(u1[r], v1[r], P.q, P.qinv);
    }
    __syncthreads();                                      // every image has been read before the inverse transforms overwrite them
    inv_core16<F, LOGS, true>(x, lds[g], tid, P, W, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    if (g < 3) {
#pragma unroll
        for (int r = 0; r < 16; r++) w[(size_t)(4 + g) * N + k * NS + tid + r * T] = F::canon_inv(x[r], P.q);
    }
}
template <class F, int LOGN>
__global__ void __launch_bounds__(1 << (LOGN - 6))
ntt_ct4_last_kernel(typename F::E *__restrict__ c0, typename F::E *__restrict__ c1, typename F::E *__restrict__ c2, const typename F::E *ws,
                    const Limb<F> *__restrict__ limbs, uint32_t L) {
    using K = Coop4<F, LOGN>;
    using E = typename F::E;
    constexpr int NS = K::NS, T = K::T, N = 1 << LOGN;
    const uint32_t tid = threadIdx.x, p = blockIdx.x / K::CWG, k = blockIdx.x % K::CWG, which = blockIdx.y;     // grid.y = 3 outputs
    const Limb<F> P = limbs[p % L];
    const E *wr = ws + (size_t)p * (7 * N) + (size_t)(4 + which) * N;
    E *out = (which == 0 ? c0 : which == 1 ? c1 : c2) + (size_t)p * N;
    const typename F::TW i2 = load_global(P.itw + 2), i3 = load_global(P.itw + 3);
#pragma unroll
    for (int m = 0; m < K::CPT; m++) {
        const uint32_t c = k * (K::CPT * T) + tid + m * T;
        E v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = wr[q * NS + c];
        F::inv_bfly(v[0], v[1], i2, P); F::inv_bfly(v[2], v[3], i3, P);
        F::inv_last(v[0], v[2], P.q, P.q2, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
        F::inv_last(v[1], v[3], P.q, P.q2, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
        for (int q = 0; q < 4; q++) out[q * NS + c] = F::canon_inv(F::regroup1(v[q], P.q, P.qinv), P.q);
    }
}

// Tensor product of FHEContext::multiply (src/fhe.cu:199-218) for few ciphertexts, outputs as COMPACT polynomials (the first half of the
// one-call multiply + relinearise): c0 = a0 b0, c1 = a0 b1 + a1 b0, c2 = a1 b1.  Same latency argument as ntt16_multiply_kernel: twice the waves of
// the 32-per-thread kernel on the same polynomial, every twiddle loaded once at the top beside the four operands.
template <class F, int LOGN>
__global__ void __launch_bounds__(Cfg16<LOGN>::T)
ntt16_ct_multiply_kernel(typename F::E *__restrict__ c0, typename F::E *__restrict__ c1, typename F::E *__restrict__ c2,
                         const char *__restrict__ a0, const char *__restrict__ a1, const char *__restrict__ b0, const char *__restrict__ b1,
                         const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = Cfg16<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::N];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)p * (C::N * 32), offc = (size_t)p * C::N;
    E A0[16], A1[16], B0[16], B1[16];
    Twiddles16<F, LOGN> W;
    W.template load<false>(tid, P);
    load16<F, LOGN>(a0 + off, tid, A0);
    load16<F, LOGN>(a1 + off, tid, A1);
    load16<F, LOGN>(b0 + off, tid, B0);
    load16<F, LOGN>(b1 + off, tid, B1);
    fwd_core16<F, LOGN>(A0, lds, tid, P, W);
    __syncthreads();
    fwd_core16<F, LOGN>(A1, lds, tid, P, W);
    __syncthreads();
    fwd_core16<F, LOGN>(B0, lds, tid, P, W);
    __syncthreads();
    fwd_core16<F, LOGN>(B1, lds, tid, P, W);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const E u0 = F::canon_fwd(A0[r], P.q, P.q2, P.qinv), u1 = F::canon_fwd(A1[r], P.q, P.q2, P.qinv);   // canonical a-side
        const E v0 = B0[r], v1 = B1[r];                                                                     // lazy b-side
        A0[r] = F::pw_mul(u0, v0, P.q, P.qinv);
        A1[r] = F::pw_mul2(u0, v1, u1, v0, P.q, P.q2, P.qinv);
        B0[r] = F::pw_mul(u1, v1, P.q, P.qinv);
    }
    // three inverse transforms; each result leaves in pattern A = natural order of a compact polynomial (index tid + r T)
    inv_core16<F, LOGN>(A0, lds, tid, P, W, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 16; r++) c0[offc + tid + (size_t)r * C::T] = F::canon_inv(A0[r], P.q);
    __syncthreads();
    inv_core16<F, LOGN>(A1, lds, tid, P, W, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 16; r++) c1[offc + tid + (size_t)r * C::T] = F::canon_inv(A1[r], P.q);
    __syncthreads();
    inv_core16<F, LOGN>(B0, lds, tid, P, W, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 16; r++) c2[offc + tid + (size_t)r * C::T] = F::canon_inv(B0[r], P.q);
}

// ---- key switching for few ciphertexts: the digit pairs of a limb go to SEPARATE workgroups ------------------------------------------
// ntt_keyswitch2_kernel runs all ceil(L K / 2) paired digit transforms of a (ciphertext, limb) one after the other in one workgroup --
// right for throughput, but with fewer workgroups than CUs the call lasts as long as that one workgroup (55 us at N = 8192, L = 4,
// w = 16: five paired transforms at one wave per SIMD).  Here a first launch gives every (ciphertext, limb, digit pair) its own workgroup:
// paired forward transform and the key products of that pair only, the two NTT-domain partial accumulators (values in [0, 2q)) written in
// register order to a workspace (slot tid + r T); a second launch sums the partials of a (ciphertext, limb) and finishes as the one-launch
// kernel does (paired inverse transform, addends, store).  Same arithmetic in the same order within a pair, and the sums of lazy values
// are reduced exactly as the running accumulators are (pw_add), so the containers are bit-identical.
// (grid.y = 2: the two digit sources of an external product -- c2 / c2b with the row tables kb, ka / kb_b, ka_b -- and 2 NP partials per limb polynomial)
template <class F, int LOGN, bool COMPACT>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, 2)
ntt_keyswitch2_part_kernel(typename F::E *__restrict__ part0, typename F::E *__restrict__ part1, const char *__restrict__ c2, const char *__restrict__ c2b,
                           const typename F::E *__restrict__ kb, const typename F::E *__restrict__ ka,
                           const typename F::E *__restrict__ kb_b, const typename F::E *__restrict__ ka_b,
                           const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[2 * C::LDS_ELEMS];
    const uint32_t LK = L * K, NP = (LK + 1) / 2, comp = blockIdx.y, NPART = NP * gridDim.y;
    const uint32_t tid = threadIdx.x, pr = blockIdx.x % NP, p = blockIdx.x / NP, b = p / L, i = p % L;     // p = ciphertext * L + limb
    if (comp) { c2 = c2b; kb = kb_b; ka = ka_b; }
    const Limb<F> P = limbs[i];
    const TableBuf C2(c2 + (size_t)b * L * (C::N * (COMPACT ? sizeof(E) : 32)));
    E acc0[32], acc1[32], d0[32], d1[32];
#pragma unroll
    for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
    const uint32_t jk = 2 * pr;
    if (jk + 1 < LK) {
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
        fwd_core2<F, LOGN, false>(d0, d1, lds, tid, P);
        mac_keys2<F, true>(acc0, acc1, d0, d1, kb, ka, ((size_t)jk * L + i) * C::N, kb, ka, ((size_t)(jk + 1) * L + i) * C::N, tid, C::T, P);
    } else {                                              // odd number of digit polynomials: the last one alone
        const uint32_t j0 = jk / K, k0 = jk % K;
        load_src_buf<F, LOGN, COMPACT>(C2, j0, tid, d0);
#pragma unroll
        for (int r = 0; r < 32; r++) d0[r] = F::digit(d0[r], k0 * w, w);
        fwd_core<F, LOGN>(d0, lds, tid, P);
        mac_keys<F>(acc0, acc1, d0, kb, ka, ((size_t)jk * L + i) * C::N, tid, C::T, P);
    }
    const size_t slot = ((size_t)p * NPART + comp * NP + pr) * C::N;
    store_A_compact<F, LOGN>(part0 + slot, tid, acc0);    // register order: the combining launch reads slot tid + r T back into register r
    store_A_compact<F, LOGN>(part1 + slot, tid, acc1);
}
template <class F, int LOGN, bool ADD_COMPACT, bool OUT_COMPACT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, 2)
ntt_keyswitch2_comb_kernel(char *c0, char *c1, const typename F::E *__restrict__ part0, const typename F::E *__restrict__ part1,
                           const char *add0, const char *add1, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t NP) {   // NP = partial pairs per limb polynomial
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[2 * C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x, i = p % L;
    const Limb<F> P = limbs[i];
    E acc0[32], acc1[32], t0[32], t1[32];
    load_A_compact<F, LOGN>(part0 + (size_t)p * NP * C::N, tid, acc0);
    load_A_compact<F, LOGN>(part1 + (size_t)p * NP * C::N, tid, acc1);
    for (uint32_t pr = 1; pr < NP; pr++) {                // the order in which the one-launch kernel accumulates its pairs
        load_A_compact<F, LOGN>(part0 + ((size_t)p * NP + pr) * C::N, tid, t0);
        load_A_compact<F, LOGN>(part1 + ((size_t)p * NP + pr) * C::N, tid, t1);
#pragma unroll
        for (int r = 0; r < 32; r++) { acc0[r] = F::pw_add(acc0[r], t0[r], P.q, P.q2); acc1[r] = F::pw_add(acc1[r], t1[r], P.q, P.q2); }
    }
    finish_pair<F, LOGN, ADD_COMPACT, OUT_COMPACT>(acc0, acc1, t0, t1, lds, tid, P, add0, add1, p, c0, c1);
}

// Key switch of few ciphertexts on the 16-per-thread transforms (N <= 2^13, 4-byte residues): the digit-PAIR workgroups above run a paired 32-per-thread
// transform at one wave per SIMD (11.9 us of the call in the trace, and 18.9 us for the paired combining launch).  Here
//   * ntt_keyswitch16_part_kernel: one workgroup per (ciphertext, limb, DIGIT): digit extraction, one 16-per-thread forward transform, the two key products of
//     that digit; the NTT-domain partials go to the workspace in register order (slot tid + r T);
//   * ntt_keyswitch16_comb_kernel: one workgroup per (ciphertext, limb, COMPONENT): sum of the L K partials, one 16-per-thread inverse transform, addend, store.
// Twice the workgroups, twice the waves each, single transforms.  Residues are exact, so the order of the (lazy) additions does not show in the canonical
// result: bit-identical containers.  The key tables are packed for the 32-per-thread kernels (register r of thread t = NTT position 32 t + r, 16-byte chunk c of
// a row at c T32 16 + t 16): this kernel's register r of thread tid is position 16 tid + r, i.e. chunk 4 (tid & 1) + (r >> 2) of thread tid >> 1.
// The external product of a blind-rotation step for few accumulators is the same two launches with TWO digit sources (grid.y = 2: the pre-rotated components
// (X^a - 1) acc_0 and (X^a - 1) acc_1, each with its own RGSW rows) and 2 L K partials per limb polynomial; the combining launch then adds the accumulator itself.
template <class F, int LOGN, bool COMPACT>
__global__ void __launch_bounds__(Cfg16<LOGN>::T)
ntt_keyswitch16_part_kernel(typename F::E *__restrict__ part0, typename F::E *__restrict__ part1, const char *__restrict__ c2, const char *__restrict__ c2b,
                            const typename F::E *__restrict__ kb, const typename F::E *__restrict__ ka,
                            const typename F::E *__restrict__ kb_b, const typename F::E *__restrict__ ka_b,
                            const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = Cfg16<LOGN>;
    using E = typename F::E;
    static_assert(sizeof(E) == 4, "packed key rows of the 4-byte residues");
    typedef E VecE __attribute__((ext_vector_type(4)));
    constexpr uint32_t T32 = NttCfg<LOGN>::T;
    __shared__ E lds[C::N];
    const uint32_t LK = L * K, comp = blockIdx.y, NPART = LK * gridDim.y;
    const uint32_t tid = threadIdx.x, g = blockIdx.x % LK, p = blockIdx.x / LK, b = p / L, i = p % L, j = g / K, k = g % K;
    const Limb<F> P = limbs[i];
    if (comp) { c2 = c2b; kb = kb_b; ka = ka_b; }
    Twiddles16<F, LOGN> W;
    W.load_forward(tid, P);
    E x[16];
    if constexpr (COMPACT) {
        const E *src = reinterpret_cast<const E *>(c2) + ((size_t)b * L + j) * C::N;
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = src[tid + r * C::T];
    } else {
        load16<F, LOGN>(c2 + ((size_t)b * L + j) * (C::N * 32), tid, x);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = F::digit(x[r], k * w, w);
    fwd_core16<F, LOGN>(x, lds, tid, P, W);
    const TableBuf KB(kb), KA(ka);
    const uint32_t voff = ((tid & 1) * 4 * T32 + (tid >> 1)) * 16, row = (uint32_t)((((size_t)g * L + i) * C::N) * sizeof(E));
    const size_t slot = ((size_t)p * NPART + comp * LK + g) * C::N;
    E *o0 = part0 + slot, *o1 = part1 + slot;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const VecE vb = KB.template load16<VecE>(voff, row + c * T32 * 16), va = KA.template load16<VecE>(voff, row + c * T32 * 16);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int r = 4 * c + e;
            o0[tid + r * C::T] = F::pw_mul(vb[e], x[r], P.q, P.qinv);
            o1[tid + r * C::T] = F::pw_mul(va[e], x[r], P.q, P.qinv);
        }
    }
}
template <class F, int LOGN, bool ADD_COMPACT, bool OUT_COMPACT = false>
__global__ void __launch_bounds__(Cfg16<LOGN>::T)
ntt_keyswitch16_comb_kernel(char *c0, char *c1, const typename F::E *__restrict__ part0, const typename F::E *__restrict__ part1,
                            const char *add0, const char *add1, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t LK) {   // LK = partials per limb polynomial
    using C = Cfg16<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::N];
    const uint32_t tid = threadIdx.x, p = blockIdx.x, comp = blockIdx.y;
    const Limb<F> P = limbs[p % L];
    const E *part = (comp ? part1 : part0) + (size_t)p * LK * C::N;
    const char *add = comp ? add1 : add0;
    Twiddles16<F, LOGN> W;
    W.load_inverse(tid, P);
    E acc[16], t[16];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = part[tid + r * C::T];
    for (uint32_t g = 1; g < LK; g++) {
#pragma unroll
        for (int r = 0; r < 16; r++) t[r] = part[(size_t)g * C::N + tid + r * C::T];
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = F::pw_add(acc[r], t[r], P.q, P.q2);
    }
    if constexpr (ADD_COMPACT) {
        const E *a = reinterpret_cast<const E *>(add) + (size_t)p * C::N;
#pragma unroll
        for (int r = 0; r < 16; r++) t[r] = a[tid + r * C::T];
    } else {
        load16<F, LOGN>(add + (size_t)p * (C::N * 32), tid, t);      // in place: the whole addend is in registers before the first store
    }
    inv_core16<F, LOGN>(acc, lds, tid, P, W, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = F::ew_add(F::canon_inv(acc[r], P.q), t[r], P.q);
    if constexpr (OUT_COMPACT) {                          // the accumulator of a blind-rotation loop stays compact between steps
        E *o = reinterpret_cast<E *>(comp ? c1 : c0) + (size_t)p * C::N;
#pragma unroll
        for (int r = 0; r < 16; r++) o[tid + r * C::T] = acc[r];
    } else {
        put16<P16A<LOGN>>(lds, tid, acc);                 // the slots this thread read last
        __syncthreads();
        store16<F, LOGN>((comp ? c1 : c0) + (size_t)p * (C::N * 32), lds, tid);
    }
}

}  // namespace fhe_dev
