// paired_ct_multiply.hip.h -- EXPERIMENT, not adopted: the tensor product with (a0, a1), (b0, b1) and (c0, c1) transformed two at a
// time (fwd_core2 from the product header + the inv_core2 below).  Parity-green on the MI355X; interleaved A/B against
// ntt_ct_multiply_kernel: N = 8192, 4 limbs, batch 1024: 776 K vs 798 K ct-mul/s (-3 %); N = 16384, 6 limbs, batch 128: 207 K vs 209 K;
// N = 4096: 1.503 M vs 1.510 M.  The tensor product is HBM-bound (0.74 of 8 TB/s), so sharing twiddle loads and barriers buys
// nothing and the second exchange buffer + 256 VGPRs cost a little.  (The same pairing pays for key switching: +4..11 %.)
// Include after ntt_lds.hip.h.
#pragma once

namespace fhe_dev {

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
template <class F, int LOGN>
__device__ __forceinline__ void inv_core2(typename F::E (&x0)[32], typename F::E (&x1)[32], typename F::E *lds0, typename F::E *lds1, uint32_t tid,
                                          const Limb<F> &P, typename F::E ninv, typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s) {
    using C = NttCfg<LOGN>;
    inv_stages2<F, LOGN, PatZ<LOGN>, 0, 4>(x0, x1, tid, P.itw, P);
    F::regroup(x0, P.q, P.qinv); F::regroup(x1, P.q, P.qinv);
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

// The tensor product with its transforms done two at a time (fwd_core2 / inv_core2): (a0, a1) and (b0, b1) forward, (c0, c1)
// inverse, c2 alone.  Same results as ntt_ct_multiply_kernel; two exchange buffers.
template <class F, int LOGN>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, 2)
ntt_ct_multiply2_kernel(char *__restrict__ c0, char *__restrict__ c1, char *__restrict__ c2,
                        const char *__restrict__ a0, const char *__restrict__ a1,
                        const char *__restrict__ b0, const char *__restrict__ b1,
                        const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[2 * C::LDS_ELEMS];
    E *lds1 = lds + C::LDS_ELEMS;
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)p * (C::N * 32);
    E A0[32], A1[32], B0[32], B1[32];
    load_A<F, LOGN>(a0 + off, tid, A0);
    load_A<F, LOGN>(a1 + off, tid, A1);
    fwd_core2<F, LOGN>(A0, A1, lds, lds1, tid, P);
    load_A<F, LOGN>(b0 + off, tid, B0);
    load_A<F, LOGN>(b1 + off, tid, B1);
    fwd_core2<F, LOGN, true>(B0, B1, lds, lds1, tid, P);
#pragma unroll
    for (int r = 0; r < 32; r++) {
        E u0 = F::canon_fwd(A0[r], P.q, P.q2, P.qinv), u1 = F::canon_fwd(A1[r], P.q, P.q2, P.qinv);   // canonical a-side
        E v0 = B0[r], v1 = B1[r];                                                             // lazy b-side (< 4q)
        E t00 = F::pw_mul(u0, v0, P.q, P.qinv);
        E t01 = F::pw_mul(u0, v1, P.q, P.qinv);
        E t10 = F::pw_mul(u1, v0, P.q, P.qinv);
        E t11 = F::pw_mul(u1, v1, P.q, P.qinv);
        A0[r] = t00;
        A1[r] = F::pw_add(t01, t10, P.q, P.q2);
        B0[r] = t11;
    }
    __syncthreads();                           // the Z-pattern reads of (b0, b1) are done before the inverse pair's first exchange
    inv_core2<F, LOGN>(A0, A1, lds, lds1, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) { A0[r] = F::canon_inv(A0[r], P.q); A1[r] = F::canon_inv(A1[r], P.q); }
    lds_put<PatA<LOGN>>(lds, tid, A0);
    lds_put<PatA<LOGN>>(lds1, tid, A1);
    __syncthreads();
    store_from_lds<F, LOGN>(c0 + off, lds, tid);
    store_from_lds<F, LOGN>(c1 + off, lds1, tid);
    __syncthreads();
    inv_core<F, LOGN>(B0, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) B0[r] = F::canon_inv(B0[r], P.q);
    lds_put<PatA<LOGN>>(lds, tid, B0);
    __syncthreads();
    store_from_lds<F, LOGN>(c2 + off, lds, tid);
}

}  // namespace fhe_dev
