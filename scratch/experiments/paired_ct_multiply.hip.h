// paired_ct_multiply.hip.h -- EXPERIMENT, not adopted: the tensor product with (a0, a1), (b0, b1) and (c0, c1) transformed two at a
// time (fwd_core2 / inv_core2 from the product header).  Parity-green on the MI355X; interleaved A/B against
// ntt_ct_multiply_kernel: N = 8192, 4 limbs, batch 1024: 776 K vs 798 K ct-mul/s (-3 %); N = 16384, 6 limbs, batch 128: 207 K vs 209 K;
// N = 4096: 1.503 M vs 1.510 M.  The tensor product is HBM-bound (0.74 of 8 TB/s), so sharing twiddle loads and barriers buys
// nothing and the second exchange buffer + 256 VGPRs cost a little.  (The same pairing pays for key switching: +4..11 %.)
// Include after ntt_lds.hip.h.
#pragma once

namespace fhe_dev {

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
