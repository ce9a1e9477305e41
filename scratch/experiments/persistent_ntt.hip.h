// persistent_ntt.hip.h -- EXPERIMENT, not adopted: forward / inverse transform kernels whose workgroups walk several polynomials and
// issue the loads of the NEXT polynomial before transforming the current one (32 more VGPRs, grid = resident workgroups).
// Parity-green on the MI355X; interleaved A/B against the one-polynomial-per-workgroup kernels (forward+inverse pairs):
// N = 8192, 4 limbs, batch 4096: 5.49-5.53 vs 5.59-5.60 TB/s; N = 16384, 6 limbs, batch 1024: 5.41-5.50 vs 5.56-5.58 TB/s.  With four
// workgroups resident per CU the hardware already overlaps one group's loads with another's butterflies and stores; the explicit
// prefetch only adds registers.  (Same outcome as the persistent fused multiply, persistent_multiply.hip.h.)  Include after ntt_lds.hip.h.
#pragma once

namespace fhe_dev {

// Persistent forms of the two transform kernels: a workgroup walks polynomials p = blockIdx.x, + gridDim.x, ... and issues the loads
// of its NEXT polynomial before it transforms the current one, so the HBM read latency of polynomial k+1 hides under the
// butterflies and the store burst of polynomial k (32 more VGPRs; the launch uses as many workgroups as fit the chip at once).
template <class F, int LOGN, int MINW = 1>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_forward_persistent_kernel(char *__restrict__ data, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t polys) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x;
    uint32_t p = blockIdx.x;
    if (p >= polys) return;
    E x[32], nx[32];
    load_A<F, LOGN>(data + (size_t)p * (C::N * 32), tid, x);
    for (;;) {
        const uint32_t pn = p + gridDim.x;
        const bool more = pn < polys;                                  // uniform over the workgroup
        if (more) load_A<F, LOGN>(data + (size_t)pn * (C::N * 32), tid, nx);
        const Limb<F> P = limbs[p % L];
        fwd_core<F, LOGN, false, true>(x, lds, tid, P);                // PRESYNC: the previous polynomial's store has read the buffer
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
        lds_put<PatZ<LOGN>>(lds, tid, x);
        __syncthreads();
        store_from_lds<F, LOGN>(data + (size_t)p * (C::N * 32), lds, tid);
        if (!more) break;
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = nx[r];
        p = pn;
    }
}
template <class F, int LOGN, int MINW = 1>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_inverse_persistent_kernel(char *__restrict__ data, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t polys) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x;
    uint32_t p = blockIdx.x;
    if (p >= polys) return;
    E x[32], nx[32];
    load_A<F, LOGN>(data + (size_t)p * (C::N * 32), tid, x);
    for (;;) {
        const uint32_t pn = p + gridDim.x;
        const bool more = pn < polys;
        if (more) load_A<F, LOGN>(data + (size_t)pn * (C::N * 32), tid, nx);
        const Limb<F> P = limbs[p % L];
        __syncthreads();                                               // the previous polynomial's store has read the buffer
        lds_put<PatA<LOGN>>(lds, tid, x);
        __syncthreads();
        lds_get<PatZ<LOGN>>(lds, tid, x);
        inv_core<F, LOGN>(x, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);   // its first exchange rewrites this thread's own Z slots
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::canon_inv(x[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, x);
        __syncthreads();
        store_from_lds<F, LOGN>(data + (size_t)p * (C::N * 32), lds, tid);
        if (!more) break;
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = nx[r];
        p = pn;
    }
}

}  // namespace fhe_dev
