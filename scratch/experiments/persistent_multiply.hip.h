// Experiment (not adopted, see README.md): persistent fused multiply with next-pair register prefetch, and the Shoup-twiddle
// 32-bit field used for the A/B against Montgomery-form twiddles.  Included by scratch/kbench.hip only.
#pragma once
#include "../../gpu-homomorphic-encryption_amd/csrc/ntt_lds.hip.h"

namespace fhe_dev {

// Shoup-form twiddles (w, floor(w*2^32/q)): 3 integer multiplies per butterfly, 8 bytes per twiddle
struct F32S : F32Base<F32S, uint2> {};

// Persistent form of the fused multiply: gridDim.x workgroups stride over the polynomials and keep the NEXT
// pair's HBM loads in flight (64 VGPRs) while the current pair is transformed, so a workgroup never sits idle
// between its store phase and its next load phase, and there is no tail of partially filled dispatch rounds.
template <class F, int LOGN, int MINW>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_multiply_persistent_kernel(char *__restrict__ res, const char *__restrict__ a, const char *__restrict__ b,
                               const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t polys) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x;
    uint32_t p = blockIdx.x;
    if (p >= polys) return;
    E x[32], y[32], xn[32], yn[32];
    load_A<F, LOGN>(a + (size_t)p * (C::N * 32), tid, x);
    load_A<F, LOGN>(b + (size_t)p * (C::N * 32), tid, y);
    for (;;) {
        const uint32_t pn = p + gridDim.x;
        const bool more = pn < polys;               // workgroup-uniform
        if (more) {
            load_A<F, LOGN>(a + (size_t)pn * (C::N * 32), tid, xn);
            load_A<F, LOGN>(b + (size_t)pn * (C::N * 32), tid, yn);
        }
        const Limb<F> P = limbs[p % L];
        fwd_core<F, LOGN>(x, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
        __syncthreads();
        fwd_core<F, LOGN>(y, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::pw_mul(x[r], y[r], P.q, P.qinv);
        inv_core<F, LOGN>(x, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::canon_inv(x[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, x);
        __syncthreads();
        store_from_lds<F, LOGN>(res + (size_t)p * (C::N * 32), lds, tid);
        if (!more) break;
        __syncthreads();                            // store_from_lds reads other threads' slots
#pragma unroll
        for (int r = 0; r < 32; r++) { x[r] = xn[r]; y[r] = yn[r]; }
        p = pn;
    }
}


}  // namespace fhe_dev
