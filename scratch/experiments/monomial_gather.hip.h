// monomial_gather.hip.h -- EXPERIMENT, not adopted: the (X^a - 1) factor of the external product applied by gathering the rotated limb
// straight from device memory (64 loads per limb) instead of parking the limb in the LDS exchange buffer and reading it back rotated
// (32 loads + 64 LDS operations + 3 barriers).  In ntt_extprod2_kernel it costs 46-58 spilled VGPRs (the 32 gather addresses on top of
// four live 32-entry arrays) and runs at 325 K vs 423 K external products/s (N = 8192, 4 limbs) and 72 K vs 92 K (N = 16384, 6 limbs).
#pragma once

namespace fhe_dev {

// (X^a - 1) * p gathered straight from device memory: the unrotated limb into `tmp`, the rotated one into `x` (a shifted contiguous
// run per wave instruction, hitting the lines the unrotated read just brought in), combined in registers -- no exchange-buffer round
// trip and none of its three barriers.  Needs a free 32-entry array, which the paired kernel has at the start of a pair.
template <class F, int LOGN>
__device__ __forceinline__ void load_monomial_gather(const char *__restrict__ poly, uint32_t tid, uint32_t a, typename F::E qj,
                                                     typename F::E (&x)[32], typename F::E (&tmp)[32]) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __builtin_amdgcn_sched_barrier(0);              // keep the 64 loads out of the previous pair's accumulation (where both digit arrays are live)
    load_A<F, LOGN>(poly, tid, tmp);
    const uint32_t k0 = tid + 2 * C::N - a;
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const uint32_t k = (k0 + (uint32_t)r * C::T) & (C::N - 1);
        x[r] = F::load_low(poly + (uint32_t)(k << 5));               // 32-bit lane offset from a wave-uniform base
    }
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const bool neg = ((k0 + (uint32_t)r * C::T) & C::N) != 0;      // bit log2(n) of (i - a) mod 2n: X^n = -1
        E v = x[r];
        if (neg) v = F::ew_sub((E)0, v, qj);
        x[r] = F::ew_sub(v, tmp[r], qj);
    }
}

}  // namespace fhe_dev
