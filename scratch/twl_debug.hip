// Debug: compare twiddles fetched through the LDS (permuted) path with the natural-order global path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../gpu-homomorphic-encryption_amd/csrc/ntt_lds.hip.h"
using namespace fhe_dev;

template <class Pat, int LOGN, int KLO, int KHI>
__device__ void check(const uint32_t *twl, const uint32_t *g, uint32_t tid, unsigned *bad, unsigned *first) {
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const uint32_t m = 1u << (LOGN - 1 - b);
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            uint32_t nat = g[m + ((Pat::base(tid) | Pat::off(r)) >> (b + 1))];
            uint32_t got = twl[m + Pat::tw_thread(tid) + tw_slot_off<Pat>(r, k)];
            if (nat != got) { if (atomicAdd(bad, 1u) == 0) { first[0] = Pat::BIT0; first[1] = k; first[2] = tid; first[3] = r; first[4] = nat; first[5] = got; } }
        }
    }
}
template <int LOGN>
__global__ void __launch_bounds__(NttCfg<LOGN>::T) dbg(const uint32_t *g, unsigned *bad, unsigned *first) {
    using C = NttCfg<LOGN>;
    __shared__ uint32_t lds[C::LDS_ELEMS];
    __shared__ uint32_t twl[C::N];
    const uint32_t tid = threadIdx.x;
    lds[tid] = tid;
    stage_twiddles<F32, LOGN, true>(twl, g, tid);
    __syncthreads();
    check<PatM<LOGN>, LOGN, 0, 4>(twl, g, tid, bad, first);
    check<PatZ<LOGN>, LOGN, 0, C::REM - 1>(twl, g, tid, bad, first);
    __syncthreads();
    stage_twiddles<F32, LOGN, false>(twl, g, tid);
    __syncthreads();
    check<PatZ<LOGN>, LOGN, 0, 4>(twl, g, tid, bad + 1, first + 8);
    check<PatY<LOGN>, LOGN, 0, 4>(twl, g, tid, bad + 1, first + 8);
}
template <int LOGN> void run() {
    const int N = 1 << LOGN;
    std::vector<uint32_t> h(N); for (int i = 0; i < N; i++) h[i] = 1000000 + i;
    uint32_t *g; unsigned *bad, *first;
    hipMalloc(&g, N * 4); hipMalloc(&bad, 8); hipMalloc(&first, 64);
    hipMemcpy(g, h.data(), N * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 8); hipMemset(first, 0, 64);
    hipLaunchKernelGGL(dbg<LOGN>, dim3(1), dim3(NttCfg<LOGN>::T), 0, 0, g, bad, first);
    unsigned hb[2], hf[16];
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 64, hipMemcpyDeviceToHost);
    printf("LOGN=%d err=%s fwd_bad=%u (BIT0=%u k=%u tid=%u r=%u nat=%u got=%u) inv_bad=%u (BIT0=%u k=%u tid=%u r=%u nat=%u got=%u)\n", LOGN, hipGetErrorString(e),
           hb[0], hf[0], hf[1], hf[2], hf[3], hf[4], hf[5], hb[1], hf[8], hf[9], hf[10], hf[11], hf[12], hf[13]);
}
int main() { run<11>(); run<12>(); run<13>(); run<14>(); return 0; }
