// ubench2.hip -- issue cost of the full-width blocks (wide_asm.inc) on gfx950 at 2 and 8 waves per SIMD (scratch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
namespace fhe_dev {
#include "../gpu-homomorphic-encryption_amd/csrc/wide_asm.inc"
}
using namespace fhe_dev;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITER = 2048;

template <int OP> __global__ void __launch_bounds__(256) k(uint64_t *out, uint32_t seed) {
    uint32_t x[16], y[16]; for (int i = 0; i < 16; i++) { x[i] = seed * (i + 3) + threadIdx.x; y[i] = seed + i * 77 + threadIdx.x * 5; }
    uint64_t lo = seed, lo2 = seed * 3; uint32_t hi = 0, hi2 = 1;
    uint32_t a[8], t[8], q[8]; for (int i = 0; i < 8; i++) { a[i] = x[i]; t[i] = y[i]; q[i] = x[i + 8] | 1; }
    for (int it = 0; it < ITER; it++) {
        if (OP == 0) { macn_9(lo, hi, x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3], x[4], y[4], x[5], y[5], x[6], y[6], x[7], y[7], x[8], y[8]);
                       macn_9(lo2, hi2, x[1], y[0], x[2], y[1], x[3], y[2], x[4], y[3], x[5], y[4], x[6], y[5], x[7], y[6], x[8], y[7], x[9], y[8]); }
        if (OP == 1) { waddsub_8(a, t, q); }
        if (OP == 2) {   // mads only (no carry folding): 18 per iteration
#pragma unroll
            for (int i = 0; i < 9; i++) { lo = (uint64_t)x[i] * y[i] + lo; lo2 = (uint64_t)x[i + 1] * y[i] + lo2; }
        }
        if (OP == 3) {   // 18 plain adds
#pragma unroll
            for (int i = 0; i < 9; i++) { x[i] += y[i]; y[i] += x[i + 1]; }
        }
    }
    uint64_t acc = lo + lo2 + hi + hi2;
    for (int i = 0; i < 8; i++) acc += a[i] + t[i] + x[i] + y[i];
    if (acc == 0x1234567) out[0] = acc;
}
template <int OP> static void run(const char *name, int instrs, int blocks_per_cu) {
    uint64_t *out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    // wave-instructions per SIMD = blocks_per_cu waves per SIMD * ITER * instrs
    double wi = (double)blocks_per_cu * ITER * instrs;
    printf("%-40s %d waves/SIMD  %8.3f ms  => %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.0 GHz)\n", name, blocks_per_cu, ms, ms * 1e6 / wi, ms * 1e6 / wi * 2.0);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<3>("18 v_add_u32", 18, w); run<2>("18 v_mad_u64_u32 (2 chains)", 18, w);
        run<0>("2 x macn_9 (18 mad + 18 addc_e64)", 36, w); run<1>("waddsub_8 (32 carry ops + 16 cndmask)", 48, w);
    }
    return 0;
}
