// ubench.hip -- VALU instruction throughput on gfx950 (scratch): ops per clock per CU for the instructions the
// modular-arithmetic kernels are made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITER = 4096, CH = 8;

template <int OP> __global__ void __launch_bounds__(256) k(uint64_t *out, uint32_t seed) {
    uint32_t a[CH]; uint64_t b[CH]; double d[CH];
    for (int i = 0; i < CH; i++) { a[i] = seed + threadIdx.x * 7 + i; b[i] = ((uint64_t)a[i] << 20) | 12345; d[i] = (double)a[i] * 1.000001; }
    uint32_t m = seed | 1; uint64_t m64 = ((uint64_t)m << 32) | m; double dm = 1.0000001, dc = 0.999;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) a[i] = a[i] * m;                                   // v_mul_lo_u32
            if (OP == 1) a[i] = __umulhi(a[i], m);                          // v_mul_hi_u32
            if (OP == 2) b[i] = (uint64_t)(uint32_t)b[i] * m + b[i];        // v_mad_u64_u32
            if (OP == 3) a[i] = a[i] + m;                                   // v_add_u32
            if (OP == 4) { uint32_t t = a[i] - m; a[i] = t < a[i] ? t : a[i]; }   // sub + min
            if (OP == 5) d[i] = __builtin_fma(d[i], dm, dc);                // v_fma_f64
            if (OP == 6) d[i] = d[i] + dc;                                  // v_add_f64
            if (OP == 7) d[i] = __builtin_rint(d[i]) * dm;                  // v_rndne_f64 + v_mul_f64
            if (OP == 8) b[i] = b[i] * m64;                                 // 64x64 low
            if (OP == 9) b[i] = __umul64hi(b[i], m64);                      // 64x64 high
            if (OP == 10) a[i] = __umul24(a[i], m);         // v_mul_u32_u24
            if (OP == 11) d[i] = d[i] * dm;                                 // v_mul_f64
        }
    }
    uint64_t acc = 0;
    for (int i = 0; i < CH; i++) acc += a[i] + b[i] + (uint64_t)d[i];
    if (acc == 0x1234567) out[0] = acc;
}
template <int OP> static void run(const char *name, int ops_per) {
    uint64_t *out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int blocks = 256 * 8;   // 8 blocks of 256 per CU = 8 waves/SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double lane_ops = (double)blocks * 256 * ITER * CH * ops_per;
    double per_cu_clk = lane_ops / (ms * 1e-3) / 256 / 2.4e9;
    printf("%-28s %8.3f ms  %7.2f Tops/s  %6.1f lane-ops/clk/CU (at 2.4 GHz)  => %.1f cycles per wave64 instr per SIMD\n", name, ms, lane_ops / ms / 1e9, per_cu_clk, 64.0 * 4 / per_cu_clk);
}
int main() {
    run<3>("v_add_u32", 1); run<4>("v_sub+v_min_u32", 2); run<0>("v_mul_lo_u32", 1); run<1>("v_mul_hi_u32", 1); run<10>("v_mul_u32_u24", 1);
    run<2>("v_mad_u64_u32", 1); run<8>("u64*u64 low (compiler seq)", 1); run<9>("u64*u64 high (compiler seq)", 1);
    run<5>("v_fma_f64", 1); run<6>("v_add_f64", 1); run<11>("v_mul_f64", 1); run<7>("v_rndne_f64+v_mul_f64", 2);
    return 0;
}
