// sbench.hip -- raw HBM streaming ceilings for the access shapes the NTT kernels use (scratch, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ v4u32 ld16(const v4u32 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st16(v4u32 *p, v4u32 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ uint32_t ld4(const uint32_t *p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// 1R:1W copy, 16B per lane, UNROLL independent loads per thread
template <bool NT, int UNROLL> __global__ void __launch_bounds__(256) copy_kernel(v4u32 *__restrict__ r, const v4u32 *__restrict__ a, size_t n16) {
    size_t base = (size_t)blockIdx.x * (256 * UNROLL) + threadIdx.x, stride = (size_t)gridDim.x * 256 * UNROLL;
    for (; base < n16; base += stride) {
        v4u32 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = ld16<NT>(a + base + u * 256);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) st16<NT>(r + base + u * 256, v[u]);
    }
}
// 2R:1W
template <bool NT, int UNROLL> __global__ void __launch_bounds__(256) add_kernel(v4u32 *__restrict__ r, const v4u32 *__restrict__ a, const v4u32 *__restrict__ b, size_t n16) {
    size_t base = (size_t)blockIdx.x * (256 * UNROLL) + threadIdx.x, stride = (size_t)gridDim.x * 256 * UNROLL;
    for (; base < n16; base += stride) {
        v4u32 v[UNROLL], w[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { v[u] = ld16<NT>(a + base + u * 256); w[u] = ld16<NT>(b + base + u * 256); }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) st16<NT>(r + base + u * 256, v[u] + w[u]);
    }
}
// read-only (2 streams), write-only
template <bool NT, int UNROLL> __global__ void __launch_bounds__(256) read_kernel(uint32_t *__restrict__ out, const v4u32 *__restrict__ a, size_t n16) {
    size_t base = (size_t)blockIdx.x * (256 * UNROLL) + threadIdx.x, stride = (size_t)gridDim.x * 256 * UNROLL;
    v4u32 acc = {0, 0, 0, 0};
    for (; base < n16; base += stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += ld16<NT>(a + base + u * 256);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
template <bool NT, int UNROLL> __global__ void __launch_bounds__(256) write_kernel(v4u32 *__restrict__ r, size_t n16) {
    size_t base = (size_t)blockIdx.x * (256 * UNROLL) + threadIdx.x, stride = (size_t)gridDim.x * 256 * UNROLL;
    v4u32 v = {(uint32_t)base, 0, 0, 0};
    for (; base < n16; base += stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) st16<NT>(r + base + u * 256, v);
    }
}
// NTT access shape: workgroup per 256 KiB "polynomial": strided dword loads (T threads x 32), LDS transpose, paired-lane 16B stores
template <bool NT, int NIN, bool DIRECT_STORE> __global__ void __launch_bounds__(256) poly_shape_kernel(char *__restrict__ r, const char *__restrict__ a, const char *__restrict__ b) {
    __shared__ uint32_t lds[8192];
    const uint32_t tid = threadIdx.x; const size_t off = (size_t)blockIdx.x * 8192 * 32;
    uint32_t x[32], y[32];
#pragma unroll
    for (int k = 0; k < 32; k++) { x[k] = ld4<NT>((const uint32_t *)(a + off + tid * 32 + (size_t)k * 8192)); y[k] = NIN > 1 ? ld4<NT>((const uint32_t *)(b + off + tid * 32 + (size_t)k * 8192)) : 0u; }
    if (DIRECT_STORE) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            v4u32 lo = {x[k] + y[k], 0, 0, 0}, hi = {0, 0, 0, 0};
            v4u32 *d = (v4u32 *)(r + off + tid * 32 + (size_t)k * 8192);
            st16<NT>(d, lo); st16<NT>(d + 1, hi);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 32; k++) lds[tid + 256 * k] = x[k] + y[k];
        __syncthreads();
        v4u32 *dst = (v4u32 *)(r + off) + tid;
#pragma unroll 16
        for (int s = 0; s < 64; s++) { uint32_t v = lds[s * 128 + (tid >> 1)]; v4u32 o = {(tid & 1) ? 0u : v, 0u, 0u, 0u}; st16<NT>(dst + s * 256, o); }
    }
}


// generalised: LOGT threads = 2^LOGT, 32 elems per thread, NTL = nt loads, NTS = nt stores, X4 = full 16B-per-lane loads of both halves
template <int LOGT, bool NTL, bool NTS, bool X4, int NIN> __global__ void __launch_bounds__(1 << LOGT) poly2_kernel(char *__restrict__ r, const char *__restrict__ a, const char *__restrict__ b) {
    constexpr int T = 1 << LOGT, N = T * 32;
    __shared__ uint32_t lds[N];
    const uint32_t tid = threadIdx.x; const size_t off = (size_t)blockIdx.x * N * 32;
    if (X4) {
        // lane-consecutive 16B loads: lane l of instr s reads half-container (s*T + l); even lanes carry the value
        const v4u32 *pa = (const v4u32 *)(a + off) + tid, *pb = (const v4u32 *)(b + off) + tid;
        v4u32 x[16];
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int k = 0; k < 16; k++) { x[k] = ld16<NTL>(pa + (size_t)(c * 16 + k) * T); if (NIN > 1) x[k] += ld16<NTL>(pb + (size_t)(c * 16 + k) * T); }
#pragma unroll
            for (int k = 0; k < 16; k++) if (!(tid & 1)) lds[((c * 16 + k) * T + tid) >> 1] = x[k].x + x[k].y;
        }
    } else {
        uint32_t x[32];
#pragma unroll
        for (int k = 0; k < 32; k++) { x[k] = ld4<NTL>((const uint32_t *)(a + off + tid * 32 + (size_t)k * T * 32)); if (NIN > 1) x[k] += ld4<NTL>((const uint32_t *)(b + off + tid * 32 + (size_t)k * T * 32)); }
#pragma unroll
        for (int k = 0; k < 32; k++) lds[tid + T * k] = x[k];
    }
    __syncthreads();
    v4u32 *dst = (v4u32 *)(r + off) + tid;
#pragma unroll 16
    for (int s = 0; s < 64; s++) { uint32_t v = lds[s * (T / 2) + (tid >> 1)]; v4u32 o = {(tid & 1) ? 0u : v, 0u, 0u, 0u}; st16<NTS>(dst + (size_t)s * T, o); }
}

template <class Fn> static float time_it(Fn fn, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) fn();
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) fn();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError());
    return ms / iters;
}
int main(int argc, char **argv) {
    size_t bytes = (size_t)1 << 30; int iters = 20;
    size_t pad = argc > 1 ? atol(argv[1]) : 0;      // byte offset between the buffers' channel phases
    char *base; CK(hipMalloc(&base, 3 * bytes + 4 * pad + (1 << 20)));
    char *a = base, *b = base + bytes + pad, *r = base + 2 * bytes + 2 * pad;
    CK(hipMemset(base, 1, 3 * bytes + 4 * pad));
    uint32_t *flag; CK(hipMalloc(&flag, 4));
    size_t n16 = bytes / 16;
    auto rep = [&](const char *name, float ms, double moved) { printf("%-44s %8.4f ms  %8.1f GB/s\n", name, ms, moved / ms / 1e6); fflush(stdout); };
#define RUN(name, moved, ...) rep(name, time_it([&] { hipLaunchKernelGGL(__VA_ARGS__); }, iters), moved)
    printf("pad = %zu bytes\n", pad);
    RUN("copy  t  u4 g2048", 2.0 * bytes, (copy_kernel<false, 4>), dim3(2048), dim3(256), 0, 0, (v4u32 *)r, (const v4u32 *)a, n16);
    RUN("copy  nt u4 g2048", 2.0 * bytes, (copy_kernel<true, 4>), dim3(2048), dim3(256), 0, 0, (v4u32 *)r, (const v4u32 *)a, n16);
    RUN("copy  nt u8 g2048", 2.0 * bytes, (copy_kernel<true, 8>), dim3(2048), dim3(256), 0, 0, (v4u32 *)r, (const v4u32 *)a, n16);
    RUN("copy  nt u4 g8192", 2.0 * bytes, (copy_kernel<true, 4>), dim3(8192), dim3(256), 0, 0, (v4u32 *)r, (const v4u32 *)a, n16);
    RUN("copy  nt u4 one-shot grid", 2.0 * bytes, (copy_kernel<true, 4>), dim3(n16 / 1024), dim3(256), 0, 0, (v4u32 *)r, (const v4u32 *)a, n16);
    RUN("add   t  u4 g2048", 3.0 * bytes, (add_kernel<false, 4>), dim3(2048), dim3(256), 0, 0, (v4u32 *)r, (const v4u32 *)a, (const v4u32 *)b, n16);
    RUN("add   nt u4 g2048", 3.0 * bytes, (add_kernel<true, 4>), dim3(2048), dim3(256), 0, 0, (v4u32 *)r, (const v4u32 *)a, (const v4u32 *)b, n16);
    RUN("add   nt u8 g2048", 3.0 * bytes, (add_kernel<true, 8>), dim3(2048), dim3(256), 0, 0, (v4u32 *)r, (const v4u32 *)a, (const v4u32 *)b, n16);
    RUN("add   nt u4 one-shot grid", 3.0 * bytes, (add_kernel<true, 4>), dim3(n16 / 1024), dim3(256), 0, 0, (v4u32 *)r, (const v4u32 *)a, (const v4u32 *)b, n16);
    RUN("read  nt u8 g2048", 1.0 * bytes, (read_kernel<true, 8>), dim3(2048), dim3(256), 0, 0, flag, (const v4u32 *)a, n16);
    RUN("read  t  u8 g2048", 1.0 * bytes, (read_kernel<false, 8>), dim3(2048), dim3(256), 0, 0, flag, (const v4u32 *)a, n16);
    RUN("write nt u8 g2048", 1.0 * bytes, (write_kernel<true, 8>), dim3(2048), dim3(256), 0, 0, (v4u32 *)r, n16);
    RUN("write t  u8 g2048", 1.0 * bytes, (write_kernel<false, 8>), dim3(2048), dim3(256), 0, 0, (v4u32 *)r, n16);
    RUN("poly-shape 2in nt lds-store", 3.0 * bytes, (poly_shape_kernel<true, 2, false>), dim3(4096), dim3(256), 0, 0, r, a, b);
    RUN("poly-shape 2in t  lds-store", 3.0 * bytes, (poly_shape_kernel<false, 2, false>), dim3(4096), dim3(256), 0, 0, r, a, b);
    RUN("poly-shape 2in nt direct-store", 3.0 * bytes, (poly_shape_kernel<true, 2, true>), dim3(4096), dim3(256), 0, 0, r, a, b);
    RUN("poly-shape 1in nt lds-store (in-place)", 2.0 * bytes, (poly_shape_kernel<true, 1, false>), dim3(4096), dim3(256), 0, 0, a, a, b);
    RUN("poly-shape 1in nt lds-store (out-of-place)", 2.0 * bytes, (poly_shape_kernel<true, 1, false>), dim3(4096), dim3(256), 0, 0, r, a, b);

    RUN("poly2 T256 ntL ntS dword 2in", 3.0 * bytes, (poly2_kernel<8, true, true, false, 2>), dim3(4096), dim3(256), 0, 0, r, a, b);
    RUN("poly2 T256 ntL tS  dword 2in", 3.0 * bytes, (poly2_kernel<8, true, false, false, 2>), dim3(4096), dim3(256), 0, 0, r, a, b);
    RUN("poly2 T256 tL  ntS dword 2in", 3.0 * bytes, (poly2_kernel<8, false, true, false, 2>), dim3(4096), dim3(256), 0, 0, r, a, b);
    RUN("poly2 T256 ntL ntS x4    2in", 3.0 * bytes, (poly2_kernel<8, true, true, true, 2>), dim3(4096), dim3(256), 0, 0, r, a, b);
    RUN("poly2 T256 ntL tS  x4    2in", 3.0 * bytes, (poly2_kernel<8, true, false, true, 2>), dim3(4096), dim3(256), 0, 0, r, a, b);
    RUN("poly2 T128 ntL ntS dword 2in", 3.0 * bytes, (poly2_kernel<7, true, true, false, 2>), dim3(8192), dim3(128), 0, 0, r, a, b);
    RUN("poly2 T512 ntL ntS dword 2in", 3.0 * bytes, (poly2_kernel<9, true, true, false, 2>), dim3(2048), dim3(512), 0, 0, r, a, b);
    RUN("poly2 T64  ntL ntS dword 2in", 3.0 * bytes, (poly2_kernel<6, true, true, false, 2>), dim3(16384), dim3(64), 0, 0, r, a, b);
    RUN("poly2 T256 ntL ntS dword 1in", 2.0 * bytes, (poly2_kernel<8, true, true, false, 1>), dim3(4096), dim3(256), 0, 0, r, a, b);
    RUN("poly2 T256 ntL tS  dword 1in", 2.0 * bytes, (poly2_kernel<8, true, false, false, 1>), dim3(4096), dim3(256), 0, 0, r, a, b);
    RUN("poly2 T256 ntL ntS x4    1in", 2.0 * bytes, (poly2_kernel<8, true, true, true, 1>), dim3(4096), dim3(256), 0, 0, r, a, b);
    RUN("add   nt u16 chunked 64KiB/WG", 3.0 * bytes, (add_kernel<true, 16>), dim3(n16 / 4096), dim3(256), 0, 0, (v4u32 *)r, (const v4u32 *)a, (const v4u32 *)b, n16);
    return 0;
}
