// kbench.hip -- kernel-variant microbenchmark (scratch; not product).  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../gpu-homomorphic-encryption_amd/csrc/host_math.hpp"
#include "experiments/persistent_multiply.hip.h"
using namespace fhe_dev;
typedef Limb<F32S> LimbS;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static uint32_t shoup32(uint64_t w, uint64_t q) { return (uint32_t)((w << 32) / q); }

static Limb<F32> *build_m(uint32_t n, uint32_t L) {
    std::vector<uint64_t> qs(L); fhe_host::find_ntt_primes(30, n, L, qs.data());
    std::vector<Limb<F32>> limbs(L);
    for (uint32_t l = 0; l < L; l++) {
        fhe_host::NttConstants c; fhe_host::build_constants(n, fhe_host::U256(qs[l]), c);
        uint64_t q = qs[l];
        std::vector<uint32_t> tw(n), itw(n);
        for (uint32_t k = 0; k < n; k++) { tw[k] = (uint32_t)((c.tw[k].w[0] << 32) % q); itw[k] = (uint32_t)((c.itw[k].w[0] << 32) % q); }
        Limb<F32> &P = limbs[l]; memset(&P, 0, sizeof P);
        P.q = q; P.q2 = 2 * q; uint32_t x = 1; for (int i = 0; i < 5; i++) x *= 2 - (uint32_t)q * x; P.qinv = 0u - x;   /* F32::mont_mul wants -q^-1 */
        auto mulq = [q](uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % q); };
        uint64_t two32 = (1ull << 32) % q, ninv = c.n_inv.w[0], w1 = c.itw[1].w[0], nw = mulq(ninv, w1);
        P.r1 = two32; P.r1_s = shoup32(two32, q); P.ninv = ninv; P.ninv_s = shoup32(ninv, q); P.ninvw = nw; P.ninvw_s = shoup32(nw, q);
        uint64_t nr = mulq(ninv, two32), nwr = mulq(nw, two32);
        P.ninv_r = nr; P.ninv_r_s = shoup32(nr, q); P.ninvw_r = nwr; P.ninvw_r_s = shoup32(nwr, q);
        void *d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, tw.data(), n * 4, hipMemcpyHostToDevice)); P.tw = (const uint32_t *)d;
        CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, itw.data(), n * 4, hipMemcpyHostToDevice)); P.itw = (const uint32_t *)d;
    }
    Limb<F32> *d; CK(hipMalloc(&d, L * sizeof(Limb<F32>))); CK(hipMemcpy(d, limbs.data(), L * sizeof(Limb<F32>), hipMemcpyHostToDevice));
    return d;
}

static LimbS *build(uint32_t n, uint32_t L, std::vector<uint64_t> &qs) {
    qs.resize(L); fhe_host::find_ntt_primes(30, n, L, qs.data());
    std::vector<LimbS> limbs(L);
    for (uint32_t l = 0; l < L; l++) {
        fhe_host::NttConstants c; fhe_host::build_constants(n, fhe_host::U256(qs[l]), c);
        uint64_t q = qs[l];
        std::vector<uint2> tw(n), itw(n);
        for (uint32_t k = 0; k < n; k++) { tw[k] = make_uint2(c.tw[k].w[0], shoup32(c.tw[k].w[0], q)); itw[k] = make_uint2(c.itw[k].w[0], shoup32(c.itw[k].w[0], q)); }
        LimbS &P = limbs[l]; memset(&P, 0, sizeof P);
        P.q = q; P.q2 = 2 * q; uint32_t x = 1; for (int i = 0; i < 5; i++) x *= 2 - (uint32_t)q * x; P.qinv = x;
        auto mulq = [q](uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % q); };
        uint64_t two32 = (1ull << 32) % q, ninv = c.n_inv.w[0], w1 = c.itw[1].w[0], nw = mulq(ninv, w1);
        P.r1 = two32; P.r1_s = shoup32(two32, q); P.ninv = ninv; P.ninv_s = shoup32(ninv, q); P.ninvw = nw; P.ninvw_s = shoup32(nw, q);
        uint64_t nr = mulq(ninv, two32), nwr = mulq(nw, two32);
        P.ninv_r = nr; P.ninv_r_s = shoup32(nr, q); P.ninvw_r = nwr; P.ninvw_r_s = shoup32(nwr, q);
        void *d; CK(hipMalloc(&d, n * 8)); CK(hipMemcpy(d, tw.data(), n * 8, hipMemcpyHostToDevice)); P.tw = (const uint2 *)d;
        CK(hipMalloc(&d, n * 8)); CK(hipMemcpy(d, itw.data(), n * 8, hipMemcpyHostToDevice)); P.itw = (const uint2 *)d;
    }
    LimbS *d; CK(hipMalloc(&d, L * sizeof(LimbS))); CK(hipMemcpy(d, limbs.data(), L * sizeof(LimbS), hipMemcpyHostToDevice));
    return d;
}

__global__ void fill_kernel(uint4 *p, size_t halves, const LimbS *limbs, uint32_t L, uint32_t log_n, uint64_t seed) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < halves; g += stride) {
        uint4 o = make_uint4(0, 0, 0, 0);
        if (!(g & 1)) {
            uint64_t z = seed + g * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            o.x = (uint32_t)(z % limbs[(uint32_t)((g >> (log_n + 1)) % L)].q);
        }
        p[g] = o;
    }
}

// streaming ceilings with the same HBM access shapes as the NTT kernels
__global__ void __launch_bounds__(256) stream_dword_kernel(v4u32 *__restrict__ r, const char *__restrict__ a, const char *__restrict__ b, size_t containers) {
    // each workgroup handles 8192 containers like one polynomial: strided dword loads, paired-lane 16B stores via LDS
    __shared__ uint32_t lds[8192];
    const uint32_t tid = threadIdx.x; const size_t p = blockIdx.x;
    const char *pa = a + p * 8192 * 32 + tid * 32, *pb = b + p * 8192 * 32 + tid * 32;
    uint32_t x[32], y[32];
#pragma unroll
    for (int k = 0; k < 32; k++) { x[k] = __builtin_nontemporal_load((const uint32_t *)(pa + (size_t)k * 8192)); y[k] = __builtin_nontemporal_load((const uint32_t *)(pb + (size_t)k * 8192)); }
#pragma unroll
    for (int k = 0; k < 32; k++) lds[tid + 256 * k] = x[k] + y[k];
    __syncthreads();
    v4u32 *dst = r + p * 16384 + tid;
#pragma unroll 16
    for (int s = 0; s < 64; s++) { uint32_t v = lds[s * 128 + (tid >> 1)]; v4u32 o = {(tid & 1) ? 0u : v, 0u, 0u, 0u}; __builtin_nontemporal_store(o, dst + s * 256); }
}
__global__ void __launch_bounds__(256) stream_x4_kernel(v4u32 *__restrict__ r, const v4u32 *__restrict__ a, const v4u32 *__restrict__ b, size_t halves) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < halves; g += stride) {
        v4u32 x = __builtin_nontemporal_load(a + g), y = __builtin_nontemporal_load(b + g);
        __builtin_nontemporal_store(x + y, r + g);
    }
}

template <class Fn> static float time_it(Fn fn, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) fn();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / iters;
}

int main(int argc, char **argv) {
    const uint32_t LOGN = 13, n = 1u << LOGN, L = 4;
    uint32_t B = argc > 1 ? atoi(argv[1]) : 1024;
    int iters = argc > 2 ? atoi(argv[2]) : 20;
    std::vector<uint64_t> qs; LimbS *limbs = build(n, L, qs);
    size_t polys = (size_t)B * L, bytes = polys * n * 32;
    char *a, *b, *r, *r2; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&r, bytes)); CK(hipMalloc(&r2, bytes));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint4 *)a, bytes / 16, limbs, L, LOGN, 1ull);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint4 *)b, bytes / 16, limbs, L, LOGN, 2ull);
    CK(hipDeviceSynchronize());
    auto report = [&](const char *name, float ms, double algo_bytes) { printf("%-34s B=%u  %8.4f ms  %8.1f GB/s (%.3f of 8 TB/s)  %.3f Mpolymul/s\n", name, B, ms, algo_bytes / ms / 1e6, algo_bytes / ms / 1e6 / 8000, B / ms / 1e3); fflush(stdout); };
    double ab = 3.0 * bytes;
    report("stream_x4 (full 32B reads)", time_it([&] { hipLaunchKernelGGL(stream_x4_kernel, dim3(8192), dim3(256), 0, 0, (v4u32 *)r, (const v4u32 *)a, (const v4u32 *)b, bytes / 16); }, iters), ab);
    report("stream_dword (NTT access shape)", time_it([&] { hipLaunchKernelGGL(stream_dword_kernel, dim3(polys), dim3(256), 0, 0, (v4u32 *)r, a, b, polys * n); }, iters), ab);
    report("multiply  lb(256,1)", time_it([&] { hipLaunchKernelGGL((ntt_multiply_kernel<F32S, 13, 1>), dim3(polys), dim3(256), 0, 0, r, a, b, limbs, L, 0u); }, iters), ab);
    CK(hipMemcpy(r2, r, bytes, hipMemcpyDeviceToDevice));
    report("multiply  lb(256,4)", time_it([&] { hipLaunchKernelGGL((ntt_multiply_kernel<F32S, 13, 4>), dim3(polys), dim3(256), 0, 0, r, a, b, limbs, L, 0u); }, iters), ab);
    CK(hipMemcpy(r2, r, bytes, hipMemcpyDeviceToDevice));
    Limb<F32> *limbs_m = build_m(n, L);
    for (int rep = 0; rep < 3; rep++) {
        report("multiply  lb(256,4) shoup tw", time_it([&] { hipLaunchKernelGGL((ntt_multiply_kernel<F32S, 13, 4>), dim3(polys), dim3(256), 0, 0, r, a, b, limbs, L, 0u); }, iters), ab);
        report("multiply  lb(256,4) mont  tw", time_it([&] { hipLaunchKernelGGL((ntt_multiply_kernel<F32, 13, 4>), dim3(polys), dim3(256), 0, 0, r, a, b, limbs_m, L, 0u); }, iters), ab);
    }
    // correctness of the last variant vs the plain kernel
    std::vector<uint32_t> h1(1 << 20), h2(1 << 20);
    CK(hipMemcpy(h1.data(), r, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), r2, h2.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), r + bytes - (4 << 20), h1.size() * 4, hipMemcpyDeviceToHost)); 
    std::vector<uint32_t> h3(1 << 20); CK(hipMemcpy(h3.data(), r2 + bytes - (4 << 20), h3.size() * 4, hipMemcpyDeviceToHost));
    printf("variant == plain (tail 4 MiB): %s\n", h1 == h3 ? "yes" : "NO");
    report("forward", time_it([&] { hipLaunchKernelGGL((ntt_forward_kernel<F32S, 13>), dim3(polys), dim3(256), 0, 0, a, limbs, L); }, iters), 2.0 * bytes);
    report("inverse", time_it([&] { hipLaunchKernelGGL((ntt_inverse_kernel<F32S, 13>), dim3(polys), dim3(256), 0, 0, a, limbs, L); }, iters), 2.0 * bytes);
    return 0;
}
