// fetch_calib.hip -- known-byte kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE per access shape on gfx950 (scratch, not product).
// Every kernel touches each byte of a BYTES-sized buffer's cache lines exactly once (the buffer is far larger than the 256 MiB Infinity
// Cache), so the compulsory HBM traffic of a launch is BYTES whatever the shape; scripts/fetch_calibration.sh runs this under
// `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` and divides.  Shapes = the load / store forms the product kernels use:
//   rd16_contig        16 B per lane, consecutive lanes consecutive 16-B words (key rows, streaming kernels)          global_load_dwordx4
//   rd4_stride32        4 B per lane out of every 32-B container (load_A of the 4-byte fields)                         global_load_dword
//   rd8_stride32        8 B per lane out of every 32-B container (load_A of the 8-byte fields)                         global_load_dwordx2
//   rd4_contig          4 B per lane, consecutive (compact workspace polynomials, 4-byte fields)                       global_load_dword
//   rd8_contig          8 B per lane, consecutive (compact workspace polynomials, 8-byte fields)                       global_load_dwordx2
//   buf4_stride32 / buf4_contig / buf8_contig / buf16_contig   the same through raw_buffer_load_b32 / b64 / b128 (TableBuf)
//   wr16_contig        16 B per lane consecutive stores (store_from_lds), nontemporal
//   wr4_contig / wr8_contig   compact stores (store_A_compact), plain
// build: hipcc -O3 --offload-arch=gfx950 -o scratch/fetch_calib scratch/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// one workgroup of 256 threads per 256 KiB block (= one N = 8192 polynomial of containers), 32 loads per thread, like load_A
template <int W, int STRIDE, bool NT>      // W = bytes per lane and load, STRIDE = byte distance between consecutive lanes
__device__ __forceinline__ uint32_t block_reads(const char *blk, uint32_t tid) {
    constexpr int PER_PASS = 256 * STRIDE, PASSES = (256 * 1024) / PER_PASS;
    uint32_t acc = 0;
#pragma unroll 32
    for (int r = 0; r < PASSES; r++) {
        const char *p = blk + (size_t)r * PER_PASS + (size_t)tid * STRIDE;
        if constexpr (W == 4) acc ^= NT ? __builtin_nontemporal_load((const uint32_t *)p) : *(const uint32_t *)p;
        else if constexpr (W == 8) { uint64_t v = NT ? __builtin_nontemporal_load((const uint64_t *)p) : *(const uint64_t *)p; acc ^= (uint32_t)v ^ (uint32_t)(v >> 32); }
        else { v4u32 v = NT ? __builtin_nontemporal_load((const v4u32 *)p) : *(const v4u32 *)p; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    return acc;
}
template <int W, int STRIDE>
__device__ __forceinline__ uint32_t block_reads_buf(const char *blk, uint32_t tid) {
    constexpr int PER_PASS = 256 * STRIDE, PASSES = (256 * 1024) / PER_PASS;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(blk), 0, 0xffffffffu, 0x00020000);
    uint32_t acc = 0;
#pragma unroll 32
    for (int r = 0; r < PASSES; r++) {
        if constexpr (W == 4) acc ^= __builtin_bit_cast(uint32_t, __builtin_amdgcn_raw_buffer_load_b32(rs, tid * STRIDE, r * PER_PASS, 0));
        else if constexpr (W == 8) { uint64_t v = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(rs, tid * STRIDE, r * PER_PASS, 0)); acc ^= (uint32_t)v ^ (uint32_t)(v >> 32); }
        else { v4u32 v = __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * STRIDE, r * PER_PASS, 0)); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    return acc;
}
#define RD_KERNEL(name, expr) \
    __global__ void __launch_bounds__(256) name(const char *__restrict__ a, uint32_t *__restrict__ sink) { \
        const char *blk = a + (size_t)blockIdx.x * (256 * 1024); const uint32_t tid = threadIdx.x; \
        uint32_t acc = expr; if (acc == 0x12345678u) sink[0] = acc; }
RD_KERNEL(rd16_contig, (block_reads<16, 16, false>(blk, tid)))
RD_KERNEL(rd16_contig_nt, (block_reads<16, 16, true>(blk, tid)))
RD_KERNEL(rd4_stride32, (block_reads<4, 32, false>(blk, tid)))
RD_KERNEL(rd4_stride32_nt, (block_reads<4, 32, true>(blk, tid)))
RD_KERNEL(rd8_stride32_nt, (block_reads<8, 32, true>(blk, tid)))
RD_KERNEL(rd4_contig, (block_reads<4, 4, false>(blk, tid)))
RD_KERNEL(rd8_contig, (block_reads<8, 8, false>(blk, tid)))
RD_KERNEL(buf4_stride32, (block_reads_buf<4, 32>(blk, tid)))
RD_KERNEL(buf8_stride32, (block_reads_buf<8, 32>(blk, tid)))
RD_KERNEL(buf4_contig, (block_reads_buf<4, 4>(blk, tid)))
RD_KERNEL(buf8_contig, (block_reads_buf<8, 8>(blk, tid)))
RD_KERNEL(buf16_contig, (block_reads_buf<16, 16>(blk, tid)))

template <int W, bool NT>
__device__ __forceinline__ void block_writes(char *blk, uint32_t tid, uint32_t seed) {
    constexpr int PER_PASS = 256 * W, PASSES = (256 * 1024) / PER_PASS;
#pragma unroll 32
    for (int r = 0; r < PASSES; r++) {
        char *p = blk + (size_t)r * PER_PASS + (size_t)tid * W;
        if constexpr (W == 4) { if (NT) __builtin_nontemporal_store(seed + r, (uint32_t *)p); else *(uint32_t *)p = seed + r; }
        else if constexpr (W == 8) { if (NT) __builtin_nontemporal_store((uint64_t)seed + r, (uint64_t *)p); else *(uint64_t *)p = (uint64_t)seed + r; }
        else { v4u32 v = {seed + r, 0, 0, 0}; if (NT) __builtin_nontemporal_store(v, (v4u32 *)p); else *(v4u32 *)p = v; }
    }
}
#define WR_KERNEL(name, W, NT) \
    __global__ void __launch_bounds__(256) name(char *__restrict__ a, uint32_t seed) { block_writes<W, NT>(a + (size_t)blockIdx.x * (256 * 1024), threadIdx.x, seed + blockIdx.x); }
WR_KERNEL(wr16_contig_nt, 16, true)
WR_KERNEL(wr16_contig, 16, false)
WR_KERNEL(wr4_contig, 4, false)
WR_KERNEL(wr8_contig, 8, false)

int main(int argc, char **argv) {
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 2;
    const size_t bytes = gib << 30;
    const unsigned blocks = (unsigned)(bytes / (256 * 1024));
    char *a; uint32_t *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(sink, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("BYTES %zu per launch (every kernel; compulsory HBM traffic of one launch)\n", bytes);
#define RUN(k, ...) do { \
        for (int it = 0; it < 3; it++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, __VA_ARGS__); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it == 2) printf("%-18s %8.3f ms  %7.1f GB/s\n", #k, ms, bytes / (ms * 1e-3) / 1e9); } } while (0)
    RUN(rd16_contig, a, sink); RUN(rd16_contig_nt, a, sink); RUN(rd4_stride32, a, sink); RUN(rd4_stride32_nt, a, sink); RUN(rd8_stride32_nt, a, sink);
    RUN(rd4_contig, a, sink); RUN(rd8_contig, a, sink); RUN(buf4_stride32, a, sink); RUN(buf8_stride32, a, sink); RUN(buf4_contig, a, sink);
    RUN(buf8_contig, a, sink); RUN(buf16_contig, a, sink);
    RUN(wr16_contig_nt, a, 7u); RUN(wr16_contig, a, 7u); RUN(wr4_contig, a, 7u); RUN(wr8_contig, a, 7u);
    CK(hipDeviceSynchronize());
    return 0;
}
