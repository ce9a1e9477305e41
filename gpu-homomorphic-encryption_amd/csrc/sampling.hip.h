// sampling.hip.h -- scheme plumbing around the hot path: samplers, modulus switching of a
// single-modulus polynomial, negacyclic folding.  Container-level kernels (32-byte uint256_t), independent of the width
// class of the transforms; none of them is performance-critical (they run once per key / ciphertext, not per product).
//
// Two kinds of entry:
//   * LITERAL restatements of the two sampler kernels the reference defines (src/polynomial.cu:113-143) -- deterministic
//     placeholders there ("Simple LCG for demonstration"), reproduced bit for bit;
//   * the samplers the reference only declares or leaves as placeholders (sample_ternary_kernel include/polynomial.cuh:129,
//     a real discrete Gaussian, a uniform sampler without modulo bias), built on a counter-based generator (SplitMix64
//     finaliser over (seed, index, stream)) so that the device and the CPU oracle produce identical polynomials from
//     integer-only arithmetic and results do not depend on the launch shape.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntt256.hip.h"

namespace fhe_dev {

// ---- counter-based generator ------------------------------------------------------------------------------------------
__device__ __host__ inline uint64_t sm64(uint64_t z) {          // SplitMix64 output function
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// 64 random bits for (seed, element index, draw number)
__device__ __host__ inline uint64_t ctr_rand(uint64_t seed, uint64_t index, uint64_t draw) {
    return sm64(sm64(seed ^ (index * 0xD1342543DE82EF95ull)) + draw);
}
enum : uint64_t { DRAW_TERNARY = 0, DRAW_CDT = 1, DRAW_SIGN = 2, DRAW_UNIFORM = 16 };

__device__ __forceinline__ u256 u256_small(uint64_t v) { u256 r; r.l[0] = v; r.l[1] = r.l[2] = r.l[3] = 0; return r; }
__device__ __forceinline__ bool lt256(const u256 &a, const u256 &b) {   // a < b
#pragma unroll
    for (int i = 3; i >= 0; i--) { if (a.l[i] != b.l[i]) return a.l[i] < b.l[i]; }
    return false;
}

// ---- literal reference samplers ---------------------------------------------------------------------------------------
// sample_uniform_kernel (src/polynomial.cu:130-143): val = ((seed + idx) * 1103515245 + 12345) % modulus.limbs[0], 64-bit wrap-around.
// sample_gaussian_kernel (src/polynomial.cu:113-128): val = (seed + idx) % modulus.limbs[0]   (a placeholder, not Gaussian).
template <int KIND>
__global__ void __launch_bounds__(256)
sample_literal_kernel(u256 *__restrict__ out, uint64_t q0, uint64_t seed, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const uint64_t idx = (uint32_t)g;                              // the reference's index is a uint32_t
        const uint64_t v = KIND == 0 ? ((seed + idx) * 1103515245ull + 12345ull) % q0 : (seed + idx) % q0;
        store_u256(out + g, u256_small(v));
    }
}

// ---- samplers on the RNS layout [batch][L][n]: the same small integer embedded in every limb ----------------------------
// MODE 0: ternary.  P(coefficient != 0) = thr / 2^32, sign uniform  (sample_ternary_kernel(result, modulus, probability, seed, n),
//         include/polynomial.cuh:129-135, declared only; FHEContext calls it with probability 0.5, src/fhe.cu:254).
// MODE 1: discrete Gaussian by inversion of a cumulative table: magnitude m = #{ j < len : r >= cdt[j] } for 64 random bits r,
//         sign from a second draw ("Real implementation needs Box-Muller transform or ziggurat", src/polynomial.cu:122-123;
//         table inversion is the integer-only, constant-table alternative: identical on every device and on the CPU).
template <int MODE>
__global__ void __launch_bounds__(256)
sample_small_kernel(u256 *__restrict__ out, const CrtLimb *__restrict__ limbs, uint32_t L, uint32_t log_n, uint64_t seed, uint64_t thr,
                    const uint64_t *__restrict__ cdt, uint32_t cdt_len, size_t count /* batch * n */) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n = (size_t)1 << log_n;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        uint64_t mag; bool neg;
        if (MODE == 0) {
            const uint64_t r = ctr_rand(seed, g, DRAW_TERNARY);
            mag = (r & 0xffffffffull) < thr ? 1 : 0;
            neg = (r >> 63) != 0;
        } else {
            const uint64_t r = ctr_rand(seed, g, DRAW_CDT);
            uint32_t m = 0;
            for (uint32_t j = 0; j < cdt_len; j++) m += r >= cdt[j] ? 1u : 0u;   // constant work per coefficient
            mag = m;
            neg = (ctr_rand(seed, g, DRAW_SIGN) >> 63) != 0;
        }
        const size_t b = g >> log_n, x = g & (n - 1);
        for (uint32_t l = 0; l < L; l++) {
            u256 v = u256_small(mag);
            if (neg && mag) sub256(v, limbs[l].q, v);                   // -m = q_l - m   (m < q_l is checked on the host)
            store_u256(out + ((b * L + l) << log_n) + x, v);
        }
    }
}

// Uniform residues in [0, q_l) by rejection on bit-length-masked draws (no modulo bias); one lane per container.
// Draw t of element g uses words DRAW_UNIFORM + 4 t .. + 4 t + 3.  After 64 rejections (probability < 2^-64) the top bit is cleared.
__global__ void __launch_bounds__(256)
sample_uniform_rns_kernel(u256 *__restrict__ out, const CrtLimb *__restrict__ limbs, uint32_t L, uint32_t log_n, uint64_t seed, size_t count /* batch * L * n */) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const u256 q = limbs[(uint32_t)((g >> log_n) % L)].q;
        int top = 3; while (top > 0 && q.l[top] == 0) top--;
        const int bits = 64 - __builtin_clzll(q.l[top]);                // q != 0
        const uint64_t mask = bits == 64 ? ~0ull : ((1ull << bits) - 1);
        u256 v;
        for (uint32_t t = 0;; t++) {
#pragma unroll
            for (int i = 0; i < 4; i++) v.l[i] = i <= top ? ctr_rand(seed, g, DRAW_UNIFORM + 4 * t + i) : 0;
            v.l[top] &= mask;
            if (lt256(v, q)) break;
            if (t == 63) { v.l[top] &= mask >> 1; break; }
        }
        store_u256(out + g, v);
    }
}

// ---- poly_mod_switch_kernel (include/polynomial.cuh:96-103, declared; called by FHEContext::decrypt, src/fhe.cu:181-184):
// "modulus switching with rounding": r[i] = round(a[i] * new_q / old_q) mod new_q = floor((a * new_q + floor(old_q / 2)) / old_q) mod new_q
// for a < old_q < 2^255 and new_q < 2^64 (the plaintext modulus t, or a word-sized ciphertext prime).  The 320-bit numerator
// is divided by restoring shift-subtract: one lane per coefficient, not a hot path.
__global__ void __launch_bounds__(256)
poly_mod_switch_kernel(u256 *__restrict__ out, const u256 *__restrict__ in, u256 old_q, uint64_t new_q, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u256 half = old_q;                                                   // floor(old_q / 2)
#pragma unroll
    for (int i = 0; i < 4; i++) half.l[i] = (old_q.l[i] >> 1) | (i < 3 ? old_q.l[i + 1] << 63 : 0);
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const u256 a = load_u256(in + g);
        uint64_t p[5]; u128_t c = 0;                                     // p = a * new_q + half   (< 2^320)
#pragma unroll
        for (int i = 0; i < 4; i++) { c += (u128_t)a.l[i] * new_q + half.l[i]; p[i] = (uint64_t)c; c >>= 64; }
        p[4] = (uint64_t)c;
        u256 rem = u256_small(0); uint64_t quo_lo = 0, quo_hi = 0;       // quotient <= new_q < 2^64 when a < old_q; kept in 128 bits anyway
        for (int bit = 319; bit >= 0; bit--) {
            const uint64_t top = rem.l[3] >> 63;                         // rem < old_q < 2^255, so this is 0; kept for safety
#pragma unroll
            for (int i = 3; i > 0; i--) rem.l[i] = (rem.l[i] << 1) | (rem.l[i - 1] >> 63);
            rem.l[0] = (rem.l[0] << 1) | ((p[bit >> 6] >> (bit & 63)) & 1);
            quo_hi = (quo_hi << 1) | (quo_lo >> 63); quo_lo <<= 1;
            if (top || !lt256(rem, old_q)) { sub256(rem, rem, old_q); quo_lo |= 1; }
        }
        const uint64_t r = (uint64_t)((((u128_t)quo_hi << 64) | quo_lo) % new_q);
        store_u256(out + g, u256_small(r));
    }
}

// ---- negacyclic_reduce_kernel (include/polynomial.cuh:105-110, declared): fold a polynomial of 2n coefficients modulo
// x^n + 1:  data[i] = sub_mod(data[i], data[i + n]) for i < n (literal sub_mod); the upper half is left as it was.
__global__ void __launch_bounds__(256)
negacyclic_reduce_kernel(u256 *__restrict__ data, u256 q, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += stride)
        store_u256(data + g, sub_mod(load_u256(data + g), load_u256(data + g + n), q));
}

}  // namespace fhe_dev
