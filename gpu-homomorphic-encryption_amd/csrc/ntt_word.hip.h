// ntt_word.hip.h -- streaming (one pass over HBM, no transform) kernels on word-sized residues: element-wise ring operations, the
// canonical-input scan, key-table packing, the general-path relinearisation pieces, RNS conversions (row N2) and the monomial
// multiply of the blind-rotation general path.  Compiled into fhe_hip.o; the LDS-resident transform kernels are in ntt_lds.hip.h.
#pragma once
#include "ntt_field.hip.h"

namespace fhe_dev {
// Element-wise kernels over [batch][L][n] containers of word-sized residues: one 16-byte half-container per lane.
// op 0: r = a*b mod q (plain product in the NTT domain); 1: a+b; 2: a-b.
template <class F, int OP>
__global__ void __launch_bounds__(256)
ew_kernel(typename F::V16 *r, const typename F::V16 *a, const typename F::V16 *b,      // no __restrict__: r may be a or b (in-place add / sub / product)
          const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t log_n, size_t halves) {
    using E = typename F::E;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < halves; g += stride) {
        E o = 0;
        if (!(g & 1)) {
            const Limb<F> &P = limbs[(uint32_t)((g >> (log_n + 1)) % L)];
            E x = F::load_low(a + g), y = F::load_low(b + g), q = P.q;
            if (OP == 0) o = F::ew_mul(x, y, P);
            else if (OP == 1) o = F::ew_add(x, y, q);
            else o = F::ew_sub(x, y, q);
        }
        __builtin_nontemporal_store(F::pack(o), r + g);
    }
}

// containers -> compact polynomial (sizeof(E) bytes per coefficient, natural order): the operand that every limb workgroup of a
// key-switch call re-reads once per digit (c2) is compacted first, so the re-reads move S/4 (S/8) instead of S.  One lane per container.
template <class F>
__global__ void __launch_bounds__(256)
compact_kernel(typename F::E *__restrict__ out, const typename F::V16 *__restrict__ in, size_t containers) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < containers; g += stride) out[g] = F::load_low(in + 2 * g);
}

// (X^shift[b] - 1) * in[b][l] on COMPACT polynomials (one lane per coefficient): the blind-rotation loop of the three-array kernels
// applies the monomial ONCE per step here instead of once per digit inside the kernel (there it costs a store / barrier / rotated
// read-back round trip through the exchange buffer and three barriers for each of the 2 * L * K digit polynomials of a workgroup).
template <class F>
__global__ void __launch_bounds__(256)
monomial_compact_kernel(typename F::E *__restrict__ out, const typename F::E *__restrict__ in, const uint32_t *__restrict__ shifts,
                        const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t log_n, size_t count) {
    using E = typename F::E;
    const uint32_t n = 1u << log_n;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const size_t poly = g >> log_n;                                   // b * L + l
        const uint32_t x = (uint32_t)(g & (n - 1));
        const uint32_t a = shifts[poly / L] & (2 * n - 1);
        uint32_t k = (x + 2 * n - a) & (2 * n - 1);
        const bool neg = k >= n; k &= n - 1;
        const E q = limbs[(uint32_t)(poly % L)].q;
        E v = in[(poly << log_n) + k];
        if (neg) v = F::ew_sub((E)0, v, q);
        out[g] = F::ew_sub(v, in[g], q);
    }
}

// Canonical-input scan: flags any container whose value is >= q or whose upper words are not zero.
template <class F>
__global__ void __launch_bounds__(256)
check_kernel(const typename F::V16 *__restrict__ a, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t log_n,
             size_t halves, uint32_t *__restrict__ flag) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < halves; g += stride) {
        typename F::V16 v = a[g];
        if (g & 1) bad |= F::any_nonzero(v);
        else bad |= F::upper_nonzero(v) || F::ge(F::low(v), limbs[(uint32_t)((g >> (log_n + 1)) % L)].q);
    }
    if (bad) atomicOr(flag, 1u);
}

// ---- fused key switching (relinearisation) -----------------------------------------------------------------------------
// Key tables in the kernel's own register order ("packed"): for level jk and limb i, element (chunk c, thread tid, e) holds
// KEY_ntt[i][tid*32 + c*VPL + e] * 2^W mod q_i, VPL = 16 / sizeof(E) values per 16-byte lane load, so a wave instruction
// reads 1 KiB contiguous and pw_mul's 2^-W cancels.  Tables for one engine total 2 * L*K * L * N * sizeof(E) bytes
// (2 MiB at N = 8192, L = 4, K = 2, F32) and stay L2-resident across the batch.
template <class F>
__global__ void __launch_bounds__(256)
pack_keys_kernel(typename F::E *__restrict__ packed, const typename F::V16 *__restrict__ keys_ntt, const Limb<F> *__restrict__ limbs,
                 uint32_t L, uint32_t log_n, uint32_t num_keys) {
    using E = typename F::E;
    constexpr uint32_t VPL = 16 / sizeof(E);
    const uint32_t n = 1u << log_n, T = n >> 5;
    const size_t total = (size_t)num_keys * L * n, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint32_t x = (uint32_t)(g & (n - 1));                  // NTT-domain index = tid*32 + r
        const size_t poly = g >> log_n;                              // jk * L + i
        const Limb<F> &P = limbs[(uint32_t)(poly % L)];
        const uint32_t tid = x >> 5, r = x & 31, c = r / VPL, e = r % VPL;
        E v = F::load_low(keys_ntt + g * 2);
        packed[poly * n + ((size_t)c * T + tid) * VPL + e] = F::to_pw_operand(v, P);
    }
}

// ---- relinearisation building blocks (general path; the fused key-switch kernels are above) --------------------
// Digit polynomials of c2 embedded in every limb: D[jk][b][i][x] = ((c2[b][j][x] >> (k*w)) & (2^w - 1)) mod q_i,
// jk = j*K + k.  One 16-byte half container per lane.
template <class F>
__global__ void __launch_bounds__(256)
digit_embed_kernel(typename F::V16 *__restrict__ D, const typename F::V16 *__restrict__ c2, const Limb<F> *__restrict__ limbs,
                   uint32_t L, uint32_t log_n, uint32_t K, uint32_t w, uint32_t batch) {
    using E = typename F::E;
    const size_t per_poly = (size_t)2 << log_n, per_ct = per_poly * L, per_digit = per_ct * batch, total = per_digit * L * K;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        E o = 0;
        if (!(g & 1)) {
            const uint32_t jk = (uint32_t)(g / per_digit); const size_t rem = g - (size_t)jk * per_digit;
            const uint32_t b = (uint32_t)(rem / per_ct); const size_t r2 = rem - (size_t)b * per_ct;
            const uint32_t i = (uint32_t)(r2 >> (log_n + 1)); const size_t x2 = r2 & (per_poly - 1);
            const uint32_t j = jk / K, k = jk % K;
            const uint64_t v = F::low(c2[((size_t)b * L + j) * per_poly + x2]);
            const uint32_t sh = k * w;
            uint64_t d = sh >= 64 ? 0 : (v >> sh);
            if (w < 64) d &= (1ull << w) - 1;
            o = F::from_u64(d, limbs[i].q);
        }
        __builtin_nontemporal_store(F::pack(o), D + g);
    }
}
// acc0[b][i][x] = sum_jk D[jk][b][i][x] * KB[jk][i][x],  acc1 likewise with KA  (all NTT-domain, canonical)
template <class F>
__global__ void __launch_bounds__(256)
relin_mac_kernel(typename F::V16 *__restrict__ acc0, typename F::V16 *__restrict__ acc1, const typename F::V16 *__restrict__ D,
                 const typename F::V16 *__restrict__ KB, const typename F::V16 *__restrict__ KA, const Limb<F> *__restrict__ limbs,
                 uint32_t L, uint32_t log_n, uint32_t LK, uint32_t batch) {
    using E = typename F::E;
    const size_t per_poly = (size_t)2 << log_n, per_ct = per_poly * L, per_digit = per_ct * batch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < per_digit; g += stride) {
        E s0 = 0, s1 = 0;
        if (!(g & 1)) {
            const size_t kidx = g % per_ct;
            const Limb<F> &P = limbs[(uint32_t)(kidx >> (log_n + 1))];
            for (uint32_t jk = 0; jk < LK; jk++) {
                const E d = F::load_low(D + (size_t)jk * per_digit + g);
                s0 = F::ew_add(s0, F::ew_mul(d, F::load_low(KB + (size_t)jk * per_ct + kidx), P), P.q);
                s1 = F::ew_add(s1, F::ew_mul(d, F::load_low(KA + (size_t)jk * per_ct + kidx), P), P.q);
            }
        }
        __builtin_nontemporal_store(F::pack(s0), acc0 + g);
        __builtin_nontemporal_store(F::pack(s1), acc1 + g);
    }
}

// ---- RNS conversions on word-sized residues (row N2): rounded drop of the last prime, Bajard fast base conversion -----------
// The container-level kernels in ntt256.hip.h do these through 256-bit Montgomery products for every width class; for word-sized
// classes the same arithmetic fits the field type and the kernels are streaming kernels.  Constants are "pw operands"
// (c * 2^W mod q for the integer fields, c for F52) so that canon(pw_mul(constant, x)) is the plain product c * x mod q.
template <class F>
__device__ __forceinline__ typename F::E mul_const(typename F::E cop, typename F::E x, const Limb<F> &P) {
    return F::canon_inv(F::pw_mul(cop, x, P.q, P.qinv), P.q);
}
// out[b][l][x] = (in[b][l][x] - [c]_{q_l}) * q_last^-1 mod q_l, c = the CENTRED residue of in[b][L-1][x] modulo q_last
// (RNSContext::mod_switch_rns, include/rns.cuh:44, declared only).  (x_l - c) * inv = x_l * inv - c * inv with c = +-mag, mag < q_last:
// a residue of ANOTHER prime of the class is a valid second operand of the constant product as it stands (no division to reduce it).
// One lane per (b, x): the last limb is loaded once and every remaining limb produced from it, full 32-byte containers stored as two
// 16-byte halves by the same lane (+22 % at N = 8192 over one lane per output half-container, which re-read the last limb per limb).
template <class F>
__global__ void __launch_bounds__(256)
rescale_word_kernel(typename F::V16 *__restrict__ out, const typename F::V16 *__restrict__ in, const Limb<F> *__restrict__ limbs,
                            const typename F::E *__restrict__ inv_ops, uint32_t L, uint32_t log_n, size_t count /* batch * n */) {
    using E = typename F::E;
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    const E q_last = limbs[L - 1].q, half = (E)(((uint64_t)q_last - 1) >> 1);
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const size_t b = g >> log_n, x = g & (n - 1);
        const E cl = F::load_low(in + (((b * L + (L - 1)) << log_n) + x) * 2);
        const bool neg = cl > half;
        const E mag = neg ? q_last - cl : cl;
        for (uint32_t l = 0; l + 1 < L; l++) {
            const Limb<F> &P = limbs[l];
            const E xl = F::load_low(in + (((b * L + l) << log_n) + x) * 2);
            const E a = mul_const<F>(inv_ops[l], xl, P), m = mul_const<F>(inv_ops[l], mag, P);
            const E o = neg ? F::ew_add(a, m, P.q) : F::ew_sub(a, m, P.q);
            typename F::V16 *dst = out + (((b * (L - 1) + l) << log_n) + x) * 2;
            __builtin_nontemporal_store(F::pack(o), dst);
            __builtin_nontemporal_store(F::pack((E)0), dst + 1);
        }
    }
}

// out[b][j][x] = sum_i ([x_i * (Q/q_i)^-1]_{q_i} mod p_j) * ((Q/q_i) mod p_j) mod p_j   (RNSContext::base_extend, include/rns.cuh:47-48, declared only).
// minv_ops[i] is an operand of source limb i, mat_ops[i * Lp + j] an operand of target limb j.  One half container of the output per lane.
// ALL_LANES: one output container per lane, stored by lane pairs (store_wave_containers): +8..10 % on the 64-bit integer fields, whose
// constant products are the cost; the 4-byte and FP64 fields are bandwidth-bound either way and keep one half container per lane
// (measured 5.0 vs 4.4 and 4.7 vs 4.5 TB/s).
template <class F, bool ALL_LANES>
__global__ void __launch_bounds__(256)
base_convert_word_kernel(typename F::V16 *__restrict__ out, const typename F::V16 *__restrict__ in, const Limb<F> *__restrict__ src, uint32_t L,
                         const Limb<F> *__restrict__ dst, uint32_t Lp, const typename F::E *__restrict__ minv_ops,
                         const typename F::E *__restrict__ mat_ops, uint32_t log_n, size_t work /* containers if ALL_LANES, else half containers */) {
    using E = typename F::E;
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < work; g += stride) {   // n is a multiple of 256: whole waves
        E o = 0;
        if (ALL_LANES || !(g & 1)) {
            const size_t c = ALL_LANES ? g : g >> 1, x = c & (n - 1), pl = c >> log_n, b = pl / Lp;
            const uint32_t j = (uint32_t)(pl % Lp);
            const Limb<F> &D = dst[j];
            for (uint32_t i = 0; i < L; i++) {
                const Limb<F> &S = src[i];
                const E ti = mul_const<F>(minv_ops[i], F::load_low(in + (((b * L + i) << log_n) + x) * 2), S);
                o = F::ew_add(o, mul_const<F>(mat_ops[(size_t)i * Lp + j], ti, D), D.q);   // t_i < q_i: a valid operand modulo p_j as it stands
            }
        }
        if constexpr (ALL_LANES) store_wave_containers<F>(out + 2 * (g - (threadIdx.x & 63)), o);
        else __builtin_nontemporal_store(F::pack(o), out + g);
    }
}

// rns[b][l][x] = values[b][x] mod q_l for ANY 256-bit value (RNS_NTTEngine::to_rns, include/ntt.cuh:114-115, declared only): the value is
// read as 256 / W words of W bits and reduced as sum_k word_k * (2^(W k) mod q_l); pow_ops[l * NW + k] is the pw operand of
// 2^(W k) mod q_l, and a word needs no reduction of its own (integer fields: operand < q, word < 2^W, product < q 2^W; FP64 field:
// 32-bit words, far below its 2^48 operand bound).
template <class F, class WT>     // WT: the word type the value is cut into (uint32_t for F32 and F52, uint64_t for F64)
__global__ void __launch_bounds__(256)
to_rns_word_kernel(typename F::V16 *__restrict__ rns, const typename F::V16 *__restrict__ values, const Limb<F> *__restrict__ limbs,
                   const typename F::E *__restrict__ pow_ops, uint32_t L, uint32_t log_n, size_t out_containers) {
    using E = typename F::E;
    constexpr int NW = 32 / sizeof(WT), HW = NW / 2;                   // words per container / per 16-byte half
    typedef WT VecW __attribute__((ext_vector_type(HW)));
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < out_containers; c += stride) {   // n is a multiple of 256: whole waves
        const size_t x = c & (n - 1), pl = c >> log_n, b = pl / L;
        const uint32_t l = (uint32_t)(pl % L);
        const Limb<F> &P = limbs[l];
        const VecW *v = reinterpret_cast<const VecW *>(values + ((b << log_n) + x) * 2);
        const VecW lo = v[0], hi = v[1];
        const E *ops = pow_ops + (size_t)l * NW;
        E o = 0;
#pragma unroll
        for (int k = 0; k < HW; k++) {
            o = F::ew_add(o, mul_const<F>(ops[k], (E)lo[k], P), P.q);
            o = F::ew_add(o, mul_const<F>(ops[HW + k], (E)hi[k], P), P.q);
        }
        store_wave_containers<F>(rns + 2 * (c - (threadIdx.x & 63)), o);
    }
}

// values[b][x] = CRT of the L residues, in [0, Q)  (RNS_NTTEngine::from_rns, include/ntt.cuh:116-117, declared only), word-sized
// classes: sum_l [x_l * (Q/q_l)^-1]_{q_l} * (Q/q_l) is accumulated as word x 256-bit products in a 320-bit register array (the sum is
// below L * Q) and brought into [0, Q) by at most L - 1 subtractions.  One lane per value; Mi[l] = Q / q_l as a plain integer.
template <class F>
__global__ void __launch_bounds__(256)
from_rns_word_kernel(u256 *__restrict__ values, const typename F::V16 *__restrict__ rns, const Limb<F> *__restrict__ limbs,
                     const typename F::E *__restrict__ minv_ops, const u256 *__restrict__ Mi, u256 Q, uint32_t L, uint32_t log_n, size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = (size_t)1 << log_n;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += stride) {
        const size_t b = g >> log_n, x = g & (n - 1);
        uint64_t acc[5] = {0, 0, 0, 0, 0};
        for (uint32_t l = 0; l < L; l++) {
            const uint64_t t = (uint64_t)mul_const<F>(minv_ops[l], F::load_low(rns + (((b * L + l) << log_n) + x) * 2), limbs[l]);
            const u256 M = Mi[l];
            u128_t c = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { c += (u128_t)t * M.l[i] + acc[i]; acc[i] = (uint64_t)c; c >>= 64; }
            acc[4] += (uint64_t)c;
        }
        for (uint32_t it = 0; it < L; it++) {                          // acc < L * Q
            bool ge = acc[4] != 0;
            if (!ge) {
                ge = true;
#pragma unroll
                for (int i = 3; i >= 0; i--) { if (acc[i] != Q.l[i]) { ge = acc[i] > Q.l[i]; break; } }
            }
            if (!ge) break;
            uint64_t borrow = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { u128_t d = (u128_t)acc[i] - Q.l[i] - borrow; acc[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1; }
            acc[4] -= borrow;
        }
        u256 r; r.l[0] = acc[0]; r.l[1] = acc[1]; r.l[2] = acc[2]; r.l[3] = acc[3];
        store_u256(values + g, r);
    }
}

// ---- blind-rotation building block: out[b][l][x] = ((X^shift[b] - 1) * in[b][l])[x] over Z_q[x]/(x^n + 1), shift in [0, 2n) ----
// (FHEContext::blind_rotate is only declared in the reference, include/fhe.cuh:139.)  One 16-byte half container per lane;
// the rotated read is a shifted contiguous run, so it stays coalesced.
template <class F>
__global__ void __launch_bounds__(256)
monomial_mul_sub_kernel(typename F::V16 *__restrict__ out, const typename F::V16 *__restrict__ in, const uint32_t *__restrict__ shifts,
                        const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t log_n, size_t halves) {
    using E = typename F::E;
    const uint32_t n = 1u << log_n;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < halves; g += stride) {
        E o = 0;
        if (!(g & 1)) {
            const size_t c = g >> 1, poly = c >> log_n;                   // container index, polynomial index b*L + l
            const uint32_t x = (uint32_t)(c & (n - 1));
            const uint32_t a = shifts[poly / L] & (2 * n - 1);
            uint32_t k = (x + 2 * n - a) & (2 * n - 1);
            const bool neg = k >= n; k &= n - 1;
            const E q = limbs[(uint32_t)(poly % L)].q;
            E v = F::load_low(in + ((poly << log_n) + k) * 2);
            if (neg) v = F::ew_sub((E)0, v, q);
            o = F::ew_sub(v, F::load_low(in + g), q);
        }
        __builtin_nontemporal_store(F::pack(o), out + g);
    }
}

}  // namespace fhe_dev
