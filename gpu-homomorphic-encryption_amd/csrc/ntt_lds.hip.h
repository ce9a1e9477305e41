// ntt_lds.hip.h -- LDS-resident negacyclic NTT / polymul kernels for word-sized RNS primes on gfx950.
//   F32 : q < 2^30, 32-bit residues, Harvey/Shoup integer butterflies          (FHE_WIDTH_32, N = 2^11 .. 2^15)
//   F52 : q < 2^43, residues held as exact integers in doubles, FMA butterflies (FHE_WIDTH_52, N = 2^11 .. 2^14)
//   F64 : q < 2^62, 64-bit residues, Harvey/Shoup integer butterflies           (FHE_WIDTH_64, N = 2^11 .. 2^14)
//
// Replaces ntt_forward_optimized_kernel / ntt_inverse_optimized_kernel / ntt_pointwise_mul_kernel /
// bit_reverse_kernel / ntt_forward_batch_kernel (kernels/ntt_kernels.cu:7-210) and the
// NTTEngine::multiply launch sequence (src/ntt.cu:49-75) for word-sized moduli.
//
// Why a narrow path is bit-exact: with reduced operands and a prime modulus every reference
// primitive returns the canonical residue (mont(x, w*R) = x*w mod q, add_mod, sub_mod), so any exact
// evaluation of the same butterfly network yields the same 256-bit containers (upper words zero).
//
// Shape (N = 2^LOGN coefficients, one workgroup of T = N/32 threads per (polynomial, limb)):
//   * HBM is touched exactly once per coefficient: a 4/8-byte load out of each 32-byte container
//     (the rest rides along in the same 128-byte lines) and one full 32-byte store.
//   * each thread owns 32 coefficients and runs 5 butterfly stages in registers (radix-32), then the
//     workgroup transposes through LDS; log2(N) = 5 + 5 + REM stages -> 3 register groups.
//   * LDS holds one residue per coefficient, padded by one element per 32 (phys = i + i/32) so that
//     all access patterns below are bank-conflict-free for 64-lane wavefronts.
//   * butterflies are Harvey lazy butterflies (twiddles: Montgomery form for F32, Shoup pairs for F64): values live in
//     [0,4q) (forward) / [0,2q) (inverse) and are made canonical once, before the store.
//   * twiddles of the first register group are wave-uniform (scalar loads); the rest are vector
//     loads from an L2-resident table shared by the whole batch.
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_field.hip.h"

namespace fhe_dev {

template <int LOGN>
struct NttCfg {
    static_assert(LOGN >= 11 && LOGN <= 15, "LDS-resident path covers 2^11 .. 2^15");
    static constexpr int N = 1 << LOGN;
    static constexpr int LOGT = LOGN - 5;
    static constexpr int T = 1 << LOGT;            // threads per workgroup
    static constexpr int REM = LOGN - 10;          // stages in the third register group (1..5)
    static constexpr int LDS_ELEMS = N + N / 32;
};

// ---- index patterns: logical index = base(tid) | off(r), physical LDS slot = pbase(tid) + poff(r) ----
// A : r <-> index bits [LOGT, LOGN)      (coalesced global order: consecutive lanes, consecutive coefficients)
template <int LOGN> struct PatA {
    static constexpr int LOGN_ = LOGN;
    using C = NttCfg<LOGN>;
    static constexpr int BIT0 = C::LOGT;
    static constexpr bool TW_UNIFORM = true;      // every stage on these bits has i >> (b+1) independent of tid
    static constexpr uint32_t TW_STRIDE = 0;
    __device__ static uint32_t tw_thread(uint32_t) { return 0; }
    __device__ static uint32_t base(uint32_t tid) { return tid; }
    __device__ static uint32_t pbase(uint32_t tid) { return tid + (tid >> 5); }
    static constexpr uint32_t off(int r) { return (uint32_t)r << C::LOGT; }
    static constexpr uint32_t poff(int r) { return (uint32_t)r * (C::T + C::T / 32); }
};
// M : r <-> index bits [REM, REM+5)      (forward middle group)
template <int LOGN> struct PatM {
    static constexpr int LOGN_ = LOGN;
    using C = NttCfg<LOGN>;
    static constexpr int BIT0 = C::REM;
    static constexpr bool TW_UNIFORM = false;
    static constexpr uint32_t TW_STRIDE = 32;                                 // distinct twiddle-owning thread groups (T >> REM)
    __device__ static uint32_t tw_thread(uint32_t tid) { return tid >> C::REM; }
    __device__ static uint32_t base(uint32_t tid) { return ((tid >> C::REM) << (C::REM + 5)) | (tid & ((1u << C::REM) - 1)); }
    __device__ static uint32_t pbase(uint32_t tid) { uint32_t b = base(tid); return b + (b >> 5); }
    static constexpr uint32_t off(int r) { return (uint32_t)r << C::REM; }
    static constexpr uint32_t poff(int r) { return off(r) + (off(r) >> 5); }
};
// Z : r <-> index bits [0, 5)            (32 consecutive coefficients per thread)
template <int LOGN> struct PatZ {
    static constexpr int LOGN_ = LOGN;
    static constexpr int BIT0 = 0;
    static constexpr bool TW_UNIFORM = false;
    static constexpr uint32_t TW_STRIDE = NttCfg<LOGN>::T;
    __device__ static uint32_t tw_thread(uint32_t tid) { return tid; }
    __device__ static uint32_t base(uint32_t tid) { return tid << 5; }
    __device__ static uint32_t pbase(uint32_t tid) { return tid * 33; }
    static constexpr uint32_t off(int r) { return (uint32_t)r; }
    static constexpr uint32_t poff(int r) { return (uint32_t)r; }
};
// Y : r <-> index bits [5, 10)           (inverse middle group)
template <int LOGN> struct PatY {
    static constexpr int LOGN_ = LOGN;
    static constexpr int BIT0 = 5;
    static constexpr bool TW_UNIFORM = false;
    static constexpr uint32_t TW_STRIDE = NttCfg<LOGN>::T / 32;
    __device__ static uint32_t tw_thread(uint32_t tid) { return tid >> 5; }
    __device__ static uint32_t base(uint32_t tid) { return ((tid >> 5) << 10) | (tid & 31); }
    __device__ static uint32_t pbase(uint32_t tid) { return (tid >> 5) * 1056 + (tid & 31); }
    static constexpr uint32_t off(int r) { return (uint32_t)r << 5; }
    static constexpr uint32_t poff(int r) { return (uint32_t)r * 33; }
};

template <class Pat, class E>
__device__ __forceinline__ void lds_put(E *lds, uint32_t tid, const E (&x)[32]) {
    E *p = lds + Pat::pbase(tid);
#pragma unroll
    for (int r = 0; r < 32; r++) p[Pat::poff(r)] = x[r];
}
template <class Pat, class E>
__device__ __forceinline__ void lds_get(const E *lds, uint32_t tid, E (&x)[32]) {
    const E *p = lds + Pat::pbase(tid);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = p[Pat::poff(r)];
}

// Two arrays through ONE exchange image of element PAIRS (x0[r], x1[r]) side by side: the paired transforms of the 4-byte residues then
// exchange with ds_write_b64 / ds_read_b64 -- half the LDS instructions of two 4-byte images (a 4-byte LDS read needs ~4 waves per
// SIMD to reach its rate, these kernels run two: MI355X guide, LDS table), same padding rule in units of pairs, same footprint.
// FHE_PAIR_EXCHANGE_B32 (compile-time A/B switch) keeps the round-2 form: two 4-byte images at lds and lds + LDS_ELEMS.
template <class Pat, class E>
__device__ __forceinline__ void lds_put2(E *lds, uint32_t tid, const E (&x0)[32], const E (&x1)[32]) {
#ifdef FHE_PAIR_EXCHANGE_B32
    lds_put<Pat>(lds, tid, x0); lds_put<Pat>(lds + NttCfg<Pat::LOGN_>::LDS_ELEMS, tid, x1);
#else
    typedef E E2 __attribute__((ext_vector_type(2)));
    E2 *p = reinterpret_cast<E2 *>(lds) + Pat::pbase(tid);
#pragma unroll
    for (int r = 0; r < 32; r++) { E2 v = {x0[r], x1[r]}; p[Pat::poff(r)] = v; }
#endif
}
template <class Pat, class E>
__device__ __forceinline__ void lds_get2(const E *lds, uint32_t tid, E (&x0)[32], E (&x1)[32]) {
#ifdef FHE_PAIR_EXCHANGE_B32
    lds_get<Pat>(lds, tid, x0); lds_get<Pat>(lds + NttCfg<Pat::LOGN_>::LDS_ELEMS, tid, x1);
#else
    typedef E E2 __attribute__((ext_vector_type(2)));
    const E2 *p = reinterpret_cast<const E2 *>(lds) + Pat::pbase(tid);
#pragma unroll
    for (int r = 0; r < 32; r++) { const E2 v = p[Pat::poff(r)]; x0[r] = v.x; x1[r] = v.y; }
#endif
}

// ---- twiddle tables ----------------------------------------------------------------------------------------
// The stage on index bit b uses w[m + (i >> (b+1))], m = N >> (b+1), i = coefficient index.  In device memory the tables are
// in that natural order (a wave's strided walk re-uses the lines its first load brought into L1; a lane-consecutive order was
// measured 5-10 % SLOWER there).  Kernels that run many transforms under one modulus (key switching, external product) copy
// the table into LDS instead, PERMUTED within each stage range so that the lanes of a wave read consecutive words (no bank
// conflicts): for a stage that runs as r-bit k of pattern Pat, j = i >> (b+1) = tt << (4-k) | rh with tt = Pat::tw_thread(tid)
// and rh = r >> (k+1); the entry sits at slot m + rh * Pat::TW_STRIDE + tt.  Forward tables use patterns Z / M, inverse
// tables Z / Y; the stages of the uniform pattern A (m <= 16) always take scalar loads from device memory.
template <class Pat>
__device__ __host__ constexpr uint32_t tw_slot_off(int r, int k) { return (uint32_t)(r >> (k + 1)) * Pat::TW_STRIDE; }
// LDS slot of natural index idx in [1, n) of the forward (fwd = true) or inverse table for n = 2^log_n
__device__ __host__ inline uint32_t tw_slot(uint32_t log_n, bool fwd, uint32_t idx) {
    const uint32_t lg = 31 - (uint32_t)__builtin_clz(idx);
    const uint32_t m = 1u << lg, j = idx - m, b = log_n - 1 - lg, rem = log_n - 10, T = 1u << (log_n - 5);
    uint32_t k, stride;
    if (fwd) {
        if (b < rem) { k = b; stride = T; }                       // pattern Z
        else if (b < rem + 5) { k = b - rem; stride = 32; }       // pattern M
        else return idx;                                          // pattern A (never read from LDS)
    } else {
        if (b < 5) { k = b; stride = T; }                         // pattern Z
        else if (b < 10) { k = b - 5; stride = T / 32; }          // pattern Y
        else return idx;
    }
    const uint32_t sh = 4 - k, tt = j >> sh, rh = j & ((1u << sh) - 1);
    return m + rh * stride + tt;
}
// Copy a table into LDS in the permuted order (coalesced reads; the scattered LDS writes happen once per workgroup and table).
template <class F, int LOGN, bool FWD>
__device__ __forceinline__ void stage_twiddles(typename F::TW *twl, const typename F::TW *__restrict__ gtw, uint32_t tid) {
    using C = NttCfg<LOGN>;
#pragma unroll 8
    for (uint32_t idx = tid; idx < (uint32_t)C::N; idx += C::T)
        if (idx >= 1) twl[tw_slot(LOGN, FWD, idx)] = gtw[idx];     // entry 0 is unused; pattern-A entries keep their natural slot
}

// Twiddle tables are reached through pointers stored in Limb<F> (device memory), so the compiler only knows them as GENERIC pointers
// and emits flat_load: that bumps lgkmcnt as well as vmcnt (every wait on an LDS read then also waits for the twiddles in flight), needs
// a 64-bit VGPR address per load and goes through the aperture check.  They are global memory: say so (global_load, SGPR base).
template <class TW>
__device__ __forceinline__ TW load_global(const TW *p) {
#ifdef FHE_FLAT_TWIDDLES      // compile-time A/B switch: generic-pointer loads as in rounds 1-2
    return *p;
#endif
    if constexpr (sizeof(TW) == 16) {
        typedef uint64_t V __attribute__((ext_vector_type(2)));
        const V v = *(const __attribute__((address_space(1))) V *)p;
        TW r; r.x = v.x; r.y = v.y; return r;
    } else {
        return *(const __attribute__((address_space(1))) TW *)p;
    }
}

// ---- register-resident butterfly stages -------------------------------------------------------------
// Forward (Cooley-Tukey, merged psi twiddles): stage on index bit b uses twiddle tw[m + (i >> (b+1))], m = N >> (b+1).
// Processes r-bits KHI down to KLO of pattern Pat.  Values stay in [0, 4q).
// TWL: `tw` is an LDS copy in the permuted order (non-uniform patterns only).
// SUB: the 2^LOGN coefficients are block number (pre - 2^k) of a larger transform of 2^(LOGN + k) coefficients whose top k stages
// ran elsewhere (word_pass_kernel); the stage on local bit b then uses the big table at (pre << (LOGN-1-b)) + (i >> (b+1)), which for
// pre = 1 is the whole-transform formula.
template <class F, int LOGN, class Pat, int KHI, int KLO, bool TWL = false, bool SUB = false, bool GTW = true>
__device__ __forceinline__ void fwd_stages(typename F::E (&x)[32], uint32_t tid, const typename F::TW *__restrict__ tw, const Limb<F> &P, uint32_t pre = 1) {
    static_assert(!(TWL && Pat::TW_UNIFORM), "uniform stages read device memory");
    static_assert(!(TWL && SUB), "sub-transforms read their twiddles from device memory");
    const uint32_t base = TWL ? Pat::tw_thread(tid) : Pat::TW_UNIFORM ? 0u : Pat::base(tid);   // uniform -> scalar twiddle loads
#pragma unroll
    for (int k = KHI; k >= KLO; k--) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + (((SUB ? pre : 1u) << (LOGN - 1 - b)) + (TWL ? base : (base >> (b + 1))));
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            typename F::TW w;
            if constexpr (TWL) w = p[tw_slot_off<Pat>(r, k)];
            else if constexpr (GTW) w = load_global(p + (Pat::off(r) >> (b + 1)));
            else w = p[Pat::off(r) >> (b + 1)];
            F::fwd_bfly(x[r], x[r | (1 << k)], w, P);
        }
    }
}
// Inverse (Gentleman-Sande): stage on index bit b uses itw[m + (i >> (b+1))].  Processes r-bits KLO up to KHI.
// Values stay in [0, 2q).
template <class F, int LOGN, class Pat, int KLO, int KHI, bool TWL = false, bool SUB = false, bool GTW = true>
__device__ __forceinline__ void inv_stages(typename F::E (&x)[32], uint32_t tid, const typename F::TW *__restrict__ itw, const Limb<F> &P, uint32_t pre = 1) {
    static_assert(!(TWL && Pat::TW_UNIFORM), "uniform stages read device memory");
    static_assert(!(TWL && SUB), "sub-transforms read their twiddles from device memory");
    const uint32_t base = TWL ? Pat::tw_thread(tid) : Pat::TW_UNIFORM ? 0u : Pat::base(tid);
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = itw + (((SUB ? pre : 1u) << (LOGN - 1 - b)) + (TWL ? base : (base >> (b + 1))));
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            typename F::TW w;
            if constexpr (TWL) w = p[tw_slot_off<Pat>(r, k)];
            else if constexpr (GTW) w = load_global(p + (Pat::off(r) >> (b + 1)));
            else w = p[Pat::off(r) >> (b + 1)];
            F::inv_bfly(x[r], x[r | (1 << k)], w, P);
        }
    }
}
// Last inverse stage (index bit LOGN-1, single twiddle itw[1]) with the n^-1 scaling folded in.
template <class F>
__device__ __forceinline__ void inv_last_stage(typename F::E (&x)[32], typename F::E q, typename F::E q2, typename F::E ninv,
                                               typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s) {
#pragma unroll
    for (int r = 0; r < 16; r++) F::inv_last(x[r], x[r | 16], q, q2, ninv, ninv_s, ninvw, ninvw_s);
}

// ---- global memory <-> registers ----------------------------------------------------------------------
// Coalesced gather of the low word(s) of each container, pattern A (lane stride = one 32-byte container).
template <class F, int LOGN>
__device__ __forceinline__ void load_A(const char *__restrict__ poly, uint32_t tid, typename F::E (&x)[32]) {
    const char *p = poly + (size_t)tid * 32;
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::load_low(p + (size_t)r * (NttCfg<LOGN>::T * 32));
}
// Compact polynomials (internal workspace only, never at the ABI): sizeof(E) bytes per coefficient, natural order.  The fused
// FHEContext::multiply (tensor product + relinearisation) hands c2 from the tensor-product kernel to the key-switch kernel in this
// form: c2 is written once as S/8 (S/4 for 8-byte residues) and each of the L limb workgroups that re-read it moves S/8 instead of S.
template <class F, int LOGN>
__device__ __forceinline__ void load_A_compact(const typename F::E *__restrict__ poly, uint32_t tid, typename F::E (&x)[32]) {
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = poly[tid + r * NttCfg<LOGN>::T];
}
template <class F, int LOGN>
__device__ __forceinline__ void store_A_compact(typename F::E *__restrict__ poly, uint32_t tid, const typename F::E (&x)[32]) {
#pragma unroll
    for (int r = 0; r < 32; r++) poly[tid + r * NttCfg<LOGN>::T] = x[r];
}
// source of a key-switch digit polynomial: 32-byte containers (the ABI's c2) or the compact workspace
template <class F, int LOGN, bool COMPACT>
__device__ __forceinline__ void load_src(const char *__restrict__ base, size_t poly_index, uint32_t tid, typename F::E (&x)[32]) {
    if constexpr (COMPACT) load_A_compact<F, LOGN>(reinterpret_cast<const typename F::E *>(base) + poly_index * NttCfg<LOGN>::N, tid, x);
    else load_A<F, LOGN>(base + poly_index * (NttCfg<LOGN>::N * 32), tid, x);
}

// load_src through a descriptor based at the first limb polynomial of a ciphertext component (`poly` = limb index within it): the 32
// loads of a thread share one VGPR offset, the per-load strides are scalar
template <class F, int LOGN, bool COMPACT>
__device__ __forceinline__ void load_src_buf(const TableBuf &B, uint32_t poly, uint32_t tid, typename F::E (&x)[32]) {
    using C = NttCfg<LOGN>;
    constexpr uint32_t STRIDE = COMPACT ? sizeof(typename F::E) : 32;
    const uint32_t voff = tid * STRIDE, base = poly * (C::N * STRIDE);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = B.template load_residue<F, COMPACT>(voff, base + r * (C::T * STRIDE));
}
// one polynomial by its own base pointer (uniform per workgroup)
template <class F, int LOGN, bool COMPACT>
__device__ __forceinline__ void load_poly_buf(const void *poly, uint32_t tid, typename F::E (&x)[32]) {
    load_src_buf<F, LOGN, COMPACT>(TableBuf(poly), 0, tid, x);
}

// Store the whole polynomial from LDS as full containers: consecutive lanes write consecutive 16-byte
// halves (even lane: {value, 0...}; odd lane: zeros), i.e. 1 KiB contiguous per wave instruction.
template <class F, int LOGN>
__device__ __forceinline__ void store_from_lds(char *__restrict__ poly, const typename F::E *lds, uint32_t tid) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    const uint32_t half = tid & 1, c0 = tid >> 1;
    const E *p = lds + c0 + (c0 >> 5);
    typename F::V16 *dst = reinterpret_cast<typename F::V16 *>(poly) + tid;
#pragma unroll 16
    for (int s = 0; s < 64; s++) {
        E v = p[s * (C::T / 2 + C::T / 64)];
        typename F::V16 o = F::pack(half ? (E)0 : v);
        __builtin_nontemporal_store(o, dst + (size_t)s * C::T);
    }
}

// The same store for callers that hold a live register array across it: a ROLLED loop of eight containers per trip (the fully
// scheduled form above keeps up to 64 addresses and values in flight and spills around a live array).
template <class F, int LOGN>
__device__ __forceinline__ void store_from_lds_rolled(char *__restrict__ poly, const typename F::E *lds, uint32_t tid) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    const uint32_t half = tid & 1, c0 = tid >> 1;
    const E *p = lds + c0 + (c0 >> 5);
    typename F::V16 *dst = reinterpret_cast<typename F::V16 *>(poly) + tid;
#pragma unroll 1
    for (int s0 = 0; s0 < 64; s0 += 8) {
        E v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = p[(s0 + j) * (C::T / 2 + C::T / 64)];
#pragma unroll
        for (int j = 0; j < 8; j++) __builtin_nontemporal_store(F::pack(half ? (E)0 : v[j]), dst + (size_t)(s0 + j) * C::T);
    }
}

// ---- whole-transform building blocks (data in registers, lds = workgroup scratch) ----------------------
// natural-order coefficients in pattern A  ->  NTT values in pattern Z, in [0, 4q)
// TWL: the non-uniform stages take their twiddles from `twl`, an LDS copy made by stage_twiddles<F, LOGN, true>.
// PRESYNC: the barrier that protects the exchange buffer from the PREVIOUS transform's last reads sits here, after the
// register-only first group, instead of at the end of the caller's loop body: the latest legal place, where it coincides with
// the transform's own first barrier (a wave that is ahead keeps computing instead of waiting early).
template <class F, int LOGN, bool TWL = false, bool PRESYNC = false, bool SUB = false, bool GTW = true>
__device__ __forceinline__ void fwd_core(typename F::E (&x)[32], typename F::E *lds, uint32_t tid, const Limb<F> &P,
                                         const typename F::TW *twl = nullptr, uint32_t pre = 1) {
    using C = NttCfg<LOGN>;
    const typename F::TW *t2 = TWL ? twl : P.tw;
    fwd_stages<F, LOGN, PatA<LOGN>, 4, 0, false, SUB, GTW>(x, tid, P.tw, P, pre);
    if constexpr (PRESYNC) __syncthreads();
    lds_put<PatA<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatM<LOGN>>(lds, tid, x);
    fwd_stages<F, LOGN, PatM<LOGN>, 4, 0, TWL, SUB, GTW>(x, tid, t2, P, pre);
    lds_put<PatM<LOGN>>(lds, tid, x);          // same slots this thread just read: no barrier needed before
    __syncthreads();
    lds_get<PatZ<LOGN>>(lds, tid, x);
    fwd_stages<F, LOGN, PatZ<LOGN>, C::REM - 1, 0, TWL, SUB, GTW>(x, tid, t2, P, pre);
}
// ---- two forward transforms under ONE modulus at once ------------------------------------------------------------------
// The key-switch and external-product kernels transform many digit polynomials under the same modulus.  Doing two of them
// in lock step shares every twiddle load (one load feeds two butterflies), every barrier and every LDS wait between the two,
// and doubles the independent work in flight per wave -- for the same register budget as holding the undecomposed limb
// beside one digit polynomial.  x0 / x1 travel through ONE exchange image of element pairs (lds_put2 / lds_get2).
template <class F, int LOGN, class Pat, int KHI, int KLO>
__device__ __forceinline__ void fwd_stages2(typename F::E (&x0)[32], typename F::E (&x1)[32], uint32_t tid, const typename F::TW *__restrict__ tw,
                                            const Limb<F> &P) {
    const uint32_t base = Pat::TW_UNIFORM ? 0u : Pat::base(tid);
#pragma unroll
    for (int k = KHI; k >= KLO; k--) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + ((1u << (LOGN - 1 - b)) + (base >> (b + 1)));
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW w = load_global(p + (Pat::off(r) >> (b + 1)));
            F::fwd_bfly(x0[r], x0[r | (1 << k)], w, P);
            F::fwd_bfly(x1[r], x1[r | (1 << k)], w, P);
        }
    }
}
// ---- twiddles issued one exchange ahead (paired transforms) --------------------------------------------------------------------
// A wave of these kernels waits on vector-memory loads 0.44 of its lifetime (rocprofv3 SQ_WAIT_INST_ANY, round 3) although it issues only
// ~440 of them: at two waves per SIMD every load that is consumed right after it is issued exposes its whole L2 latency, and the
// per-lane twiddles of a register group cannot be issued before the barrier in front of it by the compiler (a barrier fences memory).
// They depend on the thread index only, so the 31 (or 2^g - 1) twiddles of the NEXT register group are loaded into registers BEFORE
// the exchange that precedes it and arrive while the exchange is in flight.  Slot of stage k (r-bit k), twiddle j = r >> (k+1):
// (16 >> k) - 1 + j  (k = 4: slot 0; k = 3: 1, 2; k = 2: 3..6; k = 1: 7..14; k = 0: 15..30).
template <class F, int LOGN, class Pat, int KHI, int KLO>
__device__ __forceinline__ void preload_twiddles(typename F::TW (&w)[31], uint32_t tid, const typename F::TW *__restrict__ tw) {
    static_assert(!Pat::TW_UNIFORM, "uniform stages take scalar loads");
    const uint32_t base = Pat::base(tid);
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = tw + ((1u << (LOGN - 1 - b)) + (base >> (b + 1)));
#pragma unroll
        for (int j = 0; j < (16 >> k); j++) w[(16 >> k) - 1 + j] = load_global(p + j);
    }
}
template <class F, int KHI, int KLO>
__device__ __forceinline__ void fwd_stages2_pre(typename F::E (&x0)[32], typename F::E (&x1)[32], const typename F::TW (&w)[31], const Limb<F> &P) {
#pragma unroll
    for (int k = KHI; k >= KLO; k--) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW t = w[(16 >> k) - 1 + (r >> (k + 1))];
            F::fwd_bfly(x0[r], x0[r | (1 << k)], t, P);
            F::fwd_bfly(x1[r], x1[r | (1 << k)], t, P);
        }
    }
}
template <class F, int KLO, int KHI>
__device__ __forceinline__ void inv_stages2_pre(typename F::E (&x0)[32], typename F::E (&x1)[32], const typename F::TW (&w)[31], const Limb<F> &P) {
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW t = w[(16 >> k) - 1 + (r >> (k + 1))];
            F::inv_bfly(x0[r], x0[r | (1 << k)], t, P);
            F::inv_bfly(x1[r], x1[r | (1 << k)], t, P);
        }
    }
}
template <class F, int KHI, int KLO>
__device__ __forceinline__ void fwd_stages_pre(typename F::E (&x)[32], const typename F::TW (&w)[31], const Limb<F> &P) {
#pragma unroll
    for (int k = KHI; k >= KLO; k--) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            F::fwd_bfly(x[r], x[r | (1 << k)], w[(16 >> k) - 1 + (r >> (k + 1))], P);
        }
    }
}
template <class F, int KLO, int KHI>
__device__ __forceinline__ void inv_stages_pre(typename F::E (&x)[32], const typename F::TW (&w)[31], const Limb<F> &P) {
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            F::inv_bfly(x[r], x[r | (1 << k)], w[(16 >> k) - 1 + (r >> (k + 1))], P);
        }
    }
}
// fwd_core / inv_core with the twiddles of each register group issued one exchange ahead (see preload_twiddles): for kernels of the 4-byte
// field that have ~31 VGPRs to spare (the tensor product: two waves per SIMD either way)
template <class F, int LOGN>
__device__ __forceinline__ void fwd_core_pre(typename F::E (&x)[32], typename F::E *lds, uint32_t tid, const Limb<F> &P) {
    using C = NttCfg<LOGN>;
    typename F::TW w[31];
    preload_twiddles<F, LOGN, PatM<LOGN>, 4, 0>(w, tid, P.tw);
    fwd_stages<F, LOGN, PatA<LOGN>, 4, 0>(x, tid, P.tw, P);
    lds_put<PatA<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatM<LOGN>>(lds, tid, x);
    fwd_stages_pre<F, 4, 0>(x, w, P);
    preload_twiddles<F, LOGN, PatZ<LOGN>, C::REM - 1, 0>(w, tid, P.tw);
    lds_put<PatM<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatZ<LOGN>>(lds, tid, x);
    fwd_stages_pre<F, C::REM - 1, 0>(x, w, P);
}
template <class F, int LOGN>
__device__ __forceinline__ void inv_core_pre(typename F::E (&x)[32], typename F::E *lds, uint32_t tid, const Limb<F> &P,
                                             typename F::E ninv, typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s) {
    using C = NttCfg<LOGN>;
    typename F::TW w[31];
    preload_twiddles<F, LOGN, PatZ<LOGN>, 4, 0>(w, tid, P.itw);
    inv_stages_pre<F, 0, 4>(x, w, P);
    F::regroup(x, P.q, P.qinv);
    preload_twiddles<F, LOGN, PatY<LOGN>, 4, 0>(w, tid, P.itw);
    lds_put<PatZ<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatY<LOGN>>(lds, tid, x);
    inv_stages_pre<F, 0, 4>(x, w, P);
    F::regroup(x, P.q, P.qinv);
    lds_put<PatY<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatA<LOGN>>(lds, tid, x);
    inv_stages<F, LOGN, PatA<LOGN>, 5 - C::REM, 3>(x, tid, P.itw, P);
    inv_last_stage<F>(x, P.q, P.q2, ninv, ninv_s, ninvw, ninvw_s);
}
// PRESYNC as in fwd_core: the barrier that ends the previous transforms' use of the exchange buffers sits after the first group.
template <class F, int LOGN, bool PRESYNC = false>
__device__ __forceinline__ void fwd_core2(typename F::E (&x0)[32], typename F::E (&x1)[32], typename F::E *lds, uint32_t tid,
                                          const Limb<F> &P) {
    using C = NttCfg<LOGN>;
#ifdef FHE_NO_TWIDDLE_PRELOAD     // compile-time A/B switch: the round-2 form (twiddles loaded where they are used)
    fwd_stages2<F, LOGN, PatA<LOGN>, 4, 0>(x0, x1, tid, P.tw, P);
    if constexpr (PRESYNC) __syncthreads();
    lds_put2<PatA<LOGN>>(lds, tid, x0, x1);
    __syncthreads();
    lds_get2<PatM<LOGN>>(lds, tid, x0, x1);
    fwd_stages2<F, LOGN, PatM<LOGN>, 4, 0>(x0, x1, tid, P.tw, P);
    lds_put2<PatM<LOGN>>(lds, tid, x0, x1);       // same slots this thread just read: no barrier needed before
    __syncthreads();
    lds_get2<PatZ<LOGN>>(lds, tid, x0, x1);
    fwd_stages2<F, LOGN, PatZ<LOGN>, C::REM - 1, 0>(x0, x1, tid, P.tw, P);
#else
    typename F::TW w[31];
    preload_twiddles<F, LOGN, PatM<LOGN>, 4, 0>(w, tid, P.tw);          // in flight under the first register group and the first exchange
    fwd_stages2<F, LOGN, PatA<LOGN>, 4, 0>(x0, x1, tid, P.tw, P);
    if constexpr (PRESYNC) __syncthreads();
    lds_put2<PatA<LOGN>>(lds, tid, x0, x1);
    __syncthreads();
    lds_get2<PatM<LOGN>>(lds, tid, x0, x1);
    fwd_stages2_pre<F, 4, 0>(x0, x1, w, P);
    preload_twiddles<F, LOGN, PatZ<LOGN>, C::REM - 1, 0>(w, tid, P.tw);  // in flight under the second exchange
    lds_put2<PatM<LOGN>>(lds, tid, x0, x1);       // same slots this thread just read: no barrier needed before
    __syncthreads();
    lds_get2<PatZ<LOGN>>(lds, tid, x0, x1);
    fwd_stages2_pre<F, C::REM - 1, 0>(x0, x1, w, P);
#endif
}

// NTT values in pattern Z, in [0, 2q)  ->  coefficients in pattern A, in [0, 2q), scaled by the (ninv..) constants
template <class F, int LOGN, bool TWL = false, bool PRESYNC = false, bool SUB = false, bool GTW = true>
__device__ __forceinline__ void inv_core(typename F::E (&x)[32], typename F::E *lds, uint32_t tid, const Limb<F> &P,
                                         typename F::E ninv, typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s,
                                         const typename F::TW *twl = nullptr, uint32_t pre = 1) {
    using C = NttCfg<LOGN>;
    const typename F::TW *t2 = TWL ? twl : P.itw;
    inv_stages<F, LOGN, PatZ<LOGN>, 0, 4, TWL, SUB, GTW>(x, tid, t2, P, pre);
    F::regroup(x, P.q, P.qinv);
    if constexpr (PRESYNC) __syncthreads();
    lds_put<PatZ<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatY<LOGN>>(lds, tid, x);
    inv_stages<F, LOGN, PatY<LOGN>, 0, 4, TWL, SUB, GTW>(x, tid, t2, P, pre);
    F::regroup(x, P.q, P.qinv);
    lds_put<PatY<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatA<LOGN>>(lds, tid, x);
    if constexpr (SUB) {   // a block of a larger transform: bit LOGN-1 is an ordinary stage, the scaling belongs to the last pass
        inv_stages<F, LOGN, PatA<LOGN>, 5 - C::REM, 4, false, true, GTW>(x, tid, P.itw, P, pre);
        F::regroup(x, P.q, P.qinv);
    } else {
        // index bits [10, LOGN-1) <-> r-bits [5-REM, 4) ; bit LOGN-1 <-> r-bit 4 is the scaled last stage
        inv_stages<F, LOGN, PatA<LOGN>, 5 - C::REM, 3, false, false, GTW>(x, tid, P.itw, P);
        inv_last_stage<F>(x, P.q, P.q2, ninv, ninv_s, ninvw, ninvw_s);
    }
}

// two inverse transforms in lock step (see fwd_core2)
template <class F, int LOGN, class Pat, int KLO, int KHI>
__device__ __forceinline__ void inv_stages2(typename F::E (&x0)[32], typename F::E (&x1)[32], uint32_t tid, const typename F::TW *__restrict__ itw,
                                            const Limb<F> &P) {
    const uint32_t base = Pat::TW_UNIFORM ? 0u : Pat::base(tid);
#pragma unroll
    for (int k = KLO; k <= KHI; k++) {
        const int b = Pat::BIT0 + k;
        const typename F::TW *p = itw + ((1u << (LOGN - 1 - b)) + (base >> (b + 1)));
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r & (1 << k)) continue;
            const typename F::TW w = load_global(p + (Pat::off(r) >> (b + 1)));
            F::inv_bfly(x0[r], x0[r | (1 << k)], w, P);
            F::inv_bfly(x1[r], x1[r | (1 << k)], w, P);
        }
    }
}
template <class F, int LOGN, bool PRESYNC = false>
__device__ __forceinline__ void inv_core2(typename F::E (&x0)[32], typename F::E (&x1)[32], typename F::E *lds, uint32_t tid,
                                          const Limb<F> &P, typename F::E ninv, typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s) {
    using C = NttCfg<LOGN>;
#ifdef FHE_NO_TWIDDLE_PRELOAD
    inv_stages2<F, LOGN, PatZ<LOGN>, 0, 4>(x0, x1, tid, P.itw, P);
    F::regroup(x0, P.q, P.qinv); F::regroup(x1, P.q, P.qinv);
    if constexpr (PRESYNC) __syncthreads();
    lds_put2<PatZ<LOGN>>(lds, tid, x0, x1);
    __syncthreads();
    lds_get2<PatY<LOGN>>(lds, tid, x0, x1);
    inv_stages2<F, LOGN, PatY<LOGN>, 0, 4>(x0, x1, tid, P.itw, P);
#else
    typename F::TW w[31];
    preload_twiddles<F, LOGN, PatZ<LOGN>, 4, 0>(w, tid, P.itw);          // needed at once (issued together: one latency, not five)
    inv_stages2_pre<F, 0, 4>(x0, x1, w, P);
    F::regroup(x0, P.q, P.qinv); F::regroup(x1, P.q, P.qinv);
    preload_twiddles<F, LOGN, PatY<LOGN>, 4, 0>(w, tid, P.itw);          // in flight under the first exchange
    if constexpr (PRESYNC) __syncthreads();
    lds_put2<PatZ<LOGN>>(lds, tid, x0, x1);
    __syncthreads();
    lds_get2<PatY<LOGN>>(lds, tid, x0, x1);
    inv_stages2_pre<F, 0, 4>(x0, x1, w, P);
#endif
    F::regroup(x0, P.q, P.qinv); F::regroup(x1, P.q, P.qinv);
    lds_put2<PatY<LOGN>>(lds, tid, x0, x1);
    __syncthreads();
    lds_get2<PatA<LOGN>>(lds, tid, x0, x1);
    inv_stages2<F, LOGN, PatA<LOGN>, 5 - C::REM, 3>(x0, x1, tid, P.itw, P);
    inv_last_stage<F>(x0, P.q, P.q2, ninv, ninv_s, ninvw, ninvw_s);
    inv_last_stage<F>(x1, P.q, P.q2, ninv, ninv_s, ninvw, ninvw_s);
}

// =========================================================================================================
// Kernels.  grid.x = batch * L workgroups; workgroup p handles polynomial p (limb p % L).
// =========================================================================================================
template <class F, int LOGN>
__global__ void __launch_bounds__(NttCfg<LOGN>::T)
ntt_forward_kernel(char *__restrict__ data, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    char *poly = data + (size_t)p * (C::N * 32);
    E x[32];
    load_A<F, LOGN>(poly, tid, x);
    fwd_core<F, LOGN>(x, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    lds_put<PatZ<LOGN>>(lds, tid, x);      // the slots this thread read last: no barrier needed before
    __syncthreads();
    store_from_lds<F, LOGN>(poly, lds, tid);
}

template <class F, int LOGN>
__global__ void __launch_bounds__(NttCfg<LOGN>::T)
ntt_inverse_kernel(char *__restrict__ data, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    char *poly = data + (size_t)p * (C::N * 32);
    E x[32];
    load_A<F, LOGN>(poly, tid, x);
    lds_put<PatA<LOGN>>(lds, tid, x);
    __syncthreads();
    lds_get<PatZ<LOGN>>(lds, tid, x);
    inv_core<F, LOGN>(x, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::canon_inv(x[r], P.q);
    lds_put<PatA<LOGN>>(lds, tid, x);
    __syncthreads();
    store_from_lds<F, LOGN>(poly, lds, tid);
}

// NTTEngine::multiply in one launch: r = INTT(NTT(a) .* NTT(b)); HBM traffic = read a + read b + write r.
// SQUARE: b is a (the host passes the flag when the operand pointers are equal): one load, one forward transform, 2*S of traffic.
// bcast != 0: b holds ONE RNS polynomial ([L][n]) that is multiplied into every element of the batch (served from L2 after its first
// use: 2*S of HBM traffic per product).
// COMPACT_OUT: the result goes to a compact workspace polynomial (load_A_compact) -- pieces of the fused multiply + relinearise.
template <class F, int LOGN, int MINW = 1, bool SQUARE = false, bool COMPACT_OUT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_multiply_kernel(char *res, const char *a, const char *b,      // no __restrict__: res may alias a and / or b (in-place product)
                    const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t bcast) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    // Under the 128-VGPR cap of four workgroups per CU (4-byte residues) the global_load form of the twiddle loads lets the scheduler
    // hoist more of them and the kernel spills 5 VGPRs (interleaved A/B at batch 4096: 1.934 M vs 1.946 M polymul/s); this one kernel
    // keeps the generic-pointer loads and stays scratch-free.  Every other kernel gains from global loads (ct +3..7 %, key switch +8..45 %).
    constexpr bool GTW = !(sizeof(E) == 4 && MINW >= 4 && !SQUARE);
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const uint32_t limb = p % L;
    const Limb<F> P = limbs[limb];
    const size_t off = (size_t)p * (C::N * 32);
    E x[32];
    load_A<F, LOGN>(a + off, tid, x);
    if constexpr (SQUARE) {
        fwd_core<F, LOGN, false, false, false, GTW>(x, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) { const E c = F::canon_fwd(x[r], P.q, P.q2, P.qinv); x[r] = F::pw_mul(c, c, P.q, P.qinv); }
    } else {
        E y[32];
        load_A<F, LOGN>(b + (size_t)(bcast ? limb : p) * (C::N * 32), tid, y);   // issued before a's butterflies: b's HBM latency hides under them
        fwd_core<F, LOGN, false, false, false, GTW>(x, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);   // canonical: keeps x*y < q*2^W
        __syncthreads();                   // all Z-pattern reads of a are done before b overwrites the slots
        fwd_core<F, LOGN, false, false, false, GTW>(y, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::pw_mul(x[r], y[r], P.q, P.qinv);   // [0,2q), carries 2^-W
    }
    inv_core<F, LOGN, false, false, false, GTW>(x, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::canon_inv(x[r], P.q);
    if constexpr (COMPACT_OUT) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(res) + (size_t)p * C::N, tid, x);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, x);
        __syncthreads();
        store_from_lds<F, LOGN>(res + off, lds, tid);
    }
}

// ---- transforms larger than the LDS range: N = 2^(LOGN + k), k = 1..3 --------------------------------------------------------------
// The top k stages (forward) / last k stages (inverse) run as one register-only pass over global memory (word_pass_kernel below);
// the 2^k blocks of 2^LOGN consecutive coefficients are then independent sub-transforms, each one workgroup of the LDS-resident
// machinery above with the big transform's twiddles (SUB = true, pre = 2^k + block).
// What crosses between the two launches of ONE call is a COMPACT polynomial (sizeof(E) bytes per coefficient in natural order, library
// workspace, see load_A_compact) -- only the call's own operands and results are 32-byte containers.  HBM traffic (4-byte residues):
//   transform  : S + S/8 (pass) + S/8 + S (sub-transforms)                     = 2.25 S   (round 2, container workspace: 4 S)
//   multiply   : 2 (S + S/8) (top passes of a, b) + 3 S/8 (sub-multiply) + S/8 + S (last pass) = 3.75 S   (round 2: 9 S)
// and S/4 instead of S/8 per compact polynomial for the 8-byte residues (2.5 S / 4.5 S).
// grid.x = polys << k: workgroup g handles block g & (2^k - 1) of polynomial g >> k.
// SUB_FORWARD: compact in, containers out.  SUB_INVERSE: containers in, compact out.  SUB_MULTIPLY: compact a, b in, compact out (res may
// be a: a workgroup reads its whole block of both operands before it stores).
enum { SUB_FORWARD = 0, SUB_INVERSE = 1, SUB_MULTIPLY = 2 };
template <class F, int LOGN, int MODE, int MINW = 1>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_sub_kernel(char *res, const char *a, const char *b, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t k) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x >> k, blk = blockIdx.x & ((1u << k) - 1), pre = (1u << k) + blk;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)blockIdx.x * (C::N * 32);          // containers: polynomial p starts at p << (LOGN + k) containers
    const size_t offc = (size_t)blockIdx.x * C::N;                // compact: the same block in residues
    E x[32];
    if constexpr (MODE == SUB_FORWARD) {
        load_A_compact<F, LOGN>(reinterpret_cast<const E *>(a) + offc, tid, x);
        fwd_core<F, LOGN, false, false, true>(x, lds, tid, P, nullptr, pre);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
        lds_put<PatZ<LOGN>>(lds, tid, x);
        __syncthreads();
        store_from_lds<F, LOGN>(res + off, lds, tid);
    } else {
        if constexpr (MODE == SUB_INVERSE) {
            load_A<F, LOGN>(a + off, tid, x);
            lds_put<PatA<LOGN>>(lds, tid, x);
            __syncthreads();
            lds_get<PatZ<LOGN>>(lds, tid, x);
        } else {
            E y[32];
            load_A_compact<F, LOGN>(reinterpret_cast<const E *>(a) + offc, tid, x);
            load_A_compact<F, LOGN>(reinterpret_cast<const E *>(b) + offc, tid, y);
            fwd_core<F, LOGN, false, false, true>(x, lds, tid, P, nullptr, pre);
#pragma unroll
            for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
            __syncthreads();
            fwd_core<F, LOGN, false, false, true>(y, lds, tid, P, nullptr, pre);
#pragma unroll
            for (int r = 0; r < 32; r++) x[r] = F::pw_mul(x[r], y[r], P.q, P.qinv);   // carries 2^-W until the last pass (ninv_r constants)
        }
        inv_core<F, LOGN, false, false, true>(x, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s, nullptr, pre);
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = F::canon_inv(x[r], P.q);
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(res) + offc, tid, x);     // pattern A registers: consecutive lanes, consecutive words
    }
}

// One register-only pass over R <= 3 stages of a transform of 2^log_n coefficients on the field type: FWD: the top R stages (index
// bits log_n-1 .. log_n-R), natural-order canonical input; !FWD: the last R stages (the same bits, ascending) with the n^-1 scaling
// folded into the final butterfly (rconst: the constants that also absorb the 2^-W of a fused pointwise product).  Canonical
// residues in and out.
// FWD : containers (src, the caller's operand) -> compact (dst, workspace): ONE lane per column of 2^R coefficients, every store
//       instruction of a wave writes 64 consecutive residues.
// !FWD: compact (src, workspace) -> containers (dst, the caller's result): lanes work in PAIRS -- lanes 2c and 2c+1 load the same
//       compact words and run the same butterflies; the even lane stores the value half of each output container and the odd lane the
//       zero half, so a wave instruction writes 1 KiB of consecutive bytes (like store_from_lds) instead of every other 16 bytes of
//       2 KiB (5.5 instead of 4.1 TB/s on the container-to-container pass of round 2; the arithmetic is far from binding).
// grid = (columns * (FWD ? 1 : 2) / 256, polys), columns = 2^(log_n - R).
template <class F, int R, bool FWD>
__global__ void __launch_bounds__(256)
word_pass_kernel(void *dst, const void *src, const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t log_n, uint32_t rconst) {
    using E = typename F::E;
    using V = typename F::V16;
    const uint32_t g = blockIdx.x * 256 + threadIdx.x, u = FWD ? g : g >> 1, half = FWD ? 0u : g & 1;   // u < 2^log_n >> R by construction of the grid
    const uint32_t p = blockIdx.y;
    const Limb<F> P = limbs[p % L];
    const uint32_t b_lo = log_n - R;
    E x[1 << R];
    if constexpr (FWD) {
        const V *in = reinterpret_cast<const V *>(src) + ((size_t)p << (log_n + 1));
#pragma unroll
        for (int k = 0; k < (1 << R); k++) x[k] = F::load_low(in + 2 * ((size_t)u + ((size_t)k << b_lo)));
    } else {
        const E *in = reinterpret_cast<const E *>(src) + ((size_t)p << log_n);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) x[k] = in[(size_t)u + ((size_t)k << b_lo)];
    }
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int pos = FWD ? R - 1 - j : j;                                  // k-bit of this stage; index bit b_lo + pos
        if (!FWD && pos == R - 1) {                                            // bit log_n-1: single twiddle itw[1], with the scaling
#pragma unroll
            for (int k = 0; k < (1 << (R - 1)); k++)
                F::inv_last(x[k], x[k | (1 << (R - 1))], P.q, P.q2, rconst ? P.ninv_r : P.ninv, rconst ? P.ninv_r_s : P.ninv_s,
                            rconst ? P.ninvw_r : P.ninvw, rconst ? P.ninvw_r_s : P.ninvw_s);
            continue;
        }
#pragma unroll
        for (int hh = 0; hh < (1 << (R - 1)); hh++) {
            const int k = ((hh >> pos) << (pos + 1)) | (hh & ((1 << pos) - 1));
            // index bit b = b_lo + pos: twiddle (n >> (b+1)) + (i >> (b+1)) = 2^(R-1-pos) + (k >> (pos+1)): independent of the lane
            const typename F::TW w = (FWD ? P.tw : P.itw)[(1u << (R - 1 - pos)) + (k >> (pos + 1))];
            if (FWD) F::fwd_bfly(x[k], x[k | (1 << pos)], w, P);
            else F::inv_bfly(x[k], x[k | (1 << pos)], w, P);
        }
    }
    if constexpr (FWD) {
        E *out = reinterpret_cast<E *>(dst) + ((size_t)p << log_n);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) out[(size_t)u + ((size_t)k << b_lo)] = F::canon_fwd(x[k], P.q, P.q2, P.qinv);
    } else {
        V *out = reinterpret_cast<V *>(dst) + ((size_t)p << (log_n + 1));
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
            const E v = F::canon_inv(F::regroup1(x[k], P.q, P.qinv), P.q);
            __builtin_nontemporal_store(F::pack(half ? (E)0 : v), out + 2 * ((size_t)u + ((size_t)k << b_lo)) + half);
        }
    }
}

// r = a0 (*) b1 + a1 (*) b0 in one launch (the c1 term of the tensor product) for configurations whose four
// transformed operands do not fit the register file (8-byte residues at N = 2^14): at most three arrays are live.
// HBM traffic = read 4 polynomials + write 1.
template <class F, int LOGN, int MINW = 1, bool COMPACT_OUT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_mac2_kernel(char *__restrict__ res, const char *__restrict__ a0, const char *__restrict__ b1,
                const char *__restrict__ a1, const char *__restrict__ b0, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)p * (C::N * 32);
    E x[32], y[32], acc[32];
    load_A<F, LOGN>(a0 + off, tid, x);
    load_A<F, LOGN>(b1 + off, tid, y);
    fwd_core<F, LOGN>(x, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    __syncthreads();
    fwd_core<F, LOGN>(y, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 32; r++) acc[r] = F::pw_mul(x[r], y[r], P.q, P.qinv);
    load_A<F, LOGN>(a1 + off, tid, x);
    load_A<F, LOGN>(b0 + off, tid, y);
    __syncthreads();
    fwd_core<F, LOGN>(x, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = F::canon_fwd(x[r], P.q, P.q2, P.qinv);
    __syncthreads();
    fwd_core<F, LOGN>(y, lds, tid, P);
#pragma unroll
    for (int r = 0; r < 32; r++) acc[r] = F::pw_add(acc[r], F::pw_mul(x[r], y[r], P.q, P.qinv), P.q, P.q2);
    inv_core<F, LOGN>(acc, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) acc[r] = F::canon_inv(acc[r], P.q);
    if constexpr (COMPACT_OUT) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(res) + (size_t)p * C::N, tid, acc);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, acc);
        __syncthreads();
        store_from_lds<F, LOGN>(res + off, lds, tid);
    }
}

// ---- tensor product in TWO launches for configurations whose four transformed operands do not fit the register file (8-byte
// residues at N = 2^14): 7 transforms instead of the 11 of multiply(c0) + multiply(c2) + mac2(c1).
// Launch 1 (ntt_forward_compact_kernel, grid (polys, 2)): NTT(b0), NTT(b1) into compact workspace polynomials, in REGISTER order
// (slot tid + r*T holds register r of thread tid after the last forward stage, still lazy) -- only ever multiplied pointwise against
// registers of the same thread of launch 2, so no ordering is needed, and 8 instead of 32 bytes per coefficient.
// Launch 2 (ntt_ct_a_kernel): NTT(a0); c0 = INTT(A0 . B0); t = A0 . B1; NTT(a1); c1 = INTT(t + A1 . B0); c2 = INTT(A1 . B1).  At most two
// arrays are live across any transform, three in the pointwise phases.  HBM traffic: 2 S + 2 S/4 written, then 2 S + 4 S/4 read and
// 3 S (or 3 S/4, compact outputs) written: 7.5 S (5.25 S) against 11 S (8.75 S).
template <class F, int LOGN, int MINW = 1>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_forward_compact_kernel(typename F::E *__restrict__ w0, typename F::E *__restrict__ w1, const char *__restrict__ b0,
                           const char *__restrict__ b1, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    E x[32];
    load_A<F, LOGN>((blockIdx.y ? b1 : b0) + (size_t)p * (C::N * 32), tid, x);
    fwd_core<F, LOGN>(x, lds, tid, P);
    store_A_compact<F, LOGN>((blockIdx.y ? w1 : w0) + (size_t)p * C::N, tid, x);
}
template <class F, int LOGN, bool COMPACT_OUT>
__device__ __forceinline__ void ct_store(char *__restrict__ c, size_t p, typename F::E *lds, uint32_t tid, const typename F::E (&x)[32]) {
    using C = NttCfg<LOGN>;
    if constexpr (COMPACT_OUT) {
        store_A_compact<F, LOGN>(reinterpret_cast<typename F::E *>(c) + p * C::N, tid, x);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, x);      // the slots this thread read last
        __syncthreads();
        store_from_lds_rolled<F, LOGN>(c + p * (C::N * 32), lds, tid);
    }
}
template <class F, int LOGN, int MINW = 1, bool COMPACT_OUT = false, bool EARLY = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_ct_a_kernel(char *__restrict__ c0, char *__restrict__ c1, char *__restrict__ c2, const char *__restrict__ a0, const char *__restrict__ a1,
                const typename F::E *__restrict__ w0, const typename F::E *__restrict__ w1, const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)p * (C::N * 32);
    const E *B0 = w0 + (size_t)p * C::N, *B1 = w1 + (size_t)p * C::N;
    E X[32], Y[32], T[32];
    // CT_FENCE: the compiler may not move the next phase's 32 loads above the transform / store before it (they would be a third
    // live array across it: the kernel spills 600+ bytes per lane without the fences)
#define CT_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)
    load_poly_buf<F, LOGN, false>(a0 + off, tid, X);
    fwd_core<F, LOGN>(X, lds, tid, P);
    CT_FENCE();
    load_poly_buf<F, LOGN, true>(B0, tid, Y);
#pragma unroll
    for (int r = 0; r < 32; r++) { X[r] = F::canon_fwd(X[r], P.q, P.q2, P.qinv); T[r] = F::pw_mul(X[r], Y[r], P.q, P.qinv); }
    if constexpr (EARLY) load_poly_buf<F, LOGN, true>(B1, tid, Y);     // in flight under the inverse transform of c0 (a third live array)
    else CT_FENCE();
    inv_core<F, LOGN>(T, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) T[r] = F::canon_inv(T[r], P.q);
    ct_store<F, LOGN, COMPACT_OUT>(c0, p, lds, tid, T);
    CT_FENCE();
    if constexpr (!EARLY) load_poly_buf<F, LOGN, true>(B1, tid, Y);
#pragma unroll
    for (int r = 0; r < 32; r++) T[r] = F::pw_mul(X[r], Y[r], P.q, P.qinv);          // A0 . B1
    CT_FENCE();
    load_poly_buf<F, LOGN, false>(a1 + off, tid, X);
    __syncthreads();
    fwd_core<F, LOGN>(X, lds, tid, P);
    CT_FENCE();
    load_poly_buf<F, LOGN, true>(B0, tid, Y);
#pragma unroll
    for (int r = 0; r < 32; r++) { X[r] = F::canon_fwd(X[r], P.q, P.q2, P.qinv); T[r] = F::pw_add(T[r], F::pw_mul(X[r], Y[r], P.q, P.qinv), P.q, P.q2); }
    if constexpr (EARLY) load_poly_buf<F, LOGN, true>(B1, tid, Y);
    else CT_FENCE();
    inv_core<F, LOGN>(T, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) T[r] = F::canon_inv(T[r], P.q);
    ct_store<F, LOGN, COMPACT_OUT>(c1, p, lds, tid, T);
    CT_FENCE();
    if constexpr (!EARLY) load_poly_buf<F, LOGN, true>(B1, tid, Y);
#pragma unroll
    for (int r = 0; r < 32; r++) T[r] = F::pw_mul(X[r], Y[r], P.q, P.qinv);          // A1 . B1
    CT_FENCE();
    __syncthreads();
    inv_core<F, LOGN>(T, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) T[r] = F::canon_inv(T[r], P.q);
    ct_store<F, LOGN, COMPACT_OUT>(c2, p, lds, tid, T);
#undef CT_FENCE
}

// FHEContext::multiply tensor product in one launch (src/fhe.cu:199-218): 4 forward + 3 inverse transforms,
// HBM traffic = read 4 polynomials + write 3.
// SQUARE: (b0, b1) is (a0, a1) (the host passes the flag when the operand pointers are equal -- squaring a ciphertext): two loads and
// two forward transforms instead of four, c1 = 2 a0 a1; 5*S of traffic instead of 7*S.
// COMPACT_C2: all three outputs go to compact workspace polynomials (see load_A_compact) instead of container buffers: the fused
// multiply + relinearise, whose key-switch kernel reads them back as digit source (c2) and addends (c0, c1).
// transforms of the one-launch tensor product: the 4-byte field has the registers to issue each group's twiddles one exchange ahead
// in the compact-output form (fwd_core_pre / inv_core_pre: 231 VGPRs, two waves per SIMD as before); with container outputs the extra 31
// registers cost a wave per SIMD (256 VGPRs), and the 8-byte fields have none to spare: those keep the plain form
template <class F, int LOGN, bool PRE>
__device__ __forceinline__ void ct_fwd(typename F::E (&x)[32], typename F::E *lds, uint32_t tid, const Limb<F> &P) {
    if constexpr (PRE) fwd_core_pre<F, LOGN>(x, lds, tid, P); else fwd_core<F, LOGN>(x, lds, tid, P);
}
template <class F, int LOGN, bool PRE>
__device__ __forceinline__ void ct_inv(typename F::E (&x)[32], typename F::E *lds, uint32_t tid, const Limb<F> &P,
                                       typename F::E ninv, typename F::E ninv_s, typename F::E ninvw, typename F::E ninvw_s) {
    if constexpr (PRE) inv_core_pre<F, LOGN>(x, lds, tid, P, ninv, ninv_s, ninvw, ninvw_s);
    else inv_core<F, LOGN>(x, lds, tid, P, ninv, ninv_s, ninvw, ninvw_s);
}
template <class F, int LOGN, bool SQUARE = false, bool COMPACT_C2 = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, (sizeof(typename F::E) == 8 && NttCfg<LOGN>::T <= 256) ? 2 : 1)   // 8-byte residues: two workgroups per CU
ntt_ct_multiply_kernel(char *__restrict__ c0, char *__restrict__ c1, char *__restrict__ c2,
                       const char *__restrict__ a0, const char *__restrict__ a1,
                       const char *__restrict__ b0, const char *__restrict__ b1,
                       const Limb<F> *__restrict__ limbs, uint32_t L) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, p = blockIdx.x;
    const Limb<F> P = limbs[p % L];
    const size_t off = (size_t)p * (C::N * 32);
    constexpr bool CT_PRE = sizeof(E) == 4 && COMPACT_C2 && LOGN <= 14;
    E A0[32], A1[32], B0[32];
    load_A<F, LOGN>(a0 + off, tid, A0);
    load_A<F, LOGN>(a1 + off, tid, A1);
    ct_fwd<F, LOGN, CT_PRE>(A0, lds, tid, P);
    if constexpr (SQUARE) {
        __syncthreads();
        ct_fwd<F, LOGN, CT_PRE>(A1, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const E u0 = F::canon_fwd(A0[r], P.q, P.q2, P.qinv), u1 = F::canon_fwd(A1[r], P.q, P.q2, P.qinv);
            A0[r] = F::pw_mul(u0, u0, P.q, P.qinv);
            A1[r] = F::pw_mul2(u0, u1, u1, u0, P.q, P.q2, P.qinv);   // 2 a0 a1
            B0[r] = F::pw_mul(u1, u1, P.q, P.qinv);
        }
    } else {
        E B1[32];
        load_A<F, LOGN>(b0 + off, tid, B0);
        __syncthreads();
        ct_fwd<F, LOGN, CT_PRE>(A1, lds, tid, P);
        load_A<F, LOGN>(b1 + off, tid, B1);
        __syncthreads();
        ct_fwd<F, LOGN, CT_PRE>(B0, lds, tid, P);
        __syncthreads();
        ct_fwd<F, LOGN, CT_PRE>(B1, lds, tid, P);
#pragma unroll
        for (int r = 0; r < 32; r++) {
            E u0 = F::canon_fwd(A0[r], P.q, P.q2, P.qinv), u1 = F::canon_fwd(A1[r], P.q, P.q2, P.qinv);   // canonical a-side
            E v0 = B0[r], v1 = B1[r];                                                             // lazy b-side (< 4q)
            A0[r] = F::pw_mul(u0, v0, P.q, P.qinv);                  // [0,2q)
            A1[r] = F::pw_mul2(u0, v1, u1, v0, P.q, P.q2, P.qinv);   // a0*b1 + a1*b0 with one shared reduction on the 32-bit field
            B0[r] = F::pw_mul(u1, v1, P.q, P.qinv);
        }
    }
    ct_inv<F, LOGN, CT_PRE>(A0, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) A0[r] = F::canon_inv(A0[r], P.q);
    if constexpr (COMPACT_C2) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(c0) + (size_t)p * C::N, tid, A0);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, A0);
        __syncthreads();
        store_from_lds<F, LOGN>(c0 + off, lds, tid);
    }
    __syncthreads();
    ct_inv<F, LOGN, CT_PRE>(A1, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) A1[r] = F::canon_inv(A1[r], P.q);
    if constexpr (COMPACT_C2) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(c1) + (size_t)p * C::N, tid, A1);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, A1);
        __syncthreads();
        store_from_lds<F, LOGN>(c1 + off, lds, tid);
    }
    __syncthreads();
    ct_inv<F, LOGN, CT_PRE>(B0, lds, tid, P, P.ninv_r, P.ninv_r_s, P.ninvw_r, P.ninvw_r_s);
#pragma unroll
    for (int r = 0; r < 32; r++) B0[r] = F::canon_inv(B0[r], P.q);
    if constexpr (COMPACT_C2) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(c2) + (size_t)p * C::N, tid, B0);   // pattern A: consecutive lanes, consecutive words
    } else {
        lds_put<PatA<LOGN>>(lds, tid, B0);
        __syncthreads();
        store_from_lds<F, LOGN>(c2 + off, lds, tid);
    }
}

// ---- fused key switching (relinearisation): packed key tables are built by pack_keys_kernel (ntt_word.hip.h) --------------------
// One workgroup per (ciphertext b, limb i): for every limb j of c2 and every digit k, the digit polynomial is formed in
// registers, transformed under q_i, multiplied with both key halves and accumulated in the NTT domain; two inverse
// transforms and the additions to c0, c1 finish the job.  HBM traffic per ciphertext: L reads of c2 + read/write of c0, c1
// = (L + 4) * S bytes (re-reads of c2 by the L workgroups of one ciphertext mostly hit the Infinity Cache).
// SPLIT = false: one workgroup per (ciphertext b, limb i) accumulates both key halves (4 live arrays per thread; 4-byte residues).
// SPLIT = true : two workgroups per (b, i), one per key half (3 live arrays; 8-byte residues, whose four arrays would not fit the
//                register file); the digit transforms are then computed twice, which is still far cheaper than the general
//                composition's container-sized digit workspace.
// TWL = true: the forward (then the inverse) twiddle table of limb i is copied into LDS once per workgroup; the 2*L*K + 2
// transforms then take their non-uniform twiddles from LDS (~100 cycles, conflict-free in the permuted order) instead of L2
// (500+ cycles), which these register-starved kernels (2 waves per SIMD) cannot hide.  N * sizeof(TW) more LDS per workgroup.
template <class F, int LOGN, int MINW = 1, bool SPLIT = false, bool TWL = false, bool COMPACT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_keyswitch_kernel(char *c0, char *c1, const char *__restrict__ c2, const char *add0, const char *add1,   // add* = c* (in place) unless COMPACT
                     const typename F::E *__restrict__ kb, const typename F::E *__restrict__ ka,
                     const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    static_assert(!(TWL && SPLIT), "LDS twiddles are wired into the one-workgroup-per-limb form only");
    __shared__ E lds[C::LDS_ELEMS];
    __shared__ typename F::TW twl[TWL ? C::N : 1];
    // XCD-aware workgroup -> (ciphertext, limb[, half]) map: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8
    // names the L2 a workgroup shares), and the workgroups of one ciphertext all re-read the same c2, so they are given block
    // indices that are congruent mod 8 and close together: the re-reads then hit one XCD's L2 instead of crossing the
    // fabric.  Placement only changes speed, never results.
    constexpr uint32_t H = SPLIT ? 2 : 1;
    const uint32_t LH = L * H;
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * LH)) * (8 * LH);
    uint32_t b, u;
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / LH); u = s % LH; }
    else { b = bid / LH; u = bid % LH; }
    const uint32_t i = u / H, half = u % H;
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    if constexpr (!SPLIT) {
        E acc0[32], acc1[32], x[32], d[32];
#pragma unroll
        for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
        if constexpr (TWL) { stage_twiddles<F, LOGN, true>(twl, P.tw, tid); __syncthreads(); }
        for (uint32_t j = 0; j < L; j++) {
            load_src<F, LOGN, COMPACT>(c2, (size_t)b * L + j, tid, x);
            for (uint32_t k = 0; k < K; k++) {
#pragma unroll
                for (int r = 0; r < 32; r++) d[r] = F::digit(x[r], k * w, w);
                fwd_core<F, LOGN, TWL, true>(d, lds, tid, P, twl);   // PRESYNC: the previous digit's Z-pattern reads are done before this exchange
                const size_t tbl = ((size_t)(j * K + k) * L + i) * C::N;
                const VecE *pb = reinterpret_cast<const VecE *>(kb + tbl) + tid, *pa = reinterpret_cast<const VecE *>(ka + tbl) + tid;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const VecE vb = pb[c * C::T], va = pa[c * C::T];
#pragma unroll
                    for (int e = 0; e < VPL; e++) {
                        const int r = c * VPL + e;
                        acc0[r] = F::pw_add(acc0[r], F::pw_mul(vb[e], d[r], P.q, P.qinv), P.q, P.q2);
                        acc1[r] = F::pw_add(acc1[r], F::pw_mul(va[e], d[r], P.q, P.qinv), P.q, P.q2);
                    }
                }
            }
        }
        // acc0 -> coefficient domain, + c0.  With LDS twiddles every forward-table read must be over before the table is swapped;
        // otherwise the barrier before the first exchange is inv_core's PRESYNC.
        if constexpr (TWL) { __syncthreads(); stage_twiddles<F, LOGN, false>(twl, P.itw, tid); __syncthreads(); }
        inv_core<F, LOGN, TWL, !TWL>(acc0, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s, twl);
        load_src<F, LOGN, COMPACT>(add0, p, tid, x);
#pragma unroll
        for (int r = 0; r < 32; r++) acc0[r] = F::ew_add(F::canon_inv(acc0[r], P.q), x[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc0);
        __syncthreads();
        store_from_lds<F, LOGN>(c0 + (size_t)p * (C::N * 32), lds, tid);
        __syncthreads();
        inv_core<F, LOGN, TWL>(acc1, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s, twl);
        load_src<F, LOGN, COMPACT>(add1, p, tid, x);
#pragma unroll
        for (int r = 0; r < 32; r++) acc1[r] = F::ew_add(F::canon_inv(acc1[r], P.q), x[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc1);
        __syncthreads();
        store_from_lds<F, LOGN>(c1 + (size_t)p * (C::N * 32), lds, tid);
    } else {
        const E *keys = half ? ka : kb;
        char *dst = half ? c1 : c0;
        E acc[32], d[32];
#pragma unroll
        for (int r = 0; r < 32; r++) acc[r] = 0;
        for (uint32_t j = 0; j < L; j++) {
            for (uint32_t k = 0; k < K; k++) {
                // the c2 limb is re-read per digit (L2 / Infinity-Cache hits after the first) rather than held in 64 more VGPRs
                load_src<F, LOGN, COMPACT>(c2, (size_t)b * L + j, tid, d);
#pragma unroll
                for (int r = 0; r < 32; r++) d[r] = F::digit(d[r], k * w, w);
                fwd_core<F, LOGN, false, true>(d, lds, tid, P);
                __builtin_amdgcn_sched_barrier(0);   // keep the 16 key loads (64 VGPRs) from being hoisted into the transform
                const VecE *pk = reinterpret_cast<const VecE *>(keys + ((size_t)(j * K + k) * L + i) * C::N) + tid;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const VecE v = pk[c * C::T];
#pragma unroll
                    for (int e = 0; e < VPL; e++) {
                        const int r = c * VPL + e;
                        acc[r] = F::pw_add(acc[r], F::pw_mul(v[e], d[r], P.q, P.qinv), P.q, P.q2);
                    }
                }
            }
        }
        F::regroup(acc, P.q, P.qinv);              // floating-point sums of L*K products: back below q before the inverse
        inv_core<F, LOGN, false, true>(acc, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
        load_src<F, LOGN, COMPACT>(half ? add1 : add0, p, tid, d);
#pragma unroll
        for (int r = 0; r < 32; r++) acc[r] = F::ew_add(F::canon_inv(acc[r], P.q), d[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc);
        __syncthreads();
        store_from_lds<F, LOGN>(dst + (size_t)p * (C::N * 32), lds, tid);
    }
}

// Key switching for the 8-byte residues in ONE workgroup per (ciphertext, limb) with THREE live arrays: both accumulators and the digit
// polynomial; the c2 limb is re-read for every digit (compact workspace: N * 8 bytes from L2) instead of being held.  Against the SPLIT
// form of ntt_keyswitch_kernel (two workgroups per limb, one per key half, every digit transform computed twice) this halves the
// transforms; the register file is exceeded by what the compiler parks in scratch around the transforms.
template <class F, int LOGN, int MINW = 1, bool COMPACT = false, bool ADD_COMPACT = COMPACT>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_keyswitch3_kernel(char *c0, char *c1, const char *__restrict__ c2, const char *add0, const char *add1,
                      const typename F::E *__restrict__ kb, const typename F::E *__restrict__ ka,
                      const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * L)) * (8 * L);
    uint32_t b, i;
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / L); i = s % L; }
    else { b = bid / L; i = bid % L; }
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    E acc0[32], acc1[32], d[32];
#pragma unroll
    for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
    const TableBuf KB(kb), KA(ka), C2(c2 + (size_t)b * L * (C::N * (COMPACT ? sizeof(E) : 32)));   // c2 of this ciphertext
    const uint32_t voff = tid * 16;
    for (uint32_t j = 0; j < L; j++) {
        for (uint32_t k = 0; k < K; k++) {
            load_src_buf<F, LOGN, COMPACT>(C2, j, tid, d);
#pragma unroll
            for (int r = 0; r < 32; r++) d[r] = F::digit(d[r], k * w, w);
            fwd_core<F, LOGN, false, true>(d, lds, tid, P);
            __builtin_amdgcn_sched_barrier(0);   // keep the key loads out of the transform
            const uint32_t tbl = (uint32_t)((((size_t)(j * K + k) * L + i) * C::N) * sizeof(E));   // byte offset of the row (< 4 GiB: host)
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const VecE vb = KB.template load16<VecE>(voff, tbl + c * C::T * 16), va = KA.template load16<VecE>(voff, tbl + c * C::T * 16);
#pragma unroll
                for (int e = 0; e < VPL; e++) {
                    const int r = c * VPL + e;
                    acc0[r] = F::pw_add(acc0[r], F::pw_mul(vb[e], d[r], P.q, P.qinv), P.q, P.q2);
                    acc1[r] = F::pw_add(acc1[r], F::pw_mul(va[e], d[r], P.q, P.qinv), P.q, P.q2);
                }
                if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four chunks of key loads in flight at a time
            }
        }
    }
    F::regroup(acc0, P.q, P.qinv);
    inv_core<F, LOGN, false, true>(acc0, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    __builtin_amdgcn_sched_barrier(0);
    load_poly_buf<F, LOGN, ADD_COMPACT>(add0 + (size_t)p * (C::N * (ADD_COMPACT ? sizeof(E) : 32)), tid, d);
#pragma unroll
    for (int r = 0; r < 32; r++) acc0[r] = F::ew_add(F::canon_inv(acc0[r], P.q), d[r], P.q);
    lds_put<PatA<LOGN>>(lds, tid, acc0);
    __syncthreads();
    store_from_lds_rolled<F, LOGN>(c0 + (size_t)p * (C::N * 32), lds, tid);
    __builtin_amdgcn_sched_barrier(0);
    F::regroup(acc1, P.q, P.qinv);
    __syncthreads();
    inv_core<F, LOGN>(acc1, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    load_poly_buf<F, LOGN, ADD_COMPACT>(add1 + (size_t)p * (C::N * (ADD_COMPACT ? sizeof(E) : 32)), tid, d);
#pragma unroll
    for (int r = 0; r < 32; r++) acc1[r] = F::ew_add(F::canon_inv(acc1[r], P.q), d[r], P.q);
    lds_put<PatA<LOGN>>(lds, tid, acc1);
    __syncthreads();
    store_from_lds<F, LOGN>(c1 + (size_t)p * (C::N * 32), lds, tid);
}

// ---- fused blind-rotation step (external product) ------------------------------------------------------------------------
// out = in + ExtProd((X^a - 1) * in, RGSW)  for the RLWE pair in = (in0, in1), OUT OF PLACE (workgroup (b, i) reads every limb of
// in and writes limb i of out, so out must not alias in).  (FHEContext::blind_rotate is declared only, include/fhe.cuh:139.)
//   out0[i] = in0[i] + INTT_i( sum_{c in {0,1}} sum_{j,k} NTT_i(digit_k(((X^a - 1) * in_c)[j])) .* KB_c[jk][i] ),  out1 likewise with KA_c
// The monomial factor is applied while loading: coefficient x of (X^a - 1) * p is +-p[(x - a) mod n] - p[x]; the limb is loaded once
// (coalesced, pattern A), parked in the exchange buffer and read back rotated, so HBM and the texture path see one read per limb.
// 2*L*K forward + 2 inverse transforms per workgroup; HBM traffic per ciphertext: read in0, in1 (re-reads by the L workgroups of a
// ciphertext are XCD-L2 / Infinity-Cache hits, same block map as the key-switch kernel) + write out0, out1 = 4 * S.
// SPLIT as for the key-switch kernel: two workgroups per (b, i), one per output component.
template <class F, int LOGN>
__device__ __forceinline__ void rotate_through_lds(typename F::E *lds, uint32_t tid, uint32_t a, typename F::E qj, typename F::E (&x)[32]);
template <class F, int LOGN, bool COMPACT = false>      // the same through a descriptor based at the accumulator's first limb (load_src_buf)
__device__ __forceinline__ void load_monomial_A_buf(const TableBuf &B, uint32_t limb, typename F::E *lds, uint32_t tid, uint32_t a,
                                                    typename F::E qj, typename F::E (&x)[32]) {
    load_src_buf<F, LOGN, COMPACT>(B, limb, tid, x);
    rotate_through_lds<F, LOGN>(lds, tid, a, qj, x);
}
template <class F, int LOGN, bool COMPACT = false>
__device__ __forceinline__ void load_monomial_A(const char *__restrict__ base, size_t poly_index, typename F::E *lds, uint32_t tid, uint32_t a,
                                                typename F::E qj, typename F::E (&x)[32]) {
    load_src<F, LOGN, COMPACT>(base, poly_index, tid, x);   // p[i], i = tid + r*T
    rotate_through_lds<F, LOGN>(lds, tid, a, qj, x);
}
template <class F, int LOGN>
__device__ __forceinline__ void rotate_through_lds(typename F::E *lds, uint32_t tid, uint32_t a, typename F::E qj, typename F::E (&x)[32]) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __syncthreads();                                // the previous transform's last reads of the exchange buffer are over
    // UNPADDED image for this one round trip: both the store (lane-consecutive, pattern A order) and the rotated read-back
    // (lane-consecutive from an arbitrary start) are conflict-free without padding, whereas on the padded image a read that does
    // not start on a multiple of 32 crosses one pad slot inside a 32-lane group: a 2-way conflict on every instruction (round 2:
    // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.07 in the external-product kernels, the only family with conflicts).
#pragma unroll
    for (int r = 0; r < 32; r++) lds[tid + (uint32_t)r * C::T] = x[r];
    __syncthreads();
    const uint32_t k0 = tid + 2 * C::N - a;
#pragma unroll
    for (int r = 0; r < 32; r++) {
        uint32_t k = (k0 + (uint32_t)r * C::T) & (2 * C::N - 1);      // (i - a) mod 2n
        const bool neg = k >= (uint32_t)C::N;                         // X^n = -1
        k &= C::N - 1;
        E v = lds[k];                                                 // consecutive lanes, consecutive slots (the wrap at n keeps banks distinct)
        if (neg) v = F::ew_sub((E)0, v, qj);
        x[r] = F::ew_sub(v, x[r], qj);
    }
    __syncthreads();                                // rotated reads are done before the first transform reuses the buffer
}

template <class F, int LOGN, int MINW = 1, bool SPLIT = false, bool TWL = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_extprod_kernel(char *__restrict__ out0, char *__restrict__ out1, const char *__restrict__ in0, const char *__restrict__ in1,
                   const uint32_t *__restrict__ shifts,
                   const typename F::E *__restrict__ kb0, const typename F::E *__restrict__ ka0,      // rows for component 0
                   const typename F::E *__restrict__ kb1, const typename F::E *__restrict__ ka1,      // rows for component 1
                   const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    static_assert(!(TWL && SPLIT), "LDS twiddles are wired into the one-workgroup-per-limb form only");
    __shared__ E lds[C::LDS_ELEMS];
    __shared__ typename F::TW twl[TWL ? C::N : 1];
    constexpr uint32_t H = SPLIT ? 2 : 1;
    const uint32_t LH = L * H;
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * LH)) * (8 * LH);
    uint32_t b, u;
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / LH); u = s % LH; }
    else { b = bid / LH; u = bid % LH; }
    const uint32_t i = u / H, half = u % H;
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    const uint32_t a = shifts[b] & (2 * C::N - 1);
    if constexpr (!SPLIT) {
        E acc0[32], acc1[32], x[32], d[32];
#pragma unroll
        for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
        if constexpr (TWL) stage_twiddles<F, LOGN, true>(twl, P.tw, tid);     // visible after the barriers inside load_monomial_A
        for (uint32_t c = 0; c < 2; c++) {
            const char *src = c ? in1 : in0;
            const E *kb = c ? kb1 : kb0, *ka = c ? ka1 : ka0;
            for (uint32_t j = 0; j < L; j++) {
                load_monomial_A<F, LOGN>(src, (size_t)b * L + j, lds, tid, a, limbs[j].q, x);
                for (uint32_t k = 0; k < K; k++) {
#pragma unroll
                    for (int r = 0; r < 32; r++) d[r] = F::digit(x[r], k * w, w);
                    fwd_core<F, LOGN, TWL, true>(d, lds, tid, P, twl);
                    const size_t tbl = ((size_t)(j * K + k) * L + i) * C::N;
                    const VecE *pb = reinterpret_cast<const VecE *>(kb + tbl) + tid, *pa = reinterpret_cast<const VecE *>(ka + tbl) + tid;
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++) {
                        const VecE vb = pb[ch * C::T], va = pa[ch * C::T];
#pragma unroll
                        for (int e = 0; e < VPL; e++) {
                            const int r = ch * VPL + e;
                            acc0[r] = F::pw_add(acc0[r], F::pw_mul(vb[e], d[r], P.q, P.qinv), P.q, P.q2);
                            acc1[r] = F::pw_add(acc1[r], F::pw_mul(va[e], d[r], P.q, P.qinv), P.q, P.q2);
                        }
                    }
                }
            }
        }
        if constexpr (TWL) { __syncthreads(); stage_twiddles<F, LOGN, false>(twl, P.itw, tid); __syncthreads(); }
        inv_core<F, LOGN, TWL, !TWL>(acc0, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s, twl);
        load_A<F, LOGN>(in0 + (size_t)p * (C::N * 32), tid, x);
#pragma unroll
        for (int r = 0; r < 32; r++) acc0[r] = F::ew_add(F::canon_inv(acc0[r], P.q), x[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc0);
        __syncthreads();
        store_from_lds<F, LOGN>(out0 + (size_t)p * (C::N * 32), lds, tid);
        __syncthreads();
        inv_core<F, LOGN, TWL>(acc1, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s, twl);
        load_A<F, LOGN>(in1 + (size_t)p * (C::N * 32), tid, x);
#pragma unroll
        for (int r = 0; r < 32; r++) acc1[r] = F::ew_add(F::canon_inv(acc1[r], P.q), x[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc1);
        __syncthreads();
        store_from_lds<F, LOGN>(out1 + (size_t)p * (C::N * 32), lds, tid);
    } else {
        E acc[32], d[32];
#pragma unroll
        for (int r = 0; r < 32; r++) acc[r] = 0;
        for (uint32_t c = 0; c < 2; c++) {
            const char *src = c ? in1 : in0;
            const E *keys = c ? (half ? ka1 : kb1) : (half ? ka0 : kb0);
            for (uint32_t j = 0; j < L; j++) {
                const E qj = limbs[j].q;
                for (uint32_t k = 0; k < K; k++) {
                    load_monomial_A<F, LOGN>(src, (size_t)b * L + j, lds, tid, a, qj, d);
#pragma unroll
                    for (int r = 0; r < 32; r++) d[r] = F::digit(d[r], k * w, w);
                    fwd_core<F, LOGN, false, true>(d, lds, tid, P);
                    __builtin_amdgcn_sched_barrier(0);
                    const VecE *pk = reinterpret_cast<const VecE *>(keys + ((size_t)(j * K + k) * L + i) * C::N) + tid;
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++) {
                        const VecE v = pk[ch * C::T];
#pragma unroll
                        for (int e = 0; e < VPL; e++) {
                            const int r = ch * VPL + e;
                            acc[r] = F::pw_add(acc[r], F::pw_mul(v[e], d[r], P.q, P.qinv), P.q, P.q2);
                        }
                    }
                }
            }
            F::regroup(acc, P.q, P.qinv);          // floating-point sums of L*K products: back below q (no-op for the integer fields)
        }
        inv_core<F, LOGN, false, true>(acc, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
        load_A<F, LOGN>((half ? in1 : in0) + (size_t)p * (C::N * 32), tid, d);
#pragma unroll
        for (int r = 0; r < 32; r++) acc[r] = F::ew_add(F::canon_inv(acc[r], P.q), d[r], P.q);
        lds_put<PatA<LOGN>>(lds, tid, acc);
        __syncthreads();
        store_from_lds<F, LOGN>((half ? out1 : out0) + (size_t)p * (C::N * 32), lds, tid);
    }
}

// External product for the 8-byte residues at N = 2^14 in ONE workgroup per (accumulator, limb) with three live arrays (both output
// accumulators and the digit polynomial; the input limb is re-read, rotated, for every digit) -- the counterpart of
// ntt_keyswitch3_kernel: half the transforms of the SPLIT form above.  IN_COMPACT / OUT_COMPACT as in ntt_extprod2_kernel: inside fhe_blind_rotate
// the accumulator pair is compacted once (compact_kernel) and stays compact between the steps, so the L * K rotated re-reads of a limb per
// workgroup move 8 instead of 32 bytes per coefficient.
template <class F, int LOGN, int MINW = 1, bool IN_COMPACT = false, bool OUT_COMPACT = false, bool PREROT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_extprod3_kernel(char *__restrict__ out0, char *__restrict__ out1, const char *__restrict__ in0, const char *__restrict__ in1,
                    const char *__restrict__ rot0, const char *__restrict__ rot1,     // PREROT: (X^a - 1) * in, compact (monomial_compact_kernel); else unused
                    const uint32_t *__restrict__ shifts,
                    const typename F::E *__restrict__ kb0, const typename F::E *__restrict__ ka0,
                    const typename F::E *__restrict__ kb1, const typename F::E *__restrict__ ka1,
                    const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    constexpr size_t IN_POLY = C::N * (IN_COMPACT ? sizeof(E) : 32);        // bytes of one input polynomial
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    __shared__ E lds[C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * L)) * (8 * L);
    uint32_t b, i;
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / L); i = s % L; }
    else { b = bid / L; i = bid % L; }
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    const uint32_t a = shifts[b] & (2 * C::N - 1);
    E acc0[32], acc1[32], d[32];
#pragma unroll
    for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
    const uint32_t voff = tid * 16;
    for (uint32_t c = 0; c < 2; c++) {
        const TableBuf KB(c ? kb1 : kb0), KA(c ? ka1 : ka0),
                       SRC(PREROT ? (c ? rot1 : rot0) + (size_t)b * L * (C::N * sizeof(E)) : (c ? in1 : in0) + (size_t)b * L * IN_POLY);   // this accumulator's component c
        for (uint32_t j = 0; j < L; j++) {
            const E qj = limbs[j].q;
            for (uint32_t k = 0; k < K; k++) {
                // (the FP64 field keeps flat loads here: descriptor loads of the rotated limb measured 12-17 % slower on it, round 2)
                if constexpr (PREROT) load_src_buf<F, LOGN, true>(SRC, j, tid, d);       // already rotated: a plain compact load, no exchange-buffer round trip
                else if constexpr (__is_same(typename F::E, double)) load_monomial_A<F, LOGN, IN_COMPACT>(c ? in1 : in0, (size_t)b * L + j, lds, tid, a, qj, d);
                else load_monomial_A_buf<F, LOGN, IN_COMPACT>(SRC, j, lds, tid, a, qj, d);
#pragma unroll
                for (int r = 0; r < 32; r++) d[r] = F::digit(d[r], k * w, w);
                fwd_core<F, LOGN, false, true>(d, lds, tid, P);
                __builtin_amdgcn_sched_barrier(0);   // keep the key loads out of the transform
                const uint32_t tbl = (uint32_t)((((size_t)(j * K + k) * L + i) * C::N) * sizeof(E));
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) {
                    const VecE vb = KB.template load16<VecE>(voff, tbl + ch * C::T * 16), va = KA.template load16<VecE>(voff, tbl + ch * C::T * 16);
#pragma unroll
                    for (int e = 0; e < VPL; e++) {
                        const int r = ch * VPL + e;
                        acc0[r] = F::pw_add(acc0[r], F::pw_mul(vb[e], d[r], P.q, P.qinv), P.q, P.q2);
                        acc1[r] = F::pw_add(acc1[r], F::pw_mul(va[e], d[r], P.q, P.qinv), P.q, P.q2);
                    }
                    if ((ch & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        F::regroup(acc0, P.q, P.qinv);             // floating-point sums of L*K products: back below q (no-op for the integer fields)
        F::regroup(acc1, P.q, P.qinv);
    }
    inv_core<F, LOGN, false, true>(acc0, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    __builtin_amdgcn_sched_barrier(0);
    load_poly_buf<F, LOGN, IN_COMPACT>(in0 + (size_t)p * IN_POLY, tid, d);
#pragma unroll
    for (int r = 0; r < 32; r++) acc0[r] = F::ew_add(F::canon_inv(acc0[r], P.q), d[r], P.q);
    if constexpr (OUT_COMPACT) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(out0) + (size_t)p * C::N, tid, acc0);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, acc0);
        __syncthreads();
        store_from_lds_rolled<F, LOGN>(out0 + (size_t)p * (C::N * 32), lds, tid);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    inv_core<F, LOGN>(acc1, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
    load_poly_buf<F, LOGN, IN_COMPACT>(in1 + (size_t)p * IN_POLY, tid, d);
#pragma unroll
    for (int r = 0; r < 32; r++) acc1[r] = F::ew_add(F::canon_inv(acc1[r], P.q), d[r], P.q);
    if constexpr (OUT_COMPACT) {
        store_A_compact<F, LOGN>(reinterpret_cast<E *>(out1) + (size_t)p * C::N, tid, acc1);
    } else {
        lds_put<PatA<LOGN>>(lds, tid, acc1);
        __syncthreads();
        store_from_lds<F, LOGN>(out1 + (size_t)p * (C::N * 32), lds, tid);
    }
}

// ---- paired forms of the key-switch and external-product kernels (4-byte residues) --------------------------------------
// Same results as ntt_keyswitch_kernel / ntt_extprod_kernel (SPLIT = false); the digit polynomials are transformed two at a
// time (fwd_core2).  Registers: acc0, acc1, d0, d1 (the undecomposed limb is not kept: both digits of a pair are cut from the
// freshly loaded values).  LDS: two exchange buffers (66 KiB at N = 8192: two workgroups per CU; 132 KiB at N = 16384: one).
template <class F>
__device__ __forceinline__ void mac_keys(typename F::E (&acc0)[32], typename F::E (&acc1)[32], const typename F::E (&d)[32],
                                         const typename F::E *__restrict__ kb, const typename F::E *__restrict__ ka, size_t tbl, uint32_t tid,
                                         uint32_t T, const Limb<F> &P) {
    using E = typename F::E;
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    const TableBuf KB(kb), KA(ka);                          // SGPR descriptors + one shared VGPR offset: no per-chunk 64-bit addresses
    const uint32_t voff = tid * 16, row = (uint32_t)(tbl * sizeof(E));
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const VecE vb = KB.template load16<VecE>(voff, row + c * T * 16), va = KA.template load16<VecE>(voff, row + c * T * 16);
#pragma unroll
        for (int e = 0; e < VPL; e++) {
            const int r = c * VPL + e;
            acc0[r] = F::pw_add(acc0[r], F::pw_mul(vb[e], d[r], P.q, P.qinv), P.q, P.q2);
            acc1[r] = F::pw_add(acc1[r], F::pw_mul(va[e], d[r], P.q, P.qinv), P.q, P.q2);
        }
    }
}
// acc0 += kb[t0] .* d0 + kb[t1] .* d1, acc1 likewise with ka: the two products of a pair share one Montgomery reduction (mont_mul2)
// PIPE: the four 16-byte key loads of chunk c+1 are issued BEFORE the products of chunk c (explicit double buffer, pinned by scheduling
// fences): without it every chunk's loads are consumed right after they are issued and a wave exposes the L2 latency eight times per
// digit pair (the compiler's own schedule keeps at most ~1.5 chunks in flight); 16 more VGPRs.
template <class F, bool PIPE = false>
__device__ __forceinline__ void mac_keys2(typename F::E (&acc0)[32], typename F::E (&acc1)[32], const typename F::E (&d0)[32], const typename F::E (&d1)[32],
                                          const typename F::E *__restrict__ kb0, const typename F::E *__restrict__ ka0, size_t tbl0,
                                          const typename F::E *__restrict__ kb1, const typename F::E *__restrict__ ka1, size_t tbl1,
                                          uint32_t tid, uint32_t T, const Limb<F> &P) {
    using E = typename F::E;
    static_assert(sizeof(E) == 4, "mont_mul2 is a 32-bit field operation");
    constexpr int VPL = 16 / sizeof(E), NCH = 32 / VPL;
    typedef E VecE __attribute__((ext_vector_type(VPL)));
    const TableBuf KB0(kb0), KA0(ka0), KB1(kb1), KA1(ka1);
    const uint32_t voff = tid * 16, row0 = (uint32_t)(tbl0 * sizeof(E)), row1 = (uint32_t)(tbl1 * sizeof(E));
    if constexpr (PIPE) {
        VecE vb0[2], va0[2], vb1[2], va1[2];
        vb0[0] = KB0.template load16<VecE>(voff, row0); va0[0] = KA0.template load16<VecE>(voff, row0);
        vb1[0] = KB1.template load16<VecE>(voff, row1); va1[0] = KA1.template load16<VecE>(voff, row1);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int cur = c & 1, nxt = cur ^ 1;
            if (c + 1 < NCH) {
                vb0[nxt] = KB0.template load16<VecE>(voff, row0 + (c + 1) * T * 16); va0[nxt] = KA0.template load16<VecE>(voff, row0 + (c + 1) * T * 16);
                vb1[nxt] = KB1.template load16<VecE>(voff, row1 + (c + 1) * T * 16); va1[nxt] = KA1.template load16<VecE>(voff, row1 + (c + 1) * T * 16);
            }
            __builtin_amdgcn_sched_barrier(0);          // the next chunk's loads stay above this chunk's products
#pragma unroll
            for (int e = 0; e < VPL; e++) {
                const int r = c * VPL + e;
                acc0[r] = F::pw_add(acc0[r], F::mont_mul2(vb0[cur][e], d0[r], vb1[cur][e], d1[r], P.q, P.q2, P.qinv), P.q, P.q2);
                acc1[r] = F::pw_add(acc1[r], F::mont_mul2(va0[cur][e], d0[r], va1[cur][e], d1[r], P.q, P.q2, P.qinv), P.q, P.q2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const VecE vb0 = KB0.template load16<VecE>(voff, row0 + c * T * 16), va0 = KA0.template load16<VecE>(voff, row0 + c * T * 16);
            const VecE vb1 = KB1.template load16<VecE>(voff, row1 + c * T * 16), va1 = KA1.template load16<VecE>(voff, row1 + c * T * 16);
#pragma unroll
            for (int e = 0; e < VPL; e++) {
                const int r = c * VPL + e;
                acc0[r] = F::pw_add(acc0[r], F::mont_mul2(vb0[e], d0[r], vb1[e], d1[r], P.q, P.q2, P.qinv), P.q, P.q2);
                acc1[r] = F::pw_add(acc1[r], F::mont_mul2(va0[e], d0[r], va1[e], d1[r], P.q, P.q2, P.qinv), P.q, P.q2);
            }
        }
    }
}

// both accumulators back to the coefficient domain in lock step, + the addend polynomials, store  (tail of the paired kernels)
// add0 / add1: base pointers of the addend buffers (containers, or compact polynomials when COMPACT), p = polynomial index
// dst0 / dst1: base pointers of the output buffers (containers, or compact polynomials when COMPACT_OUT)
template <class F, int LOGN, bool COMPACT = false, bool COMPACT_OUT = false>
__device__ __forceinline__ void finish_pair(typename F::E (&acc0)[32], typename F::E (&acc1)[32], typename F::E (&t0)[32], typename F::E (&t1)[32],
                                            typename F::E *lds, uint32_t tid, const Limb<F> &P,
                                            const char *add0, const char *add1, size_t p, char *dst0, char *dst1) {
    load_src<F, LOGN, COMPACT>(add0, p, tid, t0);   // issued first: the HBM latency hides under the inverse transforms
    load_src<F, LOGN, COMPACT>(add1, p, tid, t1);
    inv_core2<F, LOGN, true>(acc0, acc1, lds, tid, P, P.ninv, P.ninv_s, P.ninvw, P.ninvw_s);
#pragma unroll
    for (int r = 0; r < 32; r++) {
        acc0[r] = F::ew_add(F::canon_inv(acc0[r], P.q), t0[r], P.q);
        acc1[r] = F::ew_add(F::canon_inv(acc1[r], P.q), t1[r], P.q);
    }
    if constexpr (COMPACT_OUT) {
        store_A_compact<F, LOGN>(reinterpret_cast<typename F::E *>(dst0) + p * NttCfg<LOGN>::N, tid, acc0);
        store_A_compact<F, LOGN>(reinterpret_cast<typename F::E *>(dst1) + p * NttCfg<LOGN>::N, tid, acc1);
    } else {
        typename F::E *lds0 = lds, *lds1 = lds + NttCfg<LOGN>::LDS_ELEMS;   // two 4-byte images for the container stores
        __syncthreads();                       // every pair read of the last exchange is over before the images are rewritten
        lds_put<PatA<LOGN>>(lds0, tid, acc0);
        lds_put<PatA<LOGN>>(lds1, tid, acc1);
        __syncthreads();
        store_from_lds<F, LOGN>(dst0 + p * (NttCfg<LOGN>::N * 32), lds0, tid);
        store_from_lds<F, LOGN>(dst1 + p * (NttCfg<LOGN>::N * 32), lds1, tid);
    }
}

template <class F, int LOGN, int MINW = 1, bool COMPACT = false, bool ADD_COMPACT = COMPACT>      // COMPACT: c2 is a compact polynomial; ADD_COMPACT: the addends are too
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_keyswitch2_kernel(char *c0, char *c1, const char *__restrict__ c2, const char *add0, const char *add1,   // add* = c* (in place) unless COMPACT
                      const typename F::E *__restrict__ kb, const typename F::E *__restrict__ ka,
                      const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[2 * C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * L)) * (8 * L);
    uint32_t b, i;                                        // XCD-aware map, as in ntt_keyswitch_kernel
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / L); i = s % L; }
    else { b = bid / L; i = bid % L; }
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    const TableBuf C2(c2 + (size_t)b * L * (C::N * (COMPACT ? sizeof(E) : 32)));   // descriptor based at the first limb polynomial of this ciphertext's c2
    E acc0[32], acc1[32], d0[32], d1[32];
#pragma unroll
    for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
    const uint32_t LK = L * K;
    uint32_t jk = 0;
    // PREFETCH (N <= 2^13; at 2^14 the 32 extra registers spill): the source limb of the NEXT pair is loaded before the key products of
    // the current one (nx), so its HBM / L2 latency hides under the multiply-accumulate phase instead of being exposed at the top of
    // every iteration.
    constexpr bool PREFETCH = LOGN <= 13;
    E nx[32];
    if (PREFETCH && LK >= 2) load_src_buf<F, LOGN, COMPACT>(C2, 1 / K, tid, nx);
    for (; jk + 1 < LK; jk += 2) {
        const uint32_t j0 = jk / K, k0 = jk % K, j1 = (jk + 1) / K, k1 = (jk + 1) % K;
        if constexpr (!PREFETCH) load_src_buf<F, LOGN, COMPACT>(C2, j1, tid, nx);
        if (j0 == j1) {
#pragma unroll
            for (int r = 0; r < 32; r++) d0[r] = F::digit(nx[r], k0 * w, w);
        } else {
            load_src_buf<F, LOGN, COMPACT>(C2, j0, tid, d0);
#pragma unroll
            for (int r = 0; r < 32; r++) d0[r] = F::digit(d0[r], k0 * w, w);
        }
#pragma unroll
        for (int r = 0; r < 32; r++) d1[r] = F::digit(nx[r], k1 * w, w);
        fwd_core2<F, LOGN, true>(d0, d1, lds, tid, P);
        if constexpr (PREFETCH) {
            __builtin_amdgcn_sched_barrier(0);           // not earlier: 32 more live registers inside the transform would spill
            if (jk + 3 < LK) load_src_buf<F, LOGN, COMPACT>(C2, (jk + 3) / K, tid, nx);
            __builtin_amdgcn_sched_barrier(0);
        }
        mac_keys2<F, true>(acc0, acc1, d0, d1, kb, ka, ((size_t)jk * L + i) * C::N, kb, ka, ((size_t)(jk + 1) * L + i) * C::N, tid, C::T, P);
    }
    if (jk < LK) {                                        // odd number of digit polynomials: the last one alone
        const uint32_t j0 = jk / K, k0 = jk % K;
        load_src_buf<F, LOGN, COMPACT>(C2, j0, tid, d0);
#pragma unroll
        for (int r = 0; r < 32; r++) d0[r] = F::digit(d0[r], k0 * w, w);
        fwd_core<F, LOGN, false, true>(d0, lds, tid, P);
        mac_keys<F>(acc0, acc1, d0, kb, ka, ((size_t)jk * L + i) * C::N, tid, C::T, P);
    }
    finish_pair<F, LOGN, ADD_COMPACT>(acc0, acc1, d0, d1, lds, tid, P, add0, add1, p, c0, c1);
}

// IN_COMPACT / OUT_COMPACT: the accumulator pair is read from / written to compact polynomials (load_A_compact): inside fhe_blind_rotate
// the accumulators stay in that form between the first and the last step of a loop, so a step moves 2 S/8 per limb workgroup in and
// 2 S/8 out instead of 2 S each way (the L workgroups of an accumulator each read ALL its limbs).
template <class F, int LOGN, int MINW = 1, bool IN_COMPACT = false, bool OUT_COMPACT = false>
__global__ void __launch_bounds__(NttCfg<LOGN>::T, MINW)
ntt_extprod2_kernel(char *__restrict__ out0, char *__restrict__ out1, const char *__restrict__ in0, const char *__restrict__ in1,
                    const uint32_t *__restrict__ shifts,
                    const typename F::E *__restrict__ kb0, const typename F::E *__restrict__ ka0,
                    const typename F::E *__restrict__ kb1, const typename F::E *__restrict__ ka1,
                    const Limb<F> *__restrict__ limbs, uint32_t L, uint32_t K, uint32_t w) {
    using C = NttCfg<LOGN>;
    using E = typename F::E;
    __shared__ E lds[2 * C::LDS_ELEMS];
    const uint32_t tid = threadIdx.x, bid = blockIdx.x, full = (gridDim.x / (8 * L)) * (8 * L);
    uint32_t b, i;
    if (bid < full) { const uint32_t s = bid >> 3; b = (bid & 7) + 8 * (s / L); i = s % L; }
    else { b = bid / L; i = bid % L; }
    const uint32_t p = b * L + i;
    const Limb<F> P = limbs[i];
    const uint32_t a = shifts[b] & (2 * C::N - 1);
    const size_t cbytes = (size_t)b * L * (C::N * (IN_COMPACT ? sizeof(E) : 32));
    const TableBuf IN0(in0 + cbytes), IN1(in1 + cbytes);   // descriptors based at the first limb polynomial of this accumulator's two components
    E acc0[32], acc1[32], d0[32], d1[32];
#pragma unroll
    for (int r = 0; r < 32; r++) { acc0[r] = 0; acc1[r] = 0; }
    const uint32_t LK = L * K, G = 2 * LK;                // digit polynomials of component 0, then of component 1: always an even number
    for (uint32_t g = 0; g < G; g += 2) {
        const uint32_t c0i = g / LK, jk0 = g % LK, j0 = jk0 / K, k0 = jk0 % K;
        const uint32_t c1i = (g + 1) / LK, jk1 = (g + 1) % LK, j1 = jk1 / K, k1 = jk1 % K;
        load_monomial_A_buf<F, LOGN, IN_COMPACT>(c1i ? IN1 : IN0, j1, lds, tid, a, limbs[j1].q, d1);
        if (c0i == c1i && j0 == j1) {
#pragma unroll
            for (int r = 0; r < 32; r++) d0[r] = F::digit(d1[r], k0 * w, w);
        } else {
            load_monomial_A_buf<F, LOGN, IN_COMPACT>(c0i ? IN1 : IN0, j0, lds, tid, a, limbs[j0].q, d0);
#pragma unroll
            for (int r = 0; r < 32; r++) d0[r] = F::digit(d0[r], k0 * w, w);
        }
#pragma unroll
        for (int r = 0; r < 32; r++) d1[r] = F::digit(d1[r], k1 * w, w);
        fwd_core2<F, LOGN, true>(d0, d1, lds, tid, P);
        mac_keys2<F>(acc0, acc1, d0, d1, c0i ? kb1 : kb0, c0i ? ka1 : ka0, ((size_t)jk0 * L + i) * C::N, c1i ? kb1 : kb0, c1i ? ka1 : ka0,
                     ((size_t)jk1 * L + i) * C::N, tid, C::T, P);
    }
    finish_pair<F, LOGN, IN_COMPACT, OUT_COMPACT>(acc0, acc1, d0, d1, lds, tid, P, in0, in1, p, out0, out1);
}

}  // namespace fhe_dev
