// lds_inst.hip -- one (field, LOGN) instance of the LDS-resident kernels.
// Compile with -DFHE_FIELD=F32|F52|F64|F64X -DFHE_LOGN=11..15 (the Makefile lists the instances).
#include <cstdlib>
#include "lds_launch.h"
#include "ntt_lds.hip.h"
#include "ntt_lds_small.hip.h"

#define CAT_(a, b, c) a##b##_##c
#define CAT(a, b, c) CAT_(a, b, c)

namespace fhe_dev {

// the latency kernel where it exists (a template so that the other instances do not instantiate Cfg16 at all); b == a simply loads twice
template <class F, int LOGN>
static bool launch_small_multiply(const LdsArgs &A, const Limb<F> *limbs) {
    if constexpr (lds_small_multiply(sizeof(typename F::E), LOGN)) {
        hipLaunchKernelGGL((ntt16_multiply_kernel<F, LOGN>), dim3(A.polys), dim3(Cfg16<LOGN>::T), 0, A.stream, (char *)A.r0, (const char *)A.a0,
                           (const char *)A.b0, limbs, A.L, A.b_polys ? 1u : 0u);
        return true;
    } else {
        return false;
    }
}

// a handful of polynomials: each one over four workgroups, three dependent launches (ntt_lds_small.hip.h)
template <class F, int LOGN>
static bool launch_coop4_multiply(const LdsArgs &A, const Limb<F> *limbs) {
    using E = typename F::E;
    if constexpr (lds_coop4_multiply(sizeof(E), LOGN)) {
        const dim3 grid(A.polys * 4), cgrid(A.polys * Coop4<F, LOGN>::CWG), block(Coop4<F, LOGN>::T);      // blocks: four per limb polynomial; columns: CWG
        hipLaunchKernelGGL((ntt_multiply4_top_kernel<F, LOGN>), cgrid, block, 0, A.stream, (const char *)A.a0, (const char *)A.b0, (E *)A.coop_ws, limbs, A.L, A.b_polys ? 1u : 0u);
#ifndef FHE_COOP_ONE_GROUP
        hipLaunchKernelGGL((ntt_multiply4_block_kernel<F, LOGN>), grid, dim3(2 * Coop4<F, LOGN>::T), 0, A.stream, (E *)A.coop_ws, limbs, A.L);   // two groups: the forward transforms side by side
#else
        hipLaunchKernelGGL((ntt_multiply4_block_kernel<F, LOGN>), grid, block, 0, A.stream, (E *)A.coop_ws, limbs, A.L);
#endif
        hipLaunchKernelGGL((ntt_multiply4_last_kernel<F, LOGN>), cgrid, block, 0, A.stream, (char *)A.r0, (const E *)A.coop_ws, limbs, A.L);
        return true;
    } else {
        return false;
    }
}
// tensor product with compact outputs of a handful of ciphertexts: four workgroups per (ciphertext, limb), three launches
template <class F, int LOGN>
static bool launch_coop4_ct_multiply(const LdsArgs &A, const Limb<F> *limbs) {
    using E = typename F::E;
    if constexpr (lds_coop4_multiply(sizeof(E), LOGN)) {
        const dim3 block(Coop4<F, LOGN>::T);
        hipLaunchKernelGGL((ntt_ct4_top_kernel<F, LOGN>), dim3(A.polys * Coop4<F, LOGN>::CWG, 4), block, 0, A.stream, (const char *)A.a0, (const char *)A.a1, (const char *)A.b0,
                           (const char *)A.b1, (E *)A.coop_ws, limbs, A.L);
#ifndef FHE_COOP_ONE_GROUP
        if constexpr (LOGN == 13)      // four groups of threads: the four forward transforms side by side, three inverses side by side
            hipLaunchKernelGGL((ntt_ct4_block_kernel<F, LOGN>), dim3(A.polys * 4), dim3(4 * Coop4<F, LOGN>::T), 0, A.stream, (E *)A.coop_ws, limbs, A.L);
        else
#endif
        hipLaunchKernelGGL((ntt_ct4_block1_kernel<F, LOGN>), dim3(A.polys * 4), block, 0, A.stream, (E *)A.coop_ws, limbs, A.L);
        hipLaunchKernelGGL((ntt_ct4_last_kernel<F, LOGN>), dim3(A.polys * Coop4<F, LOGN>::CWG, 3), block, 0, A.stream, (E *)A.r0, (E *)A.r1, (E *)A.r2, (const E *)A.coop_ws, limbs, A.L);
        return true;
    } else {
        return false;
    }
}
// tensor product with compact outputs for few ciphertexts (the first half of the one-call multiply + relinearise)
template <class F, int LOGN>
static bool launch_small_ct_multiply(const LdsArgs &A, const Limb<F> *limbs) {
    using E = typename F::E;
    if constexpr (lds_small_multiply(sizeof(E), LOGN)) {
        hipLaunchKernelGGL((ntt16_ct_multiply_kernel<F, LOGN>), dim3(A.polys), dim3(Cfg16<LOGN>::T), 0, A.stream, (E *)A.r0, (E *)A.r1, (E *)A.r2,
                           (const char *)A.a0, (const char *)A.a1, (const char *)A.b0, (const char *)A.b1, limbs, A.L);
        return true;
    } else {
        return false;
    }
}
// key switch of few ciphertexts: one workgroup per digit pair + a combining launch (ntt_lds_small.hip.h), where the paired kernel exists
template <class F, int LOGN>
static bool launch_split_keyswitch(const LdsArgs &A, const Limb<F> *limbs) {
    using E = typename F::E;
    if constexpr (lds_paired_keyswitch(sizeof(E), LOGN)) {
        const bool c2_compact = A.compact_c2 || A.c2_only_compact;
#ifndef FHE_SPLIT_PAIRED
        if constexpr (lds_small_multiply(sizeof(E), LOGN)) {      // N <= 2^13: one workgroup per DIGIT and per COMPONENT on the 16-per-thread transforms
            const uint32_t LK = A.L * A.K;
            const dim3 b16(Cfg16<LOGN>::T), pgrid(A.polys * LK), cgrid(A.polys, 2);
            E *part0 = (E *)A.pair_ws, *part1 = part0 + (size_t)A.polys * LK * (1u << LOGN);
            if (c2_compact)
                hipLaunchKernelGGL((ntt_keyswitch16_part_kernel<F, LOGN, true>), pgrid, b16, 0, A.stream, part0, part1, (const char *)A.a0, (const char *)nullptr, (const E *)A.kb, (const E *)A.ka,
                                   (const E *)nullptr, (const E *)nullptr, limbs, A.L, A.K, A.w);
            else
                hipLaunchKernelGGL((ntt_keyswitch16_part_kernel<F, LOGN, false>), pgrid, b16, 0, A.stream, part0, part1, (const char *)A.a0, (const char *)nullptr, (const E *)A.kb, (const E *)A.ka,
                                   (const E *)nullptr, (const E *)nullptr, limbs, A.L, A.K, A.w);
            if (A.compact_c2)    // fused multiply + relinearise: the addends are the compact c0 (a1) and c1 (b0)
                hipLaunchKernelGGL((ntt_keyswitch16_comb_kernel<F, LOGN, true>), cgrid, b16, 0, A.stream, (char *)A.r0, (char *)A.r1, (const E *)part0, (const E *)part1,
                                   (const char *)A.a1, (const char *)A.b0, limbs, A.L, LK);
            else                 // in place on the caller's containers
                hipLaunchKernelGGL((ntt_keyswitch16_comb_kernel<F, LOGN, false>), cgrid, b16, 0, A.stream, (char *)A.r0, (char *)A.r1, (const E *)part0, (const E *)part1,
                                   (const char *)A.r0, (const char *)A.r1, limbs, A.L, LK);
            return true;
        }
#endif
        const uint32_t NP = (A.L * A.K + 1) / 2;
        const dim3 block(NttCfg<LOGN>::T), pgrid(A.polys * NP), cgrid(A.polys);
        E *part0 = (E *)A.pair_ws, *part1 = part0 + (size_t)A.polys * NP * (1u << LOGN);
        if (c2_compact)
            hipLaunchKernelGGL((ntt_keyswitch2_part_kernel<F, LOGN, true>), pgrid, block, 0, A.stream, part0, part1, (const char *)A.a0, (const char *)nullptr, (const E *)A.kb, (const E *)A.ka,
                               (const E *)nullptr, (const E *)nullptr, limbs, A.L, A.K, A.w);
        else
            hipLaunchKernelGGL((ntt_keyswitch2_part_kernel<F, LOGN, false>), pgrid, block, 0, A.stream, part0, part1, (const char *)A.a0, (const char *)nullptr, (const E *)A.kb, (const E *)A.ka,
                               (const E *)nullptr, (const E *)nullptr, limbs, A.L, A.K, A.w);
        if (A.compact_c2)    // fused multiply + relinearise: the addends are the compact c0 (a1) and c1 (b0)
            hipLaunchKernelGGL((ntt_keyswitch2_comb_kernel<F, LOGN, true>), cgrid, block, 0, A.stream, (char *)A.r0, (char *)A.r1, (const E *)part0, (const E *)part1,
                               (const char *)A.a1, (const char *)A.b0, limbs, A.L, NP);
        else                 // in place on the caller's containers
            hipLaunchKernelGGL((ntt_keyswitch2_comb_kernel<F, LOGN, false>), cgrid, block, 0, A.stream, (char *)A.r0, (char *)A.r1, (const E *)part0, (const E *)part1,
                               (const char *)A.r0, (const char *)A.r1, limbs, A.L, NP);
        return true;
    } else {
        return false;
    }
}

// external product of a blind-rotation step for few accumulators (N <= 2^13, 4-byte residues): the key-switch launches above with two digit sources -- the
// pre-rotated components b0, b1 (compact) with their RGSW rows -- and the accumulator pair a0, a1 (compact) as addends; r0, r1 compact or containers
template <class F, int LOGN>
static bool launch_split_extprod(const LdsArgs &A, const Limb<F> *limbs) {
    using E = typename F::E;
    if constexpr (lds_small_multiply(sizeof(E), LOGN)) {
        const uint32_t LK = A.L * A.K;
        const dim3 b16(Cfg16<LOGN>::T), pgrid(A.polys * LK, 2), cgrid(A.polys, 2);
        E *part0 = (E *)A.pair_ws, *part1 = part0 + (size_t)A.polys * 2 * LK * (1u << LOGN);
        hipLaunchKernelGGL((ntt_keyswitch16_part_kernel<F, LOGN, true>), pgrid, b16, 0, A.stream, part0, part1, (const char *)A.b0, (const char *)A.b1, (const E *)A.kb, (const E *)A.ka,
                           (const E *)A.kb1, (const E *)A.ka1, limbs, A.L, A.K, A.w);
        if (A.out_compact)
            hipLaunchKernelGGL((ntt_keyswitch16_comb_kernel<F, LOGN, true, true>), cgrid, b16, 0, A.stream, (char *)A.r0, (char *)A.r1, (const E *)part0, (const E *)part1,
                               (const char *)A.a0, (const char *)A.a1, limbs, A.L, 2 * LK);
        else
            hipLaunchKernelGGL((ntt_keyswitch16_comb_kernel<F, LOGN, true, false>), cgrid, b16, 0, A.stream, (char *)A.r0, (char *)A.r1, (const E *)part0, (const E *)part1,
                               (const char *)A.a0, (const char *)A.a1, limbs, A.L, 2 * LK);
        return true;
    } else if constexpr (lds_paired_keyswitch(sizeof(E), LOGN)) {      // N = 2^14: one workgroup per digit PAIR of a component (paired 32-per-thread transform), paired combining launch
        const uint32_t NP = (A.L * A.K + 1) / 2;
        const dim3 block(NttCfg<LOGN>::T), pgrid(A.polys * NP, 2), cgrid(A.polys);
        E *part0 = (E *)A.pair_ws, *part1 = part0 + (size_t)A.polys * 2 * NP * (1u << LOGN);
        hipLaunchKernelGGL((ntt_keyswitch2_part_kernel<F, LOGN, true>), pgrid, block, 0, A.stream, part0, part1, (const char *)A.b0, (const char *)A.b1, (const E *)A.kb, (const E *)A.ka,
                           (const E *)A.kb1, (const E *)A.ka1, limbs, A.L, A.K, A.w);
        if (A.out_compact)
            hipLaunchKernelGGL((ntt_keyswitch2_comb_kernel<F, LOGN, true, true>), cgrid, block, 0, A.stream, (char *)A.r0, (char *)A.r1, (const E *)part0, (const E *)part1,
                               (const char *)A.a0, (const char *)A.a1, limbs, A.L, 2 * NP);
        else
            hipLaunchKernelGGL((ntt_keyswitch2_comb_kernel<F, LOGN, true, false>), cgrid, block, 0, A.stream, (char *)A.r0, (char *)A.r1, (const E *)part0, (const E *)part1,
                               (const char *)A.a0, (const char *)A.a1, limbs, A.L, 2 * NP);
        return true;
    } else {
        return false;
    }
}

void CAT(lds_launch_, FHE_FIELD, FHE_LOGN)(const LdsArgs &A) {
    using F = FHE_FIELD;
    constexpr int LOGN = FHE_LOGN;
    constexpr int MULT_MINW = F::MULT_MINW;
    const dim3 grid(A.polys), block(NttCfg<LOGN>::T);
    const Limb<F> *limbs = (const Limb<F> *)A.limbs;
    if constexpr (LOGN == 13) {          // this instance also serves N = 2^14 .. 2^16 in two passes (ntt_sub_kernel / word_pass_kernel)
        // pass: one lane per column of a 2^13-column block (forward: containers -> compact), two per column (inverse: compact -> containers)
        const dim3 fgrid((1u << 13) >> 8, A.polys), igrid((2u << 13) >> 8, A.polys), sgrid(A.polys << A.top);
        switch (A.op) {
            case LDS_PASS_FWD:
                if (A.top == 3) hipLaunchKernelGGL((word_pass_kernel<F, 3, true>), fgrid, dim3(256), 0, A.stream, A.r0, A.a0, limbs, A.L, 16u, 0u);
                else if (A.top == 2) hipLaunchKernelGGL((word_pass_kernel<F, 2, true>), fgrid, dim3(256), 0, A.stream, A.r0, A.a0, limbs, A.L, 15u, 0u);
                else hipLaunchKernelGGL((word_pass_kernel<F, 1, true>), fgrid, dim3(256), 0, A.stream, A.r0, A.a0, limbs, A.L, 14u, 0u);
                return;
            case LDS_PASS_INV:
                if (A.top == 3) hipLaunchKernelGGL((word_pass_kernel<F, 3, false>), igrid, dim3(256), 0, A.stream, A.r0, A.a0, limbs, A.L, 16u, A.rconst ? 1u : 0u);
                else if (A.top == 2) hipLaunchKernelGGL((word_pass_kernel<F, 2, false>), igrid, dim3(256), 0, A.stream, A.r0, A.a0, limbs, A.L, 15u, A.rconst ? 1u : 0u);
                else hipLaunchKernelGGL((word_pass_kernel<F, 1, false>), igrid, dim3(256), 0, A.stream, A.r0, A.a0, limbs, A.L, 14u, A.rconst ? 1u : 0u);
                return;
            case LDS_SUB_FORWARD:
                hipLaunchKernelGGL((ntt_sub_kernel<F, 13, SUB_FORWARD, MULT_MINW>), sgrid, block, 0, A.stream, (char *)A.r0, (const char *)A.a0, (const char *)nullptr, limbs, A.L, A.top);
                return;
            case LDS_SUB_INVERSE:
                hipLaunchKernelGGL((ntt_sub_kernel<F, 13, SUB_INVERSE, MULT_MINW>), sgrid, block, 0, A.stream, (char *)A.r0, (const char *)A.a0, (const char *)nullptr, limbs, A.L, A.top);
                return;
            case LDS_SUB_MULTIPLY:
                hipLaunchKernelGGL((ntt_sub_kernel<F, 13, SUB_MULTIPLY, MULT_MINW>), sgrid, block, 0, A.stream, (char *)A.r0, (const char *)A.a0, (const char *)A.b0, limbs, A.L, A.top);
                return;
            default: break;
        }
    }
    switch (A.op) {
        default: break;
        case LDS_FORWARD:
            hipLaunchKernelGGL((ntt_forward_kernel<F, LOGN>), grid, block, 0, A.stream, (char *)A.r0, limbs, A.L);
            break;
        case LDS_INVERSE:
            hipLaunchKernelGGL((ntt_inverse_kernel<F, LOGN>), grid, block, 0, A.stream, (char *)A.r0, limbs, A.L);
            break;
        case LDS_MULTIPLY:
            if (A.coop_ws && launch_coop4_multiply<F, LOGN>(A, limbs)) break;     // a handful of polynomials: four workgroups each
            if (A.small_batch && launch_small_multiply<F, LOGN>(A, limbs)) break;   // few polynomials: one workgroup's latency is what counts (ntt_lds_small.hip.h)
            if (A.square)
                hipLaunchKernelGGL((ntt_multiply_kernel<F, LOGN, MULT_MINW, true>), grid, block, 0, A.stream, (char *)A.r0, (const char *)A.a0,
                                   (const char *)A.b0, limbs, A.L, 0u);
            else
                hipLaunchKernelGGL((ntt_multiply_kernel<F, LOGN, MULT_MINW>), grid, block, 0, A.stream, (char *)A.r0, (const char *)A.a0,
                                   (const char *)A.b0, limbs, A.L, A.b_polys ? 1u : 0u);
            break;
        case LDS_CT_MULTIPLY:
            if (A.coop_ws && A.compact_c2 && !A.ws && launch_coop4_ct_multiply<F, LOGN>(A, limbs)) break;   // a handful of ciphertexts: four workgroups per limb polynomial
            if (A.small_batch && A.compact_c2 && !A.ws && launch_small_ct_multiply<F, LOGN>(A, limbs)) break;   // few ciphertexts (ntt_lds_small.hip.h)
            if (A.ws) {                  // two launches: the b-side transforms into the workspace, then one workgroup per (ciphertext, limb) does the rest
                if constexpr (lds_ct_two_launch(sizeof(typename F::E), LOGN)) {
                    using E = typename F::E;
                    E *w0 = (E *)A.ws, *w1 = w0 + (size_t)A.polys * (1u << LOGN);
                    hipLaunchKernelGGL((ntt_forward_compact_kernel<F, LOGN, MULT_MINW>), dim3(A.polys, 2), block, 0, A.stream, w0, w1,
                                       (const char *)A.b0, (const char *)A.b1, limbs, A.L);
                    if (A.compact_c2)
                        hipLaunchKernelGGL((ntt_ct_a_kernel<F, LOGN, MULT_MINW, true>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1, (char *)A.r2,
                                           (const char *)A.a0, (const char *)A.a1, (const E *)w0, (const E *)w1, limbs, A.L);
                    else
                        hipLaunchKernelGGL((ntt_ct_a_kernel<F, LOGN, MULT_MINW, false>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1, (char *)A.r2,
                                           (const char *)A.a0, (const char *)A.a1, (const E *)w0, (const E *)w1, limbs, A.L);
                }
            } else if constexpr (lds_ct_fused(sizeof(typename F::E), LOGN)) {
                if (A.compact_c2) {
                    hipLaunchKernelGGL((ntt_ct_multiply_kernel<F, LOGN, false, true>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                       (char *)A.r2, (const char *)A.a0, (const char *)A.a1, (const char *)A.b0, (const char *)A.b1,
                                       limbs, A.L);
                } else if (A.square)
                    hipLaunchKernelGGL((ntt_ct_multiply_kernel<F, LOGN, true>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                       (char *)A.r2, (const char *)A.a0, (const char *)A.a1, (const char *)A.b0, (const char *)A.b1,
                                       limbs, A.L);
                else
                    hipLaunchKernelGGL((ntt_ct_multiply_kernel<F, LOGN>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                       (char *)A.r2, (const char *)A.a0, (const char *)A.a1, (const char *)A.b0, (const char *)A.b1,
                                       limbs, A.L);
            } else if (A.compact_c2) {   // three launches with compact outputs
                hipLaunchKernelGGL((ntt_multiply_kernel<F, LOGN, MULT_MINW, false, true>), grid, block, 0, A.stream, (char *)A.r0,
                                   (const char *)A.a0, (const char *)A.b0, limbs, A.L, 0u);
                hipLaunchKernelGGL((ntt_multiply_kernel<F, LOGN, MULT_MINW, false, true>), grid, block, 0, A.stream, (char *)A.r2,
                                   (const char *)A.a1, (const char *)A.b1, limbs, A.L, 0u);
                hipLaunchKernelGGL((ntt_mac2_kernel<F, LOGN, MULT_MINW, true>), grid, block, 0, A.stream, (char *)A.r1, (const char *)A.a0,
                                   (const char *)A.b1, (const char *)A.a1, (const char *)A.b0, limbs, A.L);
            } else {   // four transformed operands exceed the register file: c0, c2 by the fused multiply, c1 by the two-product kernel
                hipLaunchKernelGGL((ntt_multiply_kernel<F, LOGN, MULT_MINW>), grid, block, 0, A.stream, (char *)A.r0,
                                   (const char *)A.a0, (const char *)A.b0, limbs, A.L, 0u);
                hipLaunchKernelGGL((ntt_multiply_kernel<F, LOGN, MULT_MINW>), grid, block, 0, A.stream, (char *)A.r2,
                                   (const char *)A.a1, (const char *)A.b1, limbs, A.L, 0u);
                hipLaunchKernelGGL((ntt_mac2_kernel<F, LOGN, MULT_MINW>), grid, block, 0, A.stream, (char *)A.r1, (const char *)A.a0,
                                   (const char *)A.b1, (const char *)A.a1, (const char *)A.b0, limbs, A.L);
            }
            break;
        case LDS_KEYSWITCH: {
            using E = typename F::E;
            if (A.pair_ws && !A.single_transforms && !A.joint3 && launch_split_keyswitch<F, LOGN>(A, limbs)) break;   // few ciphertexts
            if (A.compact_c2) {          // the host sets it only where lds_compact_c2 holds and the default kernels are selected
                if constexpr (lds_compact_c2(sizeof(E), LOGN)) {
                    if constexpr (lds_keyswitch_split(sizeof(E), LOGN)) {
                        if (A.joint3) {
                            if constexpr (lds_keyswitch_joint3(sizeof(E), LOGN))
                                hipLaunchKernelGGL((ntt_keyswitch3_kernel<F, LOGN, 2, true>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                                   (const char *)A.a0, (const char *)A.a1, (const char *)A.b0, (const E *)A.kb, (const E *)A.ka, limbs, A.L, A.K, A.w);
                        } else
                        hipLaunchKernelGGL((ntt_keyswitch_kernel<F, LOGN, 2, true, false, true>), dim3(A.polys * 2), block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                           (const char *)A.a0, (const char *)A.a1, (const char *)A.b0, (const E *)A.kb, (const E *)A.ka, limbs, A.L, A.K, A.w);
                    }
                    else
                        hipLaunchKernelGGL((ntt_keyswitch2_kernel<F, LOGN, 2, true>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                           (const char *)A.a0, (const char *)A.a1, (const char *)A.b0, (const E *)A.kb, (const E *)A.ka, limbs, A.L, A.K, A.w);
                }
            } else if constexpr (lds_keyswitch_split(sizeof(E), LOGN)) {
                if (A.joint3) {
                    if constexpr (lds_keyswitch_joint3(sizeof(E), LOGN)) {
                        if (A.c2_only_compact)
                            hipLaunchKernelGGL((ntt_keyswitch3_kernel<F, LOGN, 2, true, false>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                               (const char *)A.a0, (const char *)A.r0, (const char *)A.r1, (const E *)A.kb, (const E *)A.ka, limbs, A.L, A.K, A.w);
                        else
                            hipLaunchKernelGGL((ntt_keyswitch3_kernel<F, LOGN, 2, false>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                               (const char *)A.a0, (const char *)A.r0, (const char *)A.r1, (const E *)A.kb, (const E *)A.ka, limbs, A.L, A.K, A.w);
                    }
                } else
                    hipLaunchKernelGGL((ntt_keyswitch_kernel<F, LOGN, 2, true>), dim3(A.polys * 2), block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                       (const char *)A.a0, (const char *)A.r0, (const char *)A.r1, (const E *)A.kb, (const E *)A.ka, limbs, A.L, A.K, A.w);
            } else if (lds_paired_keyswitch(sizeof(E), LOGN) && !A.single_transforms) {
                if constexpr (lds_paired_keyswitch(sizeof(E), LOGN)) {
                    if (A.c2_only_compact)   // stand-alone relinearisation: c2 compacted by the host first, addends are the caller's containers
                        hipLaunchKernelGGL((ntt_keyswitch2_kernel<F, LOGN, 2, true, false>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                           (const char *)A.a0, (const char *)A.r0, (const char *)A.r1, (const E *)A.kb, (const E *)A.ka, limbs, A.L, A.K, A.w);
                    else
                    hipLaunchKernelGGL((ntt_keyswitch2_kernel<F, LOGN, 2>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                       (const char *)A.a0, (const char *)A.r0, (const char *)A.r1, (const E *)A.kb, (const E *)A.ka, limbs, A.L, A.K, A.w);
                }
            } else {
                if (lds_twiddles_in_lds(sizeof(E), LOGN) && !A.global_twiddles) {
                    if constexpr (lds_twiddles_in_lds(sizeof(E), LOGN))
                        hipLaunchKernelGGL((ntt_keyswitch_kernel<F, LOGN, 2, false, true>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                           (const char *)A.a0, (const char *)A.r0, (const char *)A.r1, (const E *)A.kb, (const E *)A.ka, limbs, A.L, A.K, A.w);
                } else {
                    hipLaunchKernelGGL((ntt_keyswitch_kernel<F, LOGN, 2, false>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                       (const char *)A.a0, (const char *)A.r0, (const char *)A.r1, (const E *)A.kb, (const E *)A.ka, limbs, A.L, A.K, A.w);
                }
            }
            break;
        }
        case LDS_EXTPROD: {
            using E = typename F::E;
            if (A.pair_ws && A.b0 && A.in_compact && launch_split_extprod<F, LOGN>(A, limbs)) break;      // few accumulators: one workgroup per digit / per component
            if constexpr (lds_keyswitch_split(sizeof(E), LOGN)) {
                if (A.joint3) {
                    if constexpr (lds_keyswitch_joint3(sizeof(E), LOGN)) {
#define EXTPROD3(IC, OC, PR) hipLaunchKernelGGL((ntt_extprod3_kernel<F, LOGN, 2, IC, OC, PR>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1, \
                                       (const char *)A.a0, (const char *)A.a1, (const char *)A.b0, (const char *)A.b1, A.shifts, (const E *)A.kb, (const E *)A.ka, \
                                       (const E *)A.kb1, (const E *)A.ka1, limbs, A.L, A.K, A.w)
                        if (A.b0) {                           // pre-rotated digit sources (b0, b1): the loop form of fhe_blind_rotate, always compact input
                            if (A.out_compact) EXTPROD3(true, true, true); else EXTPROD3(true, false, true);
                        }
                        else if (A.in_compact && A.out_compact) EXTPROD3(true, true, false);
                        else if (A.in_compact) EXTPROD3(true, false, false);
                        else EXTPROD3(false, false, false);   // (container input with compact output is never asked for: the host compacts first)
#undef EXTPROD3
                    }
                } else
                hipLaunchKernelGGL((ntt_extprod_kernel<F, LOGN, 2, true>), dim3(A.polys * 2), block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                   (const char *)A.a0, (const char *)A.a1, A.shifts, (const E *)A.kb, (const E *)A.ka, (const E *)A.kb1,
                                   (const E *)A.ka1, limbs, A.L, A.K, A.w);
            } else if (lds_paired_extprod(sizeof(E), LOGN) && !A.single_transforms) {
                if constexpr (lds_paired_extprod(sizeof(E), LOGN)) {
#define EXTPROD2(IC, OC) hipLaunchKernelGGL((ntt_extprod2_kernel<F, LOGN, 2, IC, OC>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1, \
                                       (const char *)A.a0, (const char *)A.a1, A.shifts, (const E *)A.kb, (const E *)A.ka, (const E *)A.kb1, \
                                       (const E *)A.ka1, limbs, A.L, A.K, A.w)
                    if (A.in_compact && A.out_compact) EXTPROD2(true, true);
                    else if (A.in_compact) EXTPROD2(true, false);
                    else if (A.out_compact) EXTPROD2(false, true);
                    else EXTPROD2(false, false);
#undef EXTPROD2
                }
            } else {
                if (lds_twiddles_in_lds(sizeof(E), LOGN) && !A.global_twiddles) {
                    if constexpr (lds_twiddles_in_lds(sizeof(E), LOGN))
                        hipLaunchKernelGGL((ntt_extprod_kernel<F, LOGN, 2, false, true>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                           (const char *)A.a0, (const char *)A.a1, A.shifts, (const E *)A.kb, (const E *)A.ka, (const E *)A.kb1,
                                           (const E *)A.ka1, limbs, A.L, A.K, A.w);
                } else {
                    hipLaunchKernelGGL((ntt_extprod_kernel<F, LOGN, 2, false>), grid, block, 0, A.stream, (char *)A.r0, (char *)A.r1,
                                       (const char *)A.a0, (const char *)A.a1, A.shifts, (const E *)A.kb, (const E *)A.ka, (const E *)A.kb1,
                                       (const E *)A.ka1, limbs, A.L, A.K, A.w);
                }
            }
            break;
        }
    }
}

}  // namespace fhe_dev

#if defined(FHE_STAMPS) && FHE_LOGN == 13
// diagnostic build only: the s_memtime stamps of the last ntt16_multiply_kernel launch of this instance
extern "C" int fhe_debug_stamps16(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fhe_dev::g_stamps16), 16 * sizeof(unsigned long long)); }
#endif
