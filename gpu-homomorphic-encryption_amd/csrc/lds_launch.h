// lds_launch.h -- host-side launch table for the LDS-resident kernels (one translation unit per
// (field, log2 n) instance so the instances compile in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fhe_dev {

enum LdsOp { LDS_FORWARD = 0, LDS_INVERSE = 1, LDS_MULTIPLY = 2, LDS_CT_MULTIPLY = 3, LDS_KEYSWITCH = 4, LDS_EXTPROD = 5,
             // transforms beyond the LDS range (N = 2^(13 + top), top = 1..3; served by the LOGN = 13 instances): the register-only pass over
             // the top stages (r0 = dst, a0 = src) and the sub-transforms of the 2^top blocks (r0 = dst, a0 = src, b0 = second operand)
             LDS_PASS_FWD = 6, LDS_PASS_INV = 7, LDS_SUB_FORWARD = 8, LDS_SUB_INVERSE = 9, LDS_SUB_MULTIPLY = 10 };

// true when the instance runs the tensor product as one fused launch; otherwise LDS_CT_MULTIPLY issues two launches (LdsArgs::ws
// set: NTT(b0), NTT(b1) into the compact workspace, then everything else; 7 transforms) or, without a workspace (testing aid
// FHE_HIP_NO_TWO_LAUNCH_CT=1), multiply(c0), multiply(c2) and the two-product kernel for c1 (three launches, 11 transforms)
constexpr bool lds_ct_fused(int elem_bytes, int log_n) { return elem_bytes == 4 ? log_n <= 14 : log_n <= 13; }   // 1024-thread blocks cap a thread at 128 VGPRs
// the two-launch form exists wherever the one-launch kernel does not, and for every size of the 8-byte residues (whose one-launch
// kernel holds four 64-register arrays and parks 1.1-1.3 KB per lane in scratch when it writes containers); LdsArgs::ws selects it
constexpr bool lds_ct_two_launch(int elem_bytes, int log_n) { return elem_bytes == 8 || !lds_ct_fused(elem_bytes, log_n); }
// key switching: 4-byte residues run one workgroup per (ciphertext, limb); 8-byte residues and 1024-thread blocks (N = 2^15)
// two, one per key half (three live arrays instead of four)
constexpr bool lds_keyswitch_split(int elem_bytes, int log_n) { return elem_bytes == 8 || log_n >= 15; }
// ... unless LdsArgs::joint3 asks for ONE workgroup per limb with both accumulators and the digit polynomial live
// (ntt_keyswitch3_kernel / ntt_extprod3_kernel: half the transforms, ~450-660 bytes per lane parked in scratch at N = 2^14), which exists
// for every LDS-resident size of the 8-byte residues and for 4-byte residues at N = 2^15 (1024-thread workgroups capped at 128 VGPRs:
// relinearisation +38 %, blind rotation +48 % over the split form); the host decides where it is used (use_joint3 in fhe_hip.hip; never under
// FHE_HIP_SPLIT_KEYSWITCH=1, a testing aid)
constexpr bool lds_keyswitch_joint3(int elem_bytes, int log_n) { return (elem_bytes == 8 && log_n <= 14) || (elem_bytes == 4 && log_n == 15); }

// key switching / external product in the one-workgroup-per-limb form: twiddle tables copied into LDS for 4-byte residues up
// to N = 2^13 (exchange buffer + table = 65 KiB per workgroup, still two workgroups per CU).  Interleaved A/B on one MI355X
// (scripts/bench_ab_twiddles.sh): external product N = 8192 +7 %, relinearisation N = 8192 +-0 %; at N = 2^14 (130 KiB, one
// workgroup per CU) it was 1-6 % slower, so that size keeps reading twiddles through L2.
constexpr bool lds_twiddles_in_lds(int elem_bytes, int log_n) { return elem_bytes == 4 && log_n <= 13; }

// key switching / external product with the digit transforms done two at a time (ntt_keyswitch2_kernel / ntt_extprod2_kernel;
// two exchange buffers: 66 KiB at N = 2^13, 132 KiB at N = 2^14) and the two products of a pair sharing one Montgomery reduction.
// Interleaved A/B on one MI355X (scripts/bench_ab_paired.sh) against the one-at-a-time kernels (with LDS twiddles up to N = 2^13):
// relinearisation +7.5 % at N = 8192 and +17 % at N = 16384; external product +7 % at N = 8192, +13 % at N = 16384, +3.5 % at N = 4096.
constexpr bool lds_paired_keyswitch(int elem_bytes, int log_n) { return elem_bytes == 4 && log_n <= 14; }
constexpr bool lds_paired_extprod(int elem_bytes, int log_n) { return elem_bytes == 4 && log_n <= 14; }

// the fused FHEContext::multiply hands c0, c1, c2 from the tensor product to the key switch as compact polynomials wherever the
// key-switch kernel exists in its default one-launch form: paired (4-byte residues up to 2^14) or split (8-byte residues, N = 2^15)
constexpr bool lds_compact_c2(int elem_bytes, int log_n) { return lds_keyswitch_split(elem_bytes, log_n) || lds_paired_keyswitch(elem_bytes, log_n); }

// LDS_MULTIPLY for few polynomials (LdsArgs::small_batch, set by the host while batch x limbs is below the CU count): the 16-per-thread
// latency kernel of ntt_lds_small.hip.h; 4-byte residues only (its 90 preloaded 4-byte twiddles fit the register file, 8-byte ones do not)
// (N = 2^14 would be 1024 threads under the 128-VGPR cap: the preloaded twiddles spill 51-163 VGPRs there, so that size keeps the 32-per-thread kernels)
constexpr bool lds_small_multiply(int elem_bytes, int log_n) { return elem_bytes == 4 && log_n <= 13; }
// ... and for a handful of polynomials one polynomial over four workgroups (ntt_multiply_coop4_kernel)
constexpr bool lds_coop4_multiply(int elem_bytes, int log_n) { return elem_bytes == 4 && (log_n == 13 || log_n == 14); }

struct LdsArgs {
    int op;
    void *r0, *r1, *r2;                  // outputs (forward / inverse: r0 is the in-place buffer)
    const void *a0, *a1, *b0, *b1;       // inputs
    const void *limbs;                   // device array of Limb<F>
    uint32_t L, polys;
    hipStream_t stream;
    const void *kb = nullptr, *ka = nullptr;   // LDS_KEYSWITCH: packed key tables (r0 = c0, r1 = c1, a0 = c2)
    uint32_t K = 0, w = 0;
    // LDS_EXTPROD (fused blind-rotation step): r0, r1 = out pair, a0, a1 = in pair, kb / ka = rows of component 0,
    // kb1 / ka1 = rows of component 1, shifts = device array of per-ciphertext monomial exponents
    const void *kb1 = nullptr, *ka1 = nullptr;
    const uint32_t *shifts = nullptr;
    bool global_twiddles = false;        // testing aid (FHE_HIP_NO_LDS_TWIDDLES=1): run the variant that reads twiddles from L2
    bool single_transforms = false;      // testing aid (FHE_HIP_NO_PAIRED_TRANSFORMS=1): one digit transform at a time
    uint32_t b_polys = 0;                // LDS_MULTIPLY: polynomials behind b0 (0 = as many as the batch; L = one RNS polynomial broadcast over the batch)
    bool square = false;                 // LDS_MULTIPLY: b0 == a0; LDS_CT_MULTIPLY: (b0, b1) == (a0, a1) -- the squaring forms of the kernels
    bool compact_c2 = false;             // fused multiply + relinearise (see lds_compact_c2): LDS_CT_MULTIPLY: r0, r1, r2 are compact workspace polynomials;
                                         // LDS_KEYSWITCH: a0 (c2) and the addends a1 (c0), b0 (c1) are, r0 / r1 are the container outputs
    bool in_compact = false, out_compact = false;   // LDS_EXTPROD (paired kernel only): the accumulator pair a0, a1 / r0, r1 is in compact form
    bool c2_only_compact = false;        // LDS_KEYSWITCH with joint3, container addends: a0 (c2) is a compact polynomial (compacted by the host first)
    bool joint3 = false;                 // LDS_KEYSWITCH where lds_keyswitch_joint3 holds: one workgroup per limb with three live arrays (ntt_keyswitch3_kernel)
    void *ws = nullptr;                  // LDS_CT_MULTIPLY where !lds_ct_fused: 2 * polys * n residues of workspace for the transformed b-side
    uint32_t top = 0;                    // LDS_PASS_* / LDS_SUB_*: number of stages above the 2^13 blocks (log2 n = 13 + top)
    bool rconst = false;                 // LDS_PASS_INV: scale with the constants that also absorb the 2^-W of a fused pointwise product
    // (new members go at the END: objects of the other instances stay layout-compatible during development, scripts/dev_relink.sh)
    bool small_batch = false;            // LDS_MULTIPLY: use the latency kernel where lds_small_multiply holds (never with compact_c2)
    void *coop_ws = nullptr;             // LDS_MULTIPLY, N = 2^13 / 2^14, 4-byte residues, a handful of polynomials: 3 * polys * n residues -- four workgroups per
                                         // polynomial in three dependent launches (ntt_multiply4_*_kernel)
    void *pair_ws = nullptr;             // LDS_KEYSWITCH, paired kernel, few ciphertexts: 2 * polys * ceil(L K / 2) * n residues -- one workgroup per digit PAIR
                                         // (ntt_keyswitch2_part_kernel) and a combining launch instead of one workgroup per (ciphertext, limb)
};

typedef void (*lds_launch_fn)(const LdsArgs &);

// width: 32, 52, 64 or 65 (= F64X, full-range 64-bit).  nullptr when the instance does not exist.
lds_launch_fn lds_lookup(int width, int log_n);

}  // namespace fhe_dev
