/*
 * fhe_hip.h -- C ABI of the MI355X (gfx950) RNS-NTT polynomial-multiply engine.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (codebasecomprehension987/gpu-homomorphic-encryption) has no FFI layer of its own: its boundary is
 * the C++ class surface fhe::NTTEngine / RNS_NTTEngine / PolynomialOps / FHEContext::multiply.
 * Every entry point below names the reference interface it replaces (file:line, relative to the
 * reference root).  The C++ mirror of those classes lives in include/fhe/ and is a header-only
 * wrapper over this ABI; INTEGRATION.md shows the binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - 256-bit values are the reference's `uint256_t` (include/bigint.cuh:9-11): 4 x uint64_t,
 *     little-endian limbs, 32 bytes, 8-byte aligned.  Host-side moduli are passed as `const uint64_t[4]`.
 *   - `d_*` arguments are raw DEVICE pointers owned by the caller (hipMalloc / fhe_hip_malloc /
 *     torch tensor storage), exactly as the reference takes cudaMalloc'd pointers (src/ntt.cu:30-75).
 *   - Polynomial data is limb-major `[batch][L][n]` containers (src/ntt.cu:161).
 *   - Coefficients handed to the NTT entry points must be canonical residues (< q_limb).
 *   - Work is enqueued on the handle's stream and NOT synchronised (callers sync, as the
 *     reference's callers do: tests/test_fhe.cu:88,97,155), unless env FHE_HIP_SYNC=1.
 *   - Every function returns 0 on success or a negative fhe_status; fhe_hip_last_error() gives the
 *     message of the calling thread's last failure.  (The reference reports no errors at all.)
 *   - A handle is bound to the device that was current at creation and is not thread-safe; distinct
 *     handles may be used from distinct threads (docs/API_REFERENCE.md:600-606).
 *   - There is NO CPU fallback: without a usable HIP device every compute entry point fails with
 *     FHE_ERR_NO_DEVICE.
 */
#ifndef FHE_HIP_H
#define FHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FHE_HIP_ABI_VERSION 1

typedef enum fhe_status {
    FHE_OK = 0,
    FHE_ERR_INVALID_ARG = -1,   /* null pointer, n not a power of two, batch == 0 ... */
    FHE_ERR_BAD_MODULUS = -2,   /* even, >= 2^255, not prime, or q != 1 (mod 2n) */
    FHE_ERR_NO_DEVICE = -3,     /* no HIP device / HIP runtime failure at init */
    FHE_ERR_HIP = -4,           /* a HIP call failed; message holds hipGetErrorString */
    FHE_ERR_UNSUPPORTED = -5,   /* size outside what the kernels were built for */
    FHE_ERR_NONCANONICAL = -6   /* fhe_*_check found coefficients >= q in a fast-path buffer */
} fhe_status;

/* Which kernel family a modulus set selected (reported, never chosen by the caller). */
typedef enum fhe_width_class {
    FHE_WIDTH_32 = 1,    /* every q < 2^30 : 32-bit lazy Shoup butterflies, whole NTT in LDS */
    FHE_WIDTH_64 = 2,    /* every q < 2^62 : 64-bit lazy butterflies, whole NTT in LDS */
    FHE_WIDTH_52 = 3,    /* every q < 2^43 : residues as exact integers in doubles, FP64 FMA butterflies, whole NTT in LDS */
    FHE_WIDTH_256 = 4,   /* anything else < 2^255 : full 4x64-bit Montgomery (R = 2^256), multi-pass */
    FHE_WIDTH_64X = 5    /* every q < 2^64 (some q >= 2^62): 64-bit canonical butterflies (carry-aware add / sub, one-word
                            Montgomery twiddles), whole NTT in LDS */
} fhe_width_class;

typedef struct fhe_ntt fhe_ntt_t;          /* replaces fhe::NTTEngine      (include/ntt.cuh:72-103)  */
typedef struct fhe_rns_ntt fhe_rns_ntt_t;  /* replaces fhe::RNS_NTTEngine  (include/ntt.cuh:106-137) */

/* ---- library / device plumbing (the reference calls the CUDA runtime directly) ------------- */
int fhe_hip_abi_version(void);
const char *fhe_hip_last_error(void);
int fhe_hip_device_count(int *count);
int fhe_hip_set_device(int device);
int fhe_hip_get_device(int *device);
int fhe_hip_device_name(char *buf, size_t buflen);            /* gcnArchName + marketing name */
int fhe_hip_malloc(void **d_ptr, size_t bytes);               /* cudaMalloc   (src/polynomial.cu:8) */
int fhe_hip_free(void *d_ptr);                                /* cudaFree     (src/polynomial.cu:13) */
int fhe_hip_memset(void *d_ptr, int value, size_t bytes);     /* cudaMemset   (src/polynomial.cu:9) */
int fhe_hip_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes);
int fhe_hip_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes);
int fhe_hip_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes);
int fhe_hip_sync(void);                                       /* cudaDeviceSynchronize (tests/test_fhe.cu:88) */

/* ---- host-side parameter maths ------------------------------------------------------------ */
/* compute_montgomery_inverse (src/bigint.cu:23-40): inv[0] = -q^-1 mod 2^64 by 6 Newton steps,
 * inv[1..3] = 0.  Literal, including the deterministic garbage for even q. */
int fhe_montgomery_inverse(const uint64_t q[4], uint64_t inv[4]);
/* compute_montgomery_params (src/bigint.cu:42-55) with r_squared actually computed. */
int fhe_montgomery_params(const uint64_t q[4], uint64_t r_squared[4], uint64_t inv[4]);
/* find_ntt_prime / generate_rns_primes (include/rns.cuh:139-149; src/rns.cu:183-209 are stubs):
 * the `count` smallest primes >= 2^(bits-1) with q = 1 (mod 2n); bits <= 64. */
int fhe_find_ntt_primes(uint32_t bits, uint32_t n, uint32_t count, uint64_t *primes_out);
/* the same search for moduli of any width the engine accepts (bits <= 255), as 4 x u64 little-endian containers */
int fhe_find_ntt_primes_wide(uint32_t bits, uint32_t n, uint32_t count, uint64_t (*primes_out)[4]);
/* find_primitive_root (src/ntt.cu:110-114 is a stub): primitive 2n-th root of unity psi, chosen as
 * the first x^((q-1)/2n), x = 2,3,..., whose n-th power is q-1. */
int fhe_find_psi(uint32_t n, const uint64_t q[4], uint64_t psi[4]);

/* ---- element-wise 256-bit modular kernels, literal reference semantics ---------------------- */
/* batch_mod_add_kernel (src/bigint.cu:171-184) / poly_add_kernel (src/polynomial.cu:70-82):
 * r[i] = add_mod(a[i], b[i], q)  (include/bigint.cuh:27-48).  stream may be NULL (default stream). */
int fhe_u256_add_mod(void *d_r, const void *d_a, const void *d_b, const uint64_t q[4], size_t count, void *stream);
/* batch_mod_sub_kernel (src/bigint.cu:186-199) / poly_sub_kernel (src/polynomial.cu:84-96). */
int fhe_u256_sub_mod(void *d_r, const void *d_a, const void *d_b, const uint64_t q[4], size_t count, void *stream);
/* batch_mod_mul_kernel (src/bigint.cu:201-214) / ntt_pointwise_mul_kernel (kernels/ntt_kernels.cu:124-137):
 * r[i] = mul_mod_montgomery(a[i], b[i], q, inv)  (include/bigint.cuh:76-140); only inv0 = inv.limbs[0] is used. */
int fhe_u256_mont_mul(void *d_r, const void *d_a, const void *d_b, const uint64_t q[4], uint64_t inv0, size_t count, void *stream);
/* poly_mul_scalar_kernel (src/polynomial.cu:98-111): r[i] = mul_mod_montgomery(a[i], scalar, q, inv). */
int fhe_u256_mont_mul_scalar(void *d_r, const void *d_a, const uint64_t scalar[4], const uint64_t q[4], uint64_t inv0, size_t count, void *stream);

/* ---- the reference's transform kernels AS WRITTEN (L1 parity) ---------------------------------------------- */
/* ntt_forward_optimized_kernel (kernels/ntt_kernels.cu:7-62) exactly as NTTEngine::forward launches it (one block of n threads,
 * src/ntt.cu:30-40, WITHOUT the out-of-bounds bit_reverse_kernel): stage schedule log_n = popc(n-1)+1, pairs
 * (k*2m + j, k*2m + j + m) where the second index < n, twiddle index j << (log_n - stage - 1), caller-supplied table of n
 * containers (the reference fills it with the placeholders [1, 1, 2, 3, ...], src/ntt.cu:86-97), literal primitives.  This is
 * what the reference's source computes on the data it is given -- not an NTT with those tables; fhe_ntt_forward is the real
 * transform.  [batch][n] polynomials, one workgroup each; n a power of two up to 65536 (the reference itself cannot launch n > 1024). */
int fhe_ref_forward_kernel_literal(void *d_data, const void *d_twiddles, const uint64_t q[4], uint64_t inv0, uint32_t n, uint32_t batch, void *stream);
/* ntt_inverse_optimized_kernel (kernels/ntt_kernels.cu:65-121): the same pairs in reverse stage order, Gentleman-Sande form,
 * then every element mont(x, n_inv). */
int fhe_ref_inverse_kernel_literal(void *d_data, const void *d_inv_twiddles, const uint64_t q[4], uint64_t inv0, const uint64_t n_inv[4],
                                   uint32_t n, uint32_t batch, void *stream);

/* ntt_stockham_kernel (kernels/ntt_kernels.cu:213-243, never launched by the reference): one out-of-place butterfly stage,
 * out[i1] = in[i1] + mont(in[i2], tw[j * n / 2m]), out[i2] = in[i1] - ..., i1 = k*2m + j, i2 = i1 + m, m = 2^stage.  The n/2
 * in-bounds butterflies are run (as written the kernel also indexes past the arrays for idx >= n/2: undefined). */
int fhe_ref_stockham_stage_literal(void *d_output, const void *d_input, const void *d_twiddles, const uint64_t q[4], uint64_t inv0, uint32_t n,
                                   uint32_t stage, uint32_t batch, void *stream);

/* bit_reverse_kernel's intent (kernels/ntt_kernels.cu:140-161): in-place bit-reversal permutation of every polynomial of a
 * [batch][n] buffer over log2(n) bits (the reference's popc(n-1)+1 bits index out of bounds).  The engine's transforms need no
 * such pass (DIF/DIT pairing); this entry converts between natural order and the order fhe_ntt_forward leaves its values in. */
int fhe_bit_reverse(void *d_data, uint32_t n, uint32_t batch, void *stream);

/* ---- single-modulus engine: fhe::NTTEngine -------------------------------------------------- */
/* NTTEngine::NTTEngine(n, modulus) (src/ntt.cu:7-22) + precompute_twiddle_factors (:77-107), with the
 * root / inverse / table placeholders replaced by real values.  q prime, q = 1 (mod 2n), q < 2^255,
 * 8 <= n <= 65536 a power of two. */
int fhe_ntt_create(fhe_ntt_t **out, uint32_t n, const uint64_t q[4]);
int fhe_ntt_destroy(fhe_ntt_t *h);                                            /* ~NTTEngine (src/ntt.cu:24-28) */
int fhe_ntt_set_stream(fhe_ntt_t *h, void *stream);   /* adopt a caller stream (e.g. torch's); NULL = back to the private one */
int fhe_ntt_width_class(const fhe_ntt_t *h);
/* NTTEngine::forward (src/ntt.cu:30-40) / forward_batch (include/ntt.cuh:87): in place, `batch`
 * polynomials contiguous [batch][n]; natural order in, merged-CT (bit-reversed) order out. */
int fhe_ntt_forward(fhe_ntt_t *h, void *d_data, uint32_t batch);
/* NTTEngine::inverse (src/ntt.cu:42-47) / inverse_batch (include/ntt.cuh:88): exact inverse of
 * forward including the n^-1 scaling (kernels/ntt_kernels.cu:117-120). */
int fhe_ntt_inverse(fhe_ntt_t *h, void *d_data, uint32_t batch);
/* ntt_pointwise_mul_kernel's intent (kernels/ntt_kernels.cu:124-137): r = a .* b mod q, plain product. */
int fhe_ntt_pointwise(fhe_ntt_t *h, void *d_r, const void *d_a, const void *d_b, uint32_t batch);
/* NTTEngine::multiply (src/ntt.cu:49-75): r = a (*) b mod (x^n + 1, q); d_a, d_b are not modified unless d_r aliases
 * one of them, which is allowed (in-place product, squaring with d_a == d_b).  One fused launch on the word-sized paths; with
 * d_a == d_b the squaring form of the kernel runs (one load, one forward transform: 2*S bytes of traffic instead of 3*S). */
int fhe_ntt_multiply(fhe_ntt_t *h, void *d_r, const void *d_a, const void *d_b, uint32_t batch);

/* ---- RNS engine: fhe::RNS_NTTEngine ---------------------------------------------------------- */
/* RNS_NTTEngine::RNS_NTTEngine(n, rns_moduli, num_primes) (src/ntt.cu:122-145): moduli are copied. */
int fhe_rns_ntt_create(fhe_rns_ntt_t **out, uint32_t n, const uint64_t (*moduli)[4], uint32_t num_primes);
/* RNSContext::RNSContext(primes) (include/rns.cuh:27-66, src/rns.cu:6-29): an RNS base WITHOUT a ring.  The handle is an engine of
 * degree n = 1: its buffers [batch][L][1] are exactly RNSContext's interleaved [count][num_primes] layout (src/rns.cu:103-104),
 * `batch` is the reference's `count`, and every container-level entry point (fhe_rns_to_rns, fhe_rns_from_rns, fhe_rns_poly_add /
 * sub, fhe_rns_ntt_pointwise, fhe_rns_mul_mont_literal, fhe_rns_rescale_drop_last, fhe_rns_fast_base_convert, the samplers) works on
 * it.  Primes: pairwise distinct odd primes < 2^255 (no congruence condition). */
int fhe_rns_base_create(fhe_rns_ntt_t **out, const uint64_t (*primes)[4], uint32_t num_primes);
int fhe_rns_ntt_destroy(fhe_rns_ntt_t *h);                                    /* src/ntt.cu:147-156 */
int fhe_rns_ntt_set_stream(fhe_rns_ntt_t *h, void *stream);
int fhe_rns_ntt_width_class(const fhe_rns_ntt_t *h);
/* Pre-sizes the library-owned workspaces for calls of up to `batch` units (fhe_ct_multiply_relin, fhe_blind_rotate, the general
 * paths), so that later calls never allocate: required before capturing such calls into a hipGraph, optional otherwise (the
 * workspaces grow on first use).  No counterpart in the reference, which mallocs and frees inside every multiply (src/ntt.cu:51-74).
 * The key-switch workspaces depend on the digit count: import the key sets BEFORE reserving.  fhe_rns_ntt_workspace_bytes reports what
 * the engine holds at the moment (device bytes in its three workspaces; tables and key sets are not counted). */
int fhe_rns_ntt_reserve(fhe_rns_ntt_t *h, uint32_t batch);
int fhe_rns_ntt_workspace_bytes(const fhe_rns_ntt_t *h, uint64_t *bytes);
/* forward_rns / inverse_rns (src/ntt.cu:158-171): data [batch][L][n]; one launch for all limbs. */
int fhe_rns_ntt_forward(fhe_rns_ntt_t *h, void *d_data, uint32_t batch);
int fhe_rns_ntt_inverse(fhe_rns_ntt_t *h, void *d_data, uint32_t batch);
int fhe_rns_ntt_pointwise(fhe_rns_ntt_t *h, void *d_r, const void *d_a, const void *d_b, uint32_t batch);
/* multiply_rns (include/ntt.cuh:124-126; declared, never defined in the reference). */
int fhe_rns_ntt_multiply(fhe_rns_ntt_t *h, void *d_r, const void *d_a, const void *d_b, uint32_t batch);
/* The same product with ONE polynomial d_b_one ([L][n]) multiplied into every element of a batch d_a ([batch][L][n]) -- what
 * FHEContext does with a key or a plaintext (pk0 * u, c_i * pt: src/fhe.cu:160-166, include/fhe.cuh:104).  The shared operand is served
 * from L2 after its first use: 2*S bytes of HBM traffic per product instead of 3*S.  d_r may alias d_a, not d_b_one. */
int fhe_rns_ntt_multiply_bcast(fhe_rns_ntt_t *h, void *d_r, const void *d_a, const void *d_b_one, uint32_t batch);
/* PolynomialOps::add / sub over RNS polynomials (src/polynomial.cu:36-52), per-limb moduli. */
int fhe_rns_poly_add(fhe_rns_ntt_t *h, void *d_r, const void *d_a, const void *d_b, uint32_t batch);
int fhe_rns_poly_sub(fhe_rns_ntt_t *h, void *d_r, const void *d_a, const void *d_b, uint32_t batch);
/* rns_mul_kernel / RNSContext::mul_rns (src/rns.cu:84-91, :160-181), LITERAL: r = mul_mod_montgomery(a, b, q_l, inv_l) per limb, i.e. the
 * product carries R^-1 = 2^-256 exactly as in the reference (fhe_rns_ntt_pointwise is the plain product).  Full-width handles only. */
int fhe_rns_mul_mont_literal(fhe_rns_ntt_t *h, void *d_r, const void *d_a, const void *d_b, uint32_t batch);
/* FHEContext::multiply tensor product (src/fhe.cu:199-218), relinearisation excluded (:220 is a stub):
 * c0 = a0*b0, c1 = a0*b1 + a1*b0, c2 = a1*b1.  4 forward + 3 inverse transforms instead of the
 * reference's 8 + 4 (results identical: modular arithmetic is exact). */
int fhe_ct_multiply(fhe_rns_ntt_t *h, void *d_c0, void *d_c1, void *d_c2,
                    const void *d_a0, const void *d_a1, const void *d_b0, const void *d_b1, uint32_t batch);
/* ---- RNS entry / exit ------------------------------------------------------------------------------ */
/* RNS_NTTEngine::to_rns (include/ntt.cuh:114; RNSContext::to_rns src/rns.cu:56-61, kernel :93-115 is a placeholder that
 * copies the value): d_rns[b][l][x] = d_values[b][x] mod q_l for ANY 256-bit value; d_values is [batch][n], d_rns is
 * [batch][L][n] (limb-major like every other buffer of this engine). */
int fhe_rns_to_rns(fhe_rns_ntt_t *h, void *d_rns, const void *d_values, uint32_t batch);
/* RNS_NTTEngine::from_rns (include/ntt.cuh:117; from_rns_crt_kernel src/rns.cu:117-141 writes zero): Chinese remainder
 * reconstruction into [0, Q), Q = prod q_l.  Needs Q < 2^255 (FHE_ERR_UNSUPPORTED otherwise). */
int fhe_rns_from_rns(fhe_rns_ntt_t *h, void *d_values, const void *d_rns, uint32_t batch);

/* RNSContext::mod_switch_rns / rns_mod_switch_kernel (include/rns.cuh:44,128-136), FHEContext::mod_switch_to_next
 * (include/fhe.cuh:109), poly_mod_switch_kernel (include/polynomial.cuh:96-103) -- all undefined in the reference:
 * drop the last prime with rounding, d_out[b][l][x] = round(C / q_last) mod q_l for l < L-1.  d_in is [batch][L][n],
 * d_out is [batch][L-1][n] (the layout of an engine built on the first L-1 primes).  Needs L >= 2. */
int fhe_rns_rescale_drop_last(fhe_rns_ntt_t *h, void *d_out, const void *d_in, uint32_t batch);

/* Fast base conversion (Bajard et al.) -- RNSContext::base_extend / fast_base_conversion_kernel (include/rns.cuh:47-48,
 * 116-125, undefined in the reference): d_out[b][j][x] = sum_i [x_i (Q/q_i)^-1]_{q_i} (Q/q_i) mod p_j for the primes p_j of
 * `target` (an engine of the same degree).  The value is that of X + alpha*Q, 0 <= alpha < L (inexact by design of the
 * method; the computation itself is deterministic).  d_in: [batch][L][n] in src's basis, d_out: [batch][L'][n] in target's. */
int fhe_rns_fast_base_convert(fhe_rns_ntt_t *src, fhe_rns_ntt_t *target, void *d_out, const void *d_in, uint32_t batch);

/* ---- relinearisation / key switching ------------------------------------------ */
/* RelinKeys (include/fhe.cuh:52-55) as produced by FHEContext::relinkey_gen (src/fhe.cu:76-111):
 * key pairs (b, a) with b = -a*s + e + g*s^2, one per decomposition level.  In the RNS representation every residue
 * polynomial c2 mod q_j is decomposed into K = ceil(bits(q_max) / decomp_bits) base-2^w digit polynomials, so there are
 * L*K levels; level j*K + k carries g = 2^(k*w) in limb j and g = 0 in the other limbs.  For L = 1 this is the
 * reference's decomposition of a single-modulus ciphertext. */
typedef struct fhe_relin_keys fhe_relin_keys_t;
int fhe_relin_num_digits(const fhe_rns_ntt_t *h, uint32_t decomp_bits, uint32_t *digits_per_limb);
/* d_keys_b / d_keys_a: host arrays of num_keys (= L*K) device pointers to [L][n] polynomials in coefficient form.
 * The keys are copied, transformed to the NTT domain once and kept inside the returned object. */
int fhe_relin_keys_create(fhe_rns_ntt_t *h, fhe_relin_keys_t **out, uint32_t decomp_bits,
                          const void *const *d_keys_b, const void *const *d_keys_a, uint32_t num_keys);
int fhe_relin_keys_destroy(fhe_relin_keys_t *rk);
/* FHEContext::relinearize (src/fhe.cu:226-235 is a stub that drops c2; algorithm: docs/ARCHITECTURE.md:319-326):
 * c0 += sum_{j,k} D_{j,k} * b_{j,k},  c1 += sum_{j,k} D_{j,k} * a_{j,k}  with D_{j,k} the digit polynomials of c2.
 * c0, c1 are [batch][L][n] and updated in place; c2 is read only. */
int fhe_ct_relinearize(fhe_rns_ntt_t *h, const fhe_relin_keys_t *rk, void *d_c0, void *d_c1, const void *d_c2, uint32_t batch);
/* FHEContext::multiply as declared (include/fhe.cuh:101-103, src/fhe.cu:199-224): tensor product followed by relinearisation,
 * (c0, c1) = relin(a (x) b), two components out.  Same result as fhe_ct_multiply + fhe_ct_relinearize; c2 stays in an internal
 * workspace (compact form on the word-sized classes), so the call moves about half the bytes of the two-call sequence.
 * Outputs must be distinct and must not alias the inputs. */
int fhe_ct_multiply_relin(fhe_rns_ntt_t *h, const fhe_relin_keys_t *rk, void *d_c0, void *d_c1, const void *d_a0, const void *d_a1,
                          const void *d_b0, const void *d_b1, uint32_t batch);

/* ---- blind-rotation inner loop ------------------------------------------------------ */
/* FHEContext::blind_rotate is only declared in the reference (include/fhe.cuh:139; pipeline prose README.md:146-159).  Its
 * inner loop is  acc <- acc + ExternalProduct((X^a - 1) * acc, RGSW(s)),  and the external product of an RLWE pair (d0, d1)
 * with an RGSW ciphertext is two key switches accumulated into one pair.  An RGSW ciphertext is therefore imported as two
 * fhe_relin_keys_t objects (rows for component 0 and for component 1, level order j*K + k as for relinearisation keys).
 *
 * d_out[b][l] = (X^shift[b] - 1) * d_in[b][l] over Z_q[x]/(x^n + 1); d_shifts is a DEVICE array of `batch` values in [0, 2n). */
int fhe_rns_monomial_mul_sub(fhe_rns_ntt_t *h, void *d_out, const void *d_in, const uint32_t *d_shifts, uint32_t batch);
/* One blind-rotation step for `batch` independent accumulators: (acc0, acc1) += ExtProd((X^a - 1) * (acc0, acc1), RGSW) with
 * per-accumulator shifts.  d_tmp0 / d_tmp1 are caller-provided scratch polynomials ([batch][L][n] each, distinct from the
 * accumulators).  Same as fhe_blind_rotate with steps = 1. */
int fhe_blind_rotate_step(fhe_rns_ntt_t *h, const fhe_relin_keys_t *rows_c0, const fhe_relin_keys_t *rows_c1, void *d_acc0, void *d_acc1,
                          const uint32_t *d_shifts, void *d_tmp0, void *d_tmp1, uint32_t batch);
/* The loop itself: for s = 0 .. steps-1:  acc <- acc + ExtProd((X^shift[s][b] - 1) * acc, RGSW_s), RGSW_s = (rows_c0[s], rows_c1[s]);
 * d_shifts is a DEVICE array [steps][batch].  On the word-sized width classes every step is ONE kernel launch that reads the
 * accumulator pair and writes the other buffer pair (ping-pong between acc and tmp; the result always ends in d_acc0 / d_acc1),
 * 4 * S bytes of HBM traffic per accumulator and step; the full-width class composes the monomial kernel and two key switches.
 * Nothing is allocated on the word-sized path, so the whole loop can be captured into a hipGraph. */
int fhe_blind_rotate(fhe_rns_ntt_t *h, const fhe_relin_keys_t *const *rows_c0, const fhe_relin_keys_t *const *rows_c1, uint32_t steps,
                     void *d_acc0, void *d_acc1, const uint32_t *d_shifts, void *d_tmp0, void *d_tmp1, uint32_t batch);

/* ---- scheme plumbing around the hot path ------------------------------------------------ */
/* sample_uniform_kernel (src/polynomial.cu:130-143), LITERAL: out[i] = ((seed + i) * 1103515245 + 12345) % q.limbs[0] in 64-bit
 * wrap-around arithmetic ("Simple LCG for demonstration"); called by FHEContext::sample_uniform_polynomial (src/fhe.cu:245-250).
 * count <= 2^32 (the reference's index is a uint32_t); stream may be NULL. */
int fhe_sample_uniform_lcg(void *d_out, const uint64_t q[4], uint64_t seed, size_t count, void *stream);
/* sample_gaussian_kernel (src/polynomial.cu:113-128), LITERAL placeholder: out[i] = (seed + i) % q.limbs[0] (sigma is ignored
 * there); called by FHEContext::sample_error_polynomial (src/fhe.cu:238-243).  Not a Gaussian: see fhe_rns_sample_gaussian. */
int fhe_sample_gaussian_placeholder(void *d_out, const uint64_t q[4], uint64_t seed, size_t count, void *stream);
/* The samplers the reference declares or leaves as placeholders, on the RNS layout [batch][L][n] (a small signed integer is
 * embedded identically in every limb: -m -> q_l - m).  Counter-based generator over (seed, coefficient index): results do not
 * depend on the launch shape and equal the CPU oracle's bit for bit.
 *   ternary : sample_ternary_kernel(result, modulus, probability, seed, n) (include/polynomial.cuh:129-135, declared only; called
 *             with probability 0.5 by src/fhe.cu:252-257): P(coefficient != 0) = probability, sign uniform.
 *   gaussian: discrete Gaussian of parameter sigma over the integers, cut at 12 sigma, by inversion of the cumulative table
 *             fhe_gaussian_cdt builds (what sample_gaussian_kernel's comment asks for, src/polynomial.cu:122-123).
 *   uniform : residues uniform in [0, q_l) per limb, rejection sampling (no modulo bias), any modulus width. */
int fhe_rns_sample_ternary(fhe_rns_ntt_t *h, void *d_out, double probability, uint64_t seed, uint32_t batch);
int fhe_rns_sample_gaussian(fhe_rns_ntt_t *h, void *d_out, double sigma, uint64_t seed, uint32_t batch);
int fhe_rns_sample_uniform(fhe_rns_ntt_t *h, void *d_out, uint64_t seed, uint32_t batch);
/* Host helper: table[k] = floor(2^64 * P(|X| <= k)), k = 0 .. len-1, len = ceil(12 sigma); table == NULL queries len. */
int fhe_gaussian_cdt(double sigma, uint64_t *table, uint32_t capacity, uint32_t *len);
/* poly_mod_switch_kernel(result, a, old_modulus, new_modulus, n) (include/polynomial.cuh:96-103, declared; launched by
 * FHEContext::decrypt, src/fhe.cu:181-184: "scale down by delta and reduce mod t"):
 * r[i] = round(a[i] * new_q / old_q) mod new_q (round half up) on single-modulus containers; a < old_q < 2^255, new_q < 2^64. */
int fhe_poly_mod_switch(void *d_r, const void *d_a, const uint64_t old_q[4], const uint64_t new_q[4], size_t count, void *stream);
/* negacyclic_reduce_kernel(data, modulus, n) (include/polynomial.cuh:105-110, declared): fold 2n coefficients modulo x^n + 1,
 * data[i] = sub_mod(data[i], data[i + n]) for i < n (the upper half is left unchanged). */
int fhe_negacyclic_reduce(void *d_data, const uint64_t q[4], size_t n, void *stream);

/* Scan a [batch][L][n] buffer for coefficients that are not canonical (>= q_limb, or non-zero
 * upper limbs on the narrow paths).  Synchronises.  FHE_OK or FHE_ERR_NONCANONICAL. */
int fhe_rns_check_canonical(fhe_rns_ntt_t *h, const void *d_data, uint32_t batch);

/* ---- timing on the handle's stream (hipEvent pair; what bench.py uses for the roofline) ------ */
typedef struct fhe_timer fhe_timer_t;
int fhe_timer_create(fhe_timer_t **out);
int fhe_timer_destroy(fhe_timer_t *t);
int fhe_rns_timer_start(fhe_rns_ntt_t *h, fhe_timer_t *t);   /* hipEventRecord(start, h->stream) */
int fhe_rns_timer_stop(fhe_rns_ntt_t *h, fhe_timer_t *t);    /* hipEventRecord(stop,  h->stream) */
int fhe_timer_elapsed_ms(fhe_timer_t *t, float *ms);         /* synchronises on stop */

#ifdef __cplusplus
}
#endif
#endif /* FHE_HIP_H */
