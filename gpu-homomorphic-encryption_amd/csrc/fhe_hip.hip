// fhe_hip.hip -- implementation of the C ABI in include/fhe_hip.h (gfx950 only, no CPU fallback).
#include "../../include/fhe_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "host_math.hpp"
#include "ntt256.hip.h"
#include "ntt_wide.hip.h"
#include "sampling.hip.h"
#include "lds_launch.h"
#include "ntt_word.hip.h"

using fhe_host::U256;

// ------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const std::string &msg) { g_last_error = msg; return code; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(FHE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

static bool env_sync() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("FHE_HIP_SYNC"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

static int ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(FHE_ERR_NO_DEVICE, std::string("no usable HIP device (hipGetDeviceCount: ") +
                                           (e == hipSuccess ? "0 devices" : hipGetErrorString(e)) +
                                           "); this library has no CPU fallback");
    }
    return FHE_OK;
}

static int post_launch(hipStream_t s, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FHE_ERR_HIP, std::string(what) + " launch: " + hipGetErrorString(e));
    if (env_sync()) {
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(FHE_ERR_HIP, std::string(what) + " sync: " + hipGetErrorString(e));
    }
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------------
// plumbing entry points
// ------------------------------------------------------------------------------------------------------
extern "C" int fhe_hip_abi_version(void) { return FHE_HIP_ABI_VERSION; }
extern "C" const char *fhe_hip_last_error(void) { return g_last_error.c_str(); }

extern "C" int fhe_hip_device_count(int *count) {
    if (!count) return fail(FHE_ERR_INVALID_ARG, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); *count = 0; return fail(FHE_ERR_NO_DEVICE, hipGetErrorString(e)); }
    *count = n;
    return FHE_OK;
}
extern "C" int fhe_hip_set_device(int device) {
    int rc = ensure_device(); if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    return FHE_OK;
}
extern "C" int fhe_hip_get_device(int *device) {
    if (!device) return fail(FHE_ERR_INVALID_ARG, "device is null");
    int rc = ensure_device(); if (rc) return rc;
    HIP_TRY(hipGetDevice(device));
    return FHE_OK;
}
extern "C" int fhe_hip_device_name(char *buf, size_t buflen) {
    if (!buf || !buflen) return fail(FHE_ERR_INVALID_ARG, "buf is null");
    int rc = ensure_device(); if (rc) return rc;
    int dev = 0; HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "%s %s (%d CUs)", prop.gcnArchName, prop.name, prop.multiProcessorCount);
    return FHE_OK;
}
extern "C" int fhe_hip_malloc(void **d_ptr, size_t bytes) {
    if (!d_ptr) return fail(FHE_ERR_INVALID_ARG, "d_ptr is null");
    int rc = ensure_device(); if (rc) return rc;
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1));
    return FHE_OK;
}
extern "C" int fhe_hip_free(void *d_ptr) { if (d_ptr) HIP_TRY(hipFree(d_ptr)); return FHE_OK; }
extern "C" int fhe_hip_memset(void *d_ptr, int value, size_t bytes) { HIP_TRY(hipMemset(d_ptr, value, bytes)); return FHE_OK; }
extern "C" int fhe_hip_memcpy_h2d(void *d, const void *h, size_t bytes) { HIP_TRY(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return FHE_OK; }
extern "C" int fhe_hip_memcpy_d2h(void *h, const void *d, size_t bytes) { HIP_TRY(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); return FHE_OK; }
extern "C" int fhe_hip_memcpy_d2d(void *d, const void *s, size_t bytes) { HIP_TRY(hipMemcpy(d, s, bytes, hipMemcpyDeviceToDevice)); return FHE_OK; }
extern "C" int fhe_hip_sync(void) { int rc = ensure_device(); if (rc) return rc; HIP_TRY(hipDeviceSynchronize()); return FHE_OK; }

// ------------------------------------------------------------------------------------------------------
// host parameter maths (no device needed)
// ------------------------------------------------------------------------------------------------------
extern "C" int fhe_montgomery_inverse(const uint64_t q[4], uint64_t inv[4]) {
    if (!q || !inv) return fail(FHE_ERR_INVALID_ARG, "null argument");
    // literal: 6 Newton steps on the low limb from x = 1, negated (src/bigint.cu:27-37); garbage for even q
    inv[0] = fhe_host::neg_inv64(q[0]); inv[1] = inv[2] = inv[3] = 0;
    return FHE_OK;
}
extern "C" int fhe_montgomery_params(const uint64_t q[4], uint64_t r_squared[4], uint64_t inv[4]) {
    if (!q || !r_squared || !inv) return fail(FHE_ERR_INVALID_ARG, "null argument");
    U256 Q = U256::from(q);
    if (!(Q.w[0] & 1) || (Q.w[3] >> 63) || Q.bit_length() < 2) return fail(FHE_ERR_BAD_MODULUS, "modulus must be odd, > 1 and < 2^255");
    fhe_host::Mod M(Q);
    std::memcpy(r_squared, M.r2.w, 32);
    return fhe_montgomery_inverse(q, inv);
}
extern "C" int fhe_find_ntt_primes(uint32_t bits, uint32_t n, uint32_t count, uint64_t *primes_out) {
    if (!primes_out || !count) return fail(FHE_ERR_INVALID_ARG, "null output / zero count");
    if (!fhe_host::find_ntt_primes(bits, n, count, primes_out))
        return fail(FHE_ERR_INVALID_ARG, "no such primes (need 4 <= bits <= 64, n a power of two, 2n < 2^(bits-1))");
    return FHE_OK;
}
extern "C" int fhe_find_ntt_primes_wide(uint32_t bits, uint32_t n, uint32_t count, uint64_t (*primes_out)[4]) {
    if (!primes_out || !count) return fail(FHE_ERR_INVALID_ARG, "null output / zero count");
    std::vector<U256> ps(count);
    if (!fhe_host::find_ntt_primes_wide(bits, n, count, ps.data()))
        return fail(FHE_ERR_INVALID_ARG, "no such primes (need 4 <= bits <= 255, n a power of two, 2n < 2^(bits-2))");
    for (uint32_t i = 0; i < count; i++) std::memcpy(primes_out[i], ps[i].w, 32);
    return FHE_OK;
}
extern "C" int fhe_find_psi(uint32_t n, const uint64_t q[4], uint64_t psi[4]) {
    if (!q || !psi) return fail(FHE_ERR_INVALID_ARG, "null argument");
    if (n < 2 || (n & (n - 1))) return fail(FHE_ERR_INVALID_ARG, "n must be a power of two");
    U256 Q = U256::from(q);
    if (!(Q.w[0] & 1) || (Q.w[3] >> 63) || !fhe_host::is_prime(Q)) return fail(FHE_ERR_BAD_MODULUS, "modulus must be an odd prime < 2^255");
    fhe_host::Mod M(Q); U256 p;
    if (fhe_host::find_psi(n, M, p) != fhe_host::BUILD_OK) return fail(FHE_ERR_BAD_MODULUS, "q != 1 (mod 2n): no primitive 2n-th root");
    std::memcpy(psi, p.w, 32);
    return FHE_OK;
}

// ------------------------------------------------------------------------------------------------------
// literal element-wise kernels
// ------------------------------------------------------------------------------------------------------
static fhe_dev::u256 to_dev(const uint64_t q[4]) { fhe_dev::u256 r; std::memcpy(r.l, q, 32); return r; }

static unsigned ew_grid(size_t items) {
    size_t blocks = (items + 255) / 256;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));   // grid-stride above 8192 blocks
}

template <int OP>
static int launch_ew256(void *d_r, const void *d_a, const void *d_b, const uint64_t q[4], const uint64_t *scalar,
                        uint64_t inv0, size_t count, void *stream, const char *what) {
    if (!d_r || !d_a || (OP != 3 && !d_b) || !q) return fail(FHE_ERR_INVALID_ARG, std::string(what) + ": null argument");
    int rc = ensure_device(); if (rc) return rc;
    if (!count) return FHE_OK;
    hipStream_t s = (hipStream_t)stream;
    fhe_dev::u256 Q = to_dev(q), S = scalar ? to_dev(scalar) : Q;
    hipLaunchKernelGGL(fhe_dev::ew256_kernel<OP>, dim3(ew_grid(count)), dim3(256), 0, s, (fhe_dev::u256 *)d_r,
                       (const fhe_dev::u256 *)d_a, (const fhe_dev::u256 *)d_b, Q, S, inv0, count);
    return post_launch(s, what);
}
extern "C" int fhe_u256_add_mod(void *r, const void *a, const void *b, const uint64_t q[4], size_t count, void *stream) {
    return launch_ew256<1>(r, a, b, q, nullptr, 0, count, stream, "fhe_u256_add_mod");
}
extern "C" int fhe_u256_sub_mod(void *r, const void *a, const void *b, const uint64_t q[4], size_t count, void *stream) {
    return launch_ew256<2>(r, a, b, q, nullptr, 0, count, stream, "fhe_u256_sub_mod");
}
extern "C" int fhe_u256_mont_mul(void *r, const void *a, const void *b, const uint64_t q[4], uint64_t inv0, size_t count, void *stream) {
    return launch_ew256<0>(r, a, b, q, nullptr, inv0, count, stream, "fhe_u256_mont_mul");
}
extern "C" int fhe_u256_mont_mul_scalar(void *r, const void *a, const uint64_t scalar[4], const uint64_t q[4], uint64_t inv0, size_t count, void *stream) {
    if (!scalar) return fail(FHE_ERR_INVALID_ARG, "scalar is null");
    return launch_ew256<3>(r, a, nullptr, q, scalar, inv0, count, stream, "fhe_u256_mont_mul_scalar");
}

// the reference's transform kernels as written (L1 parity; see ntt256.hip.h)
static int ref_literal_check(const void *d_data, const void *d_table, const uint64_t q[4], uint32_t n, uint32_t batch, const char *what) {
    if (!d_data || !d_table || !q) return fail(FHE_ERR_INVALID_ARG, std::string(what) + ": null argument");
    if (n < 2 || (n & (n - 1)) || n > 65536) return fail(FHE_ERR_INVALID_ARG, std::string(what) + ": n must be a power of two in [2, 65536]");
    if (!batch) return fail(FHE_ERR_INVALID_ARG, std::string(what) + ": batch must be >= 1");
    return ensure_device();
}
extern "C" int fhe_ref_forward_kernel_literal(void *d_data, const void *d_twiddles, const uint64_t q[4], uint64_t inv0, uint32_t n, uint32_t batch, void *stream) {
    int rc = ref_literal_check(d_data, d_twiddles, q, n, batch, "fhe_ref_forward_kernel_literal"); if (rc) return rc;
    (void)hipGetLastError();
    hipLaunchKernelGGL(fhe_dev::ref_forward_literal_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (fhe_dev::u256 *)d_data,
                       (const fhe_dev::u256 *)d_twiddles, to_dev(q), inv0, n);
    return post_launch((hipStream_t)stream, "ref_forward_literal_kernel");
}
extern "C" int fhe_ref_inverse_kernel_literal(void *d_data, const void *d_inv_twiddles, const uint64_t q[4], uint64_t inv0, const uint64_t n_inv[4],
                                              uint32_t n, uint32_t batch, void *stream) {
    int rc = ref_literal_check(d_data, d_inv_twiddles, q, n, batch, "fhe_ref_inverse_kernel_literal"); if (rc) return rc;
    if (!n_inv) return fail(FHE_ERR_INVALID_ARG, "fhe_ref_inverse_kernel_literal: n_inv is null");
    (void)hipGetLastError();
    hipLaunchKernelGGL(fhe_dev::ref_inverse_literal_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (fhe_dev::u256 *)d_data,
                       (const fhe_dev::u256 *)d_inv_twiddles, to_dev(q), inv0, to_dev(n_inv), n);
    return post_launch((hipStream_t)stream, "ref_inverse_literal_kernel");
}

extern "C" int fhe_ref_stockham_stage_literal(void *d_output, const void *d_input, const void *d_twiddles, const uint64_t q[4], uint64_t inv0, uint32_t n,
                                             uint32_t stage, uint32_t batch, void *stream) {
    int rc = ref_literal_check(d_output, d_twiddles, q, n, batch, "fhe_ref_stockham_stage_literal"); if (rc) return rc;
    if (!d_input || d_input == d_output) return fail(FHE_ERR_INVALID_ARG, "fhe_ref_stockham_stage_literal: the stage is out of place");
    if ((2u << stage) > n) return fail(FHE_ERR_INVALID_ARG, "fhe_ref_stockham_stage_literal: stage must satisfy 2^(stage+1) <= n");
    (void)hipGetLastError();
    const size_t count = (size_t)batch * (n / 2);
    hipLaunchKernelGGL(fhe_dev::ref_stockham_stage_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, (fhe_dev::u256 *)d_output,
                       (const fhe_dev::u256 *)d_input, (const fhe_dev::u256 *)d_twiddles, to_dev(q), inv0, n, stage, count);
    return post_launch((hipStream_t)stream, "ref_stockham_stage_kernel");
}

extern "C" int fhe_bit_reverse(void *d_data, uint32_t n, uint32_t batch, void *stream) {
    if (!d_data) return fail(FHE_ERR_INVALID_ARG, "fhe_bit_reverse: null argument");
    if (n < 2 || (n & (n - 1))) return fail(FHE_ERR_INVALID_ARG, "fhe_bit_reverse: n must be a power of two >= 2");
    if (!batch) return fail(FHE_ERR_INVALID_ARG, "fhe_bit_reverse: batch must be >= 1");
    int rc = ensure_device(); if (rc) return rc;
    (void)hipGetLastError();
    uint32_t log_n = 0; while ((1u << log_n) < n) log_n++;
    const size_t count = (size_t)batch * n;
    hipLaunchKernelGGL(fhe_dev::bit_reverse_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, (fhe_dev::u256 *)d_data, log_n, count);
    return post_launch((hipStream_t)stream, "bit_reverse_kernel");
}

// ------------------------------------------------------------------------------------------------------
// engine handle
// ------------------------------------------------------------------------------------------------------
struct fhe_rns_ntt {
    int device = 0;
    uint32_t n = 0, log_n = 0, L = 0;
    int width = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t aux_stream = nullptr;   // second stream of the overlapped one-call multiply (fork / join with events around it)
    hipEvent_t ev_chunk[16] = {}, ev_join = nullptr;   // tensor product of chunk i done (engine stream) -> key switch of chunk i may start (second stream)
    uint32_t small_batch_polys = 256;   // FHE_HIP_SMALL_BATCH_POLYS: fused multiply of at most this many limb polynomials runs the 16-per-thread latency kernel (0 = never)
    uint32_t coop_polys = 64;           // FHE_HIP_COOP_POLYS: fused multiply of at most this many limb polynomials (N = 2^13 / 2^14, 4-byte residues) spreads each over four workgroups in three launches (0 = never)
    uint32_t max_digits = 0, max_composed_digits = 0;   // largest digit count K of the key sets imported on this engine / of those without packed tables (fhe_rns_ntt_reserve)
    uint32_t split_pairs_polys = 128;   // FHE_HIP_SPLIT_PAIRS_POLYS: key switch (paired kernel) of at most this many limb polynomials runs one workgroup per digit pair + a combining launch (0 = never)
    bool relin_chunks_forced = false;   // FHE_HIP_RELIN_PIPELINE=1: the stand-alone relinearisation also runs as a two-stream pipeline (A/B)
    uint32_t overlap_chunks = 4;        // FHE_HIP_CT_RELIN_CHUNKS: pieces the one-call multiply is cut into (1 = one stream, as in round 2)
    void *d_limbs = nullptr;            // owned by d_tables
    void *d_wlimbs = nullptr;           // FHE_WIDTH_256: WLimb<wide_nl>[L] for the NTT kernels of ntt_wide.hip.h (owned by d_tables)
    uint32_t sub_top = 0;               // word-sized classes beyond the LDS range: log2 n = 13 + sub_top (two-pass transforms), else 0
    int wide_nl = 0;                    // FHE_WIDTH_256: 64-bit limbs per residue in those kernels (2: q < 2^127, 4: q < 2^255)
    bool wide_lazy = false;             // FHE_WIDTH_256: every modulus below 2^(64 wide_nl - 6): the lazy tile kernels (FHE_HIP_NO_WIDE_LAZY=1: the canonical ones; cross-check / A-B)
    bool wide_tiles = true;             // FHE_HIP_NO_WIDE_TILES=1: every stage as a global-memory pass (cross-check / A-B)
    bool no_square = false, single_transforms = false, global_twiddles = false, check_inputs = false, no_fused_keyswitch = false,
         no_word_conversions = false, no_fused_blind_rotate = false, no_fused_ct_relin = false, no_compact_blind_rotate = false, no_two_launch_ct = false, split_keyswitch = false, no_c2_compaction = false, no_prerotation = false;
    int ct_form_force = 0;                         // FHE_HIP_CT_FORM: 0 = by field and size, 1 = one-launch tensor product where it exists, 2 = two-launch where it exists   // environment switches, read once at creation
    std::vector<void *> d_tables;
    void *d_ws = nullptr; size_t ws_bytes = 0;
    void *d_ws2 = nullptr; size_t ws2_bytes = 0;   // c2 of the fused multiply + relinearise (compact or containers); separate from d_ws, which the general paths use
    void *d_ws3 = nullptr; size_t ws3_bytes = 0;   // compact polynomials between the two launches of a two-pass transform (sub_top != 0)
    uint32_t *d_flag = nullptr;
    std::vector<U256> moduli;
    void *d_crt = nullptr;               // CrtLimb[L], built on first use of to_rns / from_rns (owned by d_tables)
    void *d_rescale = nullptr;           // RescaleLimb[L-1], built on first use of rescale_drop_last (owned by d_tables)
    void *d_bconv = nullptr;             // ((Q/q_i) mod p_j) * R_j for the most recent base-conversion target (owned by d_tables)
    const void *bconv_target = nullptr;
    void *d_rescale_w = nullptr;         // word-sized classes: (q_last^-1 mod q_l) as pw operands, E[L-1] (owned by d_tables)
    void *d_bconv_w_minv = nullptr, *d_bconv_w_mat = nullptr;   // word-sized classes: base-conversion operands for bconv_w_target
    const void *bconv_w_target = nullptr;
    std::vector<U256> bconv_moduli, bconv_w_moduli;   // the target bases the cached matrices were built for (a handle address can be re-used)
    void *d_from_rns_w_minv = nullptr, *d_from_rns_w_M = nullptr;   // integer word classes: CRT operands and Q / q_l (owned by d_tables)
    void *d_to_rns_w = nullptr;          // integer word classes: 2^(W k) mod q_l as pw operands, E[L][256 / W] (owned by d_tables)
    fhe_dev::CrtBig crt_big;
    int crt_state = 0;                   // 0 = not built, 1 = ready, -1 = Q too large for from_rns (to_rns still fine)
    double cdt_sigma = 0; uint64_t *d_cdt = nullptr; uint32_t cdt_len = 0;   // cumulative table of the last Gaussian sampler call
};
struct fhe_ntt { fhe_rns_ntt *impl; };

static void destroy_impl(fhe_rns_ntt *h) {
    if (!h) return;
    for (void *p : h->d_tables) (void)hipFree(p);
    if (h->d_ws) (void)hipFree(h->d_ws);
    if (h->d_ws2) (void)hipFree(h->d_ws2);
    if (h->d_ws3) (void)hipFree(h->d_ws3);
    if (h->d_cdt) (void)hipFree(h->d_cdt);
    if (h->d_flag) (void)hipFree(h->d_flag);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    for (hipEvent_t e : h->ev_chunk) if (e) (void)hipEventDestroy(e);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

template <class T>
static int upload(fhe_rns_ntt *h, const std::vector<T> &v, void **out) {
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, v.size() * sizeof(T)));
    h->d_tables.push_back(d);
    HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = d;
    return FHE_OK;
}

static uint32_t shoup32(uint64_t w, uint64_t q) { return (uint32_t)((w << 32) / q); }
static uint64_t shoup64(uint64_t w, uint64_t q) { return (uint64_t)((((fhe_host::u128)w) << 64) / q); }

static int build_limbs32(fhe_rns_ntt *h, const std::vector<fhe_host::NttConstants> &cs) {
    std::vector<fhe_dev::Limb32> limbs(h->L);
    for (uint32_t l = 0; l < h->L; l++) {
        const fhe_host::NttConstants &c = cs[l];
        const uint64_t q = c.q.w[0], n = h->n;
        std::vector<uint32_t> tw(n), itw(n);               // Montgomery form: w * 2^32 mod q
        for (uint32_t k = 0; k < n; k++) {
            tw[k] = (uint32_t)((c.tw[k].w[0] << 32) % q);
            itw[k] = (uint32_t)((c.itw[k].w[0] << 32) % q);
        }
        fhe_dev::Limb32 &P = limbs[l];
        std::memset(&P, 0, sizeof(P));
        P.q = (uint32_t)q; P.q2 = (uint32_t)(2 * q);
        uint32_t x = 1; for (int i = 0; i < 5; i++) x *= 2 - (uint32_t)q * x;     // q^-1 mod 2^32
        P.qinv = 0u - x;                                                           // negated: F32::mont_mul adds m*q instead of subtracting
        const uint64_t two32 = (1ull << 32) % q, ninv = c.n_inv.w[0], w1 = c.itw[1].w[0];
        auto mulq = [q](uint64_t a, uint64_t b) { return (uint64_t)(((fhe_host::u128)a * b) % q); };
        P.r1 = (uint32_t)two32; P.r1_s = shoup32(two32, q);
        P.ninv = (uint32_t)ninv; P.ninv_s = shoup32(ninv, q);
        uint64_t nw = mulq(ninv, w1);
        P.ninvw = (uint32_t)nw; P.ninvw_s = shoup32(nw, q);
        uint64_t nr = mulq(ninv, two32), nwr = mulq(nw, two32);
        P.ninv_r = (uint32_t)nr; P.ninv_r_s = shoup32(nr, q);
        P.ninvw_r = (uint32_t)nwr; P.ninvw_r_s = shoup32(nwr, q);
        void *d = nullptr; int rc;
        if ((rc = upload(h, tw, &d))) return rc; P.tw = (const uint32_t *)d;
        if ((rc = upload(h, itw, &d))) return rc; P.itw = (const uint32_t *)d;
    }
    return upload(h, limbs, &h->d_limbs);
}

static int build_limbs64(fhe_rns_ntt *h, const std::vector<fhe_host::NttConstants> &cs) {
    std::vector<fhe_dev::Limb64> limbs(h->L);
    for (uint32_t l = 0; l < h->L; l++) {
        const fhe_host::NttConstants &c = cs[l];
        const uint64_t q = c.q.w[0], n = h->n;
        std::vector<ulonglong2> tw(n), itw(n);
        for (uint32_t k = 0; k < n; k++) {
            tw[k] = make_ulonglong2(c.tw[k].w[0], shoup64(c.tw[k].w[0], q));
            itw[k] = make_ulonglong2(c.itw[k].w[0], shoup64(c.itw[k].w[0], q));
        }
        fhe_dev::Limb64 &P = limbs[l];
        std::memset(&P, 0, sizeof(P));
        P.q = q; P.q2 = 2 * q;
        uint64_t x = 1; for (int i = 0; i < 6; i++) x *= 2 - q * x;               // q^-1 mod 2^64
        P.qinv = x;
        auto mulq = [q](uint64_t a, uint64_t b) { return (uint64_t)(((fhe_host::u128)a * b) % q); };
        const uint64_t two64 = (uint64_t)((((fhe_host::u128)1) << 64) % q), ninv = c.n_inv.w[0], w1 = c.itw[1].w[0];
        P.r1 = two64; P.r1_s = shoup64(two64, q);
        P.ninv = ninv; P.ninv_s = shoup64(ninv, q);
        uint64_t nw = mulq(ninv, w1);
        P.ninvw = nw; P.ninvw_s = shoup64(nw, q);
        uint64_t nr = mulq(ninv, two64), nwr = mulq(nw, two64);
        P.ninv_r = nr; P.ninv_r_s = shoup64(nr, q);
        P.ninvw_r = nwr; P.ninvw_r_s = shoup64(nwr, q);
        void *d = nullptr; int rc;
        if ((rc = upload(h, tw, &d))) return rc; P.tw = (const ulonglong2 *)d;
        if ((rc = upload(h, itw, &d))) return rc; P.itw = (const ulonglong2 *)d;
    }
    return upload(h, limbs, &h->d_limbs);
}

static int build_limbs64x(fhe_rns_ntt *h, const std::vector<fhe_host::NttConstants> &cs) {
    std::vector<fhe_dev::Limb64X> limbs(h->L);
    for (uint32_t l = 0; l < h->L; l++) {
        const fhe_host::NttConstants &c = cs[l];
        const uint64_t q = c.q.w[0], n = h->n;
        auto mulq = [q](uint64_t a, uint64_t b) { return (uint64_t)(((fhe_host::u128)a * b) % q); };
        const uint64_t two64 = (uint64_t)((((fhe_host::u128)1) << 64) % q), two128 = mulq(two64, two64);
        std::vector<uint64_t> tw(n), itw(n);               // Montgomery form: w * 2^64 mod q
        for (uint32_t k = 0; k < n; k++) { tw[k] = mulq(c.tw[k].w[0], two64); itw[k] = mulq(c.itw[k].w[0], two64); }
        fhe_dev::Limb64X &P = limbs[l];
        std::memset(&P, 0, sizeof(P));
        P.q = q; P.q2 = 0;                                 // 2q does not fit; nothing on this field reads it
        uint64_t x = 1; for (int i = 0; i < 6; i++) x *= 2 - q * x;               // q^-1 mod 2^64
        P.qinv = x;
        const uint64_t ninv = c.n_inv.w[0], nw = mulq(ninv, c.itw[1].w[0]);
        P.r1 = two128;
        P.ninv = mulq(ninv, two64); P.ninvw = mulq(nw, two64);
        P.ninv_r = mulq(ninv, two128); P.ninvw_r = mulq(nw, two128);
        P.r1_s = P.ninv_s = P.ninvw_s = P.ninv_r_s = P.ninvw_r_s = x;              // the companion slots carry q^-1 (F64X::inv_last)
        void *d = nullptr; int rc;
        if ((rc = upload(h, tw, &d))) return rc; P.tw = (const uint64_t *)d;
        if ((rc = upload(h, itw, &d))) return rc; P.itw = (const uint64_t *)d;
    }
    return upload(h, limbs, &h->d_limbs);
}

static int build_limbs52(fhe_rns_ntt *h, const std::vector<fhe_host::NttConstants> &cs) {
    std::vector<fhe_dev::Limb52> limbs(h->L);
    for (uint32_t l = 0; l < h->L; l++) {
        const fhe_host::NttConstants &c = cs[l];
        const double q = (double)c.q.w[0];                 // q < 2^43: exact
        const uint64_t qi = c.q.w[0], n = h->n;
        std::vector<double> tw(n), itw(n);                 // companions fl(w * fl(1/q)) are recomputed in the butterflies
        for (uint32_t k = 0; k < n; k++) { tw[k] = (double)c.tw[k].w[0]; itw[k] = (double)c.itw[k].w[0]; }
        fhe_dev::Limb52 &P = limbs[l];
        std::memset(&P, 0, sizeof(P));
        auto mulq = [qi](uint64_t a, uint64_t b) { return (uint64_t)(((fhe_host::u128)a * b) % qi); };
        const uint64_t ninv = c.n_inv.w[0], nw = mulq(ninv, c.itw[1].w[0]);
        P.q = q; P.q2 = 2 * q; P.qinv = 1.0 / q;
        P.r1 = 1.0; P.r1_s = 1.0 / q;                       // no Montgomery factor on this path
        P.ninv = (double)ninv; P.ninv_s = (double)ninv / q;
        P.ninvw = (double)nw; P.ninvw_s = (double)nw / q;
        P.ninv_r = P.ninv; P.ninv_r_s = P.ninv_s; P.ninvw_r = P.ninvw; P.ninvw_r_s = P.ninvw_s;
        void *d = nullptr; int rc;
        if ((rc = upload(h, tw, &d))) return rc; P.tw = (const double *)d;
        if ((rc = upload(h, itw, &d))) return rc; P.itw = (const double *)d;
    }
    return upload(h, limbs, &h->d_limbs);
}

static int build_limbs256(fhe_rns_ntt *h, const std::vector<fhe_host::NttConstants> &cs) {
    std::vector<fhe_dev::Limb256> limbs(h->L);
    for (uint32_t l = 0; l < h->L; l++) {
        const fhe_host::NttConstants &c = cs[l];
        fhe_host::Mod M(c.q);
        std::vector<fhe_dev::u256> tw(h->n), itw(h->n);
        for (uint32_t k = 0; k < h->n; k++) {
            U256 a = M.to_mont(c.tw[k]), b = M.to_mont(c.itw[k]);
            std::memcpy(tw[k].l, a.w, 32); std::memcpy(itw[k].l, b.w, 32);
        }
        fhe_dev::Limb256 &P = limbs[l];
        std::memset(&P, 0, sizeof(P));
        std::memcpy(P.q.l, c.q.w, 32);
        std::memcpy(P.r2.l, M.r2.w, 32);
        U256 nm = M.to_mont(c.n_inv);
        std::memcpy(P.ninv_m.l, nm.w, 32);
        P.inv0 = M.inv0;
        void *d = nullptr; int rc;
        if ((rc = upload(h, tw, &d))) return rc; P.tw_m = (const fhe_dev::u256 *)d;
        if ((rc = upload(h, itw, &d))) return rc; P.itw_m = (const fhe_dev::u256 *)d;
    }
    return upload(h, limbs, &h->d_limbs);
}

// Constants of the LDS-staged wide kernels (ntt_wide.hip.h): Montgomery radix R = 2^(64 NL).
template <int NL>
static int build_wlimbs(fhe_rns_ntt *h, const std::vector<fhe_host::NttConstants> &cs) {
    using W = fhe_dev::wint<NL>;
    std::vector<fhe_dev::WLimb<NL>> limbs(h->L);
    for (uint32_t l = 0; l < h->L; l++) {
        const fhe_host::NttConstants &c = cs[l];
        fhe_host::Mod M(c.q);
        U256 Rn = M.r1;                                   // 2^256 mod q
        if (NL == 2) { U256 t; t.w[2] = 1; Rn = M.reduce(t); }   // 2^128 mod q
        auto put = [](W &dst, const U256 &v) { for (int i = 0; i < NL; i++) { dst.w[2 * i] = (uint32_t)v.w[i]; dst.w[2 * i + 1] = (uint32_t)(v.w[i] >> 32); } };
        std::vector<W> tw(h->n), itw(h->n);
        for (uint32_t k = 0; k < h->n; k++) { put(tw[k], M.mul(c.tw[k], Rn)); put(itw[k], M.mul(c.itw[k], Rn)); }
        fhe_dev::WLimb<NL> &P = limbs[l];
        std::memset(&P, 0, sizeof(P));
        const U256 R2 = M.mul(Rn, Rn), nR = M.mul(c.n_inv, Rn);
        put(P.q, c.q); put(P.ninv_m, nR); put(P.ninv_r2, M.mul(nR, Rn)); put(P.r2, R2);
        P.qinv32 = (uint32_t)M.inv0;
        void *d = nullptr; int rc;
        if ((rc = upload(h, tw, &d))) return rc; P.tw = (const W *)d;
        if ((rc = upload(h, itw, &d))) return rc; P.itw = (const W *)d;
    }
    return upload(h, limbs, &h->d_wlimbs);
}

// base_only: an RNS base without a ring (RNSContext, include/rns.cuh:27-66): the handle is an engine of degree n = 1, whose
// buffers [batch][L][1] are exactly RNSContext's interleaved [count][num_primes] layout (src/rns.cu:103-104) and whose
// transforms are the identity (Z_q[x]/(x + 1) = Z_q), so every container-level entry point works on it unchanged.
static fhe_host::BuildStatus base_constants(const U256 &q, fhe_host::NttConstants &out) {
    if (!(q.w[0] & 1) || (q.w[3] >> 63) || q.bit_length() < 2 || !fhe_host::is_prime(q)) return fhe_host::BUILD_BAD_MODULUS;
    out.n = 1; out.log_n = 0; out.q = q;
    fhe_host::sub_to(out.psi, q, U256(1)); out.psi_inv = out.psi;     // the primitive 2nd root of unity, -1
    out.n_inv = U256(1);
    out.tw.assign(1, U256(1)); out.itw.assign(1, U256(1));
    return fhe_host::BUILD_OK;
}
static int create_impl(fhe_rns_ntt **out, uint32_t n, const uint64_t (*moduli)[4], uint32_t L, bool base_only = false) {
    if (!out || !moduli) return fail(FHE_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (base_only) n = 1;
    else if (n < 8 || n > 65536 || (n & (n - 1))) return fail(FHE_ERR_INVALID_ARG, "polynomial degree must be a power of two in [8, 65536]");
    if (L < 1 || L > 64) return fail(FHE_ERR_INVALID_ARG, "num_primes must be in [1, 64]");
    std::vector<fhe_host::NttConstants> cs(L);
    int max_bits = 0;
    for (uint32_t l = 0; l < L; l++) {
        U256 q = U256::from(moduli[l]);
        for (uint32_t k = 0; k < l; k++) if (q == cs[k].q) return fail(FHE_ERR_BAD_MODULUS, "the RNS primes must be pairwise distinct");
        fhe_host::BuildStatus st = base_only ? base_constants(q, cs[l]) : fhe_host::build_constants(n, q, cs[l]);
        if (st != fhe_host::BUILD_OK) {
            char buf[160];
            snprintf(buf, sizeof buf, "modulus %u (low limb 0x%llx) rejected: need an odd prime < 2^255 with q = 1 (mod 2n)", l,
                     (unsigned long long)q.w[0]);
            return fail(FHE_ERR_BAD_MODULUS, buf);
        }
        if (q.bit_length() > max_bits) max_bits = q.bit_length();
    }
    int rc = ensure_device(); if (rc) return rc;
    fhe_rns_ntt *h = new (std::nothrow) fhe_rns_ntt();
    if (!h) return fail(FHE_ERR_INVALID_ARG, "out of host memory");
    h->n = n; h->L = L; while ((1u << h->log_n) < n) h->log_n++;
    for (uint32_t l = 0; l < L; l++) h->moduli.push_back(cs[l].q);
    // word-sized classes: one LDS-resident kernel per transform for 2^11 .. 2^15 (4-byte residues) / 2^14 (8-byte residues), two passes
    // (top stages over global memory + 2^13-coefficient LDS blocks) up to 2^16
    const bool lds_size = h->log_n >= 11 && h->log_n <= 16;
    const char *force = getenv("FHE_HIP_FORCE_WIDTH");       // testing aid: "52" / "64" / "256" force a wider path than needed
    const int floor_w = force ? atoi(force) : 0;
    if (floor_w >= 128) h->width = FHE_WIDTH_256;      // 128: the full-width class on two 64-bit limbs (needs q < 2^127), 256: on four
    else if (lds_size && max_bits <= 30 && floor_w < 52) h->width = FHE_WIDTH_32;
    else if (lds_size && max_bits <= 43 && floor_w < 64) h->width = FHE_WIDTH_52;
    else if (lds_size && max_bits <= 62 && floor_w < 65) h->width = FHE_WIDTH_64;
    else if (lds_size && max_bits <= 64) h->width = FHE_WIDTH_64X;      // FHE_HIP_FORCE_WIDTH=65 forces it
    else h->width = FHE_WIDTH_256;
#define TRY_OR_DESTROY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { destroy_impl(h); return fail(FHE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    TRY_OR_DESTROY(hipGetDevice(&h->device));
    TRY_OR_DESTROY(hipStreamCreate(&h->own_stream));   // blocking stream, like the reference's cudaStreamCreate (src/ntt.cu:18):
                                                       // a later hipMemcpy on the null stream is ordered after our kernels
    h->stream = h->own_stream;
    TRY_OR_DESTROY(hipMalloc((void **)&h->d_flag, sizeof(uint32_t)));
    TRY_OR_DESTROY(hipMemset(h->d_flag, 0, sizeof(uint32_t)));
#undef TRY_OR_DESTROY
    rc = h->width == FHE_WIDTH_32 ? build_limbs32(h, cs) : h->width == FHE_WIDTH_52 ? build_limbs52(h, cs)
         : h->width == FHE_WIDTH_64 ? build_limbs64(h, cs) : h->width == FHE_WIDTH_64X ? build_limbs64x(h, cs) : build_limbs256(h, cs);
    if (h->width != FHE_WIDTH_256 && h->log_n > (h->width == FHE_WIDTH_32 ? 15u : 14u)) h->sub_top = h->log_n - 13;
    if (!rc && h->width == FHE_WIDTH_256) {
        h->wide_nl = (max_bits <= 127 && floor_w != 256) ? 2 : 4;
        // six spare bits above the largest modulus: the tile kernels run their butterflies without full reductions (ntt_wide.hip.h: wct_l)
        h->wide_lazy = max_bits + 6 <= 64 * h->wide_nl && !getenv("FHE_HIP_NO_WIDE_LAZY");
        rc = h->wide_nl == 2 ? build_wlimbs<2>(h, cs) : build_wlimbs<4>(h, cs);
    }
    if (rc) { destroy_impl(h); return rc; }
    h->wide_tiles = !getenv("FHE_HIP_NO_WIDE_TILES");
    h->no_square = getenv("FHE_HIP_NO_SQUARE_KERNELS") != nullptr;
    h->single_transforms = getenv("FHE_HIP_NO_PAIRED_TRANSFORMS") != nullptr;
    h->global_twiddles = getenv("FHE_HIP_NO_LDS_TWIDDLES") != nullptr;
    h->no_fused_keyswitch = getenv("FHE_HIP_NO_FUSED_KEYSWITCH") != nullptr;
    h->no_word_conversions = getenv("FHE_HIP_NO_WORD_CONVERSIONS") != nullptr;
    h->no_fused_blind_rotate = getenv("FHE_HIP_NO_FUSED_BLIND_ROTATE") != nullptr;
    h->no_fused_ct_relin = getenv("FHE_HIP_NO_FUSED_CT_RELIN") != nullptr;
    h->no_compact_blind_rotate = getenv("FHE_HIP_NO_COMPACT_BLIND_ROTATE") != nullptr;
    h->no_two_launch_ct = getenv("FHE_HIP_NO_TWO_LAUNCH_CT") != nullptr;
    h->split_keyswitch = getenv("FHE_HIP_SPLIT_KEYSWITCH") != nullptr;
    if (const char *m = getenv("FHE_HIP_SMALL_BATCH_POLYS")) { const long v = atol(m); h->small_batch_polys = v < 0 ? 0u : (uint32_t)v; }
    if (const char *m = getenv("FHE_HIP_COOP_POLYS")) { const long v = atol(m); h->coop_polys = v < 0 ? 0u : v > 64 ? 64u : (uint32_t)v; }
    if (const char *m = getenv("FHE_HIP_SPLIT_PAIRS_POLYS")) { const long v = atol(m); h->split_pairs_polys = v < 0 ? 0u : (uint32_t)v; }
    h->relin_chunks_forced = getenv("FHE_HIP_RELIN_PIPELINE") != nullptr;
    if (const char *m = getenv("FHE_HIP_CT_RELIN_CHUNKS")) { const int v = atoi(m); h->overlap_chunks = v < 1 ? 1 : v > 16 ? 16 : (uint32_t)v; }
    h->no_prerotation = getenv("FHE_HIP_NO_PREROTATION") != nullptr;       // blind-rotation loop of the three-array kernels: monomial factor inside the kernel, per digit (A/B, cross-check)
    h->no_c2_compaction = getenv("FHE_HIP_NO_C2_COMPACTION") != nullptr;   // stand-alone relinearisation of the 8-byte fields: c2 read as containers (A/B, cross-check)
    if (const char *m = getenv("FHE_HIP_CT_FORM")) h->ct_form_force = !strcmp(m, "two") ? 2 : !strcmp(m, "one") ? 1 : 0;
    { const char *e = getenv("FHE_HIP_CHECK_INPUTS"); h->check_inputs = e && e[0] == '1'; }
    *out = h;
    return FHE_OK;
}

static int check_call(const fhe_rns_ntt *h, uint32_t batch, const char *what) {
    if (!h) return fail(FHE_ERR_INVALID_ARG, std::string(what) + ": null handle");
    if (!batch) return fail(FHE_ERR_INVALID_ARG, std::string(what) + ": batch must be >= 1");
    if ((uint64_t)batch * h->L > 0x7fffffffull) return fail(FHE_ERR_INVALID_ARG, std::string(what) + ": batch * num_primes too large");
    (void)hipGetLastError();            // drop any stale sticky error so post_launch reports only this call's
    return FHE_OK;
}

// Library-owned workspaces are per ENGINE and grow on demand; calls on one engine must be ordered on one stream (they share them).
// A hipGraph captured from a call has the workspace addresses baked in, so growing (free + malloc) later would make every replay
// touch freed memory: growth is refused while the engine's stream is capturing (call fhe_rns_ntt_reserve(h, max_batch) before the
// capture; after it nothing here allocates), and a larger batch after a capture needs a re-capture -- see INTEGRATION.md.
static int grow_ws(fhe_rns_ntt *h, void **ws, size_t *have, size_t bytes) {
    if (*have >= bytes) return FHE_OK;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
        return fail(FHE_ERR_INVALID_ARG, "a library workspace would have to grow while the engine's stream is being captured into a graph: "
                                         "call fhe_rns_ntt_reserve(h, batch) for the largest batch before the capture");
    (void)hipGetLastError();
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (*ws) { HIP_TRY(hipFree(*ws)); *ws = nullptr; *have = 0; }
    HIP_TRY(hipMalloc(ws, bytes));
    *have = bytes;
    return FHE_OK;
}
// d_ws : general paths (the reference mallocs/frees per multiply, src/ntt.cu:51-74) and the transformed b-side of the two-launch tensor product
// d_ws2: c0, c1, c2 of the fused multiply + relinearise and the compact accumulators of a blind-rotation loop
// d_ws3: the compact polynomials between the two launches of a two-pass transform (N beyond the LDS range)
// second stream + events of the chunked two-stage pipelines (fhe_ct_multiply_relin, stand-alone relinearisation): created on first use
static int ensure_aux_stream(fhe_rns_ntt *h) {
    if (h->aux_stream) return FHE_OK;
    HIP_TRY(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    for (hipEvent_t &e : h->ev_chunk) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return FHE_OK;
}
static int ensure_ws(fhe_rns_ntt *h, size_t bytes) { return grow_ws(h, &h->d_ws, &h->ws_bytes, bytes); }
static int ensure_ws2(fhe_rns_ntt *h, size_t bytes) { return grow_ws(h, &h->d_ws2, &h->ws2_bytes, bytes); }
static int ensure_ws3(fhe_rns_ntt *h, size_t bytes) { return grow_ws(h, &h->d_ws3, &h->ws3_bytes, bytes); }

// ---- general (256-bit) path launchers -----------------------------------------------------------------
// One radix-2^R global-memory pass over `polys` polynomials (src -> dst).  forward: stages s0 .. s0+R-1; inverse: index bits s0 .. s0+R-1.
template <int NL, bool FWD>
static void wide_pass(fhe_rns_ntt *h, fhe_dev::u256 *dst, const fhe_dev::u256 *src, uint32_t polys, uint32_t s0, uint32_t R, uint32_t scale) {
    const uint32_t chunk_max = (65535u / h->L) * h->L;      // grid.y limit; chunks keep the limb phase (polynomial index mod L)
    const fhe_dev::WLimb<NL> *limbs = (const fhe_dev::WLimb<NL> *)h->d_wlimbs;
    for (uint32_t done = 0; done < polys;) {
        const uint32_t chunk = polys - done < chunk_max ? polys - done : chunk_max;
        fhe_dev::u256 *d = dst + (size_t)done * h->n; const fhe_dev::u256 *sp = src + (size_t)done * h->n;
        dim3 grid(((h->n >> R) + 255) / 256, chunk), block(256);
        if (R == 3) hipLaunchKernelGGL((fhe_dev::wide_pass_kernel<NL, 3, FWD>), grid, block, 0, h->stream, d, sp, limbs, h->L, h->log_n, s0, scale);
        else if (R == 2) hipLaunchKernelGGL((fhe_dev::wide_pass_kernel<NL, 2, FWD>), grid, block, 0, h->stream, d, sp, limbs, h->L, h->log_n, s0, scale);
        else hipLaunchKernelGGL((fhe_dev::wide_pass_kernel<NL, 1, FWD>), grid, block, 0, h->stream, d, sp, limbs, h->L, h->log_n, s0, scale);
        done += chunk;
    }
}
template <int NL, int MODE>
static void wide_tile(fhe_rns_ntt *h, fhe_dev::u256 *dst, const fhe_dev::u256 *src, const fhe_dev::u256 *src2, uint32_t polys, uint32_t scale) {
    const uint32_t tiles_log = h->log_n - fhe_dev::WT_LOG;
    const uint32_t chunk_max = (0x7fffffffu >> tiles_log) / h->L * h->L;
    for (uint32_t done = 0; done < polys;) {
        const uint32_t chunk = polys - done < chunk_max ? polys - done : chunk_max;
        const size_t o = (size_t)done * h->n;
        if (h->wide_lazy)
            hipLaunchKernelGGL((fhe_dev::wide_tile_kernel<NL, MODE, true>), dim3(chunk << tiles_log), dim3(fhe_dev::WT_T), 0, h->stream, dst + o, src + o,
                               src2 ? src2 + o : nullptr, (const fhe_dev::WLimb<NL> *)h->d_wlimbs, h->L, h->log_n, scale);
        else
            hipLaunchKernelGGL((fhe_dev::wide_tile_kernel<NL, MODE, false>), dim3(chunk << tiles_log), dim3(fhe_dev::WT_T), 0, h->stream, dst + o, src + o,
                               src2 ? src2 + o : nullptr, (const fhe_dev::WLimb<NL> *)h->d_wlimbs, h->L, h->log_n, scale);
        done += chunk;
    }
}
// Number of leading (forward) / trailing (inverse) stages that run as global passes: everything above the 2^11-coefficient LDS tile.
static uint32_t wide_top_stages(const fhe_rns_ntt *h) {
    return (h->wide_tiles && h->log_n >= (uint32_t)fhe_dev::WT_LOG) ? h->log_n - fhe_dev::WT_LOG : h->log_n;
}
// The top stages of a forward transform, src -> dst (src == dst allowed): R <= 3 stages per launch.
template <int NL>
static void wide_forward_top(fhe_rns_ntt *h, fhe_dev::u256 *dst, const fhe_dev::u256 *src, uint32_t polys) {
    const uint32_t top = wide_top_stages(h);
    for (uint32_t s = 0; s < top;) {
        const uint32_t R = top - s >= 3 ? 3 : top - s;
        wide_pass<NL, true>(h, dst, s ? dst : src, polys, s, R, 0);
        s += R;
    }
}
// The trailing stages of an inverse transform, in place on data, with the final scaling (1: n^-1, 2: n^-1 R for the fused products).
template <int NL>
static void wide_inverse_top(fhe_rns_ntt *h, fhe_dev::u256 *data, uint32_t polys, uint32_t scale) {
    const uint32_t top = wide_top_stages(h), b_first = h->log_n - top;
    if (!h->log_n && scale) {                               // degree-1 engine: no butterflies, only the scaling (n^-1 = 1)
        const size_t count = (size_t)polys;
        hipLaunchKernelGGL((fhe_dev::wide_scale_kernel<NL>), dim3(ew_grid(count)), dim3(256), 0, h->stream, data, data,
                           (const fhe_dev::WLimb<NL> *)h->d_wlimbs, h->L, 0u, scale, count);
    }
    for (uint32_t s = 0; s < top;) {
        const uint32_t R = top - s >= 3 ? 3 : top - s;
        wide_pass<NL, false>(h, data, data, polys, b_first + s, R, s + R == top ? scale : 0);
        s += R;
    }
}
template <int NL>
static int wide_transform_t(fhe_rns_ntt *h, fhe_dev::u256 *dst, const fhe_dev::u256 *src, uint32_t polys, bool forward, uint32_t scale) {
    const uint32_t top = wide_top_stages(h);
    const bool tiled = top < h->log_n;
    if (!h->log_n) {                                        // degree 1: the transforms are the identity (stand-alone inverse: n^-1 = 1)
        if (dst != src) HIP_TRY(hipMemcpyAsync(dst, src, (size_t)polys * 32, hipMemcpyDeviceToDevice, h->stream));
        if (!forward && scale == 2) wide_inverse_top<NL>(h, dst, polys, 2);
        return post_launch(h->stream, "wide identity");
    }
    if (forward) {
        wide_forward_top<NL>(h, dst, src, polys);
        if (tiled) wide_tile<NL, fhe_dev::TILE_FWD>(h, dst, top ? dst : src, nullptr, polys, 0);
    } else {
        if (tiled) wide_tile<NL, fhe_dev::TILE_INV>(h, dst, src, nullptr, polys, top ? 0 : scale);
        else if (dst != src) HIP_TRY(hipMemcpyAsync(dst, src, (size_t)polys * h->n * 32, hipMemcpyDeviceToDevice, h->stream));
        wide_inverse_top<NL>(h, dst, polys, scale);
    }
    return post_launch(h->stream, forward ? "wide forward" : "wide inverse");
}
// src -> dst (in place when equal); inverse: scale 1 = n^-1 (stand-alone), 2 = n^-1 R (inputs carry the R^-1 of a fused pointwise product)
static int run256_transform(fhe_rns_ntt *h, fhe_dev::u256 *dst, const fhe_dev::u256 *src, uint32_t polys, bool forward, uint32_t scale = 1) {
    return h->wide_nl == 2 ? wide_transform_t<2>(h, dst, src, polys, forward, scale) : wide_transform_t<4>(h, dst, src, polys, forward, scale);
}
static int run256_transform(fhe_rns_ntt *h, fhe_dev::u256 *data, uint32_t polys, bool forward) { return run256_transform(h, data, data, polys, forward, 1); }

template <int OP>
static int run256_ew(fhe_rns_ntt *h, void *r, const void *a, const void *b, uint32_t polys, const char *what) {
    size_t count = (size_t)polys * h->n;
    hipLaunchKernelGGL(fhe_dev::ew256_rns_kernel<OP>, dim3(ew_grid(count)), dim3(256), 0, h->stream, (fhe_dev::u256 *)r,
                       (const fhe_dev::u256 *)a, (const fhe_dev::u256 *)b, (const fhe_dev::Limb256 *)h->d_limbs, h->L, h->log_n, count);
    return post_launch(h->stream, what);
}

// ---- LDS-resident path launchers (kernels live in lds_inst.hip, one object per (field, log2 n)) ---------
static int lds_width_id(const fhe_rns_ntt *h) {
    return h->width == FHE_WIDTH_32 ? 32 : h->width == FHE_WIDTH_52 ? 52 : h->width == FHE_WIDTH_64 ? 64 : 65;
}
// Two-pass transforms of the word-sized classes (log2 n = 13 + sub_top): one launch of the LOGN = 13 instance per pass.  Between the
// two launches the polynomials are COMPACT (sizeof(residue) bytes per coefficient, d_ws3); *_compact tell which pointers are.
static size_t residue_bytes(const fhe_rns_ntt *h) { return h->width == FHE_WIDTH_32 ? 4 : 8; }
static int lds_big(fhe_rns_ntt *h, int op, void *dst, bool dst_compact, const void *src, bool src_compact, const void *src2, uint32_t polys, bool rconst,
                   const char *what) {
    fhe_dev::lds_launch_fn fn = fhe_dev::lds_lookup(lds_width_id(h), 13);
    if (!fn) return fail(FHE_ERR_UNSUPPORTED, "no LDS instance for the two-pass transform");
    const uint32_t chunk_max = (65535u / h->L) * h->L;      // grid.y of the pass kernel; chunks keep the limb phase
    const size_t dstep = (size_t)h->n * (dst_compact ? residue_bytes(h) : 32), sstep = (size_t)h->n * (src_compact ? residue_bytes(h) : 32);
    for (uint32_t done = 0; done < polys;) {
        const uint32_t chunk = polys - done < chunk_max ? polys - done : chunk_max;
        fhe_dev::LdsArgs A{op, (char *)dst + done * dstep, nullptr, nullptr, (const char *)src + done * sstep, nullptr,
                           src2 ? (const char *)src2 + done * sstep : nullptr, nullptr, h->d_limbs, h->L, chunk, h->stream};
        A.top = h->sub_top; A.rconst = rconst;
        fn(A);
        done += chunk;
    }
    return post_launch(h->stream, what);
}
static int big_forward(fhe_rns_ntt *h, void *dst, const void *src, uint32_t polys) {
    int rc = ensure_ws3(h, (size_t)polys * h->n * residue_bytes(h)); if (rc) return rc;
    if ((rc = lds_big(h, fhe_dev::LDS_PASS_FWD, h->d_ws3, true, src, false, nullptr, polys, false, "word_pass_kernel"))) return rc;
    return lds_big(h, fhe_dev::LDS_SUB_FORWARD, dst, false, h->d_ws3, true, nullptr, polys, false, "ntt_sub_kernel");
}
static int big_inverse(fhe_rns_ntt *h, void *data, uint32_t polys) {
    int rc = ensure_ws3(h, (size_t)polys * h->n * residue_bytes(h)); if (rc) return rc;
    if ((rc = lds_big(h, fhe_dev::LDS_SUB_INVERSE, h->d_ws3, true, data, false, nullptr, polys, false, "ntt_sub_kernel"))) return rc;
    return lds_big(h, fhe_dev::LDS_PASS_INV, data, false, h->d_ws3, true, nullptr, polys, false, "word_pass_kernel");
}

// Few ciphertexts on the paired key-switch kernel: one workgroup per digit pair and a combining launch (ntt_lds_small.hip.h) -- sets A.pair_ws.
static int split_pairs_workspace(fhe_rns_ntt *h, fhe_dev::LdsArgs &A) {
    if (h->width != FHE_WIDTH_32 || h->single_transforms || !fhe_dev::lds_paired_keyswitch(4, (int)h->log_n)) return FHE_OK;
    const uint32_t NP = (h->L * A.K + 1) / 2;
    if (NP < 2 || A.polys > h->split_pairs_polys) return FHE_OK;
    // partial accumulators: one pair per digit PAIR, or (N <= 2^13: the 16-per-thread form, one workgroup per digit) one pair per digit
    int rc = ensure_ws(h, 2 * (size_t)A.polys * (fhe_dev::lds_small_multiply(4, (int)h->log_n) ? h->L * A.K : NP) * h->n * 4); if (rc) return rc;
    A.pair_ws = h->d_ws;
    return FHE_OK;
}

// Key switch / external product of the 8-byte residues (and of the 4-byte residues at N = 2^15): ONE workgroup per (ciphertext, limb)
// with three live arrays (ntt_keyswitch3_kernel / ntt_extprod3_kernel) instead of the split form -- everywhere it was faster in the
// interleaved A/B (scripts/ab_keyswitch3.sh, ab_extprod3.sh, ab_joint3_small.sh; with the descriptor loads of round 2): every size for
// the lazy 64-bit field (+4...+60 %) and for the FP64 field's external product and compact-operand key switch (+10...+46 %); the FP64
// field's stand-alone key switch (container operands) from N = 2^13 (3-6 % behind at N <= 4096); the full-range 64-bit field from
// N = 2^12 (at N = 2048: key switch -3 %, external product -16 %).
static bool use_joint3(const fhe_rns_ntt *h, bool extprod, bool compact) {
    const int eb = h->width == FHE_WIDTH_32 ? 4 : 8;
    if (h->split_keyswitch || !fhe_dev::lds_keyswitch_joint3(eb, (int)h->log_n)) return false;
    if (h->width == FHE_WIDTH_52) return extprod || compact || h->log_n >= 13;
    if (h->width == FHE_WIDTH_64X) return h->log_n >= 12;
    return true;
}
// Tensor product in two launches (ntt_forward_compact_kernel + ntt_ct_a_kernel, workspace for the transformed b-side) instead of the
// one-launch kernel: always where that kernel does not exist (8-byte residues at N = 2^14, N = 2^15), and for the 8-byte residues
// where the interleaved A/B favoured it (scripts/ab_ct_form.sh, one MI355X, batch 1024, N = 8192 / 4096 / 2048): the FP64 field
// (tensor product +32 / +35 / +26 %, full multiply +11 / +10 / +2 %) and the stand-alone tensor product of the full-range 64-bit
// field (+13 / +11 / +12 %; inside the full multiply 0 / -9 / -2 %); the lazy 64-bit field keeps its one-launch kernel (-1 / +6 / -4 %).
// The squaring forms stay on the one-launch kernel (5 transforms).  FHE_HIP_CT_FORM=one|two forces a form where both exist.
static int ct_workspace(fhe_rns_ntt *h, fhe_dev::LdsArgs &A) {
    const int eb = h->width == FHE_WIDTH_32 ? 4 : 8;
    if (h->no_two_launch_ct || !fhe_dev::lds_ct_two_launch(eb, (int)h->log_n)) return FHE_OK;
    if (fhe_dev::lds_ct_fused(eb, (int)h->log_n)) {
        bool want = h->width == FHE_WIDTH_52 || (h->width == FHE_WIDTH_64X && !A.compact_c2);
        if (h->ct_form_force) want = h->ct_form_force == 2;       // FHE_HIP_CT_FORM=one|two (A/B, cross-check)
        if (!want || A.square) return FHE_OK;
    }
    int rc = ensure_ws(h, 2 * (size_t)A.polys * h->n * eb); if (rc) return rc;
    A.ws = h->d_ws;
    return FHE_OK;
}
static int lds_run(fhe_rns_ntt *h, int op, void *r0, void *r1, void *r2, const void *a0, const void *a1, const void *b0,
                   const void *b1, uint32_t polys, const char *what, uint32_t b_polys = 0) {
    fhe_dev::lds_launch_fn fn = fhe_dev::lds_lookup(lds_width_id(h), (int)h->log_n);
    if (!fn) return fail(FHE_ERR_UNSUPPORTED, "transform size outside the LDS-resident range");
    fhe_dev::LdsArgs A{op, r0, r1, r2, a0, a1, b0, b1, h->d_limbs, h->L, polys, h->stream};
    A.single_transforms = h->single_transforms;
    A.b_polys = b_polys;
    A.small_batch = op == fhe_dev::LDS_MULTIPLY && polys <= h->small_batch_polys;
    if (op == fhe_dev::LDS_MULTIPLY && polys <= h->coop_polys && fhe_dev::lds_coop4_multiply(h->width == FHE_WIDTH_32 ? 4 : 8, (int)h->log_n)) {
        // (in place is fine: every operand container is read by the first launch, the result containers are written by the third)
        int rc = ensure_ws3(h, 3 * (size_t)polys * h->n * 4); if (rc) return rc;
        A.coop_ws = h->d_ws3;
    }
    A.square = !b_polys && !h->no_square &&
               ((op == fhe_dev::LDS_MULTIPLY && a0 == b0) || (op == fhe_dev::LDS_CT_MULTIPLY && a0 == b0 && a1 == b1));
    if (op == fhe_dev::LDS_CT_MULTIPLY) { int rc = ct_workspace(h, A); if (rc) return rc; }
    fn(A);
    return post_launch(h->stream, what);
}
template <class F, int OP>
static int lds_ew(fhe_rns_ntt *h, void *r, const void *a, const void *b, uint32_t polys, const char *what) {
    using V = typename F::V16;
    size_t halves = (size_t)polys * h->n * 2;
    hipLaunchKernelGGL((fhe_dev::ew_kernel<F, OP>), dim3(ew_grid(halves)), dim3(256), 0, h->stream, (V *)r, (const V *)a,
                       (const V *)b, (const fhe_dev::Limb<F> *)h->d_limbs, h->L, h->log_n, halves);
    return post_launch(h->stream, what);
}
template <class F>
static int lds_check(fhe_rns_ntt *h, const void *d, uint32_t polys) {
    using V = typename F::V16;
    size_t halves = (size_t)polys * h->n * 2;
    hipLaunchKernelGGL((fhe_dev::check_kernel<F>), dim3(ew_grid(halves)), dim3(256), 0, h->stream, (const V *)d,
                       (const fhe_dev::Limb<F> *)h->d_limbs, h->L, h->log_n, halves, h->d_flag);
    return post_launch(h->stream, "check_kernel");
}

template <class F>
static int compact_poly_t(fhe_rns_ntt *h, void *out, const void *in, size_t containers) {
    hipLaunchKernelGGL((fhe_dev::compact_kernel<F>), dim3(ew_grid(containers)), dim3(256), 0, h->stream, (typename F::E *)out, (const typename F::V16 *)in, containers);
    return post_launch(h->stream, "compact_kernel");
}
static int compact_poly(fhe_rns_ntt *h, void *out, const void *in, size_t containers) {
    switch (h->width) {
        case FHE_WIDTH_32: return compact_poly_t<fhe_dev::F32>(h, out, in, containers);
        case FHE_WIDTH_52: return compact_poly_t<fhe_dev::F52>(h, out, in, containers);
        case FHE_WIDTH_64: return compact_poly_t<fhe_dev::F64>(h, out, in, containers);
        case FHE_WIDTH_64X: return compact_poly_t<fhe_dev::F64X>(h, out, in, containers);
        default: return fail(FHE_ERR_UNSUPPORTED, "compact polynomials exist on the word-sized classes only");
    }
}

template <class F>
static int monomial_compact_t(fhe_rns_ntt *h, void *out, const void *in, const uint32_t *shifts, size_t count) {
    hipLaunchKernelGGL((fhe_dev::monomial_compact_kernel<F>), dim3(ew_grid(count)), dim3(256), 0, h->stream, (typename F::E *)out, (const typename F::E *)in, shifts,
                       (const fhe_dev::Limb<F> *)h->d_limbs, h->L, h->log_n, count);
    return post_launch(h->stream, "monomial_compact_kernel");
}
static int monomial_compact(fhe_rns_ntt *h, void *out, const void *in, const uint32_t *shifts, size_t count) {
    switch (h->width) {
        case FHE_WIDTH_32: return monomial_compact_t<fhe_dev::F32>(h, out, in, shifts, count);
        case FHE_WIDTH_52: return monomial_compact_t<fhe_dev::F52>(h, out, in, shifts, count);
        case FHE_WIDTH_64: return monomial_compact_t<fhe_dev::F64>(h, out, in, shifts, count);
        case FHE_WIDTH_64X: return monomial_compact_t<fhe_dev::F64X>(h, out, in, shifts, count);
        default: return fail(FHE_ERR_UNSUPPORTED, "compact polynomials exist on the word-sized classes only");
    }
}

static int do_forward(fhe_rns_ntt *h, void *d_data, uint32_t batch) {
    const uint32_t polys = batch * h->L;
    if (h->sub_top) return big_forward(h, d_data, d_data, polys);
    if (h->width != FHE_WIDTH_256)
        return lds_run(h, fhe_dev::LDS_FORWARD, d_data, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, polys, "ntt_forward_kernel");
    return run256_transform(h, (fhe_dev::u256 *)d_data, polys, true);
}
static int do_inverse(fhe_rns_ntt *h, void *d_data, uint32_t batch) {
    const uint32_t polys = batch * h->L;
    if (h->sub_top) return big_inverse(h, d_data, polys);
    if (h->width != FHE_WIDTH_256)
        return lds_run(h, fhe_dev::LDS_INVERSE, d_data, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, polys, "ntt_inverse_kernel");
    return run256_transform(h, (fhe_dev::u256 *)d_data, polys, false);
}
template <int OP>
static int do_ew(fhe_rns_ntt *h, void *r, const void *a, const void *b, uint32_t batch, const char *what) {
    const uint32_t polys = batch * h->L;
    if (h->width == FHE_WIDTH_32) return lds_ew<fhe_dev::F32, OP>(h, r, a, b, polys, what);
    if (h->width == FHE_WIDTH_52) return lds_ew<fhe_dev::F52, OP>(h, r, a, b, polys, what);
    if (h->width == FHE_WIDTH_64) return lds_ew<fhe_dev::F64, OP>(h, r, a, b, polys, what);
    if (h->width == FHE_WIDTH_64X) return lds_ew<fhe_dev::F64X, OP>(h, r, a, b, polys, what);
    return run256_ew<OP>(h, r, a, b, polys, what);
}
// Full-width multiply without operand copies (the reference copies both operands first, src/ntt.cu:50-58): the top forward stages
// write the transformed operands into the workspace, one fused tile launch does the remaining forward stages of both operands, the
// pointwise product and the low inverse stages, the trailing inverse stages finish in place on the result.
template <int NL>
static int wide_multiply_t(fhe_rns_ntt *h, void *d_r, const void *d_a, const void *d_b, uint32_t polys) {
    using fhe_dev::u256;
    const uint32_t top = wide_top_stages(h);
    const bool tiled = top < h->log_n;
    const size_t bytes = (size_t)polys * h->n * 32;
    const u256 *A = (const u256 *)d_a, *B = (const u256 *)d_b;
    if (!h->log_n) return run256_ew<0>(h, d_r, d_a, d_b, polys, "pointwise");     // degree 1: the product in Z_q
    if (top) {
        int rc = ensure_ws(h, 2 * bytes); if (rc) return rc;
        u256 *wa = (u256 *)h->d_ws, *wb = (u256 *)((char *)h->d_ws + bytes);
        wide_forward_top<NL>(h, wa, A, polys);
        if (d_b != d_a) wide_forward_top<NL>(h, wb, B, polys);
        B = d_b != d_a ? wb : wa; A = wa;
    }
    if (tiled) {
        wide_tile<NL, fhe_dev::TILE_MUL>(h, (u256 *)d_r, A, B, polys, top ? 0 : 2);
    } else {                                                // n < 2^11 (or FHE_HIP_NO_WIDE_TILES): pointwise product of the transformed copies
        const size_t count = (size_t)polys * h->n;
        hipLaunchKernelGGL((fhe_dev::wide_pointwise_kernel<NL>), dim3(ew_grid(count)), dim3(256), 0, h->stream, (u256 *)d_r, A, B,
                           (const fhe_dev::WLimb<NL> *)h->d_wlimbs, h->L, h->log_n, count);
    }
    wide_inverse_top<NL>(h, (u256 *)d_r, polys, 2);
    return post_launch(h->stream, "wide multiply");
}
template <int NL>
static int wide_ct_multiply_t(fhe_rns_ntt *h, void *c0, void *c1, void *c2, const void *a0, const void *a1, const void *b0, const void *b1, uint32_t polys) {
    using fhe_dev::u256;
    // 4 forward transforms into the workspace (no copies), one pass for the three NTT-domain products, 3 inverse transforms
    const size_t bytes = (size_t)polys * h->n * 32;
    int rc = ensure_ws(h, 4 * bytes); if (rc) return rc;
    char *ws = (char *)h->d_ws;
    const void *src[4] = {a0, a1, b0, b1};
    for (int i = 0; i < 4; i++)
        if ((rc = run256_transform(h, (u256 *)(ws + i * bytes), (const u256 *)src[i], polys, true, 0))) return rc;
    const size_t count = (size_t)polys * h->n;
    hipLaunchKernelGGL((fhe_dev::wide_ct_pointwise_kernel<NL>), dim3(ew_grid(count)), dim3(256), 0, h->stream, (u256 *)c0, (u256 *)c1, (u256 *)c2,
                       (const u256 *)ws, (const u256 *)(ws + bytes), (const u256 *)(ws + 2 * bytes), (const u256 *)(ws + 3 * bytes),
                       (const fhe_dev::WLimb<NL> *)h->d_wlimbs, h->L, h->log_n, count);
    for (void *c : {c0, c1, c2})
        if ((rc = run256_transform(h, (u256 *)c, (const u256 *)c, polys, false, 2))) return rc;
    return FHE_OK;
}
static int do_multiply(fhe_rns_ntt *h, void *d_r, const void *d_a, const void *d_b, uint32_t batch) {
    const uint32_t polys = batch * h->L;
    // d_r may alias d_a and/or d_b, as in the reference (which copies its operands first, src/ntt.cu:50-58): every
    // workgroup loads both of its operand polynomials completely before its first store, and the general path works on copies.
    if (h->sub_top) {   // two-pass: the top stages of both operands go to the COMPACT workspace, one fused launch over the 2^13 blocks (compact in and out), last pass into the result
        const size_t cbytes = (size_t)polys * h->n * residue_bytes(h);
        int rc = ensure_ws3(h, 2 * cbytes); if (rc) return rc;
        char *wa = (char *)h->d_ws3, *wb = d_b != d_a ? wa + cbytes : wa;
        if ((rc = lds_big(h, fhe_dev::LDS_PASS_FWD, wa, true, d_a, false, nullptr, polys, false, "word_pass_kernel"))) return rc;
        if (d_b != d_a && (rc = lds_big(h, fhe_dev::LDS_PASS_FWD, wb, true, d_b, false, nullptr, polys, false, "word_pass_kernel"))) return rc;
        if ((rc = lds_big(h, fhe_dev::LDS_SUB_MULTIPLY, wa, true, wa, true, wb, polys, false, "ntt_sub_kernel"))) return rc;
        return lds_big(h, fhe_dev::LDS_PASS_INV, d_r, false, wa, true, nullptr, polys, true, "word_pass_kernel");
    }
    if (h->width != FHE_WIDTH_256)
        return lds_run(h, fhe_dev::LDS_MULTIPLY, d_r, nullptr, nullptr, d_a, nullptr, d_b, nullptr, polys, "ntt_multiply_kernel");
    return h->wide_nl == 2 ? wide_multiply_t<2>(h, d_r, d_a, d_b, polys) : wide_multiply_t<4>(h, d_r, d_a, d_b, polys);
}
static int do_ct_multiply(fhe_rns_ntt *h, void *c0, void *c1, void *c2, const void *a0, const void *a1, const void *b0,
                          const void *b1, uint32_t batch) {
    const uint32_t polys = batch * h->L;
    if (h->sub_top) {   // 4 forward transforms into the workspace, NTT-domain products on the field type, 3 inverse transforms
        const size_t bytes = (size_t)polys * h->n * 32;
        int rc = ensure_ws(h, 5 * bytes); if (rc) return rc;
        char *ws = (char *)h->d_ws, *T = ws + 4 * bytes;
        const void *src[4] = {a0, a1, b0, b1};
        for (int i = 0; i < 4; i++) if ((rc = big_forward(h, ws + i * bytes, src[i], polys))) return rc;
        if ((rc = do_ew<0>(h, c0, ws, ws + 2 * bytes, batch, "ct pointwise"))) return rc;
        if ((rc = do_ew<0>(h, c1, ws, ws + 3 * bytes, batch, "ct pointwise"))) return rc;
        if ((rc = do_ew<0>(h, T, ws + bytes, ws + 2 * bytes, batch, "ct pointwise"))) return rc;
        if ((rc = do_ew<1>(h, c1, c1, T, batch, "ct add"))) return rc;
        if ((rc = do_ew<0>(h, c2, ws + bytes, ws + 3 * bytes, batch, "ct pointwise"))) return rc;
        for (void *c : {c0, c1, c2}) if ((rc = big_inverse(h, c, polys))) return rc;
        return FHE_OK;
    }
    if (h->width != FHE_WIDTH_256)
        return lds_run(h, fhe_dev::LDS_CT_MULTIPLY, c0, c1, c2, a0, a1, b0, b1, polys, "ntt_ct_multiply_kernel");
    return h->wide_nl == 2 ? wide_ct_multiply_t<2>(h, c0, c1, c2, a0, a1, b0, b1, polys) : wide_ct_multiply_t<4>(h, c0, c1, c2, a0, a1, b0, b1, polys);
}

// ------------------------------------------------------------------------------------------------------
// RNS engine ABI
// ------------------------------------------------------------------------------------------------------
// FHE_HIP_CHECK_INPUTS=1 (read at engine creation): every compute entry point first scans its operands for coefficients that are not
// canonical residues (value >= q_l or non-zero upper words) and returns FHE_ERR_NONCANONICAL instead of computing on them -- the
// word-sized classes read only the low word(s) of a container, so such an operand would otherwise give a silently different
// product than the reference's full 256-bit arithmetic.  Debugging aid: one extra read of every operand and a stream sync per call.
extern "C" int fhe_rns_check_canonical(fhe_rns_ntt_t *h, const void *d_data, uint32_t batch);
static int check_inputs(fhe_rns_ntt *h, std::initializer_list<const void *> operands, uint32_t batch) {
    if (!h->check_inputs) return FHE_OK;
    for (const void *p : operands) { int rc = fhe_rns_check_canonical(h, p, batch); if (rc) return rc; }
    return FHE_OK;
}

extern "C" int fhe_rns_ntt_create(fhe_rns_ntt_t **out, uint32_t n, const uint64_t (*moduli)[4], uint32_t num_primes) {
    return create_impl(out, n, moduli, num_primes);
}
extern "C" int fhe_rns_base_create(fhe_rns_ntt_t **out, const uint64_t (*primes)[4], uint32_t num_primes) {
    return create_impl(out, 1, primes, num_primes, true);
}
extern "C" int fhe_rns_ntt_destroy(fhe_rns_ntt_t *h) { destroy_impl(h); return FHE_OK; }
extern "C" int fhe_rns_ntt_set_stream(fhe_rns_ntt_t *h, void *stream) {
    if (!h) return fail(FHE_ERR_INVALID_ARG, "null handle");
    h->stream = stream ? (hipStream_t)stream : h->own_stream;
    return FHE_OK;
}
// Pre-sizes the library-owned workspaces for calls of up to `batch` units, so that no later call allocates (hipMalloc synchronises
// and cannot be captured into a hipGraph): the compact / container workspace of fhe_ct_multiply_relin and fhe_blind_rotate, and the
// transform workspace of the general paths (full-width class, two-pass sizes).  Relinearisation on the general path sizes its digit
// workspace by itself (bounded to 1 GiB, chunked).
extern "C" int fhe_rns_ntt_reserve(fhe_rns_ntt_t *h, uint32_t batch) {
    // The union of what every entry point asks of ensure_ws / ws2 / ws3 for `batch` units (each site cited), so that none of them allocates
    // afterwards.  The key-switch workspaces depend on the digit count: the largest K of the key sets imported so far (import keys first).
    int rc = check_call(h, batch, "reserve"); if (rc) return rc;
    const size_t polys = (size_t)batch * h->L, S = (size_t)h->L * h->n * 32, eb = residue_bytes(h), cbytes = polys * h->n * eb;
    const bool lds_class = h->width != FHE_WIDTH_256 && !h->sub_top;
    const uint32_t LK = h->L * (h->max_digits ? h->max_digits : 1);
    size_t ws = 0, ws2 = (size_t)batch * S, ws3 = 0;                   // ws2: one container component (c2 of multiply + relinearise on the general path) ...
    if (6 * cbytes > ws2) ws2 = 6 * cbytes;                            // ... or 4 compact accumulators + 2 rotated ones of a blind-rotation loop on the 8-byte fields
    if (!lds_class) ws = 5 * (size_t)batch * S;                        // 4 transformed operands + one product (tensor product of the general / two-pass paths)
    else ws = 2 * cbytes;                                              // transformed b-side of the two-launch tensor product (2 compact components)
    if (h->max_composed_digits) {                                      // digit polynomials + two accumulators of the composed key switch (key sets without packed tables)
        const size_t LK = (size_t)h->L * h->max_composed_digits;
        size_t chunk = ((size_t)1 << 30) / ((LK + 2) * S); if (chunk < 1) chunk = 1; if (chunk > batch) chunk = batch;
        if ((LK + 2) * chunk * S > ws) ws = (LK + 2) * chunk * S;
    }
    // (the few-ciphertext forms are taken by every call of at most split_pairs_polys / coop_polys limb polynomials: a smaller batch than the reserved one included)
    if (lds_class && h->width == FHE_WIDTH_32 && h->split_pairs_polys) {            // few ciphertexts: one workgroup per digit pair, partial sums in the workspace
        // N <= 2^13: one pair per digit, and two digit sources in a blind-rotation step (2 L K partials per limb polynomial)
        const size_t NP = fhe_dev::lds_small_multiply(4, (int)h->log_n) ? 2 * LK : 2 * ((LK + 1) / 2), few = polys < h->split_pairs_polys ? polys : h->split_pairs_polys;
        if (2 * few * NP * h->n * 4 > ws) ws = 2 * few * NP * h->n * 4;
    }
    if (h->sub_top) ws3 = 2 * cbytes;                                  // two compact operands of a two-pass multiply
    if (lds_class && h->coop_polys && fhe_dev::lds_coop4_multiply((int)eb, (int)h->log_n))
        ws3 = 7 * (polys < h->coop_polys ? polys : h->coop_polys) * h->n * 4;       // few polynomials over four workgroups each: 3 (multiply) / 7 (tensor product) block images per limb polynomial
    if ((rc = ensure_ws(h, ws)) || (rc = ensure_ws2(h, ws2)) || (ws3 && (rc = ensure_ws3(h, ws3)))) return rc;
    return ensure_aux_stream(h);                                       // second stream + events of the chunked pipelines
}
extern "C" int fhe_rns_ntt_workspace_bytes(const fhe_rns_ntt_t *h, uint64_t *bytes) {
    if (!h || !bytes) return fail(FHE_ERR_INVALID_ARG, "workspace_bytes: null argument");
    *bytes = (uint64_t)h->ws_bytes + h->ws2_bytes + h->ws3_bytes;
    return FHE_OK;
}
extern "C" int fhe_rns_ntt_width_class(const fhe_rns_ntt_t *h) { return h ? h->width : fail(FHE_ERR_INVALID_ARG, "null handle"); }
extern "C" int fhe_rns_ntt_forward(fhe_rns_ntt_t *h, void *d_data, uint32_t batch) {
    int rc = check_call(h, batch, "forward"); if (rc) return rc;
    if (!d_data) return fail(FHE_ERR_INVALID_ARG, "forward: null data");
    if ((rc = check_inputs(h, {d_data}, batch))) return rc;
    return do_forward(h, d_data, batch);
}
extern "C" int fhe_rns_ntt_inverse(fhe_rns_ntt_t *h, void *d_data, uint32_t batch) {
    int rc = check_call(h, batch, "inverse"); if (rc) return rc;
    if (!d_data) return fail(FHE_ERR_INVALID_ARG, "inverse: null data");
    if ((rc = check_inputs(h, {d_data}, batch))) return rc;
    return do_inverse(h, d_data, batch);
}
extern "C" int fhe_rns_ntt_pointwise(fhe_rns_ntt_t *h, void *r, const void *a, const void *b, uint32_t batch) {
    int rc = check_call(h, batch, "pointwise"); if (rc) return rc;
    if (!r || !a || !b) return fail(FHE_ERR_INVALID_ARG, "pointwise: null argument");
    return do_ew<0>(h, r, a, b, batch, "pointwise");
}
extern "C" int fhe_rns_ntt_multiply(fhe_rns_ntt_t *h, void *r, const void *a, const void *b, uint32_t batch) {
    int rc = check_call(h, batch, "multiply"); if (rc) return rc;
    if (!r || !a || !b) return fail(FHE_ERR_INVALID_ARG, "multiply: null argument");
    if ((rc = check_inputs(h, {a, b}, batch))) return rc;
    return do_multiply(h, r, a, b, batch);
}
extern "C" int fhe_rns_ntt_multiply_bcast(fhe_rns_ntt_t *h, void *r, const void *a, const void *b_one, uint32_t batch) {
    int rc = check_call(h, batch, "multiply_bcast"); if (rc) return rc;
    if (!r || !a || !b_one) return fail(FHE_ERR_INVALID_ARG, "multiply_bcast: null argument");
    if (r == b_one) return fail(FHE_ERR_INVALID_ARG, "multiply_bcast: the result must not overwrite the shared operand");
    if (h->width != FHE_WIDTH_256 && !h->sub_top)   // every workgroup reads limb (p % L) of the one shared polynomial: L2 hits after the first use
        return lds_run(h, fhe_dev::LDS_MULTIPLY, r, nullptr, nullptr, a, nullptr, b_one, nullptr, batch * h->L, "ntt_multiply_kernel", h->L);
    const size_t S = (size_t)h->L * h->n * 32;
    for (uint32_t i = 0; i < batch; i++)
        if ((rc = do_multiply(h, (char *)r + i * S, (const char *)a + i * S, b_one, 1))) return rc;
    return FHE_OK;
}
extern "C" int fhe_rns_poly_add(fhe_rns_ntt_t *h, void *r, const void *a, const void *b, uint32_t batch) {
    int rc = check_call(h, batch, "poly_add"); if (rc) return rc;
    if (!r || !a || !b) return fail(FHE_ERR_INVALID_ARG, "poly_add: null argument");
    return do_ew<1>(h, r, a, b, batch, "poly_add");
}
extern "C" int fhe_rns_mul_mont_literal(fhe_rns_ntt_t *h, void *r, const void *a, const void *b, uint32_t batch) {
    int rc = check_call(h, batch, "mul_mont_literal"); if (rc) return rc;
    if (!r || !a || !b) return fail(FHE_ERR_INVALID_ARG, "mul_mont_literal: null argument");
    if (h->width != FHE_WIDTH_256) return fail(FHE_ERR_UNSUPPORTED, "mul_mont_literal: R = 2^256 Montgomery products exist on full-width handles only "
                                                                    "(an RNS base from fhe_rns_base_create, or FHE_HIP_FORCE_WIDTH=256)");
    return run256_ew<3>(h, r, a, b, batch * h->L, "mul_mont_literal");
}
extern "C" int fhe_rns_poly_sub(fhe_rns_ntt_t *h, void *r, const void *a, const void *b, uint32_t batch) {
    int rc = check_call(h, batch, "poly_sub"); if (rc) return rc;
    if (!r || !a || !b) return fail(FHE_ERR_INVALID_ARG, "poly_sub: null argument");
    return do_ew<2>(h, r, a, b, batch, "poly_sub");
}
extern "C" int fhe_ct_multiply(fhe_rns_ntt_t *h, void *c0, void *c1, void *c2, const void *a0, const void *a1,
                               const void *b0, const void *b1, uint32_t batch) {
    int rc = check_call(h, batch, "ct_multiply"); if (rc) return rc;
    if (!c0 || !c1 || !c2 || !a0 || !a1 || !b0 || !b1) return fail(FHE_ERR_INVALID_ARG, "ct_multiply: null argument");
    const void *ins[4] = {a0, a1, b0, b1}; void *outs[3] = {c0, c1, c2};
    for (void *o : outs) for (const void *i : ins) if (o == i) return fail(FHE_ERR_INVALID_ARG, "ct_multiply: outputs must not alias inputs");
    if (c0 == c1 || c0 == c2 || c1 == c2) return fail(FHE_ERR_INVALID_ARG, "ct_multiply: outputs must be distinct");
    if ((rc = check_inputs(h, {a0, a1, b0, b1}, batch))) return rc;
    return do_ct_multiply(h, c0, c1, c2, a0, a1, b0, b1, batch);
}
extern "C" int fhe_rns_check_canonical(fhe_rns_ntt_t *h, const void *d_data, uint32_t batch) {
    int rc = check_call(h, batch, "check_canonical"); if (rc) return rc;
    if (!d_data) return fail(FHE_ERR_INVALID_ARG, "check_canonical: null data");
    const uint32_t polys = batch * h->L;
    HIP_TRY(hipMemsetAsync(h->d_flag, 0, sizeof(uint32_t), h->stream));
    if (h->width == FHE_WIDTH_32) rc = lds_check<fhe_dev::F32>(h, d_data, polys);
    else if (h->width == FHE_WIDTH_52) rc = lds_check<fhe_dev::F52>(h, d_data, polys);
    else if (h->width == FHE_WIDTH_64) rc = lds_check<fhe_dev::F64>(h, d_data, polys);
    else if (h->width == FHE_WIDTH_64X) rc = lds_check<fhe_dev::F64X>(h, d_data, polys);
    else {
        size_t count = (size_t)polys * h->n;
        hipLaunchKernelGGL(fhe_dev::check256_kernel, dim3(ew_grid(count)), dim3(256), 0, h->stream, (const fhe_dev::u256 *)d_data,
                           (const fhe_dev::Limb256 *)h->d_limbs, h->L, h->log_n, count, h->d_flag);
        rc = post_launch(h->stream, "check256_kernel");
    }
    if (rc) return rc;
    uint32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, h->d_flag, sizeof flag, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return flag ? fail(FHE_ERR_NONCANONICAL, "buffer holds coefficients that are not canonical residues of their limb modulus") : FHE_OK;
}


// ------------------------------------------------------------------------------------------------------
// relinearisation / key switching (general path: digit embedding -> batched forward NTT -> MAC -> inverse)
// ------------------------------------------------------------------------------------------------------
struct fhe_relin_keys {
    fhe_rns_ntt *owner = nullptr;
    uint32_t decomp_bits = 0, K = 0, num_keys = 0;
    void *d_kb = nullptr, *d_ka = nullptr;      // [num_keys][L][n] containers, NTT domain, canonical
    void *d_pkb = nullptr, *d_pka = nullptr;    // packed tables for the fused key-switch kernel (32-bit path only)
};

static uint32_t relin_digits(const fhe_rns_ntt *h, uint32_t w) {
    int mx = 0;
    for (const U256 &q : h->moduli) mx = q.bit_length() > mx ? q.bit_length() : mx;
    return ((uint32_t)mx + w - 1) / w;
}
extern "C" int fhe_relin_num_digits(const fhe_rns_ntt_t *h, uint32_t decomp_bits, uint32_t *digits) {
    if (!h || !digits) return fail(FHE_ERR_INVALID_ARG, "null argument");
    if (decomp_bits < 1 || decomp_bits > 64) return fail(FHE_ERR_INVALID_ARG, "decomp_bits must be in [1, 64]");
    *digits = relin_digits(h, decomp_bits);
    return FHE_OK;
}
extern "C" int fhe_relin_keys_destroy(fhe_relin_keys_t *rk) {
    if (rk) {
        for (void *p : {rk->d_kb, rk->d_ka, rk->d_pkb, rk->d_pka}) if (p) (void)hipFree(p);
        delete rk;
    }
    return FHE_OK;
}

// packed key tables for the fused key-switch kernels (word-sized paths)
template <class F>
static int pack_relin_keys_t(fhe_rns_ntt *h, fhe_relin_keys *rk) {
    const size_t elems = (size_t)rk->num_keys * h->L * h->n;
    HIP_TRY(hipMalloc(&rk->d_pkb, elems * sizeof(typename F::E)));
    HIP_TRY(hipMalloc(&rk->d_pka, elems * sizeof(typename F::E)));
    hipLaunchKernelGGL((fhe_dev::pack_keys_kernel<F>), dim3(ew_grid(elems)), dim3(256), 0, h->stream, (typename F::E *)rk->d_pkb,
                       (const typename F::V16 *)rk->d_kb, (const fhe_dev::Limb<F> *)h->d_limbs, h->L, h->log_n, rk->num_keys);
    hipLaunchKernelGGL((fhe_dev::pack_keys_kernel<F>), dim3(ew_grid(elems)), dim3(256), 0, h->stream, (typename F::E *)rk->d_pka,
                       (const typename F::V16 *)rk->d_ka, (const fhe_dev::Limb<F> *)h->d_limbs, h->L, h->log_n, rk->num_keys);
    return post_launch(h->stream, "pack_keys_kernel");
}
static int pack_relin_keys(fhe_rns_ntt *h, fhe_relin_keys *rk) {
    if (h->width == FHE_WIDTH_32) return pack_relin_keys_t<fhe_dev::F32>(h, rk);
    if (h->width == FHE_WIDTH_52) return pack_relin_keys_t<fhe_dev::F52>(h, rk);
    if (h->width == FHE_WIDTH_64X) return pack_relin_keys_t<fhe_dev::F64X>(h, rk);
    return pack_relin_keys_t<fhe_dev::F64>(h, rk);
}

extern "C" int fhe_relin_keys_create(fhe_rns_ntt_t *h, fhe_relin_keys_t **out, uint32_t decomp_bits,
                                     const void *const *d_keys_b, const void *const *d_keys_a, uint32_t num_keys) {
    if (!h || !out || !d_keys_b || !d_keys_a) return fail(FHE_ERR_INVALID_ARG, "relin_keys_create: null argument");
    *out = nullptr;
    if (decomp_bits < 1 || decomp_bits > 64) return fail(FHE_ERR_INVALID_ARG, "decomp_bits must be in [1, 64]");
    const uint32_t K = relin_digits(h, decomp_bits);
    if (num_keys != h->L * K) {
        char buf[128]; snprintf(buf, sizeof buf, "relin_keys_create: expected %u keys (L = %u limbs x K = %u digits), got %u", h->L * K, h->L, K, num_keys);
        return fail(FHE_ERR_INVALID_ARG, buf);
    }
    for (uint32_t i = 0; i < num_keys; i++) if (!d_keys_b[i] || !d_keys_a[i]) return fail(FHE_ERR_INVALID_ARG, "relin_keys_create: null key pointer");
    (void)hipGetLastError();
    fhe_relin_keys *rk = new (std::nothrow) fhe_relin_keys();
    if (!rk) return fail(FHE_ERR_INVALID_ARG, "out of host memory");
    rk->owner = h; rk->decomp_bits = decomp_bits; rk->K = K; rk->num_keys = num_keys;
    if (K > h->max_digits) h->max_digits = K;                          // fhe_rns_ntt_reserve sizes the key-switch workspaces for it
    const size_t S = (size_t)h->L * h->n * 32;
    hipError_t e;
    if ((e = hipMalloc(&rk->d_kb, S * num_keys)) != hipSuccess || (e = hipMalloc(&rk->d_ka, S * num_keys)) != hipSuccess) {
        fhe_relin_keys_destroy(rk); return fail(FHE_ERR_HIP, std::string("relin_keys_create: ") + hipGetErrorString(e));
    }
    for (uint32_t i = 0; i < num_keys; i++) {
        if ((e = hipMemcpyAsync((char *)rk->d_kb + i * S, d_keys_b[i], S, hipMemcpyDeviceToDevice, h->stream)) != hipSuccess ||
            (e = hipMemcpyAsync((char *)rk->d_ka + i * S, d_keys_a[i], S, hipMemcpyDeviceToDevice, h->stream)) != hipSuccess) {
            fhe_relin_keys_destroy(rk); return fail(FHE_ERR_HIP, std::string("relin_keys_create copy: ") + hipGetErrorString(e));
        }
    }
    int rc = do_forward(h, rk->d_kb, num_keys);
    if (!rc) rc = do_forward(h, rk->d_ka, num_keys);
    // The fused kernels feed a digit of limb j (< min(2^w, q_j)) straight into limb i's lazy forward transform, whose
    // integer butterflies accept inputs below 4*q_i; bases mixing very different prime sizes go through the general
    // composition, which reduces every digit modulo q_i first.
    bool digits_fit = true;
    if (h->width == FHE_WIDTH_32 || h->width == FHE_WIDTH_64) {
        fhe_host::u128 q_min = ~(fhe_host::u128)0, q_max = 0;
        for (const U256 &q : h->moduli) { fhe_host::u128 v = q.w[0]; q_min = v < q_min ? v : q_min; q_max = v > q_max ? v : q_max; }
        fhe_host::u128 digit_bound = decomp_bits >= 64 ? q_max : (((fhe_host::u128)1 << decomp_bits) < q_max ? ((fhe_host::u128)1 << decomp_bits) : q_max);
        digits_fit = digit_bound <= 4 * q_min;
    } else if (h->width == FHE_WIDTH_64X) {   // canonical butterflies: a digit must be a residue of q_i as it stands
        fhe_host::u128 q_min = ~(fhe_host::u128)0, q_max = 0;
        for (const U256 &q : h->moduli) { fhe_host::u128 v = q.w[0]; q_min = v < q_min ? v : q_min; q_max = v > q_max ? v : q_max; }
        digits_fit = (decomp_bits >= 64 ? q_max : (((fhe_host::u128)1 << decomp_bits) < q_max ? ((fhe_host::u128)1 << decomp_bits) : q_max)) <= q_min;
    }
    // the fused kernels address a packed table through a buffer descriptor with 32-bit offsets (fhe_dev::TableBuf): a table of 4 GiB or
    // more (L*K*L*n residues: not reached by any parameter set of the reference) stays on the general composition
    const bool table_fits = (size_t)rk->num_keys * h->L * h->n * (h->width == FHE_WIDTH_32 ? 4 : 8) < ((size_t)1 << 32);
    if (!rc && h->width != FHE_WIDTH_256 && !h->sub_top && digits_fit && table_fits && !h->no_fused_keyswitch) {
        rc = pack_relin_keys(h, rk);
        if (!rc) {   // the fused kernels read only the packed tables (n * sizeof(E) bytes per key polynomial instead of n * 32): drop the
                     // container copy, so that a bootstrapping key of several hundred RGSW ciphertexts fits (hipFree waits for the packing)
            (void)hipFree(rk->d_kb); (void)hipFree(rk->d_ka);
            rk->d_kb = rk->d_ka = nullptr;
        }
    }
    if (rc) { fhe_relin_keys_destroy(rk); return rc; }
    if (!rk->d_pkb && K > h->max_composed_digits) h->max_composed_digits = K;   // this key set runs the composed key switch (digit polynomials in the workspace)
    *out = rk;
    return FHE_OK;
}

template <class F>
static int relin_embed_mac_lds(fhe_rns_ntt *h, const fhe_relin_keys *rk, char *D, char *acc0, char *acc1, const void *c2, uint32_t chunk, int phase) {
    using V = typename F::V16;
    const uint32_t LK = h->L * rk->K;
    if (phase == 0) {
        size_t total = (size_t)LK * chunk * h->L * h->n * 2;
        hipLaunchKernelGGL((fhe_dev::digit_embed_kernel<F>), dim3(ew_grid(total)), dim3(256), 0, h->stream, (V *)D, (const V *)c2,
                           (const fhe_dev::Limb<F> *)h->d_limbs, h->L, h->log_n, rk->K, rk->decomp_bits, chunk);
        return post_launch(h->stream, "digit_embed_kernel");
    }
    size_t halves = (size_t)chunk * h->L * h->n * 2;
    hipLaunchKernelGGL((fhe_dev::relin_mac_kernel<F>), dim3(ew_grid(halves)), dim3(256), 0, h->stream, (V *)acc0, (V *)acc1, (const V *)D,
                       (const V *)rk->d_kb, (const V *)rk->d_ka, (const fhe_dev::Limb<F> *)h->d_limbs, h->L, h->log_n, LK, chunk);
    return post_launch(h->stream, "relin_mac_kernel");
}
static int relin_embed_mac(fhe_rns_ntt *h, const fhe_relin_keys *rk, char *D, char *acc0, char *acc1, const void *c2, uint32_t chunk, int phase) {
    if (h->width == FHE_WIDTH_32) return relin_embed_mac_lds<fhe_dev::F32>(h, rk, D, acc0, acc1, c2, chunk, phase);
    if (h->width == FHE_WIDTH_52) return relin_embed_mac_lds<fhe_dev::F52>(h, rk, D, acc0, acc1, c2, chunk, phase);
    if (h->width == FHE_WIDTH_64) return relin_embed_mac_lds<fhe_dev::F64>(h, rk, D, acc0, acc1, c2, chunk, phase);
    if (h->width == FHE_WIDTH_64X) return relin_embed_mac_lds<fhe_dev::F64X>(h, rk, D, acc0, acc1, c2, chunk, phase);
    const uint32_t LK = h->L * rk->K;
    if (phase == 0) {
        size_t total = (size_t)LK * chunk * h->L * h->n;
        hipLaunchKernelGGL(fhe_dev::digit_embed256_kernel, dim3(ew_grid(total)), dim3(256), 0, h->stream, (fhe_dev::u256 *)D, (const fhe_dev::u256 *)c2,
                           (const fhe_dev::Limb256 *)h->d_limbs, h->L, h->log_n, rk->K, rk->decomp_bits, chunk);
        return post_launch(h->stream, "digit_embed256_kernel");
    }
    size_t count = (size_t)chunk * h->L * h->n;
    hipLaunchKernelGGL(fhe_dev::relin_mac256_kernel, dim3(ew_grid(count)), dim3(256), 0, h->stream, (fhe_dev::u256 *)acc0, (fhe_dev::u256 *)acc1,
                       (const fhe_dev::u256 *)D, (const fhe_dev::u256 *)rk->d_kb, (const fhe_dev::u256 *)rk->d_ka,
                       (const fhe_dev::Limb256 *)h->d_limbs, h->L, h->log_n, LK, chunk);
    return post_launch(h->stream, "relin_mac256_kernel");
}

extern "C" int fhe_ct_relinearize(fhe_rns_ntt_t *h, const fhe_relin_keys_t *rk, void *d_c0, void *d_c1, const void *d_c2, uint32_t batch) {
    int rc = check_call(h, batch, "ct_relinearize"); if (rc) return rc;
    if (!rk || !d_c0 || !d_c1 || !d_c2) return fail(FHE_ERR_INVALID_ARG, "ct_relinearize: null argument");
    if (rk->owner != h) return fail(FHE_ERR_INVALID_ARG, "ct_relinearize: keys were imported for a different engine");
    if (d_c0 == d_c1 || d_c0 == d_c2 || d_c1 == d_c2) return fail(FHE_ERR_INVALID_ARG, "ct_relinearize: components must be distinct buffers");
    if ((rc = check_inputs(h, {d_c0, d_c1, d_c2}, batch))) return rc;
    if (rk->d_pkb) {   // word-sized paths: one fused launch
        fhe_dev::lds_launch_fn fn = fhe_dev::lds_lookup(lds_width_id(h), (int)h->log_n);
        if (!fn) return fail(FHE_ERR_UNSUPPORTED, "transform size outside the LDS-resident range");
        fhe_dev::LdsArgs A{fhe_dev::LDS_KEYSWITCH, d_c0, d_c1, nullptr, d_c2, nullptr, nullptr, nullptr, h->d_limbs, h->L, batch * h->L, h->stream};
        A.kb = rk->d_pkb; A.ka = rk->d_pka; A.K = rk->K; A.w = rk->decomp_bits;
        A.global_twiddles = h->global_twiddles;
        A.single_transforms = h->single_transforms;
        A.joint3 = use_joint3(h, false, false);
        const bool c2_in_ws2 = h->d_ws2 && (const char *)d_c2 >= (const char *)h->d_ws2 && (const char *)d_c2 < (const char *)h->d_ws2 + h->ws2_bytes;
        const size_t eb = residue_bytes(h), S = (size_t)h->L * h->n * 32, Sc = (size_t)h->L * h->n * eb;
        const bool paired32 = h->width == FHE_WIDTH_32 && !h->single_transforms && fhe_dev::lds_paired_keyswitch(4, (int)h->log_n);
        if (((A.joint3 && h->width != FHE_WIDTH_32) || paired32) && !h->no_c2_compaction && !c2_in_ws2) {   // (c2 already in the workspace: the composed multiply + relinearise under a testing switch)
            // Every limb workgroup re-reads all of c2 (the three-array kernels once per DIGIT): compact it once (a streaming pass: S read, S/4 or
            // S/8 written) so that those re-reads move compact polynomials instead of 32-byte containers -- round 2 counters at N = 2^14, 6 x 40-bit:
            // 2.6 x the algorithmic bytes; on the 4-byte residues the container loads kept the address FIFO full 12 % of the time (round 3 SQ counters).
            // Chunks of whole ciphertexts on two streams: the compaction of chunk i+1 (HBM-bound) runs beside the key switch of chunk i.
            if ((rc = ensure_ws2(h, (size_t)batch * Sc))) return rc;
            // (measured, N = 8192 x 4 x 30-bit, batch 1024: compaction alone 773 K -> 805 K relin/s at w = 16, 948 K -> 1007 K at w = 30; with the key switch of
            //  chunk i beside the compaction of chunk i+1 on a second stream 807 K / 953 K at two chunks, 772 K / 948 K at four: one stream unless asked)
            uint32_t chunks = (paired32 && h->relin_chunks_forced) ? h->overlap_chunks : 1;
            while (chunks > 1 && ((size_t)batch * h->L / chunks < 1024 || batch < chunks)) chunks--;
            if (chunks > 1 && (rc = ensure_aux_stream(h))) return rc;
            for (uint32_t c = 0, b0 = 0; c < chunks; c++) {
                const uint32_t nb = batch / chunks + (c < batch % chunks ? 1 : 0);
                char *c2c = (char *)h->d_ws2 + (size_t)b0 * Sc;
                if ((rc = compact_poly(h, c2c, (const char *)d_c2 + (size_t)b0 * S, (size_t)nb * h->L * h->n))) return rc;
                fhe_dev::LdsArgs B = A;
                B.r0 = (char *)d_c0 + (size_t)b0 * S; B.r1 = (char *)d_c1 + (size_t)b0 * S; B.a0 = c2c; B.polys = nb * h->L; B.c2_only_compact = true;
                if (chunks == 1 && (rc = split_pairs_workspace(h, B))) return rc;
                if (chunks > 1) {
                    HIP_TRY(hipEventRecord(h->ev_chunk[c], h->stream));
                    HIP_TRY(hipStreamWaitEvent(h->aux_stream, h->ev_chunk[c], 0));
                    B.stream = h->aux_stream;
                }
                fn(B);
                if ((rc = post_launch(B.stream, "ntt_keyswitch_kernel"))) return rc;
                b0 += nb;
            }
            if (chunks > 1) {
                HIP_TRY(hipEventRecord(h->ev_join, h->aux_stream));
                HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
            }
            return FHE_OK;
        }
        if ((rc = split_pairs_workspace(h, A))) return rc;
        fn(A);
        return post_launch(h->stream, "ntt_keyswitch_kernel");
    }
    const uint32_t LK = h->L * rk->K;
    const size_t S = (size_t)h->L * h->n * 32;
    // workspace: digit polynomials D[LK][chunk] + two accumulators; bounded to ~1 GiB, the batch is processed in chunks
    uint32_t chunk = (uint32_t)(((size_t)1 << 30) / ((LK + 2) * S));
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if ((rc = ensure_ws(h, (size_t)(LK + 2) * chunk * S))) return rc;
    char *D = (char *)h->d_ws, *acc0 = D + (size_t)LK * chunk * S, *acc1 = acc0 + (size_t)chunk * S;
    for (uint32_t done = 0; done < batch; done += chunk) {
        const uint32_t nb = batch - done < chunk ? batch - done : chunk;
        const char *c2 = (const char *)d_c2 + (size_t)done * S;
        char *c0 = (char *)d_c0 + (size_t)done * S, *c1 = (char *)d_c1 + (size_t)done * S;
        if ((rc = relin_embed_mac(h, rk, D, acc0, acc1, c2, nb, 0))) return rc;
        if ((rc = do_forward(h, D, LK * nb))) return rc;
        if ((rc = relin_embed_mac(h, rk, D, acc0, acc1, c2, nb, 1))) return rc;
        if ((rc = do_inverse(h, acc0, nb))) return rc;
        if ((rc = do_inverse(h, acc1, nb))) return rc;
        if ((rc = do_ew<1>(h, c0, c0, acc0, nb, "relin add"))) return rc;
        if ((rc = do_ew<1>(h, c1, c1, acc1, nb, "relin add"))) return rc;
    }
    return FHE_OK;
}

// FHEContext::multiply as the reference declares it (src/fhe.cu:199-224: tensor product, then relinearize): (c0, c1) = relin(a (x) b).
// On the LDS-resident sizes of the word-sized classes the three components of the tensor product never take the 32-byte container
// form: the tensor-product kernel(s) write c0, c1, c2 to a compact workspace (sizeof(residue) bytes per coefficient) and the
// key-switch kernel reads its digit source (c2) and its addends (c0, c1) from there -- HBM traffic 4 S in + 2 S out + 3 compact
// components written once and read back (c2 by every limb workgroup) instead of 12 S.  Elsewhere: fhe_ct_multiply into a container
// workspace followed by fhe_ct_relinearize.  Same bits either way (tests compare both with the oracle).
extern "C" int fhe_ct_multiply_relin(fhe_rns_ntt_t *h, const fhe_relin_keys_t *rk, void *d_c0, void *d_c1, const void *d_a0, const void *d_a1,
                                     const void *d_b0, const void *d_b1, uint32_t batch) {
    int rc = check_call(h, batch, "ct_multiply_relin"); if (rc) return rc;
    if (!rk || !d_c0 || !d_c1 || !d_a0 || !d_a1 || !d_b0 || !d_b1) return fail(FHE_ERR_INVALID_ARG, "ct_multiply_relin: null argument");
    if (rk->owner != h) return fail(FHE_ERR_INVALID_ARG, "ct_multiply_relin: keys were imported for a different engine");
    const void *ins[4] = {d_a0, d_a1, d_b0, d_b1};
    for (const void *i : ins) if (d_c0 == i || d_c1 == i) return fail(FHE_ERR_INVALID_ARG, "ct_multiply_relin: outputs must not alias inputs");
    if (d_c0 == d_c1) return fail(FHE_ERR_INVALID_ARG, "ct_multiply_relin: outputs must be distinct");
    if ((rc = check_inputs(h, {d_a0, d_a1, d_b0, d_b1}, batch))) return rc;
    const int eb = h->width == FHE_WIDTH_32 ? 4 : 8;
    const bool fused = rk->d_pkb && h->width != FHE_WIDTH_256 && !h->sub_top && !h->single_transforms && !h->no_fused_ct_relin &&
                       fhe_dev::lds_compact_c2(eb, (int)h->log_n);
    const uint32_t polys = batch * h->L;
    if (fused) {
        const size_t cbytes = (size_t)polys * h->n * eb;         // one compact component
        if ((rc = ensure_ws2(h, 3 * cbytes))) return rc;
        char *c0c = (char *)h->d_ws2, *c1c = c0c + cbytes, *c2c = c1c + cbytes;
        fhe_dev::lds_launch_fn fn = fhe_dev::lds_lookup(lds_width_id(h), (int)h->log_n);
        if (!fn) return fail(FHE_ERR_UNSUPPORTED, "transform size outside the LDS-resident range");
        // The two kernels of the call sit on different roofs: the tensor product streams 4 S in at the HBM rate, the key switch (compact
        // operands) is bound by instruction issue.  The call is therefore a two-stage pipeline over chunks of whole ciphertexts: every
        // tensor product runs on the engine's stream, back to back; the key switch of chunk i runs on a second stream as soon as
        // tensor product i is done (event), i.e. beside tensor product i+1 on the same CUs.  The engine's stream joins the second one
        // at the end, so the call stays ordered on the engine's stream (and can be captured into a graph: fork / join through events).
        // Every chunk has its own slice of the compact workspace.
        uint32_t chunks = h->overlap_chunks;
        char *bws = nullptr;                                     // two-launch tensor product: the transformed b-side, sized here for the whole batch, sliced per chunk
        {
            fhe_dev::LdsArgs probe{fhe_dev::LDS_CT_MULTIPLY, c0c, c1c, c2c, d_a0, d_a1, d_b0, d_b1, h->d_limbs, h->L, polys, h->stream};
            probe.compact_c2 = true;
            if ((rc = ct_workspace(h, probe))) return rc;
            bws = (char *)probe.ws;
            if (bws && h->log_n >= 14) chunks = 1;               // 128+ KiB of LDS per workgroup: the two stages cannot share a CU anyway
        }
        while (chunks > 1 && (polys / chunks < 1024 || batch < chunks)) chunks--;   // every chunk must fill the chip: >= 256 CUs x 4 workgroups (one per limb polynomial)
        if (chunks > 1 && (rc = ensure_aux_stream(h))) return rc;
        const size_t S = (size_t)h->L * h->n * 32, Sc = (size_t)h->L * h->n * eb;   // bytes of one ciphertext component: containers / compact
        for (uint32_t c = 0, b0 = 0; c < chunks; c++) {
            const uint32_t nb = batch / chunks + (c < batch % chunks ? 1 : 0);
            const size_t o = (size_t)b0 * S, oc = (size_t)b0 * Sc;
            fhe_dev::LdsArgs A{fhe_dev::LDS_CT_MULTIPLY, c0c + oc, c1c + oc, c2c + oc, (const char *)d_a0 + o, (const char *)d_a1 + o, (const char *)d_b0 + o,
                               (const char *)d_b1 + o, h->d_limbs, h->L, nb * h->L, h->stream};
            A.compact_c2 = true;
            if (bws) A.ws = bws + 2 * oc;                        // two compact polynomials per limb polynomial of the chunk
            A.small_batch = A.polys <= h->split_pairs_polys;     // few ciphertexts: the 16-per-thread tensor product (and, below, the split key switch)
            if (chunks == 1 && !bws && A.polys <= h->coop_polys && h->width == FHE_WIDTH_32 && fhe_dev::lds_coop4_multiply(4, (int)h->log_n)) {
                if ((rc = ensure_ws3(h, 7 * (size_t)A.polys * h->n * 4))) return rc;      // a handful: four workgroups per limb polynomial, three launches
                A.coop_ws = h->d_ws3;
            }
            fn(A);
            if ((rc = post_launch(h->stream, "tensor product (compact outputs)"))) return rc;
            hipStream_t ks = h->stream;
            if (chunks > 1) {
                HIP_TRY(hipEventRecord(h->ev_chunk[c], h->stream));
                HIP_TRY(hipStreamWaitEvent(h->aux_stream, h->ev_chunk[c], 0));
                ks = h->aux_stream;
            }
            fhe_dev::LdsArgs B{fhe_dev::LDS_KEYSWITCH, (char *)d_c0 + o, (char *)d_c1 + o, nullptr, c2c + oc, c0c + oc, c1c + oc, nullptr, h->d_limbs, h->L, nb * h->L, ks};
            B.kb = rk->d_pkb; B.ka = rk->d_pka; B.K = rk->K; B.w = rk->decomp_bits; B.compact_c2 = true;
            B.joint3 = use_joint3(h, false, true);
            if (chunks == 1 && !bws && (rc = split_pairs_workspace(h, B))) return rc;     // (d_ws is free: no two-launch tensor product on the 4-byte field up to 2^14)
            fn(B);
            if ((rc = post_launch(ks, "key switch (compact operands)"))) return rc;
            b0 += nb;
        }
        if (chunks > 1) {
            HIP_TRY(hipEventRecord(h->ev_join, h->aux_stream));
            HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
        }
        return FHE_OK;
    }
    if ((rc = ensure_ws2(h, (size_t)polys * h->n * 32))) return rc;
    if ((rc = do_ct_multiply(h, d_c0, d_c1, h->d_ws2, d_a0, d_a1, d_b0, d_b1, batch))) return rc;
    return fhe_ct_relinearize(h, rk, d_c0, d_c1, h->d_ws2, batch);
}


// ------------------------------------------------------------------------------------------------------
// RNS entry / exit
// ------------------------------------------------------------------------------------------------------
static bool mul_checked(U256 &r, const U256 &a, const U256 &b) {      // r = a*b, false on overflow beyond 256 bits
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            fhe_host::u128 acc = (fhe_host::u128)a.w[i] * b.w[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)acc; carry = (uint64_t)(acc >> 64);
        }
        t[i + 4] = carry;
    }
    std::memcpy(r.w, t, 32);
    return !(t[4] | t[5] | t[6] | t[7]);
}
static int ensure_crt(fhe_rns_ntt *h) {
    if (h->crt_state) return FHE_OK;
    const uint32_t L = h->L;
    U256 Q(1); bool fits = true;
    for (uint32_t l = 0; l < L && fits; l++) { U256 t; fits = mul_checked(t, Q, h->moduli[l]); Q = t; }
    fits = fits && !(Q.w[3] >> 63);
    std::vector<fhe_dev::CrtLimb> limbs(L);
    std::memset(limbs.data(), 0, L * sizeof(fhe_dev::CrtLimb));
    for (uint32_t l = 0; l < L; l++) {
        fhe_host::Mod M(h->moduli[l]);
        std::memcpy(limbs[l].q.l, M.q.w, 32); std::memcpy(limbs[l].r2.l, M.r2.w, 32); limbs[l].inv0 = M.inv0;
    }
    if (fits) {
        fhe_host::Mod MQ(Q);
        for (uint32_t l = 0; l < L; l++) {
            U256 Mi(1);
            for (uint32_t k = 0; k < L; k++) if (k != l) { U256 t; mul_checked(t, Mi, h->moduli[k]); Mi = t; }
            fhe_host::Mod M(h->moduli[l]);
            U256 qm2; fhe_host::sub_to(qm2, M.q, U256(2));
            U256 minv_m = M.pow_m(M.to_mont(M.reduce(Mi)), qm2);                 // ((Q/q)^-1 mod q) * R
            U256 Mi_mQ = MQ.to_mont(Mi);
            std::memcpy(limbs[l].minv_m.l, minv_m.w, 32); std::memcpy(limbs[l].Mi_mQ.l, Mi_mQ.w, 32);
        }
        std::memcpy(h->crt_big.Q.l, Q.w, 32); h->crt_big.inv0 = MQ.inv0; h->crt_big._pad = 0;
    }
    int rc = upload(h, limbs, &h->d_crt); if (rc) return rc;
    h->crt_state = fits ? 1 : -1;
    return FHE_OK;
}
// "pw operand" of the constant c for limb modulus q of a word-sized class: c * 2^W mod q for the integer fields (so that the
// Montgomery product with it is the plain product), c itself for the FP64 field
template <class F> static typename F::E word_operand(uint64_t c, uint64_t q) {
    using E = typename F::E;
    if (std::is_same<F, fhe_dev::F52>::value) return (E)c;
    const unsigned W = 8 * sizeof(E);
    fhe_host::u128 v = (fhe_host::u128)(c % q);
    for (unsigned k = 0; k < W; k++) v = (v << 1) % q;
    return (E)(uint64_t)v;
}
static uint64_t inv_mod_u64(uint64_t a, uint64_t q) {       // a^(q-2) mod q, q prime
    fhe_host::u128 acc = 1, b = a % q; uint64_t e = q - 2;
    for (; e; e >>= 1) { if (e & 1) acc = acc * b % q; b = b * b % q; }
    return (uint64_t)acc;
}
template <class F, class WT>
static int to_rns_word(fhe_rns_ntt *h, void *d_rns, const void *d_values, uint32_t batch) {
    using E = typename F::E; using V = typename F::V16;
    constexpr uint32_t NW = 32 / sizeof(WT), W = 8 * sizeof(WT);
    if (!h->d_to_rns_w) {
        std::vector<E> ops((size_t)h->L * NW);
        for (uint32_t l = 0; l < h->L; l++) {
            const uint64_t q = h->moduli[l].w[0];
            fhe_host::u128 p = 1 % q;
            for (uint32_t k = 0; k < NW; k++) {
                ops[(size_t)l * NW + k] = word_operand<F>((uint64_t)p, q);
                for (uint32_t t = 0; t < W; t++) p = (p << 1) % q;              // 2^(W (k+1)) mod q
            }
        }
        int rc = upload(h, ops, &h->d_to_rns_w); if (rc) return rc;
    }
    const size_t containers = (size_t)batch * h->L * h->n;
    hipLaunchKernelGGL((fhe_dev::to_rns_word_kernel<F, WT>), dim3(ew_grid(containers)), dim3(256), 0, h->stream, (V *)d_rns, (const V *)d_values,
                       (const fhe_dev::Limb<F> *)h->d_limbs, (const E *)h->d_to_rns_w, h->L, h->log_n, containers);
    return post_launch(h->stream, "to_rns_word_kernel");
}
extern "C" int fhe_rns_to_rns(fhe_rns_ntt_t *h, void *d_rns, const void *d_values, uint32_t batch) {
    int rc = check_call(h, batch, "to_rns"); if (rc) return rc;
    if (!d_rns || !d_values || d_rns == d_values) return fail(FHE_ERR_INVALID_ARG, "to_rns: null or aliased argument");
    // (to_rns_word_kernel stores through lane pairs of WHOLE waves, store_wave_containers: every wave must cover 64 consecutive containers
    //  of one polynomial, i.e. n a multiple of 256.  Word-sized classes exist from n = 2^11, the guard keeps that an explicit condition.)
    if (!h->no_word_conversions && h->log_n >= 8) {       // word-sized classes: a streaming kernel on the field type
        if (h->width == FHE_WIDTH_32) return to_rns_word<fhe_dev::F32, uint32_t>(h, d_rns, d_values, batch);
        if (h->width == FHE_WIDTH_52) return to_rns_word<fhe_dev::F52, uint32_t>(h, d_rns, d_values, batch);
        if (h->width == FHE_WIDTH_64) return to_rns_word<fhe_dev::F64, uint64_t>(h, d_rns, d_values, batch);
        if (h->width == FHE_WIDTH_64X) return to_rns_word<fhe_dev::F64X, uint64_t>(h, d_rns, d_values, batch);
    }
    if ((rc = ensure_crt(h))) return rc;
    const size_t count = (size_t)batch * h->n;
    hipLaunchKernelGGL(fhe_dev::to_rns_kernel, dim3(ew_grid(count)), dim3(256), 0, h->stream, (fhe_dev::u256 *)d_rns, (const fhe_dev::u256 *)d_values,
                       (const fhe_dev::CrtLimb *)h->d_crt, h->L, h->log_n, count);
    return post_launch(h->stream, "to_rns_kernel");
}
template <class F>
static int from_rns_word(fhe_rns_ntt *h, void *d_values, const void *d_rns, uint32_t batch) {
    using E = typename F::E; using V = typename F::V16;
    if (!h->d_from_rns_w_minv) {
        const uint32_t L = h->L;
        std::vector<E> minv(L); std::vector<fhe_dev::u256> Ms(L);
        for (uint32_t l = 0; l < L; l++) {
            const uint64_t q = h->moduli[l].w[0];
            U256 Mi(1); fhe_host::u128 Mi_mod_q = 1;
            for (uint32_t k = 0; k < L; k++) if (k != l) { U256 t; mul_checked(t, Mi, h->moduli[k]); Mi = t; Mi_mod_q = Mi_mod_q * (h->moduli[k].w[0] % q) % q; }
            minv[l] = word_operand<F>(inv_mod_u64((uint64_t)Mi_mod_q, q), q);
            std::memcpy(Ms[l].l, Mi.w, 32);
        }
        int rc;
        if ((rc = upload(h, minv, &h->d_from_rns_w_minv)) || (rc = upload(h, Ms, &h->d_from_rns_w_M))) return rc;
    }
    const size_t count = (size_t)batch * h->n;
    hipLaunchKernelGGL((fhe_dev::from_rns_word_kernel<F>), dim3(ew_grid(count)), dim3(256), 0, h->stream, (fhe_dev::u256 *)d_values, (const V *)d_rns,
                       (const fhe_dev::Limb<F> *)h->d_limbs, (const E *)h->d_from_rns_w_minv, (const fhe_dev::u256 *)h->d_from_rns_w_M, h->crt_big.Q,
                       h->L, h->log_n, count);
    return post_launch(h->stream, "from_rns_word_kernel");
}
extern "C" int fhe_rns_from_rns(fhe_rns_ntt_t *h, void *d_values, const void *d_rns, uint32_t batch) {
    int rc = check_call(h, batch, "from_rns"); if (rc) return rc;
    if (!d_rns || !d_values || d_rns == d_values) return fail(FHE_ERR_INVALID_ARG, "from_rns: null or aliased argument");
    if ((rc = ensure_crt(h))) return rc;
    if (h->crt_state < 0) return fail(FHE_ERR_UNSUPPORTED, "from_rns: the product of the moduli must be below 2^255 to fit a 256-bit container");
    if (!h->no_word_conversions) {       // word-sized classes: word x 256-bit accumulation instead of 256-bit Montgomery products
        if (h->width == FHE_WIDTH_32) return from_rns_word<fhe_dev::F32>(h, d_values, d_rns, batch);
        if (h->width == FHE_WIDTH_52) return from_rns_word<fhe_dev::F52>(h, d_values, d_rns, batch);
        if (h->width == FHE_WIDTH_64) return from_rns_word<fhe_dev::F64>(h, d_values, d_rns, batch);
        if (h->width == FHE_WIDTH_64X) return from_rns_word<fhe_dev::F64X>(h, d_values, d_rns, batch);
    }
    const size_t count = (size_t)batch * h->n;
    hipLaunchKernelGGL(fhe_dev::from_rns_kernel, dim3(ew_grid(count)), dim3(256), 0, h->stream, (fhe_dev::u256 *)d_values, (const fhe_dev::u256 *)d_rns,
                       (const fhe_dev::CrtLimb *)h->d_crt, h->crt_big, h->L, h->log_n, count);
    return post_launch(h->stream, "from_rns_kernel");
}


template <class F>
static int rescale_word(fhe_rns_ntt *h, void *d_out, const void *d_in, uint32_t batch) {
    using E = typename F::E; using V = typename F::V16;
    if (!h->d_rescale_w) {
        std::vector<E> ops(h->L - 1);
        const uint64_t ql = h->moduli[h->L - 1].w[0];
        for (uint32_t l = 0; l + 1 < h->L; l++) { const uint64_t q = h->moduli[l].w[0]; ops[l] = word_operand<F>(inv_mod_u64(ql % q, q), q); }
        int rc = upload(h, ops, &h->d_rescale_w); if (rc) return rc;
    }
    const size_t count = (size_t)batch * h->n;
    hipLaunchKernelGGL((fhe_dev::rescale_word_kernel<F>), dim3(ew_grid(count)), dim3(256), 0, h->stream, (V *)d_out, (const V *)d_in,
                       (const fhe_dev::Limb<F> *)h->d_limbs, (const E *)h->d_rescale_w, h->L, h->log_n, count);
    return post_launch(h->stream, "rescale_word_kernel");
}
template <class F>
static int base_convert_word(fhe_rns_ntt *h, fhe_rns_ntt *t, void *d_out, const void *d_in, uint32_t batch) {
    using E = typename F::E; using V = typename F::V16;
    if (h->bconv_w_target != t || !(h->bconv_w_moduli == t->moduli)) {
        const uint32_t L = h->L, Lp = t->L;
        std::vector<E> minv(L), mat((size_t)L * Lp);
        for (uint32_t i = 0; i < L; i++) {
            const uint64_t qi = h->moduli[i].w[0];
            fhe_host::u128 Mi = 1;
            for (uint32_t k = 0; k < L; k++) if (k != i) Mi = Mi * (h->moduli[k].w[0] % qi) % qi;
            minv[i] = word_operand<F>(inv_mod_u64((uint64_t)Mi, qi), qi);
            for (uint32_t j = 0; j < Lp; j++) {
                const uint64_t pj = t->moduli[j].w[0];
                fhe_host::u128 m = 1;
                for (uint32_t k = 0; k < L; k++) if (k != i) m = m * (h->moduli[k].w[0] % pj) % pj;
                mat[(size_t)i * Lp + j] = word_operand<F>((uint64_t)m, pj);
            }
        }
        int rc;
        if ((rc = upload(h, minv, &h->d_bconv_w_minv)) || (rc = upload(h, mat, &h->d_bconv_w_mat))) return rc;   // earlier tables stay owned by d_tables
        h->bconv_w_target = t; h->bconv_w_moduli = t->moduli;
    }
    constexpr bool all_lanes = std::is_same<F, fhe_dev::F64>::value || std::is_same<F, fhe_dev::F64X>::value;
    const size_t work = (size_t)batch * t->L * h->n * (all_lanes ? 1 : 2);
    hipLaunchKernelGGL((fhe_dev::base_convert_word_kernel<F, all_lanes>), dim3(ew_grid(work)), dim3(256), 0, h->stream, (V *)d_out, (const V *)d_in,
                       (const fhe_dev::Limb<F> *)h->d_limbs, h->L, (const fhe_dev::Limb<F> *)t->d_limbs, t->L, (const E *)h->d_bconv_w_minv,
                       (const E *)h->d_bconv_w_mat, h->log_n, work);
    return post_launch(h->stream, "base_convert_word_kernel");
}

extern "C" int fhe_rns_rescale_drop_last(fhe_rns_ntt_t *h, void *d_out, const void *d_in, uint32_t batch) {
    int rc = check_call(h, batch, "rescale_drop_last"); if (rc) return rc;
    if (!d_out || !d_in || d_out == d_in) return fail(FHE_ERR_INVALID_ARG, "rescale_drop_last: null or aliased argument");
    if (h->L < 2) return fail(FHE_ERR_INVALID_ARG, "rescale_drop_last: needs at least two primes");
    if (!h->no_word_conversions) {       // word-sized classes: streaming kernels on the field type
        if (h->width == FHE_WIDTH_32) return rescale_word<fhe_dev::F32>(h, d_out, d_in, batch);
        if (h->width == FHE_WIDTH_52) return rescale_word<fhe_dev::F52>(h, d_out, d_in, batch);
        if (h->width == FHE_WIDTH_64) return rescale_word<fhe_dev::F64>(h, d_out, d_in, batch);
        if (h->width == FHE_WIDTH_64X) return rescale_word<fhe_dev::F64X>(h, d_out, d_in, batch);
    }
    if ((rc = ensure_crt(h))) return rc;
    if (!h->d_rescale) {
        std::vector<fhe_dev::RescaleLimb> rs(h->L - 1);
        const U256 &ql = h->moduli[h->L - 1];
        for (uint32_t l = 0; l + 1 < h->L; l++) {
            fhe_host::Mod M(h->moduli[l]);
            U256 qm2; fhe_host::sub_to(qm2, M.q, U256(2));
            U256 inv_m = M.pow_m(M.to_mont(M.reduce(ql)), qm2);                    // (q_last^-1 mod q_l) * R
            std::memcpy(rs[l].qlast_inv_m.l, inv_m.w, 32);
        }
        if ((rc = upload(h, rs, &h->d_rescale))) return rc;
    }
    const size_t count = (size_t)batch * h->n;
    hipLaunchKernelGGL(fhe_dev::rescale_drop_last_kernel, dim3(ew_grid(count)), dim3(256), 0, h->stream, (fhe_dev::u256 *)d_out,
                       (const fhe_dev::u256 *)d_in, (const fhe_dev::CrtLimb *)h->d_crt, (const fhe_dev::RescaleLimb *)h->d_rescale, h->L, h->log_n, count);
    return post_launch(h->stream, "rescale_drop_last_kernel");
}


extern "C" int fhe_rns_fast_base_convert(fhe_rns_ntt_t *h, fhe_rns_ntt_t *target, void *d_out, const void *d_in, uint32_t batch) {
    int rc = check_call(h, batch, "fast_base_convert"); if (rc) return rc;
    if (!target || !d_out || !d_in || d_out == d_in) return fail(FHE_ERR_INVALID_ARG, "fast_base_convert: null or aliased argument");
    if (target->n != h->n) return fail(FHE_ERR_INVALID_ARG, "fast_base_convert: source and target engines differ in degree");
    if (h->width == target->width && h->width != FHE_WIDTH_256 && !h->no_word_conversions && h->log_n >= 8) {   // whole waves per polynomial, as in to_rns
        if (h->width == FHE_WIDTH_32) return base_convert_word<fhe_dev::F32>(h, target, d_out, d_in, batch);
        if (h->width == FHE_WIDTH_52) return base_convert_word<fhe_dev::F52>(h, target, d_out, d_in, batch);
        if (h->width == FHE_WIDTH_64X) return base_convert_word<fhe_dev::F64X>(h, target, d_out, d_in, batch);
        return base_convert_word<fhe_dev::F64>(h, target, d_out, d_in, batch);
    }
    if ((rc = ensure_crt(h)) || (rc = ensure_crt(target))) return rc;
    if (h->bconv_target != target || !(h->bconv_moduli == target->moduli)) {
        std::vector<fhe_dev::u256> mat((size_t)h->L * target->L);
        for (uint32_t j = 0; j < target->L; j++) {
            fhe_host::Mod M(target->moduli[j]);
            for (uint32_t i = 0; i < h->L; i++) {
                U256 acc = M.r1;                                              // prod_{k != i} q_k mod p_j, Montgomery form
                for (uint32_t k = 0; k < h->L; k++) if (k != i) acc = M.mont(acc, M.to_mont(M.reduce(h->moduli[k])));
                std::memcpy(mat[(size_t)i * target->L + j].l, acc.w, 32);
            }
        }
        if ((rc = upload(h, mat, &h->d_bconv))) return rc;                    // earlier matrices stay owned by d_tables until destroy
        h->bconv_target = target; h->bconv_moduli = target->moduli;
    }
    const size_t count = (size_t)batch * h->n;
    hipLaunchKernelGGL(fhe_dev::fast_base_convert_kernel, dim3(ew_grid(count)), dim3(256), 0, h->stream, (fhe_dev::u256 *)d_out,
                       (const fhe_dev::u256 *)d_in, (const fhe_dev::CrtLimb *)h->d_crt, h->L, (const fhe_dev::CrtLimb *)target->d_crt, target->L,
                       (const fhe_dev::u256 *)h->d_bconv, h->log_n, count);
    return post_launch(h->stream, "fast_base_convert_kernel");
}


// ------------------------------------------------------------------------------------------------------
// blind-rotation inner loop
// ------------------------------------------------------------------------------------------------------
template <class F>
static int monomial_lds(fhe_rns_ntt *h, void *out, const void *in, const uint32_t *shifts, uint32_t batch) {
    using V = typename F::V16;
    size_t halves = (size_t)batch * h->L * h->n * 2;
    hipLaunchKernelGGL((fhe_dev::monomial_mul_sub_kernel<F>), dim3(ew_grid(halves)), dim3(256), 0, h->stream, (V *)out, (const V *)in, shifts,
                       (const fhe_dev::Limb<F> *)h->d_limbs, h->L, h->log_n, halves);
    return post_launch(h->stream, "monomial_mul_sub_kernel");
}
extern "C" int fhe_rns_monomial_mul_sub(fhe_rns_ntt_t *h, void *d_out, const void *d_in, const uint32_t *d_shifts, uint32_t batch) {
    int rc = check_call(h, batch, "monomial_mul_sub"); if (rc) return rc;
    if (!d_out || !d_in || !d_shifts || d_out == d_in) return fail(FHE_ERR_INVALID_ARG, "monomial_mul_sub: null or aliased argument");
    if (h->width == FHE_WIDTH_32) return monomial_lds<fhe_dev::F32>(h, d_out, d_in, d_shifts, batch);
    if (h->width == FHE_WIDTH_52) return monomial_lds<fhe_dev::F52>(h, d_out, d_in, d_shifts, batch);
    if (h->width == FHE_WIDTH_64) return monomial_lds<fhe_dev::F64>(h, d_out, d_in, d_shifts, batch);
    if (h->width == FHE_WIDTH_64X) return monomial_lds<fhe_dev::F64X>(h, d_out, d_in, d_shifts, batch);
    size_t count = (size_t)batch * h->L * h->n;
    hipLaunchKernelGGL(fhe_dev::monomial_mul_sub256_kernel, dim3(ew_grid(count)), dim3(256), 0, h->stream, (fhe_dev::u256 *)d_out,
                       (const fhe_dev::u256 *)d_in, d_shifts, (const fhe_dev::Limb256 *)h->d_limbs, h->L, h->log_n, count);
    return post_launch(h->stream, "monomial_mul_sub256_kernel");
}
// One step on the general composition (any width class): d = (X^a - 1) * acc, then two key switches accumulate into acc in place.
static int blind_rotate_step_general(fhe_rns_ntt_t *h, const fhe_relin_keys_t *rows_c0, const fhe_relin_keys_t *rows_c1, void *d_acc0, void *d_acc1,
                                     const uint32_t *d_shifts, void *d_tmp0, void *d_tmp1, uint32_t batch) {
    int rc;
    if ((rc = fhe_rns_monomial_mul_sub(h, d_tmp0, d_acc0, d_shifts, batch))) return rc;       // d0 = (X^a - 1) * acc0
    if ((rc = fhe_rns_monomial_mul_sub(h, d_tmp1, d_acc1, d_shifts, batch))) return rc;       // d1 = (X^a - 1) * acc1
    if ((rc = fhe_ct_relinearize(h, rows_c0, d_acc0, d_acc1, d_tmp0, batch))) return rc;      // acc += sum D(d0) * rows_c0
    return fhe_ct_relinearize(h, rows_c1, d_acc0, d_acc1, d_tmp1, batch);                     // acc += sum D(d1) * rows_c1
}
// One step as ONE launch (word-sized classes with packed rows): (out0, out1) = (in0, in1) + ExtProd((X^a - 1) * in, RGSW).
static int blind_rotate_step_fused(fhe_rns_ntt_t *h, const fhe_relin_keys_t *r0, const fhe_relin_keys_t *r1, void *out0, void *out1, const void *in0,
                                   const void *in1, const uint32_t *d_shifts, uint32_t batch, bool in_compact = false, bool out_compact = false,
                                   const void *rot0 = nullptr, const void *rot1 = nullptr, void *pair_ws = nullptr) {
    fhe_dev::lds_launch_fn fn = fhe_dev::lds_lookup(lds_width_id(h), (int)h->log_n);
    if (!fn) return fail(FHE_ERR_UNSUPPORTED, "transform size outside the LDS-resident range");
    fhe_dev::LdsArgs A{fhe_dev::LDS_EXTPROD, out0, out1, nullptr, in0, in1, rot0, rot1, h->d_limbs, h->L, batch * h->L, h->stream};   // b0, b1: pre-rotated digit sources (three-array kernel)
    A.kb = r0->d_pkb; A.ka = r0->d_pka; A.kb1 = r1->d_pkb; A.ka1 = r1->d_pka; A.K = r0->K; A.w = r0->decomp_bits; A.shifts = d_shifts;
    A.in_compact = in_compact; A.out_compact = out_compact;
    A.pair_ws = pair_ws;                 // few accumulators: partial accumulators of the one-workgroup-per-digit form
    A.joint3 = use_joint3(h, true, false);
    A.global_twiddles = h->global_twiddles;
    A.single_transforms = h->single_transforms;
    fn(A);
    return post_launch(h->stream, "ntt_extprod_kernel");
}
static int check_rows(const fhe_rns_ntt_t *h, const fhe_relin_keys_t *r0, const fhe_relin_keys_t *r1) {
    if (!r0 || !r1) return fail(FHE_ERR_INVALID_ARG, "blind_rotate: null RGSW rows");
    if (r0->owner != h || r1->owner != h) return fail(FHE_ERR_INVALID_ARG, "blind_rotate: rows were imported for a different engine");
    if (r0->decomp_bits != r1->decomp_bits || r0->K != r1->K) return fail(FHE_ERR_INVALID_ARG, "blind_rotate: the two row sets use different digit widths");
    return FHE_OK;
}
extern "C" int fhe_blind_rotate(fhe_rns_ntt_t *h, const fhe_relin_keys_t *const *rows_c0, const fhe_relin_keys_t *const *rows_c1, uint32_t steps,
                                void *d_acc0, void *d_acc1, const uint32_t *d_shifts, void *d_tmp0, void *d_tmp1, uint32_t batch) {
    int rc = check_call(h, batch, "blind_rotate"); if (rc) return rc;
    if (!d_acc0 || !d_acc1 || !d_tmp0 || !d_tmp1 || !d_shifts || (steps && (!rows_c0 || !rows_c1))) return fail(FHE_ERR_INVALID_ARG, "blind_rotate: null argument");
    {   // the four buffers must be pairwise distinct
        const void *bufs[4] = {d_acc0, d_acc1, d_tmp0, d_tmp1};
        for (int x = 0; x < 4; x++) for (int y = x + 1; y < 4; y++)
            if (bufs[x] == bufs[y]) return fail(FHE_ERR_INVALID_ARG, "blind_rotate: accumulators and scratch must be distinct buffers");
    }
    if ((rc = check_inputs(h, {d_acc0, d_acc1}, batch))) return rc;
    bool fused = h->width != FHE_WIDTH_256 && !h->no_fused_blind_rotate;
    for (uint32_t s = 0; s < steps; s++) {
        if ((rc = check_rows(h, rows_c0[s], rows_c1[s]))) return rc;
        fused = fused && rows_c0[s]->d_pkb && rows_c1[s]->d_pkb;
    }
    if (!fused) {
        for (uint32_t s = 0; s < steps; s++)
            if ((rc = blind_rotate_step_general(h, rows_c0[s], rows_c1[s], d_acc0, d_acc1, d_shifts + (size_t)s * batch, d_tmp0, d_tmp1, batch))) return rc;
        return FHE_OK;
    }
    // Few accumulators (4-byte residues, N <= 2^13; what a bootstrapping of one or a few ciphertexts looks like): with one workgroup per (accumulator, limb)
    // a step lasts as long as that workgroup's 2 L K / 2 paired transforms back to back (103 us per external product at N = 8192, L = 4, w = 16, batch 1).
    // Here a step is three launches: the monomial factor (X^a - 1) once per step (a streaming pass), one workgroup per (accumulator, limb, component, DIGIT) on the
    // 16-per-thread forward transform with that digit's two key products, and one workgroup per (accumulator, limb, output component) that sums the 2 L K partials,
    // runs one inverse transform and adds the accumulator (ntt_keyswitch16_{part,comb}_kernel).  The pair stays compact between the steps.
    // (N = 2^14: the same three launches on the paired 32-per-thread transforms, one workgroup per digit PAIR of a component)
    if (steps >= 1 && h->width == FHE_WIDTH_32 && !h->single_transforms && !h->no_compact_blind_rotate && !h->no_prerotation && fhe_dev::lds_paired_keyswitch(4, (int)h->log_n) &&
        h->split_pairs_polys && batch * h->L <= h->split_pairs_polys) {
        const size_t cbytes = (size_t)batch * h->L * h->n * 4, count = (size_t)batch * h->L * h->n;
        if ((rc = ensure_ws2(h, 6 * cbytes))) return rc;
        char *w0 = (char *)h->d_ws2;
        char *pp[2][2] = {{w0, w0 + cbytes}, {w0 + 2 * cbytes, w0 + 3 * cbytes}};
        char *rot0 = w0 + 4 * cbytes, *rot1 = w0 + 5 * cbytes;       // (X^a - 1) * acc of the current step
        uint32_t kmax = 0;
        for (uint32_t s = 0; s < steps; s++) kmax = rows_c0[s]->K > kmax ? rows_c0[s]->K : kmax;
        const size_t parts = fhe_dev::lds_small_multiply(4, (int)h->log_n) ? 2 * (size_t)h->L * kmax : 2 * (((size_t)h->L * kmax + 1) / 2);
        if ((rc = ensure_ws(h, 2 * count * parts * 4))) return rc;      // two partial accumulators per (limb polynomial, component, digit or digit pair)
        if ((rc = compact_poly(h, pp[1][0], d_acc0, count))) return rc;
        if ((rc = compact_poly(h, pp[1][1], d_acc1, count))) return rc;
        for (uint32_t s = 0; s < steps; s++) {
            const bool last = s + 1 == steps;
            const void *i0 = pp[(s + 1) & 1][0], *i1 = pp[(s + 1) & 1][1];
            void *o0 = last ? d_acc0 : pp[s & 1][0], *o1 = last ? d_acc1 : pp[s & 1][1];
            const uint32_t *sh = d_shifts + (size_t)s * batch;
            if ((rc = monomial_compact(h, rot0, i0, sh, count)) || (rc = monomial_compact(h, rot1, i1, sh, count))) return rc;
            if ((rc = blind_rotate_step_fused(h, rows_c0[s], rows_c1[s], o0, o1, i0, i1, sh, batch, true, !last, rot0, rot1, h->d_ws))) return rc;
        }
        return FHE_OK;
    }
    // Paired kernel (4-byte residues up to N = 2^14): the accumulator pair lives in COMPACT form for the whole call (workspace ping-pong, 4
    // bytes per coefficient): the L limb workgroups of an accumulator each read all of it, which in container form is 3x the algorithmic
    // traffic (profiles/r02_blindrotate_*) and made the first step of a loop 40 % slower than the others (1114 vs 785 us at N = 16384 x 6,
    // profiles/r03_blindrotate_n16384_summary.txt).  Since round 3 the pair is compacted first (one streaming pass), every step reads compact
    // input, the last one writes the caller's containers.  The caller's scratch pair is not touched.
    if (steps >= 1 && h->width == FHE_WIDTH_32 && !h->single_transforms && !h->no_compact_blind_rotate && fhe_dev::lds_paired_extprod(4, (int)h->log_n)) {
        const size_t cbytes = (size_t)batch * h->L * h->n * 4, count = (size_t)batch * h->L * h->n;
        if ((rc = ensure_ws2(h, 4 * cbytes))) return rc;
        char *w0 = (char *)h->d_ws2;
        char *pp[2][2] = {{w0, w0 + cbytes}, {w0 + 2 * cbytes, w0 + 3 * cbytes}};
        if ((rc = compact_poly(h, pp[1][0], d_acc0, count))) return rc;
        if ((rc = compact_poly(h, pp[1][1], d_acc1, count))) return rc;
        for (uint32_t s = 0; s < steps; s++) {
            const bool last = s + 1 == steps;
            const void *i0 = pp[(s + 1) & 1][0], *i1 = pp[(s + 1) & 1][1];
            void *o0 = last ? d_acc0 : pp[s & 1][0], *o1 = last ? d_acc1 : pp[s & 1][1];
            if ((rc = blind_rotate_step_fused(h, rows_c0[s], rows_c1[s], o0, o1, i0, i1, d_shifts + (size_t)s * batch, batch, true, !last))) return rc;
        }
        return FHE_OK;
    }
    // Three-array kernel (8-byte residues; 4-byte residues at N = 2^15): every limb workgroup re-reads each limb of the accumulator pair once
    // per DIGIT (rotated), L * K * 2 reads per workgroup -- as containers that was several times the algorithmic traffic.  The pair is
    // compacted once (compact_kernel), every step reads compact input, all but the last write compact output (workspace ping-pong).
    if (steps >= 1 && use_joint3(h, true, false) && !h->no_compact_blind_rotate) {
        const size_t eb = residue_bytes(h), cbytes = (size_t)batch * h->L * h->n * eb, count = (size_t)batch * h->L * h->n;
        if ((rc = ensure_ws2(h, 6 * cbytes))) return rc;
        char *w0 = (char *)h->d_ws2;
        char *pp[2][2] = {{w0, w0 + cbytes}, {w0 + 2 * cbytes, w0 + 3 * cbytes}};
        char *rot0 = w0 + 4 * cbytes, *rot1 = w0 + 5 * cbytes;       // (X^a - 1) * acc of the current step
        if ((rc = compact_poly(h, pp[1][0], d_acc0, count))) return rc;
        if ((rc = compact_poly(h, pp[1][1], d_acc1, count))) return rc;
        for (uint32_t s = 0; s < steps; s++) {
            const bool last = s + 1 == steps;
            const void *i0 = pp[(s + 1) & 1][0], *i1 = pp[(s + 1) & 1][1];
            void *o0 = last ? d_acc0 : pp[s & 1][0], *o1 = last ? d_acc1 : pp[s & 1][1];
            const uint32_t *sh = d_shifts + (size_t)s * batch;
            // the monomial factor once per step (a streaming pass over two compact polynomials) instead of once per digit inside the kernel
            const bool prerot = !h->no_prerotation;
            if (prerot && ((rc = monomial_compact(h, rot0, i0, sh, count)) || (rc = monomial_compact(h, rot1, i1, sh, count)))) return rc;
            if ((rc = blind_rotate_step_fused(h, rows_c0[s], rows_c1[s], o0, o1, i0, i1, sh, batch, true, !last, prerot ? rot0 : nullptr, prerot ? rot1 : nullptr))) return rc;
        }
        return FHE_OK;
    }
    // ping-pong between (acc0, acc1) and (tmp0, tmp1): one launch per step, 4*S bytes of HBM traffic per accumulator and step
    void *cur0 = d_acc0, *cur1 = d_acc1, *nxt0 = d_tmp0, *nxt1 = d_tmp1;
    for (uint32_t s = 0; s < steps; s++) {
        if ((rc = blind_rotate_step_fused(h, rows_c0[s], rows_c1[s], nxt0, nxt1, cur0, cur1, d_shifts + (size_t)s * batch, batch))) return rc;
        std::swap(cur0, nxt0); std::swap(cur1, nxt1);
    }
    if (cur0 != d_acc0) {   // odd number of steps: the result sits in the scratch pair
        const size_t bytes = (size_t)batch * h->L * h->n * 32;
        HIP_TRY(hipMemcpyAsync(d_acc0, cur0, bytes, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(d_acc1, cur1, bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    return FHE_OK;
}
extern "C" int fhe_blind_rotate_step(fhe_rns_ntt_t *h, const fhe_relin_keys_t *rows_c0, const fhe_relin_keys_t *rows_c1, void *d_acc0, void *d_acc1,
                                     const uint32_t *d_shifts, void *d_tmp0, void *d_tmp1, uint32_t batch) {
    return fhe_blind_rotate(h, &rows_c0, &rows_c1, 1, d_acc0, d_acc1, d_shifts, d_tmp0, d_tmp1, batch);
}

// ------------------------------------------------------------------------------------------------------
// scheme plumbing around the hot path: samplers, single-modulus modulus switch, negacyclic fold
// ------------------------------------------------------------------------------------------------------
template <int KIND>
static int sample_literal(void *d_out, const uint64_t q[4], uint64_t seed, size_t count, void *stream, const char *what) {
    if (!d_out || !q) return fail(FHE_ERR_INVALID_ARG, std::string(what) + ": null argument");
    if (!q[0]) return fail(FHE_ERR_BAD_MODULUS, std::string(what) + ": modulus.limbs[0] is zero (the reference would divide by zero)");
    if (count > 0xffffffffull) return fail(FHE_ERR_INVALID_ARG, std::string(what) + ": the reference indexes with a uint32_t");
    int rc = ensure_device(); if (rc) return rc;
    if (!count) return FHE_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(fhe_dev::sample_literal_kernel<KIND>, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, (fhe_dev::u256 *)d_out, q[0], seed, count);
    return post_launch((hipStream_t)stream, what);
}
extern "C" int fhe_sample_uniform_lcg(void *d_out, const uint64_t q[4], uint64_t seed, size_t count, void *stream) {
    return sample_literal<0>(d_out, q, seed, count, stream, "fhe_sample_uniform_lcg");
}
extern "C" int fhe_sample_gaussian_placeholder(void *d_out, const uint64_t q[4], uint64_t seed, size_t count, void *stream) {
    return sample_literal<1>(d_out, q, seed, count, stream, "fhe_sample_gaussian_placeholder");
}

// Cumulative table of the discrete Gaussian D_sigma over the integers, cut at 12 sigma:  w_k = exp(-k^2 / (2 sigma^2)),
// Z = w_0 + 2 sum_{k=1..len} w_k, table[k] = floor(2^64 * P(|X| <= k)) for k = 0 .. len-1 (clamped to 2^64 - 1).
// Plain IEEE double arithmetic in a fixed order, so the CPU oracle's table is identical.
extern "C" int fhe_gaussian_cdt(double sigma, uint64_t *table, uint32_t capacity, uint32_t *len_out) {
    if (!len_out) return fail(FHE_ERR_INVALID_ARG, "gaussian_cdt: null argument");
    if (!(sigma > 0) || sigma > 1e6) return fail(FHE_ERR_INVALID_ARG, "gaussian_cdt: sigma must be in (0, 1e6]");
    uint32_t len = (uint32_t)std::ceil(sigma * 12.0);
    if (len < 1) len = 1;
    *len_out = len;
    if (!table) return FHE_OK;                                    // size query
    if (capacity < len) return fail(FHE_ERR_INVALID_ARG, "gaussian_cdt: table too small");
    const double den = (2.0 * sigma) * sigma;
    std::vector<double> w(len + 1);
    for (uint32_t k = 0; k <= len; k++) w[k] = std::exp(-((double)k * (double)k) / den);
    double Z = w[0];
    for (uint32_t k = 1; k <= len; k++) Z += 2.0 * w[k];
    double cum = 0;
    for (uint32_t k = 0; k < len; k++) {
        const double term = (k == 0 ? w[0] : 2.0 * w[k]) / Z;
        cum += term;
        const double scaled = cum * 18446744073709551616.0;
        table[k] = scaled >= 18446744073709551615.0 ? ~0ull : (uint64_t)scaled;
    }
    return FHE_OK;
}
static bool small_fits(const fhe_rns_ntt *h, uint64_t max_magnitude) {   // max_magnitude < every q_l ?
    for (const U256 &q : h->moduli) if (!(q.w[1] | q.w[2] | q.w[3]) && q.w[0] <= max_magnitude) return false;
    return true;
}
extern "C" int fhe_rns_sample_ternary(fhe_rns_ntt_t *h, void *d_out, double probability, uint64_t seed, uint32_t batch) {
    int rc = check_call(h, batch, "rns_sample_ternary"); if (rc) return rc;
    if (!d_out) return fail(FHE_ERR_INVALID_ARG, "rns_sample_ternary: null output");
    if (!(probability >= 0.0 && probability <= 1.0)) return fail(FHE_ERR_INVALID_ARG, "rns_sample_ternary: probability must be in [0, 1]");
    if ((rc = ensure_crt(h))) return rc;
    const uint64_t thr = (uint64_t)(probability * 4294967296.0);
    const size_t count = (size_t)batch * h->n;
    hipLaunchKernelGGL(fhe_dev::sample_small_kernel<0>, dim3(ew_grid(count)), dim3(256), 0, h->stream, (fhe_dev::u256 *)d_out,
                       (const fhe_dev::CrtLimb *)h->d_crt, h->L, h->log_n, seed, thr, (const uint64_t *)nullptr, 0u, count);
    return post_launch(h->stream, "sample_small_kernel<ternary>");
}
extern "C" int fhe_rns_sample_gaussian(fhe_rns_ntt_t *h, void *d_out, double sigma, uint64_t seed, uint32_t batch) {
    int rc = check_call(h, batch, "rns_sample_gaussian"); if (rc) return rc;
    if (!d_out) return fail(FHE_ERR_INVALID_ARG, "rns_sample_gaussian: null output");
    if (h->cdt_sigma != sigma || !h->d_cdt) {
        uint32_t len = 0;
        if ((rc = fhe_gaussian_cdt(sigma, nullptr, 0, &len))) return rc;
        if (!small_fits(h, len)) return fail(FHE_ERR_INVALID_ARG, "rns_sample_gaussian: 12 sigma does not fit below the smallest modulus");
        std::vector<uint64_t> t(len);
        if ((rc = fhe_gaussian_cdt(sigma, t.data(), len, &len))) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));                 // an earlier launch may still read the old table
        if (h->d_cdt) { HIP_TRY(hipFree(h->d_cdt)); h->d_cdt = nullptr; }
        HIP_TRY(hipMalloc((void **)&h->d_cdt, len * sizeof(uint64_t)));
        HIP_TRY(hipMemcpy(h->d_cdt, t.data(), len * sizeof(uint64_t), hipMemcpyHostToDevice));
        h->cdt_sigma = sigma; h->cdt_len = len;
    }
    if ((rc = ensure_crt(h))) return rc;
    const size_t count = (size_t)batch * h->n;
    hipLaunchKernelGGL(fhe_dev::sample_small_kernel<1>, dim3(ew_grid(count)), dim3(256), 0, h->stream, (fhe_dev::u256 *)d_out,
                       (const fhe_dev::CrtLimb *)h->d_crt, h->L, h->log_n, seed, (uint64_t)0, (const uint64_t *)h->d_cdt, h->cdt_len, count);
    return post_launch(h->stream, "sample_small_kernel<gaussian>");
}
extern "C" int fhe_rns_sample_uniform(fhe_rns_ntt_t *h, void *d_out, uint64_t seed, uint32_t batch) {
    int rc = check_call(h, batch, "rns_sample_uniform"); if (rc) return rc;
    if (!d_out) return fail(FHE_ERR_INVALID_ARG, "rns_sample_uniform: null output");
    if ((rc = ensure_crt(h))) return rc;
    const size_t count = (size_t)batch * h->L * h->n;
    hipLaunchKernelGGL(fhe_dev::sample_uniform_rns_kernel, dim3(ew_grid(count)), dim3(256), 0, h->stream, (fhe_dev::u256 *)d_out,
                       (const fhe_dev::CrtLimb *)h->d_crt, h->L, h->log_n, seed, count);
    return post_launch(h->stream, "sample_uniform_rns_kernel");
}
extern "C" int fhe_poly_mod_switch(void *d_r, const void *d_a, const uint64_t old_q[4], const uint64_t new_q[4], size_t count, void *stream) {
    if (!d_r || !d_a || !old_q || !new_q) return fail(FHE_ERR_INVALID_ARG, "poly_mod_switch: null argument");
    U256 O = U256::from(old_q);
    if (O.bit_length() < 2 || (O.w[3] >> 63)) return fail(FHE_ERR_BAD_MODULUS, "poly_mod_switch: old modulus must be in [2, 2^255)");
    if (new_q[1] | new_q[2] | new_q[3]) return fail(FHE_ERR_UNSUPPORTED, "poly_mod_switch: the new modulus must be below 2^64");
    if (!new_q[0]) return fail(FHE_ERR_BAD_MODULUS, "poly_mod_switch: new modulus is zero");
    int rc = ensure_device(); if (rc) return rc;
    if (!count) return FHE_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(fhe_dev::poly_mod_switch_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, (fhe_dev::u256 *)d_r,
                       (const fhe_dev::u256 *)d_a, to_dev(old_q), new_q[0], count);
    return post_launch((hipStream_t)stream, "poly_mod_switch_kernel");
}
extern "C" int fhe_negacyclic_reduce(void *d_data, const uint64_t q[4], size_t n, void *stream) {
    if (!d_data || !q) return fail(FHE_ERR_INVALID_ARG, "negacyclic_reduce: null argument");
    int rc = ensure_device(); if (rc) return rc;
    if (!n) return FHE_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(fhe_dev::negacyclic_reduce_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (fhe_dev::u256 *)d_data, to_dev(q), n);
    return post_launch((hipStream_t)stream, "negacyclic_reduce_kernel");
}

// ------------------------------------------------------------------------------------------------------
// single-modulus engine ABI = RNS engine with one limb
// ------------------------------------------------------------------------------------------------------
extern "C" int fhe_ntt_create(fhe_ntt_t **out, uint32_t n, const uint64_t q[4]) {
    if (!out || !q) return fail(FHE_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    uint64_t m[1][4]; std::memcpy(m[0], q, 32);
    fhe_rns_ntt *impl = nullptr;
    int rc = create_impl(&impl, n, m, 1); if (rc) return rc;
    fhe_ntt *h = new (std::nothrow) fhe_ntt{impl};
    if (!h) { destroy_impl(impl); return fail(FHE_ERR_INVALID_ARG, "out of host memory"); }
    *out = h;
    return FHE_OK;
}
extern "C" int fhe_ntt_destroy(fhe_ntt_t *h) { if (h) { destroy_impl(h->impl); delete h; } return FHE_OK; }
extern "C" int fhe_ntt_set_stream(fhe_ntt_t *h, void *stream) { return h ? fhe_rns_ntt_set_stream(h->impl, stream) : fail(FHE_ERR_INVALID_ARG, "null handle"); }
extern "C" int fhe_ntt_width_class(const fhe_ntt_t *h) { return h ? h->impl->width : fail(FHE_ERR_INVALID_ARG, "null handle"); }
extern "C" int fhe_ntt_forward(fhe_ntt_t *h, void *d, uint32_t batch) { return h ? fhe_rns_ntt_forward(h->impl, d, batch) : fail(FHE_ERR_INVALID_ARG, "null handle"); }
extern "C" int fhe_ntt_inverse(fhe_ntt_t *h, void *d, uint32_t batch) { return h ? fhe_rns_ntt_inverse(h->impl, d, batch) : fail(FHE_ERR_INVALID_ARG, "null handle"); }
extern "C" int fhe_ntt_pointwise(fhe_ntt_t *h, void *r, const void *a, const void *b, uint32_t batch) {
    return h ? fhe_rns_ntt_pointwise(h->impl, r, a, b, batch) : fail(FHE_ERR_INVALID_ARG, "null handle");
}
extern "C" int fhe_ntt_multiply(fhe_ntt_t *h, void *r, const void *a, const void *b, uint32_t batch) {
    return h ? fhe_rns_ntt_multiply(h->impl, r, a, b, batch) : fail(FHE_ERR_INVALID_ARG, "null handle");
}

// ------------------------------------------------------------------------------------------------------
// timers
// ------------------------------------------------------------------------------------------------------
struct fhe_timer { hipEvent_t start, stop; };
extern "C" int fhe_timer_create(fhe_timer_t **out) {
    if (!out) return fail(FHE_ERR_INVALID_ARG, "null argument");
    int rc = ensure_device(); if (rc) return rc;
    fhe_timer *t = new (std::nothrow) fhe_timer();
    if (!t) return fail(FHE_ERR_INVALID_ARG, "out of host memory");
    HIP_TRY(hipEventCreate(&t->start)); HIP_TRY(hipEventCreate(&t->stop));
    *out = t;
    return FHE_OK;
}
extern "C" int fhe_timer_destroy(fhe_timer_t *t) {
    if (t) { (void)hipEventDestroy(t->start); (void)hipEventDestroy(t->stop); delete t; }
    return FHE_OK;
}
extern "C" int fhe_rns_timer_start(fhe_rns_ntt_t *h, fhe_timer_t *t) {
    if (!h || !t) return fail(FHE_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipEventRecord(t->start, h->stream)); return FHE_OK;
}
extern "C" int fhe_rns_timer_stop(fhe_rns_ntt_t *h, fhe_timer_t *t) {
    if (!h || !t) return fail(FHE_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipEventRecord(t->stop, h->stream)); return FHE_OK;
}
extern "C" int fhe_timer_elapsed_ms(fhe_timer_t *t, float *ms) {
    if (!t || !ms) return fail(FHE_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipEventSynchronize(t->stop));
    HIP_TRY(hipEventElapsedTime(ms, t->start, t->stop));
    return FHE_OK;
}
