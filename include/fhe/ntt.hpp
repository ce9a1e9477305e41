// fhe/ntt.hpp -- mirror of the reference's fhe::NTTEngine / fhe::RNS_NTTEngine (include/ntt.cuh:72-137)
// over the HIP C ABI.  Same constructor and method signatures; d_* are raw device pointers owned by
// the caller; transforms are in place; work is enqueued on the engine's stream, callers synchronise.
#pragma once
#include <vector>

#include "bigint.hpp"

namespace fhe {

class NTTEngine {
public:
    NTTEngine(uint32_t polynomial_degree, const uint256_t &modulus) : n_(polynomial_degree), modulus_(modulus) {
        check(fhe_ntt_create(&h_, polynomial_degree, modulus.limbs), "NTTEngine");      // src/ntt.cu:7-22
    }
    ~NTTEngine() { fhe_ntt_destroy(h_); }
    NTTEngine(const NTTEngine &) = delete;
    NTTEngine &operator=(const NTTEngine &) = delete;

    void forward(uint256_t *d_data) { check(fhe_ntt_forward(h_, d_data, 1), "NTTEngine::forward"); }          // src/ntt.cu:30-40
    void inverse(uint256_t *d_data) { check(fhe_ntt_inverse(h_, d_data, 1), "NTTEngine::inverse"); }          // src/ntt.cu:42-47
    void multiply(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b) {                            // src/ntt.cu:49-75
        check(fhe_ntt_multiply(h_, d_result, d_a, d_b, 1), "NTTEngine::multiply");
    }
    void forward_batch(uint256_t *d_data, uint32_t batch_size) { check(fhe_ntt_forward(h_, d_data, batch_size), "NTTEngine::forward_batch"); }  // include/ntt.cuh:87
    void inverse_batch(uint256_t *d_data, uint32_t batch_size) { check(fhe_ntt_inverse(h_, d_data, batch_size), "NTTEngine::inverse_batch"); }  // include/ntt.cuh:88
    // additions (not in the reference): batched multiply and the plain NTT-domain product
    void multiply_batch(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b, uint32_t batch_size) {
        check(fhe_ntt_multiply(h_, d_result, d_a, d_b, batch_size), "NTTEngine::multiply_batch");
    }
    void pointwise(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b, uint32_t batch_size = 1) {
        check(fhe_ntt_pointwise(h_, d_result, d_a, d_b, batch_size), "NTTEngine::pointwise");
    }
    void set_stream(void *hip_stream) { check(fhe_ntt_set_stream(h_, hip_stream), "NTTEngine::set_stream"); }
    uint32_t degree() const { return n_; }
    const uint256_t &modulus() const { return modulus_; }
    int width_class() const { return fhe_ntt_width_class(h_); }

private:
    uint32_t n_;
    uint256_t modulus_;
    fhe_ntt_t *h_ = nullptr;
};

class RNS_NTTEngine {
public:
    RNS_NTTEngine(uint32_t polynomial_degree, const uint256_t *rns_moduli, uint32_t num_primes)   // src/ntt.cu:122-145
        : n_(polynomial_degree), num_primes_(num_primes), moduli_(rns_moduli, rns_moduli + num_primes) {
        check(fhe_rns_ntt_create(&h_, polynomial_degree, reinterpret_cast<const uint64_t(*)[4]>(moduli_.data()), num_primes), "RNS_NTTEngine");
    }
    ~RNS_NTTEngine() { fhe_rns_ntt_destroy(h_); }
    RNS_NTTEngine(const RNS_NTTEngine &) = delete;
    RNS_NTTEngine &operator=(const RNS_NTTEngine &) = delete;

    // data limb-major [num_primes][n] (src/ntt.cu:161); `batch` > 1 = [batch][num_primes][n]
    void forward_rns(uint256_t *d_rns_data, uint32_t batch = 1) { check(fhe_rns_ntt_forward(h_, d_rns_data, batch), "forward_rns"); }   // src/ntt.cu:158-164
    void inverse_rns(uint256_t *d_rns_data, uint32_t batch = 1) { check(fhe_rns_ntt_inverse(h_, d_rns_data, batch), "inverse_rns"); }   // src/ntt.cu:166-171
    void multiply_rns(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b, uint32_t batch = 1) {                                // include/ntt.cuh:124-126
        check(fhe_rns_ntt_multiply(h_, d_result, d_a, d_b, batch), "multiply_rns");
    }
    // addition (not in the reference): ONE polynomial d_b_one multiplied into every element of the batch d_a (a key, a plaintext)
    void multiply_rns_bcast(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b_one, uint32_t batch) {
        check(fhe_rns_ntt_multiply_bcast(h_, d_result, d_a, d_b_one, batch), "multiply_rns_bcast");
    }
    // include/ntt.cuh:114-117 (undefined in the reference): d_data is [batch][n] 256-bit integers, d_rns_data [batch][L][n]
    void to_rns(uint256_t *d_rns_data, const uint256_t *d_data, uint32_t batch = 1) { check(fhe_rns_to_rns(h_, d_rns_data, d_data, batch), "to_rns"); }
    void from_rns(uint256_t *d_data, const uint256_t *d_rns_data, uint32_t batch = 1) { check(fhe_rns_from_rns(h_, d_data, d_rns_data, batch), "from_rns"); }
    // RNSContext::base_extend (include/rns.cuh:47-48): fast base conversion into `target`'s primes; d_out is [batch][L'][n]
    void base_extend(uint256_t *d_out, const uint256_t *d_in, RNS_NTTEngine &target, uint32_t batch = 1) {
        check(fhe_rns_fast_base_convert(h_, target.h_, d_out, d_in, batch), "base_extend");
    }
    // RNSContext::mod_switch_rns (include/rns.cuh:44): drop the last prime with rounding; d_out is [batch][L-1][n]
    void rescale_drop_last(uint256_t *d_out, const uint256_t *d_in, uint32_t batch = 1) { check(fhe_rns_rescale_drop_last(h_, d_out, d_in, batch), "rescale_drop_last"); }
    void pointwise_rns(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b, uint32_t batch = 1) {
        check(fhe_rns_ntt_pointwise(h_, d_result, d_a, d_b, batch), "pointwise_rns");
    }
    void add_rns(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b, uint32_t batch = 1) { check(fhe_rns_poly_add(h_, d_result, d_a, d_b, batch), "add_rns"); }
    void sub_rns(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b, uint32_t batch = 1) { check(fhe_rns_poly_sub(h_, d_result, d_a, d_b, batch), "sub_rns"); }
    // c0 = a0 b0, c1 = a0 b1 + a1 b0, c2 = a1 b1  (FHEContext::multiply, src/fhe.cu:199-218)
    void tensor_multiply(uint256_t *d_c0, uint256_t *d_c1, uint256_t *d_c2, const uint256_t *d_a0, const uint256_t *d_a1,
                         const uint256_t *d_b0, const uint256_t *d_b1, uint32_t batch = 1) {
        check(fhe_ct_multiply(h_, d_c0, d_c1, d_c2, d_a0, d_a1, d_b0, d_b1, batch), "tensor_multiply");
    }
    void check_canonical(const uint256_t *d_data, uint32_t batch = 1) { check(fhe_rns_check_canonical(h_, d_data, batch), "check_canonical"); }
    void set_stream(void *hip_stream) { check(fhe_rns_ntt_set_stream(h_, hip_stream), "RNS_NTTEngine::set_stream"); }

    uint32_t degree() const { return n_; }
    uint32_t num_primes() const { return num_primes_; }
    const std::vector<uint256_t> &moduli() const { return moduli_; }
    int width_class() const { return fhe_rns_ntt_width_class(h_); }
    fhe_rns_ntt_t *handle() { return h_; }

private:
    uint32_t n_, num_primes_;
    std::vector<uint256_t> moduli_;
    fhe_rns_ntt_t *h_ = nullptr;
};

// find_primitive_root (include/ntt.cuh:140; src/ntt.cu:110-114 returns 3): a real primitive 2n-th root.
inline uint256_t find_primitive_root(uint32_t n, const uint256_t &modulus) {
    uint256_t psi;
    check(fhe_find_psi(n, modulus.limbs, psi.limbs), "find_primitive_root");
    return psi;
}

}  // namespace fhe
