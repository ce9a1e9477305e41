// fhe/fhe.hpp -- mirror of the parts of fhe::FHEContext on the multiply path (include/fhe.cuh:15-148,
// src/fhe.cu:7-52,187-235): SecurityParams, SchemeParams, Ciphertext, RelinKeys, FHEContext::add /
// multiply / relinearize.  Key generation, encoding, encryption, rotations and bootstrapping are out
// of scope of this engine.
#pragma once
#include <algorithm>
#include <memory>
#include <vector>

#include "polynomial.hpp"

namespace fhe {

struct SecurityParams {          // include/fhe.cuh:15-21
    uint32_t lambda;
    uint32_t poly_degree;
    uint32_t log_q;
    float sigma;
    uint32_t hamming_weight;
};

struct PublicKey { Polynomial *pk0; Polynomial *pk1; };      // include/fhe.cuh:41-44
struct RelinKeys {                                            // include/fhe.cuh:52-55
    std::vector<PublicKey *> rlk_keys;
    uint32_t decomp_bits = 16;
};

struct Ciphertext {                                           // include/fhe.cuh:63-69
    std::vector<Polynomial *> components;
    uint32_t level = 0;
    float noise_budget = 0.f;
    bool is_ntt_form = false;
};

struct SchemeParams {                                         // include/fhe.cuh:24-38 (the parts that are used)
    SecurityParams security;
    uint32_t n;
    std::vector<uint256_t> rns_moduli;     // q = prod(rns_moduli); the reference hard-codes the even q = 2^60 (src/fhe.cu:13)
    RNS_NTTEngine *rns_ntt;
};

class FHEContext {
public:
    // Honours poly_degree and log_q (the reference ignores log_q): L = ceil(log_q / 30) NTT primes of
    // ceil(log_q / L) bits each, the smallest ones >= 2^(bits-1) with q = 1 (mod 2n).
    explicit FHEContext(const SecurityParams &params) {
        uint32_t L = std::max(1u, (params.log_q + 29) / 30);
        uint32_t bits = std::max(20u, (params.log_q + L - 1) / L);
        std::vector<uint64_t> primes(L);
        check(fhe_find_ntt_primes(bits, params.poly_degree, L, primes.data()), "FHEContext: prime search");
        std::vector<uint256_t> moduli;
        for (uint64_t p : primes) moduli.emplace_back(p);
        init(params, moduli);
    }
    // Explicit RNS basis (e.g. BASELINE config 4: N = 16384, 6 limbs).
    FHEContext(const SecurityParams &params, const std::vector<uint256_t> &rns_moduli) { init(params, rns_moduli); }
    ~FHEContext() { delete params_.rns_ntt; }
    FHEContext(const FHEContext &) = delete;
    FHEContext &operator=(const FHEContext &) = delete;

    Polynomial *new_polynomial() const { return new Polynomial(params_.n, params_.rns_moduli[0], (uint32_t)params_.rns_moduli.size()); }

    // src/fhe.cu:187-197
    void add(Ciphertext &result, const Ciphertext &a, const Ciphertext &b) {
        size_t num = std::max(a.components.size(), b.components.size());
        ensure_components(result, num);
        for (size_t i = 0; i < num; i++) {
            if (i < a.components.size() && i < b.components.size())
                params_.rns_ntt->add_rns(result.components[i]->coeffs, a.components[i]->coeffs, b.components[i]->coeffs);
            else {
                const Polynomial *src = i < a.components.size() ? a.components[i] : b.components[i];
                check(fhe_hip_memcpy_d2d(result.components[i]->coeffs, src->coeffs, src->count() * sizeof(uint256_t)), "add copy");
            }
        }
        result.noise_budget = std::min(a.noise_budget, b.noise_budget);
        result.level = std::max(a.level, b.level);
    }

    // src/fhe.cu:199-224: tensor product in ONE fused launch (4 forward + 3 inverse transforms instead of
    // the reference's 8 + 4), then relinearize().
    void multiply(Ciphertext &result, const Ciphertext &a, const Ciphertext &b, const RelinKeys &rlk) {
        if (a.components.size() != 2 || b.components.size() != 2) throw std::runtime_error("FHEContext::multiply: 2-component ciphertexts expected");
        ensure_components(result, 3);
        params_.rns_ntt->tensor_multiply(result.components[0]->coeffs, result.components[1]->coeffs, result.components[2]->coeffs,
                                         a.components[0]->coeffs, a.components[1]->coeffs, b.components[0]->coeffs, b.components[1]->coeffs);
        relinearize(result, rlk);
        result.noise_budget = a.noise_budget + b.noise_budget + 10;   // the reference's rough estimate (src/fhe.cu:222)
        result.level = std::max(a.level, b.level);
    }

    // src/fhe.cu:226-235 is a stub that drops c2 (which breaks decryption).  Until the key-switch row
    //  lands this keeps all three components, so no information is lost.
    void relinearize(Ciphertext &ct, const RelinKeys &rlk) { (void)ct; (void)rlk; }

    const SchemeParams &params() const { return params_; }

private:
    SchemeParams params_;

    void init(const SecurityParams &params, const std::vector<uint256_t> &moduli) {
        params_.security = params;
        params_.n = params.poly_degree;
        params_.rns_moduli = moduli;
        params_.rns_ntt = new RNS_NTTEngine(params_.n, params_.rns_moduli.data(), (uint32_t)moduli.size());
    }
    void ensure_components(Ciphertext &ct, size_t num) {
        while (ct.components.size() < num) ct.components.push_back(new_polynomial());   // the reference `new`s and never frees (src/fhe.cu:202-205)
    }
};

}  // namespace fhe
