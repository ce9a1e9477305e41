// fhe/polynomial.hpp -- mirror of fhe::Polynomial / fhe::PolynomialOps (include/polynomial.cuh:10-59)
// for the operations on the multiply path: add, sub, mul_ntt, mul_scalar.
#pragma once
#include <algorithm>

#include "ntt.hpp"

namespace fhe {

// Owning device buffer.  The reference allocates degree+1 containers and adds over degree+1
// (src/polynomial.cu:8,37) while its NTT works on `degree` coefficients; here `degree`
// is the ring dimension n and exactly n coefficients (x num_limbs for RNS polynomials) are held.
struct Polynomial {
    uint256_t *coeffs;   // device memory, [num_limbs][degree]
    uint32_t degree;
    uint256_t modulus;   // first limb's modulus for RNS polynomials
    bool is_ntt_form;
    uint32_t num_limbs;

    Polynomial(uint32_t deg, const uint256_t &mod, uint32_t limbs = 1)
        : coeffs(nullptr), degree(deg), modulus(mod), is_ntt_form(false), num_limbs(limbs) {
        coeffs = device_alloc((size_t)deg * limbs);                                    // src/polynomial.cu:8
        check(fhe_hip_memset(coeffs, 0, (size_t)deg * limbs * sizeof(uint256_t)), "Polynomial memset");   // :9
    }
    ~Polynomial() { device_free(coeffs); }                                            // src/polynomial.cu:12-14
    Polynomial(const Polynomial &) = delete;                                           // the reference's shallow copies double-free
    Polynomial &operator=(const Polynomial &) = delete;
    size_t count() const { return (size_t)degree * num_limbs; }
};

class PolynomialOps {
public:
    // non-owning `ntt`, exactly like the reference (include/polynomial.cuh:23)
    PolynomialOps(uint32_t max_degree, const uint256_t &modulus, NTTEngine *ntt)
        : max_degree_(max_degree), modulus_(modulus), mont_params_(compute_montgomery_params(modulus)), ntt_engine_(ntt) {}

    void add(Polynomial &result, const Polynomial &a, const Polynomial &b) {          // src/polynomial.cu:36-43
        batch_mod_add(result.coeffs, a.coeffs, b.coeffs, modulus_, std::min(a.count(), b.count()));
    }
    void sub(Polynomial &result, const Polynomial &a, const Polynomial &b) {          // src/polynomial.cu:45-52
        batch_mod_sub(result.coeffs, a.coeffs, b.coeffs, modulus_, std::min(a.count(), b.count()));
    }
    void mul_ntt(Polynomial &result, const Polynomial &a, const Polynomial &b) {      // src/polynomial.cu:54-58
        if (ntt_engine_) ntt_engine_->multiply(result.coeffs, a.coeffs, b.coeffs);
    }
    // literal poly_mul_scalar_kernel: result = mont(a, scalar) (carries R^-1, src/polynomial.cu:98-111)
    void mul_scalar(Polynomial &result, const Polynomial &a, const uint256_t &scalar) {
        check(fhe_u256_mont_mul_scalar(result.coeffs, a.coeffs, scalar.limbs, modulus_.limbs, mont_params_.inv.limbs[0], a.count(), nullptr),
              "PolynomialOps::mul_scalar");
    }
    // mul / mul_negacyclic (include/polynomial.cuh:29,38-39, undefined in the reference) are what mul_ntt computes here:
    // the product in Z_q[x]/(x^n + 1).
    void mul(Polynomial &result, const Polynomial &a, const Polynomial &b) { mul_ntt(result, a, b); }
    void mul_negacyclic(Polynomial &result, const Polynomial &a, const Polynomial &b) { mul_ntt(result, a, b); }
    // mod_switch (include/polynomial.cuh:41-42, undefined; its kernel poly_mod_switch_kernel is what FHEContext::decrypt
    // launches, src/fhe.cu:181-184): result[i] = round(a[i] * new_modulus / modulus) mod new_modulus, new_modulus < 2^64.
    void mod_switch(Polynomial &result, const Polynomial &a, const uint256_t &new_modulus) {
        check(fhe_poly_mod_switch(result.coeffs, a.coeffs, modulus_.limbs, new_modulus.limbs, a.count(), nullptr), "PolynomialOps::mod_switch");
    }

private:
    uint32_t max_degree_;
    uint256_t modulus_;
    MontgomeryParams mont_params_;
    NTTEngine *ntt_engine_;
};

}  // namespace fhe
