// fhe/rns.hpp -- mirror of the reference's fhe::RNSContext (include/rns.cuh:20-66, src/rns.cu:6-91) over the HIP C ABI.
// Same constructor and method signatures; buffers are the reference's interleaved layout [count][num_primes] of 32-byte
// containers (src/rns.cu:103-104: value_idx = idx / num_primes, prime_idx = idx % num_primes).
//
// What is literal and what is intent:
//   add_rns / sub_rns / mul_rns -- the reference defines these kernels (src/rns.cu:143-181; sub by analogy): add_mod, sub_mod and
//     mul_mod_montgomery per limb, reproduced bit for bit (mul_rns therefore carries R^-1 = 2^-256, like the reference's).
//   to_rns / from_rns / mod_switch_rns / base_extend -- placeholders or declarations only in the reference (src/rns.cu:93-141,
//     include/rns.cuh:44-48): implemented as the real conversions (exact residues, CRT, rounded drop of the last prime,
//     Bajard fast base conversion).
#pragma once
#include <memory>
#include <vector>

#include "bigint.hpp"

namespace fhe {

struct RNSBase {                              // include/rns.cuh:9-17 (the parts a caller reads)
    std::vector<uint256_t> primes;
    uint32_t num_primes = 0;
};

class RNSContext {
public:
    explicit RNSContext(const std::vector<uint256_t> &primes) {                                   // src/rns.cu:6-29
        base_.primes = primes; base_.num_primes = (uint32_t)primes.size();
        check(fhe_rns_base_create(&h_, reinterpret_cast<const uint64_t(*)[4]>(base_.primes.data()), base_.num_primes), "RNSContext");
    }
    ~RNSContext() { fhe_rns_ntt_destroy(h_); }
    RNSContext(const RNSContext &) = delete;
    RNSContext &operator=(const RNSContext &) = delete;

    // values[count] -> residues[count][num_primes]: exact residues of any 256-bit value (src/rns.cu:56-63 launches a placeholder copy)
    void to_rns(uint256_t *d_rns_residues, const uint256_t *d_values, uint32_t count) { check(fhe_rns_to_rns(h_, d_rns_residues, d_values, count), "RNSContext::to_rns"); }
    // CRT reconstruction modulo the product of the primes (which must stay below 2^255); src/rns.cu:65-72 launches a placeholder
    void from_rns(uint256_t *d_values, const uint256_t *d_rns_residues, uint32_t count) { check(fhe_rns_from_rns(h_, d_values, d_rns_residues, count), "RNSContext::from_rns"); }
    void add_rns(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b, uint32_t count) {   // src/rns.cu:74-81, kernel :143-158 (literal add_mod)
        check(fhe_rns_poly_add(h_, d_result, d_a, d_b, count), "RNSContext::add_rns");
    }
    void sub_rns(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b, uint32_t count) {   // include/rns.cuh:40 (declared): literal sub_mod
        check(fhe_rns_poly_sub(h_, d_result, d_a, d_b, count), "RNSContext::sub_rns");
    }
    void mul_rns(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b, uint32_t count) {   // src/rns.cu:83-90, kernel :160-181 (literal Montgomery product)
        check(fhe_rns_mul_mont_literal(h_, d_result, d_a, d_b, count), "RNSContext::mul_rns");
    }
    // addition (not in the reference): the plain product a*b mod q_l
    void mul_rns_plain(uint256_t *d_result, const uint256_t *d_a, const uint256_t *d_b, uint32_t count) {
        check(fhe_rns_ntt_pointwise(h_, d_result, d_a, d_b, count), "RNSContext::mul_rns_plain");
    }

    // include/rns.cuh:43-45 (declared): level l keeps the first num_primes - l primes; going from old_level to new_level drops the
    // last prime new_level - old_level times with rounding (x -> round(x / q_last) in the remaining base).  d_result holds
    // [count][num_primes - new_level] containers.
    void mod_switch_rns(uint256_t *d_result, const uint256_t *d_input, uint32_t old_level, uint32_t new_level, uint32_t count) {
        if (new_level < old_level || new_level >= base_.num_primes) throw std::runtime_error("RNSContext::mod_switch_rns: need old_level <= new_level < num_primes");
        const uint32_t L0 = base_.num_primes - old_level;
        if (new_level == old_level) { check(fhe_hip_memcpy_d2d(d_result, d_input, (size_t)count * L0 * sizeof(uint256_t)), "mod_switch_rns copy"); return; }
        void *tmp[2] = {nullptr, nullptr};
        const uint256_t *src = d_input;
        for (uint32_t lvl = old_level; lvl < new_level; lvl++) {
            const bool last = lvl + 1 == new_level;
            uint256_t *dst = d_result;
            if (!last) {
                void *&t = tmp[(lvl - old_level) & 1];
                if (!t) check(fhe_hip_malloc(&t, (size_t)count * (base_.num_primes - lvl - 1) * sizeof(uint256_t)), "mod_switch_rns workspace");
                dst = static_cast<uint256_t *>(t);
            }
            check(fhe_rns_rescale_drop_last(level_handle(lvl), dst, src, count), "RNSContext::mod_switch_rns");
            src = dst;
        }
        check(fhe_hip_sync(), "mod_switch_rns sync");
        for (void *t : tmp) if (t) fhe_hip_free(t);
    }
    // include/rns.cuh:47-48 (declared; the reference passes the target RNSBase): fast base conversion into the target's base,
    // d_extended = [count][target.num_primes()].  The result is x + alpha * Q for some 0 <= alpha < num_primes (Bajard et al.).
    void base_extend(uint256_t *d_extended, const uint256_t *d_input, RNSContext &target, uint32_t count) {
        check(fhe_rns_fast_base_convert(h_, target.h_, d_extended, d_input, count), "RNSContext::base_extend");
    }

    uint32_t num_primes() const { return base_.num_primes; }
    const RNSBase &base() const { return base_; }
    fhe_rns_ntt_t *handle() { return h_; }

private:
    RNSBase base_;
    fhe_rns_ntt_t *h_ = nullptr;
    std::vector<std::unique_ptr<RNSContext>> levels_;     // levels_[l - 1] = base over the first num_primes - l primes

    fhe_rns_ntt_t *level_handle(uint32_t level) {
        if (level == 0) return h_;
        if (levels_.size() < level) levels_.resize(level);
        if (!levels_[level - 1])
            levels_[level - 1].reset(new RNSContext(std::vector<uint256_t>(base_.primes.begin(), base_.primes.end() - level)));
        return levels_[level - 1]->h_;
    }
};

}  // namespace fhe
