// fhe/bigint.hpp -- host-side mirror of the reference's include/bigint.cuh for the HIP engine.
// Same names and layout (fhe::uint256_t = 4 x u64 little-endian, include/bigint.cuh:9-24;
// MontgomeryParams, include/bigint.cuh:167-173); the device primitives live behind the C ABI.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>

#include "../fhe_hip.h"

namespace fhe {

struct uint256_t {
    uint64_t limbs[4];  // little-endian: limbs[0] is least significant
    uint256_t() : limbs{0, 0, 0, 0} {}
    uint256_t(uint64_t val) : limbs{val, 0, 0, 0} {}
    uint256_t(uint64_t l0, uint64_t l1, uint64_t l2, uint64_t l3) : limbs{l0, l1, l2, l3} {}
    bool operator==(const uint256_t &o) const {
        return limbs[0] == o.limbs[0] && limbs[1] == o.limbs[1] && limbs[2] == o.limbs[2] && limbs[3] == o.limbs[3];
    }
    bool operator!=(const uint256_t &o) const { return !(*this == o); }
};
static_assert(sizeof(uint256_t) == 32, "uint256_t must be the 32-byte container of the reference");

// The reference reports no errors at all (every CUDA return code is dropped); the mirror throws.
inline void check(int status, const char *what) {
    if (status != FHE_OK) throw std::runtime_error(std::string(what) + ": " + fhe_hip_last_error());
}

struct MontgomeryParams {
    uint256_t modulus;
    uint256_t r_squared;  // R^2 mod N, R = 2^256 (really computed; the reference leaves 1, src/bigint.cu:49)
    uint256_t inv;        // inv.limbs[0] = -N^-1 mod 2^64, upper limbs 0 (src/bigint.cu:23-40)
};

inline uint256_t compute_montgomery_inverse(const uint256_t &modulus) {
    uint256_t inv;
    check(fhe_montgomery_inverse(modulus.limbs, inv.limbs), "compute_montgomery_inverse");
    return inv;
}

inline MontgomeryParams compute_montgomery_params(const uint256_t &modulus) {
    MontgomeryParams p;
    p.modulus = modulus;
    check(fhe_montgomery_params(modulus.limbs, p.r_squared.limbs, p.inv.limbs), "compute_montgomery_params");
    return p;
}

// Device buffer helpers in the spirit of the reference's raw cudaMalloc/cudaMemcpy use.
inline uint256_t *device_alloc(size_t count) {
    void *p = nullptr;
    check(fhe_hip_malloc(&p, count * sizeof(uint256_t)), "fhe_hip_malloc");
    return static_cast<uint256_t *>(p);
}
inline void device_free(uint256_t *p) { if (p) fhe_hip_free(p); }
inline void copy_to_device(uint256_t *d, const uint256_t *h, size_t count) { check(fhe_hip_memcpy_h2d(d, h, count * sizeof(uint256_t)), "memcpy h2d"); }
inline void copy_to_host(uint256_t *h, const uint256_t *d, size_t count) { check(fhe_hip_memcpy_d2h(h, d, count * sizeof(uint256_t)), "memcpy d2h"); }
inline void device_synchronize() { check(fhe_hip_sync(), "fhe_hip_sync"); }

// batch_mod_{add,sub,mul}_kernel (src/bigint.cu:171-214) as host-callable batch primitives.
inline void batch_mod_add(uint256_t *d_r, const uint256_t *d_a, const uint256_t *d_b, const uint256_t &q, size_t count) {
    check(fhe_u256_add_mod(d_r, d_a, d_b, q.limbs, count, nullptr), "batch_mod_add");
}
inline void batch_mod_sub(uint256_t *d_r, const uint256_t *d_a, const uint256_t *d_b, const uint256_t &q, size_t count) {
    check(fhe_u256_sub_mod(d_r, d_a, d_b, q.limbs, count, nullptr), "batch_mod_sub");
}
inline void batch_mod_mul(uint256_t *d_r, const uint256_t *d_a, const uint256_t *d_b, const uint256_t &q, const uint256_t &inv, size_t count) {
    check(fhe_u256_mont_mul(d_r, d_a, d_b, q.limbs, inv.limbs[0], count, nullptr), "batch_mod_mul");
}

}  // namespace fhe
