// host_math.hpp -- host-side 256-bit parameter maths for the engine (product code, no oracle use).
//
// Fills in what the reference leaves as placeholders: R^2 mod q (src/bigint.cu:49), mod_inverse and
// find_primitive_root (src/ntt.cu:110-119), the twiddle tables (src/ntt.cu:86-97), the NTT-prime
// search (src/rns.cu:183-209).  All arithmetic is exact integer arithmetic.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace fhe_host {

typedef unsigned __int128 u128;

struct U256 {
    uint64_t w[4];
    U256() : w{0, 0, 0, 0} {}
    explicit U256(uint64_t v) : w{v, 0, 0, 0} {}
    static U256 from(const uint64_t q[4]) { U256 r; std::memcpy(r.w, q, 32); return r; }
    bool operator==(const U256 &o) const { return std::memcmp(w, o.w, 32) == 0; }
    bool operator!=(const U256 &o) const { return !(*this == o); }
    bool is_zero() const { return !(w[0] | w[1] | w[2] | w[3]); }
    bool bit(int i) const { return (w[i >> 6] >> (i & 63)) & 1; }
    int bit_length() const {
        for (int i = 3; i >= 0; i--) if (w[i]) return 64 * i + 64 - __builtin_clzll(w[i]);
        return 0;
    }
};

inline int cmp(const U256 &a, const U256 &b) {
    for (int i = 3; i >= 0; i--) { if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1; }
    return 0;
}
inline unsigned add_to(U256 &r, const U256 &a, const U256 &b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.w[i] + b.w[i]; r.w[i] = (uint64_t)c; c >>= 64; }
    return (unsigned)c;
}
inline unsigned sub_to(U256 &r, const U256 &a, const U256 &b) {
    unsigned br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a.w[i] - b.w[i] - br; r.w[i] = (uint64_t)d; br = (unsigned)(d >> 64) & 1; }
    return br;
}
inline U256 shr(const U256 &a, unsigned s) {   // 0 <= s < 256
    U256 r; unsigned ws = s >> 6, bs = s & 63;
    for (int i = 0; i < 4; i++) {
        uint64_t lo = (i + ws < 4) ? a.w[i + ws] : 0, hi = (i + ws + 1 < 4) ? a.w[i + ws + 1] : 0;
        r.w[i] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
    }
    return r;
}

// -q^-1 mod 2^64 for odd q (exact; equals the reference's 6-step Newton result for odd q).
inline uint64_t neg_inv64(uint64_t q0) {
    uint64_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - q0 * x;
    return (uint64_t)0 - x;
}

// Modular arithmetic context for an odd modulus q < 2^255 (values kept fully reduced).
struct Mod {
    U256 q; uint64_t inv0; U256 r1, r2;   // R mod q, R^2 mod q, R = 2^256
    explicit Mod(const U256 &q_) : q(q_), inv0(neg_inv64(q_.w[0])) {
        U256 x(1);
        for (int i = 0; i < 512; i++) { x = add(x, x); if (i == 255) r1 = x; }
        r2 = x;
    }
    U256 add(const U256 &a, const U256 &b) const {
        U256 s, t; unsigned c = add_to(s, a, b);
        if (c || cmp(s, q) >= 0) { sub_to(t, s, q); return t; }
        return s;
    }
    U256 sub(const U256 &a, const U256 &b) const {
        U256 d; if (sub_to(d, a, b)) { U256 t; add_to(t, d, q); return t; }
        return d;
    }
    // a*b*R^-1 mod q (operand-interleaved Montgomery; a, b < q).
    U256 mont(const U256 &a, const U256 &b) const {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) { c += (u128)a.w[j] * b.w[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * inv0;
            c = ((u128)m * q.w[0] + t[0]) >> 64;
            for (int j = 1; j < 4; j++) { c += (u128)m * q.w[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        U256 u; std::memcpy(u.w, t, 32);
        if (t[4] || cmp(u, q) >= 0) { U256 d; sub_to(d, u, q); return d; }
        return u;
    }
    U256 to_mont(const U256 &a) const { return mont(a, r2); }
    U256 from_mont(const U256 &a) const { return mont(a, U256(1)); }
    U256 mul(const U256 &a, const U256 &b) const { return mont(mont(a, b), r2); }      // plain a*b mod q
    U256 pow_m(const U256 &base_m, const U256 &e) const {                              // Montgomery in/out
        U256 acc = r1;
        for (int i = e.bit_length() - 1; i >= 0; i--) { acc = mont(acc, acc); if (e.bit(i)) acc = mont(acc, base_m); }
        return acc;
    }
    U256 pow(const U256 &base, const U256 &e) const { return from_mont(pow_m(to_mont(base), e)); }
    U256 reduce(const U256 &a) const {   // a mod q for arbitrary a < 2^256 (shift-subtract)
        U256 r;
        for (int i = 255; i >= 0; i--) { r = add(r, r); if (a.bit(i)) r = add(r, U256(1)); }
        return r;
    }
};

// Miller-Rabin: the first 12 primes as bases are a proof below 3.3e24; 24 more fixed pseudo-random
// bases beyond that (error < 4^-36).
inline bool is_prime(const U256 &n) {
    static const uint64_t small[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n.bit_length() <= 6) {
        for (uint64_t p : small) if (n.w[0] == p) return true;
        if (n.w[0] < 41) return false;
    }
    if (!(n.w[0] & 1)) return false;
    if (n.w[3] >> 63) return false;   // outside the engine's domain
    Mod M(n);
    U256 nm1; sub_to(nm1, n, U256(1));
    int s = 0; while (!nm1.bit(s)) s++;
    U256 d = shr(nm1, s), nm1_m = M.to_mont(nm1);
    std::vector<U256> bases;
    for (uint64_t p : small) bases.push_back(U256(p));
    if (n.bit_length() > 80) {
        uint64_t st = 0x9E3779B97F4A7C15ull ^ n.w[0];
        for (int k = 0; k < 24; k++) {
            U256 b;
            for (int i = 0; i < 4; i++) {
                st += 0x9E3779B97F4A7C15ull; uint64_t z = st;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                b.w[i] = z ^ (z >> 31);
            }
            b = M.reduce(b);
            if (b.bit_length() > 1) bases.push_back(b);
        }
    }
    for (const U256 &a : bases) {
        U256 ar = M.reduce(a);
        if (ar.is_zero()) continue;
        U256 x = M.pow_m(M.to_mont(ar), d);
        if (x == M.r1 || x == nm1_m) continue;
        bool comp = true;
        for (int r = 1; r < s && comp; r++) { x = M.mont(x, x); if (x == nm1_m) comp = false; }
        if (comp) return false;
    }
    return true;
}

inline uint32_t bitrev(uint32_t x, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

// Everything an NTT of size n modulo q needs, in plain (non-Montgomery) form.
struct NttConstants {
    uint32_t n, log_n;
    U256 q, psi, psi_inv, n_inv;
    std::vector<U256> tw, itw;   // tw[k] = psi^bitrev(k), itw[k] = psi^-bitrev(k), k in [0, n)
};

enum BuildStatus { BUILD_OK = 0, BUILD_BAD_N = 1, BUILD_BAD_MODULUS = 2 };

inline BuildStatus find_psi(uint32_t n, const Mod &M, U256 &psi_out) {
    uint32_t log_n = 0; while ((1u << log_n) < n) log_n++;
    U256 qm1; sub_to(qm1, M.q, U256(1));
    if (qm1.w[0] & (2ull * n - 1)) return BUILD_BAD_MODULUS;
    U256 e = shr(qm1, log_n + 1), n_u(n), qm1_m = M.to_mont(qm1);
    for (uint64_t g = 2; g < 100000; g++) {
        U256 c_m = M.pow_m(M.to_mont(U256(g)), e);
        if (M.pow_m(c_m, n_u) == qm1_m) { psi_out = M.from_mont(c_m); return BUILD_OK; }
    }
    return BUILD_BAD_MODULUS;
}

inline BuildStatus build_constants(uint32_t n, const U256 &q, NttConstants &out) {
    if (n < 2 || (n & (n - 1))) return BUILD_BAD_N;
    if (!(q.w[0] & 1) || (q.w[3] >> 63) || q.bit_length() < 2) return BUILD_BAD_MODULUS;
    if (!is_prime(q)) return BUILD_BAD_MODULUS;
    Mod M(q);
    out.n = n; out.q = q; out.log_n = 0; while ((1u << out.log_n) < n) out.log_n++;
    BuildStatus st = find_psi(n, M, out.psi);
    if (st != BUILD_OK) return st;
    U256 psi_m = M.to_mont(out.psi);
    U256 ipsi_m = M.pow_m(psi_m, U256(2ull * n - 1));
    out.psi_inv = M.from_mont(ipsi_m);
    U256 qm2; sub_to(qm2, q, U256(2));
    out.n_inv = M.pow(U256(n), qm2);
    out.tw.assign(n, U256()); out.itw.assign(n, U256());
    U256 pw = M.r1, ipw = M.r1;
    for (uint32_t k = 0; k < n; k++) {
        uint32_t s = bitrev(k, out.log_n);
        out.tw[s] = M.from_mont(pw); out.itw[s] = M.from_mont(ipw);
        pw = M.mont(pw, psi_m); ipw = M.mont(ipw, ipsi_m);
    }
    return BUILD_OK;
}

// `count` smallest primes >= 2^(bits-1) with q = 1 (mod 2n); bits in [lb(2n)+2, 64].
inline bool find_ntt_primes(uint32_t bits, uint32_t n, uint32_t count, uint64_t *out) {
    if (bits < 4 || bits > 64 || n < 2 || (n & (n - 1))) return false;
    u128 step = 2ull * (u128)n, lo = (u128)1 << (bits - 1), hi = (u128)1 << bits;
    if (step >= lo) return false;
    u128 q = (lo / step) * step + 1;
    if (q < lo) q += step;
    uint32_t found = 0;
    for (; found < count && q < hi; q += step) {
        U256 c((uint64_t)q);
        if (is_prime(c)) out[found++] = (uint64_t)q;
    }
    return found == count;
}

// The same search for any width: `count` smallest primes >= 2^(bits-1) with q = 1 (mod 2n), bits in [lb(2n)+2, 255].
inline bool find_ntt_primes_wide(uint32_t bits, uint32_t n, uint32_t count, U256 *out) {
    if (bits < 4 || bits > 255 || n < 2 || (n & (n - 1))) return false;
    uint32_t log2n = 1; while ((1u << log2n) < 2 * n) log2n++;
    if (log2n + 1 >= bits) return false;
    U256 q, step((uint64_t)2 * n), hi;
    q.w[(bits - 1) >> 6] = 1ull << ((bits - 1) & 63);          // 2^(bits-1) = 0 (mod 2n)
    if (bits < 256) { if (bits == 255) { hi.w[3] = 1ull << 63; } else hi.w[bits >> 6] = 1ull << (bits & 63); }
    add_to(q, q, U256(1));
    uint32_t found = 0;
    for (; found < count && cmp(q, hi) < 0; add_to(q, q, step))
        if (is_prime(q)) out[found++] = q;
    return found == count;
}

}  // namespace fhe_host
