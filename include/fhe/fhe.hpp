// fhe/fhe.hpp -- mirror of the parts of fhe::FHEContext on the multiply path (include/fhe.cuh:15-148,
// src/fhe.cu:7-52,187-235): SecurityParams, SchemeParams, Ciphertext, RelinKeys, FHEContext::add /
// multiply / relinearize.  Key generation, encoding, encryption, rotations and bootstrapping are out
// of scope of this engine.
#pragma once
#include <algorithm>
#include <memory>
#include <random>
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
struct SecretKey { Polynomial *sk; };                         // include/fhe.cuh:47-49
struct RelinKeys {                                            // include/fhe.cuh:52-55
    std::vector<PublicKey *> rlk_keys;     // level j*K + k: (b, a) with b = -a*s + e + g*s^2, g = 2^(k*w) in limb j, 0 elsewhere
    uint32_t decomp_bits = 16;
    // engine-side copy (NTT domain, packed for the fused key-switch kernel); built on first use
    mutable fhe_relin_keys_t *imported = nullptr;
    mutable const void *imported_for = nullptr;
    RelinKeys() = default;
    RelinKeys(const RelinKeys &) = delete;
    RelinKeys &operator=(const RelinKeys &) = delete;
    ~RelinKeys() { fhe_relin_keys_destroy(imported); for (PublicKey *k : rlk_keys) { if (k) { delete k->pk0; delete k->pk1; delete k; } } }
};

struct Plaintext {                                            // include/fhe.cuh:72-75
    Polynomial *poly = nullptr;
    bool is_ntt_form = false;
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
    uint64_t t = 65537;                    // plaintext modulus (src/fhe.cu:14); t = 1 (mod 2n) gives n SIMD slots
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
    // sub / add_plain / sub_plain / multiply_plain (include/fhe.cuh:98-104, declared only in the reference).  Plaintexts are the
    // encode()d polynomials (reduced mod t in every limb); with the mirror's BGV-style encryption c0 + c1*s = m + t*e they act
    // on the message directly: component-wise subtraction, +-pt on c0, every component times pt.
    void sub(Ciphertext &result, const Ciphertext &a, const Ciphertext &b) {
        if (a.components.size() != b.components.size()) throw std::runtime_error("FHEContext::sub: ciphertexts of different size");
        ensure_components(result, a.components.size());
        for (size_t i = 0; i < a.components.size(); i++)
            params_.rns_ntt->sub_rns(result.components[i]->coeffs, a.components[i]->coeffs, b.components[i]->coeffs);
        result.noise_budget = std::min(a.noise_budget, b.noise_budget);
        result.level = std::max(a.level, b.level);
    }
    void add_plain(Ciphertext &result, const Ciphertext &ct, const Plaintext &pt) { plain_addsub(result, ct, pt, true); }
    void sub_plain(Ciphertext &result, const Ciphertext &ct, const Plaintext &pt) { plain_addsub(result, ct, pt, false); }
    void multiply_plain(Ciphertext &result, const Ciphertext &ct, const Plaintext &pt) {
        ensure_components(result, ct.components.size());
        for (size_t i = 0; i < ct.components.size(); i++)
            params_.rns_ntt->multiply_rns(result.components[i]->coeffs, ct.components[i]->coeffs, pt.poly->coeffs);
        result.noise_budget = ct.noise_budget; result.level = ct.level;
    }

    // src/fhe.cu:199-224: tensor product (4 forward + 3 inverse transforms instead of the reference's 8 + 4), then relinearisation.
    // With keys the two steps are ONE ABI call (fhe_ct_multiply_relin: c2 never leaves the library's workspace); with an empty
    // RelinKeys the result keeps its three components, as relinearize() does.
    void multiply(Ciphertext &result, const Ciphertext &a, const Ciphertext &b, const RelinKeys &rlk) {
        if (a.components.size() != 2 || b.components.size() != 2) throw std::runtime_error("FHEContext::multiply: 2-component ciphertexts expected");
        if (rlk.rlk_keys.empty()) {
            ensure_components(result, 3);
            params_.rns_ntt->tensor_multiply(result.components[0]->coeffs, result.components[1]->coeffs, result.components[2]->coeffs,
                                             a.components[0]->coeffs, a.components[1]->coeffs, b.components[0]->coeffs, b.components[1]->coeffs);
        } else {
            import_relin_keys(rlk);
            ensure_components(result, 2);
            while (result.components.size() > 2) { delete result.components.back(); result.components.pop_back(); }
            check(fhe_ct_multiply_relin(params_.rns_ntt->handle(), rlk.imported, result.components[0]->coeffs, result.components[1]->coeffs,
                                        a.components[0]->coeffs, a.components[1]->coeffs, b.components[0]->coeffs, b.components[1]->coeffs, 1),
                  "FHEContext::multiply");
            device_synchronize();
        }
        result.noise_budget = a.noise_budget + b.noise_budget + 10;   // the reference's rough estimate (src/fhe.cu:222)
        result.level = std::max(a.level, b.level);
    }

    // src/fhe.cu:226-235 is a stub that drops c2 (which breaks decryption); docs/ARCHITECTURE.md:319-326 gives the
    // intended algorithm, implemented here as RNS digit decomposition + key switching (fhe_ct_relinearize).
    // With no keys (an empty RelinKeys) the ciphertext keeps its three components, so no information is lost.
    void relinearize(Ciphertext &ct, const RelinKeys &rlk) {
        if (ct.components.size() <= 2) return;                                              // src/fhe.cu:227
        if (rlk.rlk_keys.empty()) return;
        if (ct.components.size() != 3) throw std::runtime_error("FHEContext::relinearize: 3-component ciphertext expected");
        import_relin_keys(rlk);
        check(fhe_ct_relinearize(params_.rns_ntt->handle(), rlk.imported, ct.components[0]->coeffs, ct.components[1]->coeffs,
                                 ct.components[2]->coeffs, 1), "FHEContext::relinearize");
        device_synchronize();
        delete ct.components[2];
        ct.components.resize(2);                                                            // src/fhe.cu:234
    }

    // hands the key polynomials to the engine once (transformed and packed inside the library), cached in the RelinKeys object
    void import_relin_keys(const RelinKeys &rlk) {
        if (rlk.imported && rlk.imported_for == params_.rns_ntt) return;
        fhe_relin_keys_destroy(rlk.imported); rlk.imported = nullptr;
        std::vector<const void *> kb, ka;
        for (const PublicKey *k : rlk.rlk_keys) { kb.push_back(k->pk0->coeffs); ka.push_back(k->pk1->coeffs); }
        check(fhe_relin_keys_create(params_.rns_ntt->handle(), &rlk.imported, rlk.decomp_bits, kb.data(), ka.data(), (uint32_t)kb.size()),
              "FHEContext: relinearisation key import");
        rlk.imported_for = params_.rns_ntt;
    }

    // number of key levels relinkey_gen must produce for this context: L limbs x ceil(bits(q_max) / decomp_bits) digits
    uint32_t relin_levels(uint32_t decomp_bits) const {
        uint32_t K = 0;
        check(fhe_relin_num_digits(params_.rns_ntt->handle(), decomp_bits, &K), "relin_levels");
        return K * (uint32_t)params_.rns_moduli.size();
    }

    // FHEContext::relinkey_gen (src/fhe.cu:76-111): rlk[level] = (-a*s + e + g*s^2, a).  `a` uniform and `e` small are drawn
    // on the host from `rng` (a std::mt19937_64-like generator; the reference's device samplers are toys and out of
    // scope); the polynomial products run on the engine.  `noise_scale` multiplies e (1 = the reference's BFV-style
    // keys; t for BGV-style keys whose noise must vanish mod t).
    template <class Rng>
    void relinkey_gen(RelinKeys &rlk, const SecretKey &sk, uint32_t decomp_bits, Rng &rng, uint64_t noise_scale = 1, int noise_bound = 3) {
        rlk.decomp_bits = decomp_bits;
        const uint32_t L = (uint32_t)params_.rns_moduli.size(), n = params_.n, levels = relin_levels(decomp_bits), K = levels / L;
        RNS_NTTEngine &E = *params_.rns_ntt;
        std::unique_ptr<Polynomial> s2(new_polynomial()), tmp(new_polynomial());
        E.multiply_rns(s2->coeffs, sk.sk->coeffs, sk.sk->coeffs);                              // s^2 (src/fhe.cu:80-81)
        std::vector<uint256_t> h_s2((size_t)L * n), h_a((size_t)L * n), h_e((size_t)L * n), h_g((size_t)L * n);
        device_synchronize();
        copy_to_host(h_s2.data(), s2->coeffs, h_s2.size());
        for (uint32_t j = 0; j < L; j++)
            for (uint32_t k = 0; k < K; k++) {
                PublicKey *key = new PublicKey{new_polynomial(), new_polynomial()};
                std::fill(h_g.begin(), h_g.end(), uint256_t());
                const uint64_t qj = params_.rns_moduli[j].limbs[0];
                uint64_t g = 1;
                for (uint32_t i = 0; i < k * decomp_bits; i++) g = (g << 1) % qj;           // 2^(k*w) mod q_j  (src/fhe.cu:98)
                std::vector<int> e(n);
                for (uint32_t x = 0; x < n; x++) e[x] = (int)(rng() % (2 * noise_bound + 1)) - noise_bound;
                for (uint32_t l = 0; l < L; l++) {
                    const uint64_t q = params_.rns_moduli[l].limbs[0];
                    for (uint32_t x = 0; x < n; x++) {
                        h_a[(size_t)l * n + x] = uint256_t(rng() % q);
                        const uint64_t mag = (uint64_t)(e[x] < 0 ? -e[x] : e[x]) * (noise_scale % q) % q;
                        h_e[(size_t)l * n + x] = uint256_t(e[x] < 0 ? (q - mag) % q : mag);
                    }
                }
                for (uint32_t x = 0; x < n; x++)
                    h_g[(size_t)j * n + x] = uint256_t((uint64_t)((unsigned __int128)h_s2[(size_t)j * n + x].limbs[0] * g % qj));
                copy_to_device(key->pk1->coeffs, h_a.data(), h_a.size());
                E.multiply_rns(tmp->coeffs, key->pk1->coeffs, sk.sk->coeffs);                 // a*s (src/fhe.cu:104)
                copy_to_device(key->pk0->coeffs, h_e.data(), h_e.size());
                E.sub_rns(key->pk0->coeffs, key->pk0->coeffs, tmp->coeffs);                   // e - a*s (:105)
                copy_to_device(tmp->coeffs, h_g.data(), h_g.size());
                E.add_rns(key->pk0->coeffs, key->pk0->coeffs, tmp->coeffs);                   // + g*s^2 (:106)
                device_synchronize();
                rlk.rlk_keys.push_back(key);
            }
    }

    // ---- scheme plumbing: BGV-flavoured so that tensor product + relinearisation decrypt correctly --------
    // (the reference's keygen / encrypt / decrypt, src/fhe.cu:54-185, mix BFV scaling with an even modulus and call
    // undefined kernels; here noise is a multiple of t and plaintexts sit in the low bits: c0 + c1*s = m + t*e).
    // Sampling is done on the host with a seeded std::mt19937_64 (the reference's device samplers are toys, out of scope).
    void seed(uint64_t s) { rng_.seed(s); }
    // true: keygen / encrypt draw their polynomials with the DEVICE samplers below (seeds taken from the context's generator)
    void device_sampling(bool on) { device_sampling_ = on; }

    // FHEContext::sample_error_polynomial / sample_uniform_polynomial / sample_ternary_polynomial (src/fhe.cu:237-257) on the
    // device: discrete Gaussian of parameter security.sigma, uniform residues, ternary with P(nonzero) = 0.5 -- the
    // distributions the reference names (its kernels are placeholders or undefined; the literal placeholders are
    // fhe_sample_uniform_lcg / fhe_sample_gaussian_placeholder).  Seeds come from the context generator where the reference calls rand().
    void sample_error_polynomial(Polynomial &p) {
        check(fhe_rns_sample_gaussian(params_.rns_ntt->handle(), p.coeffs, (double)params_.security.sigma, rng_(), 1), "sample_error_polynomial");
    }
    void sample_uniform_polynomial(Polynomial &p) { check(fhe_rns_sample_uniform(params_.rns_ntt->handle(), p.coeffs, rng_(), 1), "sample_uniform_polynomial"); }
    void sample_ternary_polynomial(Polynomial &p) { check(fhe_rns_sample_ternary(params_.rns_ntt->handle(), p.coeffs, 0.5, rng_(), 1), "sample_ternary_polynomial"); }

    void keygen(PublicKey &pk, SecretKey &sk) {                                            // src/fhe.cu:54-74
        sk.sk = new_polynomial(); pk.pk0 = new_polynomial(); pk.pk1 = new_polynomial();
        draw_ternary(*sk.sk);                                                               // ternary secret (:57)
        draw_uniform(*pk.pk1);                                                              // :64
        std::unique_ptr<Polynomial> as(new_polynomial());
        params_.rns_ntt->multiply_rns(as->coeffs, pk.pk1->coeffs, sk.sk->coeffs);           // :71
        draw_scaled_error(*pk.pk0);                                                         // t*e (:67-68)
        params_.rns_ntt->sub_rns(pk.pk0->coeffs, pk.pk0->coeffs, as->coeffs);               // pk0 = t*e - pk1*sk (:72)
        device_synchronize();
    }

    void relinkey_gen(RelinKeys &rlk, const SecretKey &sk, uint32_t decomp_bits = 16) {    // src/fhe.cu:76 signature
        relinkey_gen(rlk, sk, decomp_bits, rng_, params_.t);
    }

    // SIMD-slot encoding (the reference's encode scales by delta, src/fhe.cu:113-136, and its BatchEncoder is a passthrough,
    // :267-279; the expectations of its test -- element-wise products -- need real slots): m(zeta_i) = values[i].
    void encode(Plaintext &pt, const std::vector<uint64_t> &values) {
        const uint32_t n = params_.n; const uint64_t t = params_.t;
        if ((t - 1) % (2ull * n)) throw std::runtime_error("FHEContext::encode: t must be 1 (mod 2n) for slot encoding");
        std::vector<uint64_t> m(n, 0);
        for (size_t i = 0; i < values.size() && i < n; i++) m[i] = values[i] % t;
        slot_transform(m, true);
        std::vector<long long> sm(m.begin(), m.end());
        if (!pt.poly) pt.poly = new_polynomial();
        upload_signed(*pt.poly, sm);
        pt.is_ntt_form = false;
    }
    void decode(std::vector<uint64_t> &values, const Plaintext &pt) {
        const uint32_t n = params_.n, L = (uint32_t)params_.rns_moduli.size();
        std::vector<uint256_t> h((size_t)L * n);
        device_synchronize(); copy_to_host(h.data(), pt.poly->coeffs, h.size());
        values.assign(n, 0);
        for (uint32_t i = 0; i < n; i++) values[i] = h[i].limbs[0] % params_.t;              // plaintexts are stored reduced mod t in every limb
        slot_transform(values, false);
    }

    void encrypt(Ciphertext &ct, const Plaintext &pt, const PublicKey &pk) {               // src/fhe.cu:138-169
        ensure_components(ct, 2);
        RNS_NTTEngine &E = *params_.rns_ntt;
        std::unique_ptr<Polynomial> u(new_polynomial()), e(new_polynomial());
        draw_ternary(*u);
        E.multiply_rns(ct.components[0]->coeffs, pk.pk0->coeffs, u->coeffs);                 // pk0*u (:160)
        E.multiply_rns(ct.components[1]->coeffs, pk.pk1->coeffs, u->coeffs);                 // pk1*u (:165)
        draw_scaled_error(*e);
        E.add_rns(ct.components[0]->coeffs, ct.components[0]->coeffs, e->coeffs);            // + t*e1
        E.add_rns(ct.components[0]->coeffs, ct.components[0]->coeffs, pt.poly->coeffs);      // + m  (:161-162)
        device_synchronize();
        draw_scaled_error(*e);
        E.add_rns(ct.components[1]->coeffs, ct.components[1]->coeffs, e->coeffs);            // + t*e2 (:166)
        device_synchronize();
        ct.level = 0; ct.noise_budget = 0; ct.is_ntt_form = false;
    }

    // c0 + c1*s (+ c2*s^2), CRT to the centred integer through fhe_rns_from_rns, reduced mod t  (src/fhe.cu:171-185)
    void decrypt(Plaintext &pt, const Ciphertext &ct, const SecretKey &sk) {
        const uint32_t n = params_.n, L = (uint32_t)params_.rns_moduli.size();
        RNS_NTTEngine &E = *params_.rns_ntt;
        std::unique_ptr<Polynomial> acc(new_polynomial()), sp(new_polynomial()), tmp(new_polynomial());
        check(fhe_hip_memcpy_d2d(acc->coeffs, ct.components[0]->coeffs, acc->count() * sizeof(uint256_t)), "decrypt copy");
        check(fhe_hip_memcpy_d2d(sp->coeffs, sk.sk->coeffs, sp->count() * sizeof(uint256_t)), "decrypt copy");
        for (size_t k = 1; k < ct.components.size(); k++) {
            E.multiply_rns(tmp->coeffs, ct.components[k]->coeffs, sp->coeffs);
            E.add_rns(acc->coeffs, acc->coeffs, tmp->coeffs);
            if (k + 1 < ct.components.size()) E.multiply_rns(sp->coeffs, sp->coeffs, sk.sk->coeffs);   // s^(k+1), in place
        }
        uint256_t *d_int = device_alloc(n);
        E.from_rns(d_int, acc->coeffs);
        std::vector<uint256_t> v(n);
        device_synchronize(); copy_to_host(v.data(), d_int, n); device_free(d_int);
        // Q and Q/2 as 256-bit integers
        uint64_t Q[4] = {1, 0, 0, 0};
        for (uint32_t l = 0; l < L; l++) mul_small(Q, params_.rns_moduli[l].limbs[0]);
        uint64_t half[4]; for (int i = 0; i < 4; i++) half[i] = (Q[i] >> 1) | (i < 3 ? Q[i + 1] << 63 : 0);
        const uint64_t t = params_.t, q_mod_t = mod_small(Q, t);
        std::vector<long long> m(n);
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t r = mod_small(v[i].limbs, t);
            m[i] = (long long)(greater(v[i].limbs, half) ? (r + t - q_mod_t) % t : r);         // value - Q when above Q/2
        }
        if (!pt.poly) pt.poly = new_polynomial();
        upload_signed(*pt.poly, m);
        pt.is_ntt_form = false;
    }

    const SchemeParams &params() const { return params_; }

private:
    SchemeParams params_;
    std::mt19937_64 rng_{0x5EED0000ull};
    bool device_sampling_ = false;

    void plain_addsub(Ciphertext &result, const Ciphertext &ct, const Plaintext &pt, bool add_it) {
        ensure_components(result, ct.components.size());
        RNS_NTTEngine &E = *params_.rns_ntt;
        if (add_it) E.add_rns(result.components[0]->coeffs, ct.components[0]->coeffs, pt.poly->coeffs);
        else E.sub_rns(result.components[0]->coeffs, ct.components[0]->coeffs, pt.poly->coeffs);
        for (size_t i = 1; i < ct.components.size(); i++)
            if (result.components[i] != ct.components[i])
                check(fhe_hip_memcpy_d2d(result.components[i]->coeffs, ct.components[i]->coeffs, ct.components[i]->count() * sizeof(uint256_t)), "plain op copy");
        result.noise_budget = ct.noise_budget; result.level = ct.level;
    }

    // ---- where keygen / encrypt get their random polynomials: host generator (default) or the device samplers --------------
    void draw_ternary(Polynomial &p) { if (device_sampling_) sample_ternary_polynomial(p); else upload_signed(p, sample_small(1)); }
    void draw_uniform(Polynomial &p) { if (device_sampling_) sample_uniform_polynomial(p); else upload_uniform(p); }
    // t * e, e small: BGV-style noise (a multiple of the plaintext modulus)
    void draw_scaled_error(Polynomial &p) {
        if (!device_sampling_) { upload_signed(p, sample_small(3), params_.t); return; }
        sample_error_polynomial(p);
        const uint32_t n = params_.n, L = (uint32_t)params_.rns_moduli.size();
        for (uint32_t l = 0; l < L; l++) {   // limb l *= t: literal Montgomery product with the scalar t * 2^256 mod q_l (poly_mul_scalar_kernel)
            const uint256_t &q = params_.rns_moduli[l];
            if (q.limbs[1] | q.limbs[2] | q.limbs[3]) throw std::runtime_error("FHEContext: device sampling expects word-sized RNS primes");
            const uint64_t q0 = q.limbs[0], tr = (uint64_t)((unsigned __int128)(params_.t % q0) * pow_mod(2, 256, q0) % q0);
            uint64_t inv[4]; check(fhe_montgomery_inverse(q.limbs, inv), "montgomery inverse");
            const uint256_t scalar(tr);
            check(fhe_u256_mont_mul_scalar(p.coeffs + (size_t)l * n, p.coeffs + (size_t)l * n, scalar.limbs, q.limbs, inv[0], n, nullptr),   // legacy default stream: ordered with the engine's blocking stream
                  "FHEContext: scale error by t");
        }
    }

    // ---- host-side helpers of the plumbing above --------------------------------------------------------------------------
    std::vector<long long> sample_small(int bound) {
        std::vector<long long> v(params_.n);
        for (auto &x : v) x = (long long)(rng_() % (2 * bound + 1)) - bound;
        return v;
    }
    // signed integers (times `scale`) -> residues in every limb -> device
    void upload_signed(Polynomial &p, const std::vector<long long> &v, uint64_t scale = 1) {
        const uint32_t n = params_.n, L = (uint32_t)params_.rns_moduli.size();
        std::vector<uint256_t> h((size_t)L * n);
        for (uint32_t l = 0; l < L; l++) {
            const uint64_t q = params_.rns_moduli[l].limbs[0];
            for (uint32_t i = 0; i < n; i++) {
                const uint64_t mag = (uint64_t)((unsigned __int128)(uint64_t)(v[i] < 0 ? -v[i] : v[i]) % q * (scale % q) % q);
                h[(size_t)l * n + i] = uint256_t(v[i] < 0 ? (q - mag) % q : mag);
            }
        }
        copy_to_device(p.coeffs, h.data(), h.size());
    }
    void upload_uniform(Polynomial &p) {
        const uint32_t n = params_.n, L = (uint32_t)params_.rns_moduli.size();
        std::vector<uint256_t> h((size_t)L * n);
        for (uint32_t l = 0; l < L; l++) for (uint32_t i = 0; i < n; i++) h[(size_t)l * n + i] = uint256_t(rng_() % params_.rns_moduli[l].limbs[0]);
        copy_to_device(p.coeffs, h.data(), h.size());
    }
    static uint64_t pow_mod(uint64_t b, uint64_t e, uint64_t m) {
        unsigned __int128 acc = 1, bb = b % m;
        for (; e; e >>= 1) { if (e & 1) acc = acc * bb % m; bb = bb * bb % m; }
        return (uint64_t)acc;
    }
    // forward = false: coefficients -> values at the odd powers of a primitive 2n-th root mod t; forward = true: the inverse map.
    // O(n^2 / 64)-free: a plain O(n log n) negacyclic NTT over Z_t on the host.
    void slot_transform(std::vector<uint64_t> &a, bool inverse) const {
        const uint32_t n = params_.n; const uint64_t t = params_.t;
        uint64_t g = 2;                                           // find a generator-derived primitive 2n-th root of unity mod t
        uint64_t psi = 0;
        for (;; g++) { psi = pow_mod(g, (t - 1) / (2ull * n), t); if (pow_mod(psi, n, t) == t - 1) break; }
        const uint64_t ipsi = pow_mod(psi, 2ull * n - 1, t), ninv = pow_mod(n, t - 2, t);
        auto mulm = [t](uint64_t x, uint64_t y) { return (uint64_t)((unsigned __int128)x * y % t); };
        if (!inverse) { uint64_t pw = 1; for (uint32_t i = 0; i < n; i++) { a[i] = mulm(a[i], pw); pw = mulm(pw, psi); } }   // twist
        // cyclic NTT with omega = psi^2 (or its inverse), bit-reversal + iterative Cooley-Tukey
        const uint64_t omega = inverse ? mulm(ipsi, ipsi) : mulm(psi, psi);
        uint32_t lg = 0; while ((1u << lg) < n) lg++;
        for (uint32_t i = 0; i < n; i++) { uint32_t r = 0; for (uint32_t b = 0; b < lg; b++) r |= ((i >> b) & 1) << (lg - 1 - b); if (i < r) std::swap(a[i], a[r]); }
        for (uint32_t len = 2; len <= n; len <<= 1) {
            const uint64_t wl = pow_mod(omega, n / len, t);
            for (uint32_t i = 0; i < n; i += len) {
                uint64_t w = 1;
                for (uint32_t j = 0; j < len / 2; j++) {
                    const uint64_t u = a[i + j], v = mulm(a[i + j + len / 2], w);
                    a[i + j] = (u + v) % t; a[i + j + len / 2] = (u + t - v) % t; w = mulm(w, wl);
                }
            }
        }
        if (inverse) { uint64_t pw = 1; for (uint32_t i = 0; i < n; i++) { a[i] = mulm(mulm(a[i], ninv), pw); pw = mulm(pw, ipsi); } }   // untwist, scale
    }
    static void mul_small(uint64_t x[4], uint64_t m) {
        unsigned __int128 c = 0;
        for (int i = 0; i < 4; i++) { c += (unsigned __int128)x[i] * m; x[i] = (uint64_t)c; c >>= 64; }
    }
    static uint64_t mod_small(const uint64_t x[4], uint64_t m) {
        unsigned __int128 r = 0;
        for (int i = 3; i >= 0; i--) r = ((r << 64) | x[i]) % m;
        return (uint64_t)r;
    }
    static bool greater(const uint64_t a[4], const uint64_t b[4]) {
        for (int i = 3; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i];
        return false;
    }

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
