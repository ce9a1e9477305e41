// test_fhe_mirror.cpp -- the reference's test scenarios (tests/test_fhe.cu:24-318) on the hot path, run
// through the C++ mirror classes (include/fhe/*.hpp) and ASSERTED (the reference only prints).
// Expected values are computed here on the host with plain 64/128-bit arithmetic (schoolbook).
//   ./test_fhe_mirror             full run (needs the MI355X)
//   ./test_fhe_mirror --host-only parameter maths only (no device)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>
#include <vector>

#include "fhe/fhe.hpp"
#include "fhe/rns.hpp"

using namespace fhe;
typedef unsigned __int128 u128;

#define REQUIRE(cond)                                                                       \
    do {                                                                                    \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
    } while (0)

static std::vector<uint64_t> schoolbook_negacyclic(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b, uint64_t q) {
    size_t n = a.size();
    std::vector<uint64_t> r(n, 0);
    for (size_t i = 0; i < n; i++) {
        if (!a[i]) continue;
        for (size_t j = 0; j < n; j++) {
            uint64_t p = (uint64_t)((u128)a[i] * b[j] % q);
            size_t k = i + j;
            if (k < n) r[k] = (r[k] + p) % q;
            else r[k - n] = (r[k - n] + q - p) % q;
        }
    }
    return r;
}
static std::vector<uint256_t> widen(const std::vector<uint64_t> &v) { return std::vector<uint256_t>(v.begin(), v.end()); }

static void test_host_parameters() {
    std::cout << "Testing host parameter maths..." << std::endl;
    MontgomeryParams p = compute_montgomery_params(uint256_t(12289));
    REQUIRE(p.inv.limbs[0] == 0x2faf01aff7002fffull && p.inv.limbs[1] == 0);
    REQUIRE(p.r_squared.limbs[0] == 10889);                                          // R^2 mod 12289
    REQUIRE(compute_montgomery_inverse(uint256_t(1ull << 60)).limbs[0] == 0xffffffffffffffc0ull);   // even-modulus garbage, literal
    uint256_t psi = find_primitive_root(1024, uint256_t(12289));
    uint64_t x = 1; for (int i = 0; i < 1024; i++) x = x * psi.limbs[0] % 12289;
    REQUIRE(x == 12288);                                                             // psi^n = -1
    bool threw = false;
    try { find_primitive_root(1024, uint256_t(12291)); } catch (const std::runtime_error &) { threw = true; }
    REQUIRE(threw);
    std::cout << "  ok" << std::endl;
}

// tests/test_fhe.cu:24-63
static void test_bigint_arithmetic() {
    std::cout << "Testing BigInt Arithmetic..." << std::endl;
    uint256_t *d_a = device_alloc(1), *d_b = device_alloc(1), *d_r = device_alloc(1);
    uint256_t h_a(12345, 0, 0, 0), h_b(67890, 0, 0, 0), h_mod(100000, 0, 0, 0), h_r;
    copy_to_device(d_a, &h_a, 1); copy_to_device(d_b, &h_b, 1);
    batch_mod_add(d_r, d_a, d_b, h_mod, 1); device_synchronize(); copy_to_host(&h_r, d_r, 1);
    std::cout << "  Addition: 12345 + 67890 = " << h_r.limbs[0] << " (mod 100000)" << std::endl;
    REQUIRE(h_r == uint256_t(80235));
    batch_mod_sub(d_r, d_a, d_b, h_mod, 1); device_synchronize(); copy_to_host(&h_r, d_r, 1);
    std::cout << "  Subtraction: 12345 - 67890 = " << h_r.limbs[0] << " (mod 100000)" << std::endl;
    REQUIRE(h_r == uint256_t(44455));
    device_free(d_a); device_free(d_b); device_free(d_r);
}

// tests/test_fhe.cu:65-124
static void test_ntt_transform() {
    std::cout << "Testing NTT Transform..." << std::endl;
    const uint32_t N = 1024;
    uint256_t modulus(12289, 0, 0, 0);
    NTTEngine ntt(N, modulus);
    std::vector<uint256_t> h_data(N), h_fwd(N), h_back(N);
    for (uint32_t i = 0; i < N; i++) h_data[i] = uint256_t(i + 1);
    uint256_t *d = device_alloc(N);
    copy_to_device(d, h_data.data(), N);
    ntt.forward(d); device_synchronize(); copy_to_host(h_fwd.data(), d, N);
    ntt.inverse(d); device_synchronize(); copy_to_host(h_back.data(), d, N);
    bool changed = false;
    for (uint32_t i = 0; i < N; i++) { REQUIRE(h_back[i] == h_data[i]); REQUIRE(h_fwd[i].limbs[0] < 12289); changed |= h_fwd[i] != h_data[i]; }
    REQUIRE(changed);
    // X[0] = sum_j x[j] psi^j  (bitrev(0) = 0)
    uint64_t psi = find_primitive_root(N, modulus).limbs[0], acc = 0, pw = 1;
    for (uint32_t j = 0; j < N; j++) { acc = (acc + (j + 1) * pw) % 12289; pw = pw * psi % 12289; }
    REQUIRE(h_fwd[0].limbs[0] == acc);
    device_free(d);
    std::cout << "  round trip ok, X[0] matches the definition" << std::endl;
}

// tests/test_fhe.cu:126-167 (the reference never reads the product back)
static void test_polynomial_multiplication() {
    std::cout << "Testing Polynomial Multiplication..." << std::endl;
    const uint32_t N = 2048;
    uint256_t modulus(40961, 0, 0, 0);
    NTTEngine ntt(N, modulus);
    PolynomialOps ops(N, modulus, &ntt);
    std::vector<uint64_t> a(N), b(N);
    srand(1);
    for (uint32_t i = 0; i < N; i++) { a[i] = rand() % 100; b[i] = rand() % 100; }
    Polynomial pa(N, modulus), pb(N, modulus), pr(N, modulus), ps(N, modulus);
    auto wa = widen(a), wb = widen(b);
    copy_to_device(pa.coeffs, wa.data(), N); copy_to_device(pb.coeffs, wb.data(), N);
    ops.mul_ntt(pr, pa, pb);
    ops.add(ps, pa, pb);
    device_synchronize();
    std::vector<uint256_t> got(N), sum(N), a_after(N);
    copy_to_host(got.data(), pr.coeffs, N); copy_to_host(sum.data(), ps.coeffs, N); copy_to_host(a_after.data(), pa.coeffs, N);
    auto want = schoolbook_negacyclic(a, b, 40961);
    for (uint32_t i = 0; i < N; i++) {
        REQUIRE(got[i] == uint256_t(want[i]));
        REQUIRE(sum[i] == uint256_t((a[i] + b[i]) % 40961));
        REQUIRE(a_after[i] == uint256_t(a[i]));                                    // operands preserved (src/ntt.cu:50-58)
    }
    ops.mul(ps, pa, pb); device_synchronize(); copy_to_host(sum.data(), ps.coeffs, N);        // mul == mul_ntt == mul_negacyclic
    for (uint32_t i = 0; i < N; i++) REQUIRE(sum[i] == got[i]);
    ops.mod_switch(ps, pa, uint256_t(257)); device_synchronize(); copy_to_host(sum.data(), ps.coeffs, N);
    for (uint32_t i = 0; i < N; i++) REQUIRE(sum[i] == uint256_t(((a[i] * 257 + 40961 / 2) / 40961) % 257));   // round(a * t / q) mod t
    std::cout << "  product matches schoolbook, operands preserved; mod_switch rounds" << std::endl;
}

// tests/test_fhe.cu:169-273, multiply path only: the tensor product of two 2-component ciphertexts
static void test_fhe_multiply() {
    std::cout << "Testing FHEContext::multiply (tensor product)..." << std::endl;
    SecurityParams sp{128, 4096, 120, 3.2f, 64};
    FHEContext ctx(sp);
    const SchemeParams &P = ctx.params();
    const uint32_t N = P.n, L = (uint32_t)P.rns_moduli.size();
    REQUIRE(L == 4);
    for (auto &q : P.rns_moduli) REQUIRE(q.limbs[0] >> 29 == 1 && q.limbs[0] % (2 * N) == 1);
    Ciphertext A, B, C, S;
    std::vector<std::vector<uint64_t>> host[4];     // a0, a1, b0, b1 : [L][N]
    srand(7);
    for (int c = 0; c < 4; c++) {
        Polynomial *p = ctx.new_polynomial();
        host[c].assign(L, std::vector<uint64_t>(N));
        std::vector<uint256_t> flat((size_t)L * N);
        for (uint32_t l = 0; l < L; l++)
            for (uint32_t i = 0; i < N; i++) {
                uint64_t v = (((uint64_t)rand() << 20) ^ (uint64_t)rand()) % P.rns_moduli[l].limbs[0];
                host[c][l][i] = v; flat[(size_t)l * N + i] = uint256_t(v);
            }
        copy_to_device(p->coeffs, flat.data(), flat.size());
        (c < 2 ? A : B).components.push_back(p);
    }
    RelinKeys rlk;
    ctx.multiply(C, A, B, rlk);
    ctx.add(S, A, B);
    device_synchronize();
    REQUIRE(C.components.size() == 3 && S.components.size() == 2);
    std::vector<uint256_t> c[3];
    for (int k = 0; k < 3; k++) { c[k].resize((size_t)L * N); copy_to_host(c[k].data(), C.components[k]->coeffs, c[k].size()); }
    for (uint32_t l : {0u, L - 1}) {
        uint64_t q = P.rns_moduli[l].limbs[0];
        auto c0 = schoolbook_negacyclic(host[0][l], host[2][l], q);
        auto t1 = schoolbook_negacyclic(host[0][l], host[3][l], q);
        auto t2 = schoolbook_negacyclic(host[1][l], host[2][l], q);
        auto c2 = schoolbook_negacyclic(host[1][l], host[3][l], q);
        for (uint32_t i = 0; i < N; i++) {
            REQUIRE(c[0][(size_t)l * N + i] == uint256_t(c0[i]));
            REQUIRE(c[1][(size_t)l * N + i] == uint256_t((t1[i] + t2[i]) % q));
            REQUIRE(c[2][(size_t)l * N + i] == uint256_t(c2[i]));
        }
    }
    std::vector<uint256_t> s0((size_t)L * N);
    copy_to_host(s0.data(), S.components[0]->coeffs, s0.size());
    for (uint32_t l = 0; l < L; l++)
        for (uint32_t i = 0; i < N; i += 97) REQUIRE(s0[(size_t)l * N + i] == uint256_t((host[0][l][i] + host[2][l][i]) % P.rns_moduli[l].limbs[0]));
    std::cout << "  c0, c1, c2 match schoolbook on limbs 0 and " << L - 1 << std::endl;
}


// FHEContext::multiply + relinearize + the decryption identity: with BGV-style ciphertexts (c0 + c1*s = m + t*e) the
// relinearised product must decrypt to m1 (*) m2 mod t.  Key generation through FHEContext::relinkey_gen, decryption
// here on the host (per-limb phase, centred CRT for two limbs).
static void test_fhe_multiply_relinearize() {
    std::cout << "Testing FHEContext::multiply + relinearize (decrypts to the plaintext product)..." << std::endl;
    const uint32_t N = 2048; const uint64_t t = 257;
    SecurityParams sp{128, N, 60, 3.2f, 64};                       // log_q = 60 -> 2 limbs of 30 bits
    FHEContext ctx(sp);
    const SchemeParams &P = ctx.params();
    const uint32_t L = (uint32_t)P.rns_moduli.size();
    REQUIRE(L == 2);
    const uint64_t q0 = P.rns_moduli[0].limbs[0], q1 = P.rns_moduli[1].limbs[0];
    std::mt19937_64 rng(42);
    // secret key, plaintexts, errors
    std::vector<int> s(N); for (auto &v : s) v = (int)(rng() % 3) - 1;
    std::vector<uint64_t> m1(N), m2(N); for (uint32_t i = 0; i < N; i++) { m1[i] = rng() % t; m2[i] = rng() % t; }
    auto to_rns = [&](const std::vector<long long> &v) {
        std::vector<uint256_t> out((size_t)L * N);
        for (uint32_t l = 0; l < L; l++) { const long long q = (long long)P.rns_moduli[l].limbs[0]; for (uint32_t i = 0; i < N; i++) out[(size_t)l * N + i] = uint256_t((uint64_t)(((v[i] % q) + q) % q)); }
        return out;
    };
    SecretKey sk{ctx.new_polynomial()};
    { std::vector<long long> sv(s.begin(), s.end()); auto r = to_rns(sv); copy_to_device(sk.sk->coeffs, r.data(), r.size()); }
    RNS_NTTEngine &E = *P.rns_ntt;
    auto encrypt = [&](const std::vector<uint64_t> &m, Ciphertext &ct) {       // c1 = a uniform, c0 = m + t*e - a*s
        Polynomial *c0 = ctx.new_polynomial(), *c1 = ctx.new_polynomial();
        std::unique_ptr<Polynomial> as(ctx.new_polynomial());
        std::vector<uint256_t> a((size_t)L * N);
        for (uint32_t l = 0; l < L; l++) for (uint32_t i = 0; i < N; i++) a[(size_t)l * N + i] = uint256_t(rng() % P.rns_moduli[l].limbs[0]);
        std::vector<long long> me(N); for (uint32_t i = 0; i < N; i++) me[i] = (long long)m[i] + (long long)t * ((long long)(rng() % 7) - 3);
        auto r = to_rns(me);
        copy_to_device(c1->coeffs, a.data(), a.size()); copy_to_device(c0->coeffs, r.data(), r.size());
        E.multiply_rns(as->coeffs, c1->coeffs, sk.sk->coeffs);
        E.sub_rns(c0->coeffs, c0->coeffs, as->coeffs);
        device_synchronize();
        ct.components = {c0, c1};
    };
    Ciphertext A, B, C;
    encrypt(m1, A); encrypt(m2, B);
    RelinKeys rlk;
    ctx.relinkey_gen(rlk, sk, 16, rng, /*noise_scale=*/t);
    REQUIRE(rlk.rlk_keys.size() == ctx.relin_levels(16) && rlk.rlk_keys.size() == 4);
    ctx.multiply(C, A, B, rlk);
    REQUIRE(C.components.size() == 2);                                       // relinearised (src/fhe.cu:234)
    // decrypt: phase = c0 + c1*s per limb, CRT to the centred integer, mod t
    std::unique_ptr<Polynomial> ph(ctx.new_polynomial());
    E.multiply_rns(ph->coeffs, C.components[1]->coeffs, sk.sk->coeffs);
    E.add_rns(ph->coeffs, ph->coeffs, C.components[0]->coeffs);
    device_synchronize();
    std::vector<uint256_t> h((size_t)L * N); copy_to_host(h.data(), ph->coeffs, h.size());
    const unsigned __int128 Q = (unsigned __int128)q0 * q1;
    uint64_t q0inv_mod_q1 = 1; { uint64_t b = q0 % q1, e = q1 - 2; unsigned __int128 acc = 1, bb = b; while (e) { if (e & 1) acc = acc * bb % q1; bb = bb * bb % q1; e >>= 1; } q0inv_mod_q1 = (uint64_t)acc; }
    auto want = schoolbook_negacyclic(m1, m2, t);
    for (uint32_t i = 0; i < N; i++) {
        const uint64_t r0 = h[i].limbs[0], r1 = h[(size_t)N + i].limbs[0];
        // x = r0 + q0 * ((r1 - r0) * q0^-1 mod q1)
        const uint64_t diff = (r1 + q1 - r0 % q1) % q1;
        unsigned __int128 x = (unsigned __int128)r0 + (unsigned __int128)q0 * (uint64_t)((unsigned __int128)diff * q0inv_mod_q1 % q1);
        long long mt;
        if (x > Q / 2) { unsigned __int128 neg = Q - x; mt = (long long)((t - (uint64_t)(neg % t)) % t); } else mt = (long long)(uint64_t)(x % t);
        REQUIRE((uint64_t)mt == want[i]);
    }
    delete sk.sk;
    std::cout << "  Dec(relin(Enc(m1) x Enc(m2))) == m1 (*) m2 mod " << t << " on all " << N << " coefficients" << std::endl;
}


// tests/test_fhe.cu:169-273 through the mirror's FHEContext alone: keygen, relinkey_gen(16), encode {5,10,15,20} and
// {3,6,9,12}, encrypt, add / multiply (+ relinearize), decrypt, decode -- and the expectations the reference only prints
// (8 16 24 32 and 15 60 135 240) are asserted.
static void test_fhe_operations() {
    std::cout << "Testing FHE Operations (reference scenario through the mirror)..." << std::endl;
    SecurityParams sp{128, 4096, 120, 3.2f, 64};
    FHEContext ctx(sp);
    ctx.seed(2026);
    PublicKey pk; SecretKey sk; RelinKeys rlk;
    ctx.keygen(pk, sk);
    ctx.relinkey_gen(rlk, sk, 16);
    REQUIRE(rlk.rlk_keys.size() == 8);                                     // 4 limbs x 2 digits of 16 bits
    Plaintext pa, pb, pr;
    ctx.encode(pa, {5, 10, 15, 20});
    ctx.encode(pb, {3, 6, 9, 12});
    std::vector<uint64_t> back; ctx.decode(back, pa);
    REQUIRE(back[0] == 5 && back[1] == 10 && back[2] == 15 && back[3] == 20 && back[4] == 0);
    Ciphertext ca, cb, csum, cprod;
    ctx.encrypt(ca, pa, pk); ctx.encrypt(cb, pb, pk);
    std::vector<uint64_t> out;
    ctx.decrypt(pr, ca, sk); ctx.decode(out, pr);
    REQUIRE(out[0] == 5 && out[1] == 10 && out[2] == 15 && out[3] == 20);
    ctx.add(csum, ca, cb);
    ctx.decrypt(pr, csum, sk); ctx.decode(out, pr);
    std::cout << "  Addition result: " << out[0] << " " << out[1] << " " << out[2] << " " << out[3] << " (expected: 8 16 24 32)" << std::endl;
    REQUIRE(out[0] == 8 && out[1] == 16 && out[2] == 24 && out[3] == 32);
    ctx.multiply(cprod, ca, cb, rlk);
    REQUIRE(cprod.components.size() == 2);
    ctx.decrypt(pr, cprod, sk); ctx.decode(out, pr);
    std::cout << "  Multiplication result: " << out[0] << " " << out[1] << " " << out[2] << " " << out[3] << " (expected: 15 60 135 240)" << std::endl;
    REQUIRE(out[0] == 15 && out[1] == 60 && out[2] == 135 && out[3] == 240);
    for (size_t i = 4; i < out.size(); i++) REQUIRE(out[i] == 0);
    // sub / add_plain / sub_plain / multiply_plain (declared only in the reference, include/fhe.cuh:98-104)
    {
        Ciphertext cd, cp;
        ctx.sub(cd, cb, ca);                                                // slots wrap mod t: 3 - 5 = t - 2
        ctx.decrypt(pr, cd, sk); ctx.decode(out, pr);
        const uint64_t t = ctx.params().t;
        REQUIRE(out[0] == t - 2 && out[1] == t - 4 && out[2] == t - 6 && out[3] == t - 8 && out[4] == 0);
        ctx.add_plain(cp, ca, pb);
        ctx.decrypt(pr, cp, sk); ctx.decode(out, pr);
        REQUIRE(out[0] == 8 && out[1] == 16 && out[2] == 24 && out[3] == 32);
        ctx.sub_plain(cp, ca, pb);
        ctx.decrypt(pr, cp, sk); ctx.decode(out, pr);
        REQUIRE(out[0] == 2 && out[1] == 4 && out[2] == 6 && out[3] == 8);
        ctx.multiply_plain(cp, ca, pb);
        ctx.decrypt(pr, cp, sk); ctx.decode(out, pr);
        REQUIRE(out[0] == 15 && out[1] == 60 && out[2] == 135 && out[3] == 240 && out[4] == 0);
    }
    // depth 2: (a*b)*a = 75 600 2025 4800
    Ciphertext c3;
    ctx.multiply(c3, cprod, ca, rlk);
    ctx.decrypt(pr, c3, sk); ctx.decode(out, pr);
    REQUIRE(out[0] == 75 && out[1] == 600 && out[2] == 2025 && out[3] == 4800);
    delete pk.pk0; delete pk.pk1; delete sk.sk; delete pa.poly; delete pb.poly; delete pr.poly;
}

// The same scenario with every random polynomial drawn by the DEVICE samplers (row N4): ternary secret and encryption
// randomness, discrete-Gaussian noise of parameter sigma = 3.2 scaled by t, uniform public-key mask (relinkey_gen keeps its host generator).
static void test_fhe_operations_device_sampling() {
    std::cout << "Testing FHE Operations with device samplers..." << std::endl;
    SecurityParams sp{128, 4096, 120, 3.2f, 64};
    FHEContext ctx(sp);
    ctx.seed(77); ctx.device_sampling(true);
    // the samplers themselves: ternary values are 0, 1 or q - 1 in every limb, about half of them non-zero
    std::unique_ptr<Polynomial> s(ctx.new_polynomial());
    ctx.sample_ternary_polynomial(*s); device_synchronize();
    const uint32_t N = 4096; const size_t L = ctx.params().rns_moduli.size();
    std::vector<uint256_t> h(L * N); copy_to_host(h.data(), s->coeffs, h.size());
    size_t nz = 0;
    for (uint32_t i = 0; i < N; i++) {
        const uint64_t v0 = h[i].limbs[0], q0 = ctx.params().rns_moduli[0].limbs[0];
        REQUIRE(v0 == 0 || v0 == 1 || v0 == q0 - 1);
        for (size_t l = 1; l < L; l++) {
            const uint64_t ql = ctx.params().rns_moduli[l].limbs[0], vl = h[l * N + i].limbs[0];
            REQUIRE((v0 == 0 && vl == 0) || (v0 == 1 && vl == 1) || (v0 == q0 - 1 && vl == ql - 1));
        }
        nz += v0 != 0;
    }
    REQUIRE(nz > N * 4 / 10 && nz < N * 6 / 10);
    ctx.sample_error_polynomial(*s); device_synchronize(); copy_to_host(h.data(), s->coeffs, h.size());
    double var = 0;
    for (uint32_t i = 0; i < N; i++) {
        const uint64_t v0 = h[i].limbs[0], q0 = ctx.params().rns_moduli[0].limbs[0];
        const double c = v0 > q0 / 2 ? -(double)(q0 - v0) : (double)v0;
        REQUIRE(c >= -39 && c <= 39);                                       // cut at 12 sigma
        var += c * c;
    }
    var /= N;
    REQUIRE(var > 3.2 * 3.2 * 0.8 && var < 3.2 * 3.2 * 1.2);
    PublicKey pk; SecretKey sk; RelinKeys rlk;
    ctx.keygen(pk, sk);
    ctx.relinkey_gen(rlk, sk, 16);
    Plaintext pa, pb, pr;
    ctx.encode(pa, {5, 10, 15, 20});
    ctx.encode(pb, {3, 6, 9, 12});
    Ciphertext ca, cb, csum, cprod;
    ctx.encrypt(ca, pa, pk); ctx.encrypt(cb, pb, pk);
    std::vector<uint64_t> out;
    ctx.add(csum, ca, cb);
    ctx.decrypt(pr, csum, sk); ctx.decode(out, pr);
    REQUIRE(out[0] == 8 && out[1] == 16 && out[2] == 24 && out[3] == 32);
    ctx.multiply(cprod, ca, cb, rlk);
    ctx.decrypt(pr, cprod, sk); ctx.decode(out, pr);
    std::cout << "  Multiplication result: " << out[0] << " " << out[1] << " " << out[2] << " " << out[3] << " (expected: 15 60 135 240)" << std::endl;
    REQUIRE(out[0] == 15 && out[1] == 60 && out[2] == 135 && out[3] == 240);
    for (size_t i = 4; i < out.size(); i++) REQUIRE(out[i] == 0);
    delete pk.pk0; delete pk.pk1; delete sk.sk; delete pa.poly; delete pb.poly; delete pr.poly;
}

// fhe::RNSContext (include/rns.cuh:27-66): interleaved [count][num_primes] buffers, literal add / mul kernels, real conversions.
static void test_rns_context() {
    std::cout << "Testing RNSContext..." << std::endl;
    const std::vector<uint64_t> ps = {12289, 40961, 1073479681ull, 1152921504606584833ull};      // any distinct odd primes: no NTT condition
    std::vector<uint256_t> primes(ps.begin(), ps.end());
    RNSContext rns(primes);
    REQUIRE(rns.num_primes() == 4 && rns.base().primes[1] == uint256_t(40961));
    const uint32_t count = 1000, L = 4;
    std::mt19937_64 rng(5);
    unsigned __int128 Q = 1; for (uint64_t p : ps) Q *= p;                                       // ~2^119
    std::vector<unsigned __int128> va(count), vb(count);
    std::vector<uint256_t> ha(count), hb(count);
    for (uint32_t i = 0; i < count; i++) {
        va[i] = (((unsigned __int128)rng() << 64) | rng()) % Q; vb[i] = (((unsigned __int128)rng() << 64) | rng()) % Q;
        ha[i] = uint256_t((uint64_t)va[i], (uint64_t)(va[i] >> 64), 0, 0); hb[i] = uint256_t((uint64_t)vb[i], (uint64_t)(vb[i] >> 64), 0, 0);
    }
    uint256_t *dA = device_alloc(count), *dB = device_alloc(count), *dRA = device_alloc((size_t)count * L), *dRB = device_alloc((size_t)count * L),
              *dR = device_alloc((size_t)count * L), *dV = device_alloc(count);
    copy_to_device(dA, ha.data(), count); copy_to_device(dB, hb.data(), count);
    rns.to_rns(dRA, dA, count); rns.to_rns(dRB, dB, count);
    std::vector<uint256_t> r((size_t)count * L);
    device_synchronize(); copy_to_host(r.data(), dRA, r.size());
    for (uint32_t i = 0; i < count; i++) for (uint32_t l = 0; l < L; l++) REQUIRE(r[(size_t)i * L + l] == uint256_t((uint64_t)(va[i] % ps[l])));   // interleaved layout
    rns.from_rns(dV, dRA, count);
    std::vector<uint256_t> back(count); device_synchronize(); copy_to_host(back.data(), dV, count);
    for (uint32_t i = 0; i < count; i++) REQUIRE(back[i] == ha[i]);
    // add / sub / plain product commute with the CRT; mul_rns is the literal Montgomery product (carries 2^-256 per limb)
    rns.add_rns(dR, dRA, dRB, count); rns.from_rns(dV, dR, count); device_synchronize(); copy_to_host(back.data(), dV, count);
    for (uint32_t i = 0; i < count; i++) { unsigned __int128 w = (va[i] + vb[i]) % Q; REQUIRE(back[i] == uint256_t((uint64_t)w, (uint64_t)(w >> 64), 0, 0)); }
    rns.sub_rns(dR, dRA, dRB, count); rns.from_rns(dV, dR, count); device_synchronize(); copy_to_host(back.data(), dV, count);
    for (uint32_t i = 0; i < count; i++) { unsigned __int128 w = (va[i] + Q - vb[i]) % Q; REQUIRE(back[i] == uint256_t((uint64_t)w, (uint64_t)(w >> 64), 0, 0)); }
    rns.mul_rns_plain(dR, dRA, dRB, count); device_synchronize(); copy_to_host(r.data(), dR, r.size());
    for (uint32_t i = 0; i < count; i++) for (uint32_t l = 0; l < L; l++)
        REQUIRE(r[(size_t)i * L + l] == uint256_t((uint64_t)((unsigned __int128)(uint64_t)(va[i] % ps[l]) * (uint64_t)(vb[i] % ps[l]) % ps[l])));
    rns.mul_rns(dR, dRA, dRB, count); device_synchronize(); copy_to_host(r.data(), dR, r.size());
    auto pow_mod = [](unsigned __int128 b, unsigned e, uint64_t m) { unsigned __int128 a = 1; b %= m; for (; e; e >>= 1) { if (e & 1) a = a * b % m; b = b * b % m; } return (uint64_t)a; };
    for (uint32_t l = 0; l < L; l++) {
        const uint64_t q = ps[l];
        unsigned __int128 base = pow_mod(2, 256, q), acc = 1; uint64_t e = q - 2;      // acc = 2^-256 mod q = (2^256)^(q-2)
        for (; e; e >>= 1) { if (e & 1) acc = acc * base % q; base = base * base % q; }
        for (uint32_t i = 0; i < count; i += 97) {
            const unsigned __int128 want = (unsigned __int128)(uint64_t)(va[i] % q) * (uint64_t)(vb[i] % q) % q * acc % q;
            REQUIRE(r[(size_t)i * L + l] == uint256_t((uint64_t)want));
        }
    }
    // mod_switch_rns: two levels down = round(round(x / q3) / q2) in the base {q0, q1}
    uint256_t *dS = device_alloc((size_t)count * 2);
    rns.mod_switch_rns(dS, dRA, 0, 2, count);
    std::vector<uint256_t> s2((size_t)count * 2); device_synchronize(); copy_to_host(s2.data(), dS, s2.size());
    for (uint32_t i = 0; i < count; i += 13) {
        unsigned __int128 x = va[i];
        // centred rounding as fhe_rns_rescale_drop_last defines it: (x - [x]_q) / q with [x]_q the centred residue
        for (int drop = 3; drop >= 2; drop--) {
            const uint64_t q = ps[drop]; uint64_t c = (uint64_t)(x % q);
            if (c > q / 2) x = (x + (q - c)) / q; else x = (x - c) / q;
        }
        REQUIRE(s2[(size_t)i * 2] == uint256_t((uint64_t)(x % ps[0])) && s2[(size_t)i * 2 + 1] == uint256_t((uint64_t)(x % ps[1])));
    }
    // base_extend into a disjoint base: x + alpha * Q with 0 <= alpha < L
    RNSContext target(std::vector<uint256_t>{uint256_t(65537), uint256_t(786433)});
    uint256_t *dE = device_alloc((size_t)count * 2);
    rns.base_extend(dE, dRA, target, count);
    std::vector<uint256_t> ext((size_t)count * 2); device_synchronize(); copy_to_host(ext.data(), dE, ext.size());
    for (uint32_t i = 0; i < count; i += 7) {
        bool ok = false;
        for (unsigned alpha = 0; alpha < L && !ok; alpha++)
            ok = ext[(size_t)i * 2] == uint256_t((uint64_t)((va[i] % 65537 + (unsigned __int128)alpha * (Q % 65537)) % 65537)) &&
                 ext[(size_t)i * 2 + 1] == uint256_t((uint64_t)((va[i] % 786433 + (unsigned __int128)alpha * (Q % 786433)) % 786433));
        REQUIRE(ok);
    }
    for (uint256_t *p : {dA, dB, dRA, dRB, dR, dV, dS, dE}) device_free(p);
    std::cout << "  to_rns / from_rns / add / sub / mul (literal + plain) / mod_switch_rns / base_extend ok" << std::endl;
}

// tests/test_fhe.cu:275-318 shape (N = 8192), timing the multiply path instead of encrypt
static void benchmark_multiply() {
    std::cout << "Benchmark: ciphertext tensor product, N = 8192, log_q = 120" << std::endl;
    SecurityParams sp{128, 8192, 120, 3.2f, 64};
    FHEContext ctx(sp);
    Ciphertext A, B, C;
    for (int c = 0; c < 4; c++) (c < 2 ? A : B).components.push_back(ctx.new_polynomial());
    RelinKeys rlk;
    ctx.multiply(C, A, B, rlk); device_synchronize();
    auto t0 = std::chrono::high_resolution_clock::now();
    const int iters = 100;
    for (int i = 0; i < iters; i++) ctx.multiply(C, A, B, rlk);
    device_synchronize();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    std::cout << "  " << ms / iters << " ms per multiply (batch 1, launch-bound), " << iters * 1000.0 / ms << " ops/s" << std::endl;
}

int main(int argc, char **argv) {
    test_host_parameters();
    if (argc > 1 && !std::strcmp(argv[1], "--host-only")) { std::cout << "host-only: PASSED" << std::endl; return 0; }
    int count = 0;
    check(fhe_hip_device_count(&count), "device count");
    REQUIRE(count > 0);
    char name[256]; check(fhe_hip_device_name(name, sizeof name), "device name");
    std::cout << "Device: " << name << std::endl;
    test_bigint_arithmetic();
    test_ntt_transform();
    test_polynomial_multiplication();
    test_fhe_multiply();
    test_fhe_multiply_relinearize();
    test_fhe_operations();
    test_fhe_operations_device_sampling();
    test_rns_context();
    benchmark_multiply();
    std::cout << "ALL PASSED" << std::endl;
    return 0;
}
