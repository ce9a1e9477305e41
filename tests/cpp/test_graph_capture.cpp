// test_graph_capture.cpp -- the engine's launches are capturable into a hipGraph (no allocation, synchronisation or
// host-blocking call on the word-sized hot path): tensor product + relinearisation captured once on a caller-owned
// stream, replayed, and compared with the directly launched result.  Build: hipcc (needs the HIP runtime API).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fhe_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
#define FHE_OK_(x) do { int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "fhe error %d (%s) at line %d\n", rc_, fhe_hip_last_error(), __LINE__); std::exit(1); } } while (0)

// usage: test_graph_capture [prime_bits n batch]   (default 30 8192 8; the test also runs 30 8192 1 -- the few-ciphertext forms,
// whose block images live in a third workspace -- and 40 16384 2 -- the 8-byte residues, whose blind-rotation loop keeps six compact polynomials)
int main(int argc, char **argv) {
    const uint32_t bits = argc > 3 ? (uint32_t)std::atoi(argv[1]) : 30, n = argc > 3 ? (uint32_t)std::atoi(argv[2]) : 8192, batch = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 8;
    constexpr uint32_t L = 4; const uint32_t w = 16;
    uint64_t primes[L]; FHE_OK_(fhe_find_ntt_primes(bits, n, L, primes));
    uint64_t moduli[L][4]; for (uint32_t l = 0; l < L; l++) { moduli[l][0] = primes[l]; moduli[l][1] = moduli[l][2] = moduli[l][3] = 0; }
    fhe_rns_ntt_t *h = nullptr; FHE_OK_(fhe_rns_ntt_create(&h, n, moduli, L));
    hipStream_t s; HIP_OK(hipStreamCreate(&s));
    FHE_OK_(fhe_rns_ntt_set_stream(h, s));

    const size_t poly = (size_t)L * n * 4, bytes = (size_t)batch * poly * 8;      // u64 words
    std::vector<uint64_t> host((size_t)batch * poly, 0);
    auto fill = [&](uint64_t seed) { for (uint32_t b = 0; b < batch; b++) for (uint32_t l = 0; l < L; l++) for (uint32_t x = 0; x < n; x++) {
        seed = seed * 6364136223846793005ull + 1442695040888963407ull; host[((size_t)(b * L + l) * n + x) * 4] = (seed >> 20) % primes[l]; } };
    void *in[4], *out[3], *ref[3];
    for (int i = 0; i < 4; i++) { HIP_OK(hipMalloc(&in[i], bytes)); fill(1000 + i); HIP_OK(hipMemcpy(in[i], host.data(), bytes, hipMemcpyHostToDevice)); }
    for (int i = 0; i < 3; i++) { HIP_OK(hipMalloc(&out[i], bytes)); HIP_OK(hipMalloc(&ref[i], bytes)); }
    uint32_t K = 0; FHE_OK_(fhe_relin_num_digits(h, w, &K));
    std::vector<void *> kb(L * K), ka(L * K);
    for (uint32_t i = 0; i < L * K; i++) {
        HIP_OK(hipMalloc(&kb[i], poly * 8)); HIP_OK(hipMalloc(&ka[i], poly * 8));
        fill(5000 + i); HIP_OK(hipMemcpy(kb[i], host.data(), poly * 8, hipMemcpyHostToDevice));
        fill(9000 + i); HIP_OK(hipMemcpy(ka[i], host.data(), poly * 8, hipMemcpyHostToDevice));
    }
    fhe_relin_keys_t *rk = nullptr; FHE_OK_(fhe_relin_keys_create(h, &rk, w, kb.data(), ka.data(), L * K));

    // reference result by direct launches
    FHE_OK_(fhe_ct_multiply(h, ref[0], ref[1], ref[2], in[0], in[1], in[2], in[3], batch));
    FHE_OK_(fhe_ct_relinearize(h, rk, ref[0], ref[1], ref[2], batch));
    HIP_OK(hipStreamSynchronize(s));

    // capture the same two calls
    hipGraph_t graph; hipGraphExec_t exec;
    HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    FHE_OK_(fhe_ct_multiply(h, out[0], out[1], out[2], in[0], in[1], in[2], in[3], batch));
    FHE_OK_(fhe_ct_relinearize(h, rk, out[0], out[1], out[2], batch));
    HIP_OK(hipStreamEndCapture(s, &graph));
    size_t nodes = 0; HIP_OK(hipGraphGetNodes(graph, nullptr, &nodes));
    HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 3; i++) HIP_OK(hipMemsetAsync(out[i], 0xFF, bytes, s));
        HIP_OK(hipGraphLaunch(exec, s));
    }
    HIP_OK(hipStreamSynchronize(s));
    std::vector<uint64_t> a(host.size()), b(host.size());
    for (int i = 0; i < 2; i++) {
        HIP_OK(hipMemcpy(a.data(), out[i], bytes, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(b.data(), ref[i], bytes, hipMemcpyDeviceToHost));
        if (std::memcmp(a.data(), b.data(), bytes) != 0) { std::fprintf(stderr, "graph replay differs from direct launch (component %d)\n", i); return 1; }
    }
    std::printf("graph capture ok: %zu kernel nodes, replay bit-identical to direct launches\n", nodes);

    // ---- FHEContext::multiply as ONE call (compact workspace inside the library): reserve first, then the call allocates nothing --------
    FHE_OK_(fhe_rns_ntt_reserve(h, batch));
    hipGraph_t g1; hipGraphExec_t e1;
    HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    FHE_OK_(fhe_ct_multiply_relin(h, rk, out[0], out[1], in[0], in[1], in[2], in[3], batch));
    HIP_OK(hipStreamEndCapture(s, &g1));
    size_t nodes1 = 0; HIP_OK(hipGraphGetNodes(g1, nullptr, &nodes1));
    HIP_OK(hipGraphInstantiate(&e1, g1, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 2; i++) HIP_OK(hipMemsetAsync(out[i], 0xEE, bytes, s));
        HIP_OK(hipGraphLaunch(e1, s));
    }
    HIP_OK(hipStreamSynchronize(s));
    for (int i = 0; i < 2; i++) {
        HIP_OK(hipMemcpy(a.data(), out[i], bytes, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(b.data(), ref[i], bytes, hipMemcpyDeviceToHost));
        if (std::memcmp(a.data(), b.data(), bytes) != 0) { std::fprintf(stderr, "one-call multiply graph replay differs from the two direct calls (component %d)\n", i); return 1; }
    }
    std::printf("one-call ciphertext multiply captured: %zu nodes, replay bit-identical to ct_multiply + relinearize\n", nodes1);

    // ---- the blind-rotation loop (5 steps: odd, so the final copy-back is part of the graph too) ------------------------------
    const uint32_t steps = 5;
    std::vector<uint32_t> h_sh((size_t)steps * batch);
    for (size_t i = 0; i < h_sh.size(); i++) h_sh[i] = (uint32_t)((i * 2654435761u) % (2 * n));
    uint32_t *d_sh; HIP_OK(hipMalloc((void **)&d_sh, h_sh.size() * 4)); HIP_OK(hipMemcpy(d_sh, h_sh.data(), h_sh.size() * 4, hipMemcpyHostToDevice));
    std::vector<const fhe_relin_keys_t *> r0(steps, rk), r1(steps, rk);
    void *acc[2], *tmp[2], *want[2];
    for (int i = 0; i < 2; i++) { HIP_OK(hipMalloc(&acc[i], bytes)); HIP_OK(hipMalloc(&tmp[i], bytes)); HIP_OK(hipMalloc(&want[i], bytes)); }
    for (int i = 0; i < 2; i++) HIP_OK(hipMemcpyAsync(want[i], in[i], bytes, hipMemcpyDeviceToDevice, s));
    FHE_OK_(fhe_blind_rotate(h, r0.data(), r1.data(), steps, want[0], want[1], d_sh, tmp[0], tmp[1], batch));
    HIP_OK(hipStreamSynchronize(s));
    hipGraph_t g2; hipGraphExec_t e2;
    HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 2; i++) HIP_OK(hipMemcpyAsync(acc[i], in[i], bytes, hipMemcpyDeviceToDevice, s));
    FHE_OK_(fhe_blind_rotate(h, r0.data(), r1.data(), steps, acc[0], acc[1], d_sh, tmp[0], tmp[1], batch));
    HIP_OK(hipStreamEndCapture(s, &g2));
    size_t nodes2 = 0; HIP_OK(hipGraphGetNodes(g2, nullptr, &nodes2));
    HIP_OK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) HIP_OK(hipGraphLaunch(e2, s));
    HIP_OK(hipStreamSynchronize(s));
    for (int i = 0; i < 2; i++) {
        HIP_OK(hipMemcpy(a.data(), acc[i], bytes, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(b.data(), want[i], bytes, hipMemcpyDeviceToHost));
        if (std::memcmp(a.data(), b.data(), bytes) != 0) { std::fprintf(stderr, "blind-rotation graph replay differs from direct launches (component %d)\n", i); return 1; }
    }
    std::printf("blind-rotation loop captured: %zu nodes for %u steps, replay bit-identical\n", nodes2, steps);
    fhe_relin_keys_destroy(rk); fhe_rns_ntt_destroy(h);
    return 0;
}
