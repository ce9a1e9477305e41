"""CPU tests that pin the oracle (oracle/fhe_oracle.c) before anything is compared against it.

Level L0/L1: the reference's own known answers (tests/golden/reference_kats.json) + an independent
Python big-int closed form.  Level L2: the direct O(n^2) mathematics.
"""
import json
import os
import random

import numpy as np
import pytest

import ntt_math as nm

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kats():
    with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
        return json.load(f)


# ---------------------------------------------------------------------------------- L0 known answers
def test_reference_test_file_known_answers(oracle, kats):
    for c in kats["reference_tests"]["add_mod"]:
        assert oracle.add_mod(int(c["a"]), int(c["b"]), int(c["q"])) == int(c["r"])
    for c in kats["reference_tests"]["sub_mod"]:
        assert oracle.sub_mod(int(c["a"]), int(c["b"]), int(c["q"])) == int(c["r"])


def test_survey_primitive_kats(oracle, kats):
    for c in kats["survey_appendix_b"]["primitives"]:
        q = int(c["q"])
        assert oracle.mont_inverse(q) == int(c["inv0"], 16)
        assert oracle.mont_mul(5, 7, q) == int(c["mont_5_7"])
        assert oracle.mont_mul(q - 1, q - 1, q) == int(c["mont_qm1_qm1"])
        assert oracle.add_mod(q - 1, q - 1, q) == int(c["add_qm1_qm1"])
        assert oracle.sub_mod(0, 1, q) == int(c["sub_0_1"])
    e = kats["survey_appendix_b"]["even_modulus_inverse"]
    assert oracle.mont_inverse(int(e["q"])) == int(e["inv0"], 16)
    k = kats["survey_appendix_b"]["constants"]
    assert pow(2, 256, 12289) == int(k["R_mod_12289"])
    assert pow(2, 512, 12289) == int(k["R2_mod_12289"])
    assert pow(2, 256, 40961) == int(k["R_mod_40961"])


def test_survey_literal_kernel_kats(oracle, kats):
    """L1: reference kernels with the placeholder tables the reference really builds."""
    for c in kats["survey_appendix_b"]["literal_kernels_placeholder_tables"]:
        n, q = c["n"], int(c["q"])
        tw = oracle.ref_placeholder_table(n)
        x = oracle.to_limbs(range(1, n + 1))
        f = oracle.ref_forward_kernel(x, tw, q)
        assert oracle.from_limbs(f)[:8] == c["forward_first8"]
        i = oracle.ref_inverse_kernel(f, tw, q, int(c["n_inv"]))
        assert oracle.from_limbs(i)[:8] == c["then_inverse_first8"]


# ---------------------------------------------------------------------------------- L0 closed form
def _moduli():
    rng = random.Random(1234)
    ms = [12289, 40961, (1 << 39) + 1, 100000 + 1]
    ms += nm.ntt_primes(30, 8192, 2) + nm.ntt_primes(60, 8192, 1)
    for bits in (61, 64, 65, 120, 128, 129, 192, 193, 254, 255):
        ms.append(rng.getrandbits(bits) | (1 << (bits - 1)) | 1)
    return ms


def test_primitives_match_closed_form(oracle):
    rng = random.Random(99)
    for q in _moduli():
        edge = [0, 1, 2, q - 1, q - 2, q // 2, q, q + 1, (1 << 256) - 1, (1 << 255), rng.getrandbits(256)]
        ops = [(a, b) for a in edge for b in edge]
        ops += [(rng.randrange(q), rng.randrange(q)) for _ in range(200)]
        ops += [(rng.getrandbits(256), rng.getrandbits(256)) for _ in range(50)]   # unreduced: still literal
        inv0 = nm.mont_inverse_ref(q)
        assert oracle.mont_inverse(q) == inv0
        A = oracle.to_limbs([a for a, _ in ops]); B = oracle.to_limbs([b for _, b in ops])
        got_add = oracle.from_limbs(oracle.batch_add(A, B, q))
        got_sub = oracle.from_limbs(oracle.batch_sub(A, B, q))
        got_mul = oracle.from_limbs(oracle.batch_mont(A, B, q))
        for (a, b), ga, gs, gm in zip(ops, got_add, got_sub, got_mul):
            assert ga == nm.add_mod_ref(a, b, q)
            assert gs == nm.sub_mod_ref(a, b, q)
            assert gm == nm.mont_mul_ref(a, b, q, inv0)
        # reduced operands: Montgomery product is a*b*R^-1 mod q, canonical
        rinv = pow(nm.R, -1, q) if nm.is_prime(q) or np.gcd(q % (1 << 62), 2) == 1 else None
        try:
            rinv = pow(nm.R, -1, q)
        except ValueError:
            rinv = None
        if rinv is not None:
            for _ in range(50):
                a, b = rng.randrange(q), rng.randrange(q)
                assert oracle.mont_mul(a, b, q) == a * b * rinv % q


def test_butterflies_match_closed_form(oracle):
    rng = random.Random(7)
    for q in _moduli():
        for _ in range(40):
            a, b, w = rng.randrange(q), rng.randrange(q), rng.randrange(q)
            assert oracle.ct_butterfly(a, b, w, q) == nm.ct_ref(a, b, w, q)
            assert oracle.gs_butterfly(a, b, w, q) == nm.gs_ref(a, b, w, q)


def test_garbage_even_modulus_is_literal(oracle):
    """q = 2^60 (what FHEContext really passes, src/fhe.cu:13): deterministic garbage, still literal."""
    q = 1 << 60
    rng = random.Random(5)
    for _ in range(50):
        a, b = rng.randrange(q), rng.randrange(q)
        assert oracle.mont_mul(a, b, q) == nm.mont_mul_ref(a, b, q)


# ---------------------------------------------------------------------------------- L1 with real tables
def test_literal_kernels_with_proper_tables_are_a_cyclic_dft(oracle):
    """with tw[k] = psi^k * R the reference forward kernel is the cyclic DFT (omega =
    psi^2) of the bit-reversed input, and the inverse kernel undoes it."""
    q = 12289
    for n in (8, 64, 1024):
        psi = nm.find_psi(n, q)
        Rm = nm.R % q
        tw = oracle.to_limbs([pow(psi, k, q) * Rm % q for k in range(n)])
        itw = oracle.to_limbs([pow(psi, -k, q) * Rm % q for k in range(n)])
        n_inv_m = pow(n, -1, q) * Rm % q
        rng = random.Random(n)
        x = [rng.randrange(q) for _ in range(n)]
        bits = n.bit_length() - 1
        xb = [x[nm.bitrev(i, bits)] for i in range(n)]
        f = oracle.from_limbs(oracle.ref_forward_kernel(oracle.to_limbs(xb), tw, q))
        if n <= 64:
            om = psi * psi % q
            want = [sum(x[j] * pow(om, j * k, q) for j in range(n)) % q for k in range(n)]
            assert f == want
        back = oracle.from_limbs(oracle.ref_inverse_kernel(oracle.to_limbs(f), itw, q, n_inv_m))
        # The reference never un-permutes after the GS network; the data comes back bit-reversed (D6).
        assert back == xb or back == x


# ---------------------------------------------------------------------------------- L2 intended maths
@pytest.mark.parametrize("n,q", [(8, 12289), (64, 12289), (256, 40961), (128, None), (64, "wide")])
def test_forward_matches_direct_definition(oracle, n, q):
    if q is None:
        q = nm.ntt_primes(60, n, 1)[0]
    if q == "wide":
        q = nm.ntt_primes(250, n, 1)[0]
    plan = oracle.Plan(n, q)
    psi = nm.find_psi(n, q)
    assert plan.psi == psi
    bits = n.bit_length() - 1
    for k in (1, 2, 3, n - 1):
        assert plan.twiddle(k) == pow(psi, nm.bitrev(k, bits), q)
    rng = random.Random(n)
    x = [rng.randrange(q) for _ in range(n)]
    got = oracle.from_limbs(plan.forward(oracle.to_limbs(x)))
    assert got == nm.negacyclic_ntt_direct(x, q, psi)
    assert oracle.from_limbs(plan.inverse(oracle.to_limbs(got))) == x


def test_reference_roundtrip_case(oracle):
    """tests/test_fhe.cu:65-120: N = 1024, q = 12289, data i+1, INTT(NTT(x)) == x."""
    plan = oracle.Plan(1024, 12289)
    x = oracle.to_limbs(range(1, 1025))
    y = plan.forward(x)
    assert not np.array_equal(x, y)
    assert np.array_equal(plan.inverse(y), x)


@pytest.mark.parametrize("n,bits", [(8, 14), (256, 30), (256, 60), (128, 250)])
def test_polymul_matches_schoolbook(oracle, n, bits):
    q = 12289 if bits == 14 else nm.ntt_primes(bits, n, 1)[0]
    plan = oracle.Plan(n, q)
    rng = random.Random(bits)
    a = [rng.randrange(q) for _ in range(n)]; b = [rng.randrange(q) for _ in range(n)]
    A, B = oracle.to_limbs(a), oracle.to_limbs(b)
    want = nm.negacyclic_mul_direct(a, b, q)
    assert oracle.from_limbs(plan.polymul(A, B)) == want
    assert oracle.from_limbs(plan.schoolbook(A, B)) == want
    assert np.array_equal(A, oracle.to_limbs(a))       # operands preserved (src/ntt.cu:50-58)


def test_polymul_reference_shape_n2048(oracle):
    """tests/test_fhe.cu:126-167: N = 2048, q = 40961, coefficients rand()%100 (result never read back
    there; here it is checked against the C schoolbook oracle)."""
    plan = oracle.Plan(2048, 40961)
    rng = random.Random(2048)
    A = oracle.to_limbs(rng.randrange(100) for _ in range(2048))
    B = oracle.to_limbs(rng.randrange(100) for _ in range(2048))
    assert np.array_equal(plan.polymul(A, B), plan.schoolbook(A, B))


def test_rns_layout_and_threads(oracle):
    n, L, batch = 64, 3, 4
    moduli = nm.ntt_primes(30, n, L)
    rp = oracle.RnsPlan(n, moduli)
    rng = np.random.default_rng(3)
    a = np.zeros((batch, L, n, 4), np.uint64); b = np.zeros_like(a)
    for l, q in enumerate(moduli):
        a[:, l, :, 0] = rng.integers(0, q, (batch, n), dtype=np.uint64)
        b[:, l, :, 0] = rng.integers(0, q, (batch, n), dtype=np.uint64)
    r1 = rp.polymul(a, b, threads=1); r2 = rp.polymul(a, b, threads=2)
    assert np.array_equal(r1, r2)
    for bi in range(batch):
        for l in range(L):
            assert np.array_equal(r1[bi, l], rp.plans[l].polymul(a[bi, l], b[bi, l]))
    f = rp.forward(a, threads=2)
    assert np.array_equal(rp.inverse(f, threads=2), a)
    c0, c1, c2 = rp.ct_multiply(a, b, b, a, threads=2)
    assert np.array_equal(c0, rp.polymul(a, b)) and np.array_equal(c2, rp.polymul(b, a))
    q0 = moduli[0]
    want_c1 = oracle.batch_add(np.ascontiguousarray(rp.polymul(a, a)[:, 0]), np.ascontiguousarray(rp.polymul(b, b)[:, 0]), q0)
    assert np.array_equal(c1[:, 0], want_c1)


def test_plan_rejects_bad_moduli(oracle):
    with pytest.raises(ValueError):
        oracle.Plan(1024, 12289 + 2)       # not 1 mod 2n
    with pytest.raises(ValueError):
        oracle.Plan(1000, 12289)           # n not a power of two
    with pytest.raises(ValueError):
        oracle.Plan(8, (1 << 39) + 1)      # composite (3^2 * 2731 * 22366891): what src/rns.cu:199-204 returns


# ---------------------------------------------------------------------------------- N1: relinearisation
@pytest.mark.parametrize("n,bits,L,w,t", [(64, 30, 2, 16, 257), (64, 30, 3, 30, 257), (32, 60, 1, 16, 193), (32, 40, 2, 20, 193)])
def test_relinearisation_decrypts_like_the_three_component_ciphertext(oracle, n, bits, L, w, t):
    """Dec(relin(c0,c1,c2)) == Dec(c0,c1,c2) == m1*m2, with the toy BGV of tests/bgv_toy.py (big-int CRT)."""
    import bgv_toy
    moduli = nm.ntt_primes(bits, n, L)
    S = bgv_toy.ToyBGV(n, moduli, t, seed=n + L)
    rp = oracle.RnsPlan(n, moduli)
    rng = random.Random(5)
    m1 = [rng.randrange(t) for _ in range(n)]; m2 = [rng.randrange(t) for _ in range(n)]
    a0, a1 = S.encrypt(m1); b0, b1 = S.encrypt(m2)
    A0, A1, B0, B1 = (bgv_toy.to_limb_array(x) for x in (a0, a1, b0, b1))
    c0, c1, c2 = rp.ct_multiply(A0, A1, B0, B1)
    want = nm.negacyclic_mul_direct(m1, m2, t)
    assert S.decrypt([bgv_toy.from_limb_array(c) for c in (c0, c1, c2)]) == want
    kb, ka, K = S.relin_keys(w)
    assert K == rp.num_digits(w)
    KB = [bgv_toy.to_limb_array(k)[0] for k in kb]; KA = [bgv_toy.to_limb_array(k)[0] for k in ka]
    r0, r1 = rp.relinearize(w, c0, c1, c2, KB, KA, threads=2)
    assert S.decrypt([bgv_toy.from_limb_array(r0), bgv_toy.from_limb_array(r1)]) == want
    assert not np.array_equal(r0, c0)


def test_reference_fhe_expectation_15_60_135_240(oracle):
    """tests/test_fhe.cu:169-273: {5,10,15,20} x {3,6,9,12} -> 15 60 135 240, and {..}+{..} -> 8 16 24 32, through
    tensor product + relinearisation (the reference prints these expectations but cannot reach them)."""
    import bgv_toy
    n, t, w = 64, 257, 16
    moduli = nm.ntt_primes(30, n, 2)
    S = bgv_toy.ToyBGV(n, moduli, t, seed=3)
    rp = oracle.RnsPlan(n, moduli)
    m1 = S.slot_encode([5, 10, 15, 20]); m2 = S.slot_encode([3, 6, 9, 12])
    a0, a1 = S.encrypt(m1); b0, b1 = S.encrypt(m2)
    A0, A1, B0, B1 = (bgv_toy.to_limb_array(x) for x in (a0, a1, b0, b1))
    c0, c1, c2 = rp.ct_multiply(A0, A1, B0, B1)
    kb, ka, K = S.relin_keys(w)
    r0, r1 = rp.relinearize(w, c0, c1, c2, [bgv_toy.to_limb_array(k)[0] for k in kb], [bgv_toy.to_limb_array(k)[0] for k in ka])
    got = S.slot_decode(S.decrypt([bgv_toy.from_limb_array(r0), bgv_toy.from_limb_array(r1)]))
    assert got[:4] == [15, 60, 135, 240] and not any(got[4:])
    s0 = oracle.batch_add(np.ascontiguousarray(A0[0, 0]), np.ascontiguousarray(B0[0, 0]), moduli[0])
    sum_ct = [S.add(a0, b0), S.add(a1, b1)]
    assert S.slot_decode(S.decrypt(sum_ct))[:4] == [8, 16, 24, 32]
    assert [int(v) for v in s0[:, 0]] == sum_ct[0][0]


# ---------------------------------------------------------------------------------- RNS entry / exit (row a18)
@pytest.mark.parametrize("n,bits,L", [(16, 30, 4), (16, 60, 4), (8, 120, 2), (32, 30, 8), (8, 250, 1)])
def test_to_rns_and_crt_match_big_integers(oracle, n, bits, L):
    moduli = nm.ntt_primes(bits, n, L)
    Q = 1
    for q in moduli:
        Q *= q
    rp = oracle.RnsPlan(n, moduli)
    rng = random.Random(bits * L)
    vals = [rng.randrange(Q) for _ in range(2 * n - 3)] + [0, Q - 1, 1]
    V = oracle.to_limbs(vals).reshape(2, n, 4)
    R = rp.to_rns(V)
    for b in range(2):
        for l, q in enumerate(moduli):
            assert oracle.from_limbs(R[b, l]) == [v % q for v in vals[b * n:(b + 1) * n]]
    assert oracle.from_limbs(rp.from_rns(R)) == vals
    # to_rns accepts any 256-bit integer (values >= Q simply wrap mod each prime)
    big = [rng.getrandbits(256) for _ in range(n)]
    Rb = rp.to_rns(oracle.to_limbs(big).reshape(1, n, 4))
    for l, q in enumerate(moduli):
        assert oracle.from_limbs(Rb[0, l]) == [v % q for v in big]


def test_crt_rejects_too_large_bases(oracle):
    n = 8
    rp = oracle.RnsPlan(n, nm.ntt_primes(60, n, 5))            # 300 bits
    with pytest.raises(ValueError):
        rp.from_rns(np.zeros((1, 5, n, 4), np.uint64))


@pytest.mark.parametrize("n,bits,L", [(16, 30, 3), (16, 60, 2), (8, 120, 2), (16, 30, 5)])
def test_rescale_drop_last_is_rounded_division(oracle, n, bits, L):
    """(C - r) / q_last with the centred remainder r, i.e. round(C / q_last), limb-wise."""
    moduli = nm.ntt_primes(bits, n, L)
    Q = 1
    for q in moduli:
        Q *= q
    ql = moduli[-1]
    rp = oracle.RnsPlan(n, moduli)
    rng = random.Random(bits + L)
    vals = [rng.randrange(Q) for _ in range(2 * n - 4)] + [0, Q - 1, ql // 2, ql // 2 + 1]
    R = rp.to_rns(oracle.to_limbs(vals).reshape(2, n, 4))
    out = rp.rescale_drop_last(R)
    for b in range(2):
        for i, C in enumerate(vals[b * n:(b + 1) * n]):
            r = C % ql
            if r > ql // 2:
                r -= ql
            want = (C - r) // ql
            assert (C - r) % ql == 0 and abs(want * ql - C) <= ql // 2
            for l, q in enumerate(moduli[:-1]):
                assert oracle.from_limbs(out[b, l, i:i + 1])[0] == want % q


@pytest.mark.parametrize("n,bits,L,bits2,Lp", [(16, 30, 3, 30, 4), (16, 30, 4, 60, 2), (8, 60, 2, 30, 3), (8, 120, 2, 250, 1)])
def test_fast_base_conversion_matches_its_definition(oracle, n, bits, L, bits2, Lp):
    """y_j = sum_i [x_i (Q/q_i)^-1]_{q_i} (Q/q_i) mod p_j = (X + alpha Q) mod p_j with 0 <= alpha < L."""
    src = nm.ntt_primes(bits, n, L)
    dst = [p for p in nm.ntt_primes(bits2, n, Lp + L) if p not in src][:Lp]
    Q = 1
    for q in src:
        Q *= q
    S, D = oracle.RnsPlan(n, src), oracle.RnsPlan(n, dst)
    rng = random.Random(bits + bits2)
    vals = [rng.randrange(Q) for _ in range(n - 2)] + [0, Q - 1]
    R = S.to_rns(oracle.to_limbs(vals).reshape(1, n, 4))
    Y = S.fast_base_convert(D, R)
    for i, X in enumerate(vals):
        t = [(X % q) * pow(Q // q, -1, q) % q for q in src]
        full = sum(ti * (Q // q) for ti, q in zip(t, src))
        alpha, rem = divmod(full - X, Q)
        assert rem == 0 and 0 <= alpha < L
        for j, p in enumerate(dst):
            assert oracle.from_limbs(Y[0, j, i:i + 1])[0] == full % p


# ---------------------------------------------------------------------------------- N3: blind-rotation inner loop
def test_monomial_mul_sub_matches_definition(oracle):
    n = 32; moduli = nm.ntt_primes(30, n, 2); rp = oracle.RnsPlan(n, moduli)
    from workload import rns_poly
    x = rns_poly(5, moduli, n, 4)
    shifts = [0, 1, n + 3, 2 * n - 1]
    out = rp.monomial_mul_sub(x, shifts)
    for b, a in enumerate(shifts):
        mono = [0] * n
        sign = 1 if a < n else -1
        mono[a % n] = sign
        for l, q in enumerate(moduli):
            p = [int(v) for v in x[b, l, :, 0]]
            want = [(u - v) % q for u, v in zip(nm.negacyclic_mul_direct([m % q for m in mono], p, q), p)]
            assert [int(v) for v in out[b, l, :, 0]] == want


def test_blind_rotate_steps_rotate_the_plaintext(oracle):
    """acc <- acc + ExtProd((X^a - 1) acc, RGSW(s)) multiplies the encrypted polynomial by X^(a*s): three steps with
    secret bits 1, 0, 1 (toy BGV, big-integer decryption)."""
    import bgv_toy
    n, t, w = 64, 257, 16
    moduli = nm.ntt_primes(30, n, 2)
    S = bgv_toy.ToyBGV(n, moduli, t, seed=9)
    rp = oracle.RnsPlan(n, moduli)
    rng = random.Random(3)
    m = [rng.randrange(t) for _ in range(n)]
    c0, c1 = S.encrypt(m)
    A0, A1 = bgv_toy.to_limb_array(c0), bgv_toy.to_limb_array(c1)
    total = 0
    for bit, a in ((1, 5), (0, 17), (1, n + 9)):
        r0, r1, K = S.rgsw(bit, w)
        rows = [tuple([bgv_toy.to_limb_array(k)[0] for k in part] for part in r) for r in (r0, r1)]
        A0, A1 = rp.blind_rotate_step(w, A0, A1, [a], rows[0], rows[1])
        total += a * bit
        mono = [0] * n; mono[total % n] = 1 if (total // n) % 2 == 0 else t - 1
        want = nm.negacyclic_mul_direct(mono, m, t)
        assert S.decrypt([bgv_toy.from_limb_array(A0), bgv_toy.from_limb_array(A1)]) == want


# ---------------------------------------------------------------------------------- N4: samplers, modulus switch, fold
def test_literal_samplers_match_the_reference_formulas(oracle):
    """sample_uniform_kernel / sample_gaussian_kernel (src/polynomial.cu:113-143) are deterministic placeholders: restated
    literally, checked against the formulas in Python integers (64-bit wrap-around)."""
    M = (1 << 64) - 1
    for q, seed, count in [(12289, 1804289383, 1024), ((1 << 60), 846930886, 33), ((1 << 64) - 59, (1 << 64) - 5, 100), (2, 0, 7)]:
        u = oracle.from_limbs(oracle.sample_uniform_lcg(q, seed, count))
        g = oracle.from_limbs(oracle.sample_gaussian_placeholder(q, seed, count))
        q0 = q & M
        assert u == [((((seed + i) & M) * 1103515245 + 12345) & M) % q0 for i in range(count)]
        assert g == [((seed + i) & M) % q0 for i in range(count)]


def _sm64(z):
    M = (1 << 64) - 1
    z = (z + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def _ctr(seed, index, draw):
    M = (1 << 64) - 1
    return _sm64((_sm64(seed ^ ((index * 0xD1342543DE82EF95) & M)) + draw) & M)


def test_counter_generator_matches_its_specification(oracle):
    for seed, idx, draw in [(0, 0, 0), (1, 2, 3), ((1 << 64) - 1, 123456789, 16), (42, 1 << 40, 2)]:
        assert oracle.ctr_rand(seed, idx, draw) == _ctr(seed, idx, draw)
    assert _sm64(0) == 0xE220A8397B1DCDAF          # SplitMix64's first output for seed 0 (published test vector)


def test_ternary_and_gaussian_samplers(oracle):
    n = 4096; moduli = nm.ntt_primes(30, n, 2) + nm.ntt_primes(40, n, 1); rp = oracle.RnsPlan(n, moduli)
    t = rp.sample_ternary(0.5, seed=7, batch=4)
    vals = []
    for g in range(4 * n):
        b, x = divmod(g, n)
        r = _ctr(7, g, 0); mag = 1 if (r & 0xffffffff) < (1 << 31) else 0; neg = r >> 63
        for l, q in enumerate(moduli):
            want = 0 if not mag else (q - 1 if neg else 1)
            assert int(t[b, l, x, 0]) == want and not t[b, l, x, 1:].any()
        vals.append(-mag if neg else mag)
    nz = sum(1 for v in vals if v)
    assert abs(nz / len(vals) - 0.5) < 0.03 and abs(sum(vals)) < 4 * (len(vals) ** 0.5)
    assert not rp.sample_ternary(0.0, 1, 1)[..., 0].any()
    # discrete Gaussian: the table is a CDF, samples have the right variance, limbs agree
    sigma = 3.2
    cdt = oracle.gaussian_cdt(sigma)
    assert len(cdt) == 39 and all(a <= b for a, b in zip(cdt, cdt[1:])) and cdt[-1] >= (1 << 64) - (1 << 20)
    import math
    Z = 1 + 2 * sum(math.exp(-k * k / (2 * sigma * sigma)) for k in range(1, 40))
    assert abs(cdt[0] / 2.0 ** 64 - 1 / Z) < 1e-12
    g = rp.sample_gaussian(sigma, seed=11, batch=8)
    c = g[:, 0, :, 0].astype(np.int64).reshape(-1); q0 = moduli[0]
    c = np.where(c > q0 // 2, c - q0, c)
    assert abs(c.mean()) < 0.1 and abs(c.var() - sigma * sigma) < 0.5 and np.abs(c).max() <= len(cdt)
    for l, q in enumerate(moduli):
        cl = g[:, l, :, 0].astype(np.int64).reshape(-1); cl = np.where(cl > q // 2, cl - q, cl)
        assert np.array_equal(cl, c)


def test_uniform_sampler_is_canonical_and_unbiased(oracle):
    n = 2048; moduli = nm.ntt_primes(30, n, 1) + nm.ntt_primes(60, n, 1) + nm.ntt_primes(250, n, 1)
    rp = oracle.RnsPlan(n, moduli)
    u = rp.sample_uniform(seed=5, batch=3)
    for l, q in enumerate(moduli):
        v = oracle.from_limbs(u[:, l])
        assert all(0 <= x < q for x in v)
        assert abs(sum(v) / len(v) / q - 0.5) < 0.02
    # first element by hand: masked draws until one is below q
    q = moduli[0]; bits = q.bit_length(); t = 0
    while True:
        r = _ctr(5, 0, 16 + 4 * t) & ((1 << bits) - 1)
        if r < q:
            break
        t += 1
    assert int(u[0, 0, 0, 0]) == r
    assert not np.array_equal(u, rp.sample_uniform(seed=6, batch=3))


def test_poly_mod_switch_rounds_like_big_integers(oracle):
    rng = random.Random(77)
    cases = [(nm.ntt_primes(60, 4096, 1)[0], 65537), ((1 << 120) + 451, 257), ((1 << 254) + 79, (1 << 64) - 59), (1000003, 2), (12289, 12289)]
    for old_q, new_q in cases:
        a = [0, 1, old_q - 1, old_q // 2, old_q // 2 + 1, old_q // new_q, (old_q // (2 * new_q)), (old_q // (2 * new_q)) + 1] + [rng.randrange(old_q) for _ in range(56)]
        got = oracle.from_limbs(oracle.poly_mod_switch(oracle.to_limbs(a), old_q, new_q))
        assert got == [((x * new_q + old_q // 2) // old_q) % new_q for x in a]


def test_negacyclic_reduce_folds_the_upper_half(oracle):
    rng = random.Random(3)
    n, q = 64, 12289
    d = [rng.randrange(q) for _ in range(2 * n)]
    out = oracle.from_limbs(oracle.negacyclic_reduce(oracle.to_limbs(d), q))
    assert out[:n] == [(d[i] - d[i + n]) % q for i in range(n)] and out[n:] == d[n:]
    # consistent with the negacyclic product: fold(schoolbook over Z[x]) == polymul mod x^n + 1
    a = [rng.randrange(q) for _ in range(n)]; b = [rng.randrange(q) for _ in range(n)]
    full = [0] * (2 * n)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            full[i + j] = (full[i + j] + x * y) % q
    assert oracle.from_limbs(oracle.negacyclic_reduce(oracle.to_limbs(full), q))[:n] == nm.negacyclic_mul_direct(a, b, q)


def test_stockham_stages_equal_the_in_place_forward_kernel(oracle):
    """ntt_stockham_kernel (kernels/ntt_kernels.cu:213-243) stage by stage is the same butterfly network as the in-place forward
    kernel; only the table indexing differs: Stockham reads tw[j * n / 2m], the in-place kernel (with its log_n = log2(n) + 1)
    reads tw[j * n / m], i.e. twice the index -- so the two agree when the in-place kernel is given the table spread out by 2."""
    rng = random.Random(12)
    for n, q in [(8, 12289), (64, 40961), (256, nm.ntt_primes(60, 4096, 1)[0])]:
        t = [rng.randrange(q) for _ in range(n)]
        tw = oracle.to_limbs(t)
        spread = [0] * n
        for k in range(n // 2):
            spread[2 * k] = t[k]
        x = oracle.to_limbs([rng.randrange(q) for _ in range(n)])
        y = x
        for stage in range(n.bit_length() - 1):
            y = oracle.ref_stockham_stage(y, tw, q, stage)
        assert np.array_equal(y, oracle.ref_forward_kernel(x, oracle.to_limbs(spread), q))


def test_word_sized_cpu_port_equals_the_wide_oracle(oracle):
    """orc_rns_polymul_narrow (64-bit residues) == orc_rns_polymul (256-bit containers) on 30-, 40- and 60-bit primes."""
    from workload import rns_poly
    for n, bits, L in [(64, 30, 3), (2048, 40, 2), (1024, 60, 2)]:
        moduli = nm.ntt_primes(bits, n, L); rp = oracle.RnsPlan(n, moduli)
        a, b = rns_poly(71, moduli, n, 3), rns_poly(72, moduli, n, 3)
        assert np.array_equal(rp.polymul_narrow(a, b, threads=2), rp.polymul(a, b, threads=2))
    with pytest.raises(ValueError):
        n = 64; moduli = nm.ntt_primes(250, n, 1)
        oracle.RnsPlan(n, moduli).polymul_narrow(rns_poly(1, moduli, n, 1), rns_poly(2, moduli, n, 1))
