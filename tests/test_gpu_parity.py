"""GPU parity: the HIP engine, called through the C ABI (ctypes), against the CPU oracle on the same
seeded inputs.  Integer work: every comparison is bit-exact on whole arrays."""
import os
import random

import numpy as np
import pytest

import ntt_math as nm
from workload import rns_poly

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(pkg):
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the MI355X box")
    return pkg


def _up(pkg, arr):
    return pkg.DeviceBuffer.from_numpy(arr)


# ------------------------------------------------------------------------------------ L0 primitives
def _l0_moduli():
    rng = random.Random(4321)
    ms = [100000, 12289, 40961, (1 << 39) + 1, 1 << 60]
    ms += nm.ntt_primes(30, 8192, 1) + nm.ntt_primes(60, 8192, 1)
    for bits in (64, 65, 128, 129, 192, 254, 255):
        ms.append(rng.getrandbits(bits) | (1 << (bits - 1)) | 1)
    return ms


def test_elementwise_primitives_literal(eng, oracle):
    """batch_mod_{add,sub,mul}_kernel semantics incl. unreduced operands and the even modulus 2^60."""
    rng = random.Random(11)
    for q in _l0_moduli():
        edge = [0, 1, q - 1, q, q + 1, (1 << 256) - 1, 1 << 255, 12345, 67890]
        pairs = [(a, b) for a in edge for b in edge]
        pairs += [(rng.randrange(q), rng.randrange(q)) for _ in range(3000)]
        pairs += [(rng.getrandbits(256), rng.getrandbits(256)) for _ in range(500)]
        A = oracle.to_limbs([a for a, _ in pairs]); B = oracle.to_limbs([b for _, b in pairs])
        dA, dB = _up(eng, A), _up(eng, B)
        dR = eng.DeviceBuffer(A.nbytes)
        inv0 = eng.montgomery_inverse(q) & ((1 << 64) - 1)
        assert inv0 == oracle.mont_inverse(q)
        eng.u256_add_mod(dR, dA, dB, q, len(pairs)); assert np.array_equal(dR.download(), oracle.batch_add(A, B, q))
        eng.u256_sub_mod(dR, dA, dB, q, len(pairs)); assert np.array_equal(dR.download(), oracle.batch_sub(A, B, q))
        eng.u256_mont_mul(dR, dA, dB, q, inv0, len(pairs)); assert np.array_equal(dR.download(), oracle.batch_mont(A, B, q, inv0))
        s = rng.randrange(q)
        S = oracle.to_limbs([s] * len(pairs))
        eng.u256_mont_mul_scalar(dR, dA, s, q, inv0, len(pairs))
        assert np.array_equal(dR.download(), oracle.batch_mont(A, S, q, inv0))


def test_reference_test_known_answers_on_device(eng, oracle):
    """tests/test_fhe.cu:24-63: 12345 +/- 67890 mod 100000 through the device kernels."""
    A, B = oracle.to_limbs([12345]), oracle.to_limbs([67890])
    dA, dB, dR = _up(eng, A), _up(eng, B), eng.DeviceBuffer(32)
    eng.u256_add_mod(dR, dA, dB, 100000, 1); assert oracle.from_limbs(dR.download()) == [80235]
    eng.u256_sub_mod(dR, dA, dB, 100000, 1); assert oracle.from_limbs(dR.download()) == [44455]


# ------------------------------------------------------------------------------------ transforms
CASES = [
    # (n, moduli spec, batch, expected width class)
    (1024, [12289], 2, 4),                      # the reference's test_ntt_transform shape -> general path
    (64, ("bits", 250, 2), 2, 4),
    (4096, ("bits", 250, 1), 1, 4),
    (2048, [40961], 3, 1),                      # test_polynomial_multiplication shape -> 32-bit LDS path
    (4096, ("bits", 30, 2), 2, 1),
    (8192, ("bits", 30, 4), 3, 1),              # BASELINE config 2 shape
    (16384, ("bits", 30, 3), 2, 1),
    (32768, ("bits", 30, 1), 2, 1),
    (2048, ("bits", 60, 2), 2, 2),
    (8192, ("bits", 60, 2), 2, 2),
    (16384, ("bits", 40, 6), 1, 3),             # BASELINE config 4 shape (40-bit primes) -> FP64-FMA path
    (8192, ("bits", 43, 2), 2, 3),              # widest primes the FP64 path accepts
    (2048, ("bits", 31, 2), 2, 3),              # just above the 32-bit path
    (4096, ("bits", 44, 1), 1, 2),              # just above the FP64 path -> 64-bit integer path
    (8192, ("bits", 64, 1), 1, 5),              # 64-bit prime: too wide for the lazy 64-bit path -> full-range 64-bit path (F64X)
    (2048, ("bits", 63, 2), 2, 5),
    (16384, ("bits", 64, 2), 1, 5),
    (4096, ("mix", (64, 1), (50, 1), (63, 1)), 2, 5),   # one wide prime pulls the whole basis onto the full-range path
    # beyond the LDS range (N = 2^16 on 4-byte residues, 2^15 / 2^16 on 8-byte residues): two-pass transforms on the field type
    (32768, ("bits", 64, 1), 1, 5),
    (32768, ("bits", 40, 2), 1, 3),
    (65536, ("bits", 43, 1), 1, 3),
    (32768, ("bits", 60, 2), 1, 2),
    (65536, ("bits", 62, 1), 2, 2),
    (65536, ("bits", 64, 1), 1, 5),
    (65536, ("bits", 30, 3), 2, 1),
    (65536, ("bits", 30, 1), 1, 1),             # larger than LDS: two-pass transform
    # full-width class, LDS-staged tiles of 2^11 coefficients (ntt_wide.hip.h): two-limb (q < 2^127) and four-limb residues,
    # one tile per polynomial, 1-3 top stages in one global pass, 4-5 in two
    (2048, ("bits", 120, 2), 2, 4),
    (2048, ("bits", 255, 1), 3, 4),
    (8192, ("bits", 127, 2), 2, 4),
    (8192, ("bits", 250, 2), 2, 4),             # the shape of the round-1 full-width bench line
    (16384, ("bits", 200, 1), 1, 4),
    (32768, ("bits", 128, 1), 1, 4),
    (65536, ("bits", 100, 1), 1, 4),
    (4096, ("mix", (250, 1), (30, 1), (100, 1)), 2, 4),
]


def _moduli(spec, n):
    if isinstance(spec, tuple) and spec[0] == "mix":
        return [q for bits, cnt in spec[1:] for q in nm.ntt_primes(bits, n, cnt)]
    if isinstance(spec, tuple):
        return nm.ntt_primes(spec[1], n, spec[2])
    return list(spec)


@pytest.mark.parametrize("n,spec,batch,width", CASES)
def test_forward_inverse_multiply_match_oracle(eng, oracle, monkeypatch, n, spec, batch, width):
    moduli = _moduli(spec, n)
    L = len(moduli)
    e = eng.RnsNttEngine(n, moduli)
    assert e.width_class == width
    rp = oracle.RnsPlan(n, moduli)
    a = rns_poly(1, moduli, n, batch); b = rns_poly(2, moduli, n, batch)
    dA, dB = _up(eng, a), _up(eng, b)
    dR = eng.DeviceBuffer(a.nbytes)
    shape = a.shape
    # forward
    e.forward(dA, batch)
    fa = dA.download(shape)
    want_fa = rp.forward(a, threads=8)
    assert np.array_equal(fa, want_fa)
    # inverse returns the input
    e.inverse(dA, batch)
    assert np.array_equal(dA.download(shape), a)
    # inverse of arbitrary NTT-domain data
    dT = _up(eng, b); e.inverse(dT, batch)
    assert np.array_equal(dT.download(shape), rp.inverse(b, threads=8))
    # pointwise (plain product)
    e.pointwise(dR, dA, dB, batch)
    want_pw = np.stack([np.stack([rp.plans[l].pointwise(np.ascontiguousarray(a[bi, l]), np.ascontiguousarray(b[bi, l]))
                                  for l in range(L)]) for bi in range(batch)])
    assert np.array_equal(dR.download(shape), want_pw)
    # multiply: operands preserved, result == oracle polymul
    e.multiply(dR, dA, dB, batch)
    want_mul = rp.polymul(a, b, threads=8)
    assert np.array_equal(dR.download(shape), want_mul)
    assert np.array_equal(dA.download(shape), a) and np.array_equal(dB.download(shape), b)
    if width == eng.WIDTH_32 and n <= 8192:
        # few polynomials run the 16-per-thread latency kernel (ntt_lds_small.hip.h) by default: the throughput kernel on the same operands,
        # and the latency kernel forced for any batch, must give the same containers
        for polys in ("0", "1000000"):           # (the default engine above spread each polynomial over four workgroups at N = 8192: ntt_multiply4_*_kernel)
            monkeypatch.setenv("FHE_HIP_SMALL_BATCH_POLYS", polys)
            monkeypatch.setenv("FHE_HIP_COOP_POLYS", "0")
            e2 = eng.RnsNttEngine(n, moduli)
            monkeypatch.delenv("FHE_HIP_SMALL_BATCH_POLYS"); monkeypatch.delenv("FHE_HIP_COOP_POLYS")
            dR.zero(); e2.multiply(dR, dA, dB, batch)
            assert np.array_equal(dR.download(shape), want_mul), polys
            dC = _up(eng, a); e2.multiply(dC, dC, dB, batch)                    # in place on the first operand
            assert np.array_equal(dC.download(shape), want_mul), polys
    # add / sub
    e.poly_add(dR, dA, dB, batch)
    want = np.stack([np.stack([oracle.batch_add(np.ascontiguousarray(a[bi, l]), np.ascontiguousarray(b[bi, l]), moduli[l])
                               for l in range(L)]) for bi in range(batch)])
    assert np.array_equal(dR.download(shape), want)
    e.poly_sub(dR, dA, dB, batch)
    want = np.stack([np.stack([oracle.batch_sub(np.ascontiguousarray(a[bi, l]), np.ascontiguousarray(b[bi, l]), moduli[l])
                               for l in range(L)]) for bi in range(batch)])
    assert np.array_equal(dR.download(shape), want)
    e.check_canonical(dR, batch)


@pytest.mark.parametrize("n,L,batch", [(8192, 4, 1), (8192, 4, 4), (8192, 1, 3), (16384, 6, 2), (16384, 3, 1), (8192, 2, 8)])
def test_multiply_one_polynomial_over_four_workgroups(eng, oracle, monkeypatch, n, L, batch):
    """ntt_multiply4_{top,block,last}_kernel (a handful of polynomials at N = 2^13 / 2^14 on 4-byte residues): each polynomial over four workgroups in three
    dependent launches (top two stages by columns, 16-per-thread sub-transforms + pointwise product on the four blocks, last two inverse stages by columns).
    Repeated calls, in-place products, a broadcast operand, and the same product from the one-workgroup kernels (FHE_HIP_COOP_POLYS=0); all equal to the oracle."""
    moduli = _moduli(("bits", 30, L), n)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    monkeypatch.setenv("FHE_HIP_COOP_POLYS", "0")
    e0 = eng.RnsNttEngine(n, moduli)
    monkeypatch.delenv("FHE_HIP_COOP_POLYS")
    for rep in range(3):
        a = rns_poly(70 + rep, moduli, n, batch); b = rns_poly(80 + rep, moduli, n, batch)
        dA, dB, dR = _up(eng, a), _up(eng, b), eng.DeviceBuffer(a.nbytes)
        want = rp.polymul(a, b, threads=8)
        for _ in range(4):                                        # back-to-back launches on one stream
            e.multiply(dR, dA, dB, batch)
        assert np.array_equal(dR.download(a.shape), want), rep
        assert np.array_equal(dA.download(a.shape), a) and np.array_equal(dB.download(a.shape), b)
        dR.zero(); e0.multiply(dR, dA, dB, batch)
        assert np.array_equal(dR.download(a.shape), want)
        dC = _up(eng, a); e.multiply(dC, dC, dB, batch)           # in place on the first operand
        assert np.array_equal(dC.download(a.shape), want)
        dC = _up(eng, a); e.multiply(dC, dC, dC, batch)           # in-place squaring
        assert np.array_equal(dC.download(a.shape), rp.polymul(a, a, threads=8))
    one = rns_poly(99, moduli, n, 1)
    dOne = _up(eng, one); e.multiply_bcast(dR, dA, dOne, batch)
    assert np.array_equal(dR.download(a.shape), rp.polymul(a, np.repeat(one, batch, axis=0), threads=8))


@pytest.mark.parametrize("n,spec,batch", [(2048, [40961], 2), (8192, ("bits", 30, 4), 2), (4096, ("bits", 60, 2), 1),
                                          (16384, ("bits", 40, 6), 1), (256, ("bits", 250, 2), 1), (8192, ("bits", 64, 2), 1), (16384, ("bits", 64, 1), 1), (8192, ("bits", 250, 1), 2), (4096, ("bits", 100, 2), 1), (65536, ("bits", 30, 2), 1), (32768, ("bits", 40, 2), 1)])
def test_ct_multiply_matches_oracle(eng, oracle, n, spec, batch):
    """FHEContext::multiply tensor product (src/fhe.cu:199-218)."""
    moduli = _moduli(spec, n)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    a0, a1, b0, b1 = (rns_poly(s, moduli, n, batch) for s in (3, 4, 5, 6))
    d = [_up(eng, x) for x in (a0, a1, b0, b1)]
    c = [eng.DeviceBuffer(a0.nbytes) for _ in range(3)]
    e.ct_multiply(c[0], c[1], c[2], d[0], d[1], d[2], d[3], batch)
    w0, w1, w2 = rp.ct_multiply(a0, a1, b0, b1, threads=8)
    assert np.array_equal(c[0].download(a0.shape), w0)
    assert np.array_equal(c[1].download(a0.shape), w1)
    assert np.array_equal(c[2].download(a0.shape), w2)
    for buf, src in zip(d, (a0, a1, b0, b1)):
        assert np.array_equal(buf.download(a0.shape), src)


def test_single_modulus_engine_reference_scenarios(eng, oracle):
    """tests/test_fhe.cu:65-167 through the NTTEngine-shaped ABI, asserting instead of printing."""
    # test_ntt_transform: N = 1024, q = 12289, data i+1, forward then inverse
    e = eng.NttEngine(1024, 12289)
    x = oracle.to_limbs(range(1, 1025)); d = _up(eng, x)
    e.forward(d); y = d.download()
    assert np.array_equal(y, oracle.Plan(1024, 12289).forward(x)) and not np.array_equal(y, x)
    e.inverse(d); assert np.array_equal(d.download(), x)
    # test_polynomial_multiplication: N = 2048, q = 40961, coefficients < 100
    rng = random.Random(5)
    a = oracle.to_limbs(rng.randrange(100) for _ in range(2048)); b = oracle.to_limbs(rng.randrange(100) for _ in range(2048))
    e2 = eng.NttEngine(2048, 40961)
    dA, dB, dR = _up(eng, a), _up(eng, b), eng.DeviceBuffer(a.nbytes)
    e2.multiply(dR, dA, dB)
    p = oracle.Plan(2048, 40961)
    assert np.array_equal(dR.download(), p.schoolbook(a, b))


@pytest.mark.parametrize("force,width", [("64", 2), ("65", 5), ("128", 4), ("256", 4)])
def test_forced_wider_paths_agree(eng, oracle, monkeypatch, force, width):
    """The same 40-bit basis through the 64-bit integer path and the full-width path (FHE_HIP_FORCE_WIDTH)."""
    n = 4096; moduli = nm.ntt_primes(40, n, 2); batch = 2
    monkeypatch.setenv("FHE_HIP_FORCE_WIDTH", force)
    e = eng.RnsNttEngine(n, moduli)
    assert e.width_class == width
    rp = oracle.RnsPlan(n, moduli)
    a = rns_poly(31, moduli, n, batch); b = rns_poly(32, moduli, n, batch)
    dA, dB, dR = _up(eng, a), _up(eng, b), eng.DeviceBuffer(a.nbytes)
    e.multiply(dR, dA, dB, batch)
    assert np.array_equal(dR.download(a.shape), rp.polymul(a, b, threads=8))
    e.forward(dA, batch)
    assert np.array_equal(dA.download(a.shape), rp.forward(a, threads=8))


@pytest.mark.parametrize("n,bits,L", [(8192, 30, 2), (8192, 43, 2), (16384, 40, 2), (8192, 62, 1), (2048, 30, 1), (8192, 64, 1), (2048, 63, 2), (16384, 64, 1)])
def test_extreme_value_polynomials(eng, oracle, n, bits, L):
    """Worst-case magnitudes for the lazy / floating-point ranges: every coefficient q-1, alternating 0 / q-1,
    and q-1 against random (forward, inverse, fused multiply, tensor product)."""
    moduli = nm.ntt_primes(bits, n, L)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    top = np.zeros((1, L, n, 4), np.uint64); alt = top.copy()
    for l, q in enumerate(moduli):
        top[0, l, :, 0] = q - 1
        alt[0, l, ::2, 0] = q - 1
    rnd = rns_poly(41, moduli, n, 1)
    dR = eng.DeviceBuffer(top.nbytes)
    for x in (top, alt):
        d = _up(eng, x); e.forward(d, 1)
        assert np.array_equal(d.download(top.shape), rp.forward(x, threads=8))
        d = _up(eng, x); e.inverse(d, 1)
        assert np.array_equal(d.download(top.shape), rp.inverse(x, threads=8))
    for x, y in ((top, top), (top, alt), (alt, rnd), (top, rnd)):
        e.multiply(dR, _up(eng, x), _up(eng, y), 1)
        assert np.array_equal(dR.download(top.shape), rp.polymul(x, y, threads=8))
    c = [eng.DeviceBuffer(top.nbytes) for _ in range(3)]
    e.ct_multiply(c[0], c[1], c[2], _up(eng, top), _up(eng, alt), _up(eng, top), _up(eng, rnd), 1)
    w = rp.ct_multiply(top, alt, top, rnd, threads=8)
    for got, want in zip(c, w):
        assert np.array_equal(got.download(top.shape), want)


# ------------------------------------------------------------------------------------ edge cases
def test_edge_polynomials(eng, oracle):
    n = 8192; moduli = nm.ntt_primes(30, n, 2); L = 2
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    zero = np.zeros((1, L, n, 4), np.uint64)
    one = zero.copy(); one[:, :, 0, 0] = 1
    top = zero.copy()
    for l, q in enumerate(moduli):
        top[0, l, :, 0] = q - 1
    xn1 = zero.copy(); xn1[:, :, n - 1, 0] = 1             # x^(n-1)
    r = rns_poly(9, moduli, n, 1)
    dR = eng.DeviceBuffer(zero.nbytes)
    for a, b in ((zero, r), (one, r), (top, top), (xn1, xn1), (xn1, r)):
        e.multiply(dR, _up(eng, a), _up(eng, b), 1)
        assert np.array_equal(dR.download(zero.shape), rp.polymul(a, b))
    # x^(n-1) * x^(n-1) = x^(2n-2) = -x^(n-2)
    got = dR  # last result is xn1 * r; recompute the monomial square explicitly
    e.multiply(dR, _up(eng, xn1), _up(eng, xn1), 1)
    res = dR.download(zero.shape)
    for l, q in enumerate(moduli):
        assert int(res[0, l, n - 2, 0]) == q - 1 and int(res[0, l].sum()) == q - 1


def test_noncanonical_input_is_detected(eng):
    n = 2048; moduli = [40961]
    e = eng.RnsNttEngine(n, moduli)
    x = np.zeros((1, 1, n, 4), np.uint64); x[0, 0, 5, 0] = 40961
    with pytest.raises(eng.FheError) as ei:
        e.check_canonical(_up(eng, x), 1)
    assert ei.value.code == -6
    x[0, 0, 5, 0] = 3; x[0, 0, 7, 2] = 1
    with pytest.raises(eng.FheError):
        e.check_canonical(_up(eng, x), 1)
    x[0, 0, 7, 2] = 0
    e.check_canonical(_up(eng, x), 1)


def test_error_reporting(eng):
    with pytest.raises(eng.FheError) as ei:
        eng.RnsNttEngine(8192, [12289])            # 12289 != 1 mod 16384
    assert ei.value.code == -2
    with pytest.raises(eng.FheError) as ei:
        eng.RnsNttEngine(1000, [12289])
    assert ei.value.code == -1
    e = eng.RnsNttEngine(2048, [40961])
    buf = eng.DeviceBuffer(2048 * 32)
    with pytest.raises(eng.FheError):
        e.ct_multiply(buf, buf, buf, buf, buf, buf, buf, 1)   # aliased tensor-product buffers
    with pytest.raises(eng.FheError):
        e.forward(buf, 0)                          # empty batch


# ------------------------------------------------------------------------------------ full-size properties
def test_full_size_properties_config2(eng, oracle):
    """BASELINE config 2 at bench scale (N = 8192, 4 limbs, batch 256): size-independent properties +
    oracle spot checks on sampled polynomials."""
    n, L, batch = 8192, 4, 256
    moduli = nm.ntt_primes(30, n, L)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    a = rns_poly(21, moduli, n, batch); b = rns_poly(22, moduli, n, batch)
    dA, dB = _up(eng, a), _up(eng, b)
    dR, dS = eng.DeviceBuffer(a.nbytes), eng.DeviceBuffer(a.nbytes)
    # round trip
    e.forward(dA, batch); e.inverse(dA, batch)
    assert np.array_equal(dA.download(a.shape), a)
    # commutativity and distributivity: a*(b+a) == a*b + a*a
    e.multiply(dR, dA, dB, batch); ab = dR.download(a.shape)
    e.multiply(dR, dB, dA, batch); assert np.array_equal(dR.download(a.shape), ab)
    e.poly_add(dS, dA, dB, batch); e.multiply(dR, dA, dS, batch); lhs = dR.download(a.shape)
    e.multiply(dS, dA, dA, batch); dAB = _up(eng, ab); e.poly_add(dR, dAB, dS, batch)
    assert np.array_equal(dR.download(a.shape), lhs)
    # fused multiply == forward, pointwise, inverse
    dFa, dFb = _up(eng, a), _up(eng, b)
    e.forward(dFa, batch); e.forward(dFb, batch); e.pointwise(dR, dFa, dFb, batch); e.inverse(dR, batch)
    assert np.array_equal(dR.download(a.shape), ab)
    # oracle on a sample of the batch
    for bi in (0, 97, 255):
        want = rp.polymul(np.ascontiguousarray(a[bi:bi + 1]), np.ascontiguousarray(b[bi:bi + 1]), threads=8)
        assert np.array_equal(ab[bi:bi + 1], want)
    e.check_canonical(dR, batch)


def test_golden_digests_on_device(eng, oracle, golden_dir):
    """Committed digests (tests/golden/l2_digests.json, made by tests/golden/make_golden.py from the oracle)."""
    import hashlib, json
    with open(os.path.join(golden_dir, "l2_digests.json")) as f:
        G = json.load(f)
    for c in G["cases"]:
        n, moduli, batch = c["n"], [int(q) for q in c["moduli"]], c["batch"]
        e = eng.RnsNttEngine(n, moduli)
        a = rns_poly(c["seed_a"], moduli, n, batch); b = rns_poly(c["seed_b"], moduli, n, batch)
        dA, dB, dR = _up(eng, a), _up(eng, b), eng.DeviceBuffer(a.nbytes)
        e.multiply(dR, dA, dB, batch)
        assert hashlib.sha256(dR.download(a.shape).tobytes()).hexdigest() == c["polymul_sha256"]
        e.forward(dA, batch)
        fa = dA.download(a.shape)
        assert hashlib.sha256(fa.tobytes()).hexdigest() == c["forward_sha256"]
        assert [int(v) for v in fa[0, 0, :8, 0]] == c["forward_first8"]


# ------------------------------------------------------------------------------------ N1: relinearisation
def _random_keys(moduli, n, count, seed):
    return [rns_poly(seed + 17 * i, moduli, n, 1)[0] for i in range(count)]


@pytest.mark.parametrize("n,spec,w,batch", [(2048, [40961], 8, 2), (8192, ("bits", 30, 4), 16, 3), (4096, ("bits", 30, 2), 30, 2),
                                            (4096, ("bits", 40, 3), 20, 2), (2048, ("bits", 60, 2), 32, 1), (256, ("bits", 250, 2), 64, 2),
                                            (1024, [12289], 16, 2), (2048, ("bits", 64, 2), 32, 2), (4096, ("bits", 63, 1), 16, 1),
                                            (2048, ("mix", (64, 1), (62, 1)), 64, 1), (65536, ("bits", 30, 2), 16, 1), (32768, ("bits", 40, 1), 20, 2)])
@pytest.mark.parametrize("single", [False, True])
def test_relinearize_matches_oracle(eng, oracle, monkeypatch, n, spec, w, batch, single):
    """FHEContext::relinearize semantics on every width class, arbitrary key material; digit transforms two at a
    time (default where available) and one at a time."""
    if single:
        monkeypatch.setenv("FHE_HIP_NO_PAIRED_TRANSFORMS", "1")
    moduli = _moduli(spec, n); L = len(moduli)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    K = e.relin_num_digits(w)
    assert K == rp.num_digits(w)
    kb = _random_keys(moduli, n, L * K, 100); ka = _random_keys(moduli, n, L * K, 900)
    c0, c1, c2 = (rns_poly(s, moduli, n, batch) for s in (51, 52, 53))
    dkb = [_up(eng, k) for k in kb]; dka = [_up(eng, k) for k in ka]
    rk = e.import_relin_keys(w, dkb, dka)
    d0, d1, d2 = _up(eng, c0), _up(eng, c1), _up(eng, c2)
    e.relinearize(rk, d0, d1, d2, batch)
    w0, w1 = rp.relinearize(w, c0, c1, c2, kb, ka, threads=8)
    assert np.array_equal(d0.download(c0.shape), w0)
    assert np.array_equal(d1.download(c0.shape), w1)
    assert np.array_equal(d2.download(c0.shape), c2)
    with pytest.raises(eng.FheError):
        e.import_relin_keys(w, dkb[:-1], dka[:-1])


@pytest.mark.parametrize("n,spec,w,batch", [(16384, ("bits", 40, 3), 16, 3), (8192, ("bits", 40, 3), 20, 9), (16384, ("bits", 60, 2), 32, 2),
                                            (4096, ("bits", 64, 2), 32, 5), (8192, ("bits", 62, 1), 16, 2),
                                            (8192, ("bits", 30, 4), 16, 5), (16384, ("bits", 30, 3), 30, 2), (2048, ("bits", 30, 5), 8, 11)])
@pytest.mark.parametrize("compaction", [True, False, "pipeline", "one-launch", "one-launch-containers"])
def test_relinearize_with_and_without_c2_compaction(eng, oracle, monkeypatch, n, spec, w, batch, compaction):
    """Stand-alone relinearisation: c2 is compacted once (compact_kernel) and the key-switch kernel (three-array on the 8-byte fields, paired on the
    4-byte one) re-reads the compact copy; FHE_HIP_NO_C2_COMPACTION=1 keeps the container reads of round 2; FHE_HIP_RELIN_PIPELINE=1 runs compaction
    and key switch as a two-stream pipeline where the batch allows.  All against the oracle; c2 is left untouched."""
    if compaction == "pipeline":
        monkeypatch.setenv("FHE_HIP_RELIN_PIPELINE", "1")
    elif compaction in ("one-launch", "one-launch-containers"):   # the throughput form of the paired key switch (few ciphertexts split its digit pairs by default)
        monkeypatch.setenv("FHE_HIP_SPLIT_PAIRS_POLYS", "0")
        if compaction == "one-launch-containers":
            monkeypatch.setenv("FHE_HIP_NO_C2_COMPACTION", "1")
    elif not compaction:
        monkeypatch.setenv("FHE_HIP_NO_C2_COMPACTION", "1")
    moduli = _moduli(spec, n); L = len(moduli)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    K = e.relin_num_digits(w)
    kb = _random_keys(moduli, n, L * K, 300); ka = _random_keys(moduli, n, L * K, 700)
    c0, c1, c2 = (rns_poly(s, moduli, n, batch) for s in (61, 62, 63))
    rk = e.import_relin_keys(w, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka])
    d0, d1, d2 = _up(eng, c0), _up(eng, c1), _up(eng, c2)
    e.relinearize(rk, d0, d1, d2, batch)
    w0, w1 = rp.relinearize(w, c0, c1, c2, kb, ka, threads=8)
    assert np.array_equal(d0.download(c0.shape), w0) and np.array_equal(d1.download(c0.shape), w1)
    assert np.array_equal(d2.download(c0.shape), c2)
    # a second call with a smaller batch re-uses the (larger) workspace
    d0, d1 = _up(eng, c0), _up(eng, c1)
    e.relinearize(rk, d0, d1, d2, 1)
    assert np.array_equal(d0.download(c0.shape)[:1], w0[:1]) and np.array_equal(d0.download(c0.shape)[1:], c0[1:])


@pytest.mark.parametrize("bits", [30, 40, 60, 64])
def test_relinearize_general_path_on_word_sized_moduli(eng, oracle, monkeypatch, bits):
    """The unfused composition (digit embedding, batched NTT, MAC) must agree with the fused key-switch kernels."""
    monkeypatch.setenv("FHE_HIP_NO_FUSED_KEYSWITCH", "1")
    n, w, batch = 4096, 16, 2
    moduli = nm.ntt_primes(bits, n, 3); L = 3
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    K = e.relin_num_digits(w)
    kb = _random_keys(moduli, n, L * K, 300); ka = _random_keys(moduli, n, L * K, 700)
    c0, c1, c2 = (rns_poly(s, moduli, n, batch) for s in (61, 62, 63))
    rk = e.import_relin_keys(w, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka])
    d0, d1, d2 = _up(eng, c0), _up(eng, c1), _up(eng, c2)
    e.relinearize(rk, d0, d1, d2, batch)
    w0, w1 = rp.relinearize(w, c0, c1, c2, kb, ka, threads=8)
    assert np.array_equal(d0.download(c0.shape), w0) and np.array_equal(d1.download(c0.shape), w1)


def test_reference_fhe_scenario_on_gpu(eng, oracle):
    """tests/test_fhe.cu:169-273 at its own size (N = 4096, t = 65537) on the multiply path: encrypt on the host (toy BGV,
    test code), tensor product + relinearisation on the GPU, decrypt on the host: 15 60 135 240 and 8 16 24 32."""
    import bgv_toy
    n, t, w = 4096, 65537, 16
    moduli = eng.find_ntt_primes(30, n, 4)                      # log_q = 120
    rp = oracle.RnsPlan(n, moduli)

    def fast_mul(x, y):
        return bgv_toy.from_limb_array(rp.polymul(bgv_toy.to_limb_array(x), bgv_toy.to_limb_array(y), threads=8))

    S = bgv_toy.ToyBGV(n, moduli, t, seed=11, fast_mul=fast_mul)
    m1 = S.slot_encode([5, 10, 15, 20]); m2 = S.slot_encode([3, 6, 9, 12])
    a0, a1 = S.encrypt(m1); b0, b1 = S.encrypt(m2)
    kb, ka, K = S.relin_keys(w)
    e = eng.RnsNttEngine(n, moduli)
    d = [_up(eng, bgv_toy.to_limb_array(x)) for x in (a0, a1, b0, b1)]
    c = [eng.DeviceBuffer(d[0].nbytes) for _ in range(3)]
    rk = e.import_relin_keys(w, [_up(eng, bgv_toy.to_limb_array(k)[0]) for k in kb], [_up(eng, bgv_toy.to_limb_array(k)[0]) for k in ka])
    e.ct_multiply(c[0], c[1], c[2], d[0], d[1], d[2], d[3], 1)
    shape = (1, 4, n, 4)
    three = [bgv_toy.from_limb_array(x.download(shape)) for x in c]
    assert S.slot_decode(S.decrypt(three))[:4] == [15, 60, 135, 240]
    e.relinearize(rk, c[0], c[1], c[2], 1)
    two = [bgv_toy.from_limb_array(x.download(shape)) for x in c[:2]]
    got = S.slot_decode(S.decrypt(two))
    assert got[:4] == [15, 60, 135, 240] and not any(got[4:])
    s0, s1 = eng.DeviceBuffer(d[0].nbytes), eng.DeviceBuffer(d[0].nbytes)
    e.poly_add(s0, d[0], d[2], 1); e.poly_add(s1, d[1], d[3], 1)
    summed = [bgv_toy.from_limb_array(x.download(shape)) for x in (s0, s1)]
    assert S.slot_decode(S.decrypt(summed))[:4] == [8, 16, 24, 32]


# ------------------------------------------------------------------------------------ robustness
def test_small_transform_sizes_and_many_limbs(eng, oracle):
    """n from 8 upward (general path below 2^11) and an RNS basis of 12 limbs."""
    for n in (8, 16, 128, 512):
        q = nm.ntt_primes(20, n, 1)[0]
        e = eng.NttEngine(n, q); p = oracle.Plan(n, q)
        a = rns_poly(71, [q], n, 3)[:, 0]; b = rns_poly(72, [q], n, 3)[:, 0]
        dA, dB, dR = _up(eng, a), _up(eng, b), eng.DeviceBuffer(a.nbytes)
        e.multiply(dR, dA, dB, 3)
        got = dR.download(a.shape)
        for i in range(3):
            assert np.array_equal(got[i], p.polymul(np.ascontiguousarray(a[i]), np.ascontiguousarray(b[i])))
    n, L = 2048, 12
    moduli = nm.ntt_primes(30, n, L)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    a = rns_poly(73, moduli, n, 2); b = rns_poly(74, moduli, n, 2)
    dR = eng.DeviceBuffer(a.nbytes)
    e.multiply(dR, _up(eng, a), _up(eng, b), 2)
    assert np.array_equal(dR.download(a.shape), rp.polymul(a, b, threads=8))


def test_mixed_width_basis_uses_the_widest_class(eng, oracle):
    """A basis mixing a 30-bit and a 40-bit prime runs on the FP64 path; 30-bit + 60-bit on the 64-bit integer path."""
    n = 4096
    for moduli, width in ((nm.ntt_primes(30, n, 1) + nm.ntt_primes(40, n, 1), 3), (nm.ntt_primes(30, n, 1) + nm.ntt_primes(60, n, 1), 2)):
        e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
        assert e.width_class == width
        a = rns_poly(75, moduli, n, 2); b = rns_poly(76, moduli, n, 2)
        dR = eng.DeviceBuffer(a.nbytes)
        e.multiply(dR, _up(eng, a), _up(eng, b), 2)
        assert np.array_equal(dR.download(a.shape), rp.polymul(a, b, threads=8))


def test_general_path_batches_beyond_one_grid_dimension(eng, oracle):
    """The full-width path splits launches at 65535 polynomials (grid.y limit): 70000 polynomials of n = 8."""
    n = 8; q = nm.ntt_primes(70, n, 1)[0]; batch = 70000
    e = eng.NttEngine(n, q); p = oracle.Plan(n, q)
    assert e.width_class == 4
    a = rns_poly(77, [q], n, batch)[:, 0]
    d = _up(eng, a); e.forward(d, batch)
    got = d.download(a.shape)
    for i in (0, 65534, 65535, 65536, 69999):
        assert np.array_equal(got[i], p.forward(np.ascontiguousarray(a[i])))
    e.inverse(d, batch)
    assert np.array_equal(d.download(a.shape), a)


def test_two_engines_and_caller_stream_interleave(eng, oracle):
    """Two engines (different moduli) used alternately on their own streams, results downloaded after one sync."""
    n = 8192
    m1, m2 = nm.ntt_primes(30, n, 2), nm.ntt_primes(29, n, 2)
    e1, e2 = eng.RnsNttEngine(n, m1), eng.RnsNttEngine(n, m2)
    r1, r2 = oracle.RnsPlan(n, m1), oracle.RnsPlan(n, m2)
    a1, b1 = rns_poly(81, m1, n, 4), rns_poly(82, m1, n, 4)
    a2, b2 = rns_poly(83, m2, n, 4), rns_poly(84, m2, n, 4)
    d = [_up(eng, x) for x in (a1, b1, a2, b2)]
    o1, o2 = eng.DeviceBuffer(a1.nbytes), eng.DeviceBuffer(a2.nbytes)
    for _ in range(3):
        e1.multiply(o1, d[0], d[1], 4); e2.multiply(o2, d[2], d[3], 4)
    eng.capi.sync()
    assert np.array_equal(o1.download(a1.shape), r1.polymul(a1, b1, threads=8))
    assert np.array_equal(o2.download(a2.shape), r2.polymul(a2, b2, threads=8))


def test_repeated_calls_are_deterministic(eng):
    n = 8192; moduli = nm.ntt_primes(30, n, 4)
    e = eng.RnsNttEngine(n, moduli)
    a = rns_poly(91, moduli, n, 64); b = rns_poly(92, moduli, n, 64)
    dA, dB, dR = _up(eng, a), _up(eng, b), eng.DeviceBuffer(a.nbytes)
    e.multiply(dR, dA, dB, 64); first = dR.download(a.shape).copy()
    for _ in range(5):
        dR.zero(); e.multiply(dR, dA, dB, 64)
        assert np.array_equal(dR.download(a.shape), first)


# ------------------------------------------------------------------------------------ RNS entry / exit (row a18)
@pytest.mark.parametrize("n,bits,L", [(8192, 30, 4), (2048, 60, 4), (64, 120, 2), (1024, 30, 8), (256, 250, 1), (4096, 40, 6), (2048, 62, 3), (2048, 64, 3)])
@pytest.mark.parametrize("word", [True, False])
def test_to_rns_from_rns_match_oracle(eng, oracle, monkeypatch, n, bits, L, word):
    """word = True: to_rns on the integer word classes runs the streaming kernel on the field type; False: the 256-bit container kernel."""
    import random as _r
    if not word:
        monkeypatch.setenv("FHE_HIP_NO_WORD_CONVERSIONS", "1")
    moduli = nm.ntt_primes(bits, n, L)
    Q = 1
    for q in moduli:
        Q *= q
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    rng = _r.Random(n + L); batch = 2
    vals = [rng.randrange(Q) for _ in range(batch * n - 3)] + [0, Q - 1, 1]
    V = oracle.to_limbs(vals).reshape(batch, n, 4)
    dV = _up(eng, V); dR = eng.DeviceBuffer(batch * L * n * 32)
    e.to_rns(dR, dV, batch)
    R = dR.download((batch, L, n, 4))
    assert np.array_equal(R, rp.to_rns(V))
    dBack = eng.DeviceBuffer(V.nbytes)
    e.from_rns(dBack, dR, batch)
    assert np.array_equal(dBack.download(V.shape), V) and np.array_equal(dBack.download(V.shape), rp.from_rns(R))
    # arbitrary 256-bit inputs wrap modulo each prime
    big = oracle.to_limbs([rng.getrandbits(256) for _ in range(batch * n)]).reshape(batch, n, 4)
    e.to_rns(dR, _up(eng, big), batch)
    assert np.array_equal(dR.download((batch, L, n, 4)), rp.to_rns(big))
    # CRT of a product computed limb-wise equals the product of the integers reduced mod (x^n + 1, Q): spot-check coefficient 0
    a = rns_poly(95, moduli, n, 1); b = rns_poly(96, moduli, n, 1)
    dP = eng.DeviceBuffer(a.nbytes); e.multiply(dP, _up(eng, a), _up(eng, b), 1)
    dC = eng.DeviceBuffer(n * 32); e.from_rns(dC, dP, 1)
    assert np.array_equal(dC.download((1, n, 4)), rp.from_rns(rp.polymul(a, b, threads=8)))


def test_from_rns_rejects_oversized_basis(eng):
    n = 2048; e = eng.RnsNttEngine(n, nm.ntt_primes(60, n, 5))
    buf = eng.DeviceBuffer(5 * n * 32); out = eng.DeviceBuffer(n * 32)
    with pytest.raises(eng.FheError) as ei:
        e.from_rns(out, buf, 1)
    assert ei.value.code == -5
    e.to_rns(buf, out, 1)          # to_rns has no such limit


@pytest.mark.parametrize("n,bits,L", [(8192, 30, 2), (4096, 40, 2), (2048, 60, 1), (256, 250, 1), (2048, 64, 2), (2048, 250, 1), (8192, 100, 1), (65536, 30, 1), (32768, 62, 1)])
def test_in_place_multiply_and_squaring(eng, oracle, n, bits, L):
    """Like the reference (which copies its operands, src/ntt.cu:50-58) the result may alias an operand."""
    moduli = nm.ntt_primes(bits, n, L)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    a = rns_poly(97, moduli, n, 3); b = rns_poly(98, moduli, n, 3)
    want = rp.polymul(a, b, threads=8); sq = rp.polymul(a, a, threads=8)
    dA, dB = _up(eng, a), _up(eng, b)
    e.multiply(dA, dA, dB, 3)                       # r aliases a
    assert np.array_equal(dA.download(a.shape), want) and np.array_equal(dB.download(a.shape), b)
    dA = _up(eng, a); e.multiply(dB, dA, dB, 3)     # r aliases b
    assert np.array_equal(dB.download(a.shape), want)
    dA = _up(eng, a); e.multiply(dA, dA, dA, 3)     # in-place square
    assert np.array_equal(dA.download(a.shape), sq)


@pytest.mark.parametrize("n,bits,L,batch", [(2048, 30, 2, 19), (2048, 30, 3, 40), (2048, 40, 2, 21), (2048, 60, 1, 17), (2048, 64, 1, 17)])
def test_relinearize_xcd_mapped_batches(eng, oracle, n, bits, L, batch):
    """Batches large enough that the key-switch kernel's XCD-aware block -> (ciphertext, limb) map is in effect for most
    workgroups and the identity map for the tail (batch not a multiple of 8)."""
    moduli = nm.ntt_primes(bits, n, L); w = 16
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    K = e.relin_num_digits(w)
    kb = _random_keys(moduli, n, L * K, 1100); ka = _random_keys(moduli, n, L * K, 1900)
    c0, c1, c2 = (rns_poly(s, moduli, n, batch) for s in (151, 152, 153))
    rk = e.import_relin_keys(w, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka])
    d0, d1, d2 = _up(eng, c0), _up(eng, c1), _up(eng, c2)
    e.relinearize(rk, d0, d1, d2, batch)
    w0, w1 = rp.relinearize(w, c0, c1, c2, kb, ka, threads=8)
    assert np.array_equal(d0.download(c0.shape), w0) and np.array_equal(d1.download(c0.shape), w1)


def test_relinearize_mixed_prime_sizes(eng, oracle):
    """A basis mixing 20-bit and 30-bit primes with w = 30: digits of the wide limbs exceed the lazy range of the narrow
    limb, so the engine must take the general composition (which reduces each digit first)."""
    n, w, batch = 2048, 30, 3
    moduli = nm.ntt_primes(20, n, 1) + nm.ntt_primes(30, n, 2); L = 3
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    assert e.width_class == 1
    K = e.relin_num_digits(w)
    kb = _random_keys(moduli, n, L * K, 2100); ka = _random_keys(moduli, n, L * K, 2900)
    c0, c1, c2 = (rns_poly(s, moduli, n, batch) for s in (251, 252, 253))
    top = c2.copy()
    for l, q in enumerate(moduli):
        top[:, l, :, 0] = q - 1                      # largest digits
    rk = e.import_relin_keys(w, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka])
    for cc2 in (c2, top):
        d0, d1, d2 = _up(eng, c0), _up(eng, c1), _up(eng, cc2)
        e.relinearize(rk, d0, d1, d2, batch)
        w0, w1 = rp.relinearize(w, c0, c1, cc2, kb, ka, threads=8)
        assert np.array_equal(d0.download(c0.shape), w0) and np.array_equal(d1.download(c0.shape), w1)
    # and with w = 16 the fused kernel is in range again
    K = e.relin_num_digits(16)
    kb = _random_keys(moduli, n, L * K, 3100); ka = _random_keys(moduli, n, L * K, 3900)
    rk = e.import_relin_keys(16, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka])
    d0, d1, d2 = _up(eng, c0), _up(eng, c1), _up(eng, top)
    e.relinearize(rk, d0, d1, d2, batch)
    w0, w1 = rp.relinearize(16, c0, c1, top, kb, ka, threads=8)
    assert np.array_equal(d0.download(c0.shape), w0) and np.array_equal(d1.download(c0.shape), w1)


def test_randomized_configurations(eng, oracle):
    """Seeded sweep over (n, prime width, limbs, batch, operation) -- every width class, sizes on both sides of the LDS
    range, odd batches -- each compared bit-exactly with the oracle."""
    rng = random.Random(20260101)
    widths = [20, 25, 29, 30, 31, 36, 40, 43, 44, 50, 58, 62, 63, 64, 90, 128, 200, 250]
    for trial in range(28):
        log_n = rng.choice([3, 5, 8, 10, 11, 11, 12, 12, 13, 13, 14])
        n = 1 << log_n
        bits = rng.choice(widths)
        if bits < log_n + 3:
            bits = log_n + 4
        L = rng.choice([1, 1, 2, 3, 4, 5]) if bits < 100 else rng.choice([1, 2])
        batch = rng.choice([1, 2, 3, 5, 7])
        if n * L * batch * (4 if bits > 64 else 1) > 300000:
            batch = 1
        moduli = nm.ntt_primes(bits, n, L)
        e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
        a = rns_poly(1000 + trial, moduli, n, batch); b = rns_poly(2000 + trial, moduli, n, batch)
        dA, dB, dR = _up(eng, a), _up(eng, b), eng.DeviceBuffer(a.nbytes)
        tag = f"trial {trial}: n={n} bits={bits} L={L} batch={batch} width={e.width_class}"
        op = trial % 4
        if op == 0:
            e.multiply(dR, dA, dB, batch)
            assert np.array_equal(dR.download(a.shape), rp.polymul(a, b, threads=8)), tag
        elif op == 1:
            e.forward(dA, batch)
            assert np.array_equal(dA.download(a.shape), rp.forward(a, threads=8)), tag
            e.inverse(dA, batch)
            assert np.array_equal(dA.download(a.shape), a), tag
        elif op == 2:
            c = [eng.DeviceBuffer(a.nbytes) for _ in range(3)]
            e.ct_multiply(c[0], c[1], c[2], dA, dB, dB, dA, batch)
            w = rp.ct_multiply(a, b, b, a, threads=8)
            for got, want in zip(c, w):
                assert np.array_equal(got.download(a.shape), want), tag
        else:
            wbits = rng.choice([8, 16, 30, 64])
            K = e.relin_num_digits(wbits)
            kb = _random_keys(moduli, n, L * K, 5000 + trial); ka = _random_keys(moduli, n, L * K, 6000 + trial)
            rk = e.import_relin_keys(wbits, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka])
            c2 = rns_poly(3000 + trial, moduli, n, batch); d2 = _up(eng, c2)
            e.relinearize(rk, dA, dB, d2, batch)
            w0, w1 = rp.relinearize(wbits, a, b, c2, kb, ka, threads=8)
            assert np.array_equal(dA.download(a.shape), w0) and np.array_equal(dB.download(a.shape), w1), tag + f" w={wbits}"


def test_largest_lds_size_tensor_and_keyswitch(eng, oracle):
    """N = 2^15 (1024-thread workgroups): the tensor product runs as three launches and key switching in its split form."""
    n, L, w = 32768, 2, 16
    moduli = nm.ntt_primes(30, n, L)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    a0, a1, b0, b1 = (rns_poly(s, moduli, n, 1) for s in (301, 302, 303, 304))
    d = [_up(eng, x) for x in (a0, a1, b0, b1)]
    c = [eng.DeviceBuffer(a0.nbytes) for _ in range(3)]
    e.ct_multiply(c[0], c[1], c[2], d[0], d[1], d[2], d[3], 1)
    w0, w1, w2 = rp.ct_multiply(a0, a1, b0, b1, threads=8)
    for got, want in zip(c, (w0, w1, w2)):
        assert np.array_equal(got.download(a0.shape), want)
    K = e.relin_num_digits(w)
    kb = _random_keys(moduli, n, L * K, 4100); ka = _random_keys(moduli, n, L * K, 4900)
    rk = e.import_relin_keys(w, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka])
    e.relinearize(rk, c[0], c[1], c[2], 1)
    r0, r1 = rp.relinearize(w, w0, w1, w2, kb, ka, threads=8)
    assert np.array_equal(c[0].download(a0.shape), r0) and np.array_equal(c[1].download(a0.shape), r1)


@pytest.mark.parametrize("n,bits,L", [(8192, 30, 4), (2048, 60, 3), (64, 120, 2), (4096, 40, 6), (1024, 30, 2), (16384, 30, 6), (2048, 62, 2), (2048, 64, 3)])
@pytest.mark.parametrize("word", [True, False])
def test_rescale_drop_last_matches_oracle(eng, oracle, monkeypatch, n, bits, L, word):
    """Modulus switching by dropping the last prime (rounded division), then the result is usable by an engine on L-1 primes.
    word = True: the streaming kernels on the field type (word-sized classes); False: the container-level 256-bit kernels."""
    if not word:
        monkeypatch.setenv("FHE_HIP_NO_WORD_CONVERSIONS", "1")
    moduli = nm.ntt_primes(bits, n, L)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    batch = 3
    c = rns_poly(401, moduli, n, batch)
    for l, q in enumerate(moduli):                      # exercise both signs of the centred remainder and its extremes
        c[0, l, :4] = oracle.to_limbs([0, q - 1, q // 2, q // 2 + 1])
    dIn = _up(eng, c); dOut = eng.DeviceBuffer(batch * (L - 1) * n * 32)
    e.rescale_drop_last(dOut, dIn, batch)
    got = dOut.download((batch, L - 1, n, 4))
    assert np.array_equal(got, rp.rescale_drop_last(c))
    assert np.array_equal(dIn.download(c.shape), c)
    e2 = eng.RnsNttEngine(n, moduli[:-1]); e2.check_canonical(dOut, batch)
    with pytest.raises(eng.FheError):
        eng.RnsNttEngine(n, moduli[:1]).rescale_drop_last(dOut, dIn, 1)


@pytest.mark.parametrize("n,bits,L,bits2,Lp", [(8192, 30, 4, 30, 5), (2048, 30, 3, 60, 2), (1024, 60, 2, 40, 3), (64, 120, 2, 250, 1), (4096, 40, 3, 40, 2),
                                               (2048, 60, 2, 60, 3), (16384, 30, 6, 30, 2), (2048, 64, 2, 64, 3), (2048, 64, 2, 60, 2)])
@pytest.mark.parametrize("word", [True, False])
def test_fast_base_conversion_matches_oracle(eng, oracle, monkeypatch, n, bits, L, bits2, Lp, word):
    """word = True: same-class word-sized bases take the streaming kernel on the field type; mixed classes and word = False the 256-bit one."""
    if not word:
        monkeypatch.setenv("FHE_HIP_NO_WORD_CONVERSIONS", "1")
    src = nm.ntt_primes(bits, n, L)
    dst = [p for p in nm.ntt_primes(bits2, n, Lp + L) if p not in src][:Lp]
    e, t = eng.RnsNttEngine(n, src), eng.RnsNttEngine(n, dst)
    S, D = oracle.RnsPlan(n, src), oracle.RnsPlan(n, dst)
    batch = 2
    x = rns_poly(501, src, n, batch)
    for l, q in enumerate(src):
        x[0, l, :2] = oracle.to_limbs([0, q - 1])
    dX = _up(eng, x); dY = eng.DeviceBuffer(batch * Lp * n * 32)
    e.fast_base_convert(t, dY, dX, batch)
    assert np.array_equal(dY.download((batch, Lp, n, 4)), S.fast_base_convert(D, x))
    t.check_canonical(dY, batch)
    # converting to a second target re-derives the matrix
    e.fast_base_convert(e, eng.DeviceBuffer(x.nbytes), dX, batch)


# ------------------------------------------------------------------------------------ N3: blind-rotation inner loop
@pytest.mark.parametrize("n,spec,w,batch", [(8192, ("bits", 30, 4), 16, 5), (4096, ("bits", 40, 2), 20, 3), (2048, ("bits", 60, 1), 32, 2),
                                            (256, ("bits", 250, 1), 64, 2), (2048, ("bits", 64, 2), 32, 2)])
def test_blind_rotate_step_matches_oracle(eng, oracle, n, spec, w, batch):
    moduli = _moduli(spec, n); L = len(moduli)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    K = e.relin_num_digits(w)
    rows = [(_random_keys(moduli, n, L * K, 7000 + 100 * c), _random_keys(moduli, n, L * K, 8000 + 100 * c)) for c in range(2)]
    imported = [e.import_relin_keys(w, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka]) for kb, ka in rows]
    a0, a1 = rns_poly(601, moduli, n, batch), rns_poly(602, moduli, n, batch)
    shifts = np.array([0, 1, n - 1, n, 2 * n - 1, 12345 % (2 * n)][:batch] + [7] * max(0, batch - 6), dtype=np.uint32)[:batch]
    dSh = eng.DeviceBuffer.from_numpy(shifts)
    # the monomial kernel alone
    dT = eng.DeviceBuffer(a0.nbytes); dA0, dA1 = _up(eng, a0), _up(eng, a1)
    e.monomial_mul_sub(dT, dA0, dSh, batch)
    assert np.array_equal(dT.download(a0.shape), rp.monomial_mul_sub(a0, shifts))
    # two full steps
    dT1 = eng.DeviceBuffer(a0.nbytes)
    w0, w1 = a0, a1
    for _ in range(2):
        e.blind_rotate_step(imported[0], imported[1], dA0, dA1, dSh, dT, dT1, batch)
        w0, w1 = rp.blind_rotate_step(w, w0, w1, shifts, rows[0], rows[1], threads=8)
    assert np.array_equal(dA0.download(a0.shape), w0) and np.array_equal(dA1.download(a0.shape), w1)


@pytest.mark.parametrize("n,spec,w,batch,steps", [(8192, ("bits", 30, 4), 16, 9, 3), (2048, ("bits", 30, 2), 30, 17, 2), (16384, ("bits", 30, 3), 16, 2, 1), (16384, ("bits", 30, 2), 8, 3, 3),
                                                  (32768, ("bits", 30, 2), 16, 2, 2), (4096, ("bits", 40, 2), 20, 3, 3), (2048, ("bits", 60, 2), 32, 2, 2),
                                                  (256, ("bits", 250, 1), 64, 2, 2), (2048, ("bits", 64, 2), 32, 2, 3),
                                                  (16384, ("bits", 40, 2), 20, 2, 2), (16384, ("bits", 60, 1), 32, 1, 3)])      # three-array / split kernels
@pytest.mark.parametrize("fused", [True, False, "single", "containers", "split", "no-prerotation", "one-workgroup-per-limb"])
def test_blind_rotate_loop_matches_oracle(eng, oracle, monkeypatch, n, spec, w, batch, steps, fused):
    """fhe_blind_rotate: `steps` external products with a different RGSW row set and different shifts per step; the fused
    one-launch-per-step path (ping-pong buffers, odd and even step counts; digit transforms two at a time -- the default where
    that kernel exists -- and one at a time) and the general composition all equal the oracle."""
    if fused is False:
        monkeypatch.setenv("FHE_HIP_NO_FUSED_BLIND_ROTATE", "1")
    elif fused == "single":
        monkeypatch.setenv("FHE_HIP_NO_PAIRED_TRANSFORMS", "1")
    elif fused == "split":          # 8-byte residues at N = 2^14: two workgroups per limb instead of one with three live arrays
        monkeypatch.setenv("FHE_HIP_SPLIT_KEYSWITCH", "1")
    elif fused == "containers":     # fused steps, accumulators ping-pong through the caller's container buffers instead of the compact workspace
        monkeypatch.setenv("FHE_HIP_NO_COMPACT_BLIND_ROTATE", "1")
    elif fused == "no-prerotation":  # three-array kernels: the monomial factor applied per digit inside the kernel instead of once per step by monomial_compact_kernel
        monkeypatch.setenv("FHE_HIP_NO_PREROTATION", "1")
    elif fused == "one-workgroup-per-limb":   # 4-byte residues, N <= 2^13, few accumulators: the default is one workgroup per digit + one per component (three launches
        monkeypatch.setenv("FHE_HIP_SPLIT_PAIRS_POLYS", "0")   # per step); this keeps the paired one-launch kernel these shapes ran before
    moduli = _moduli(spec, n); L = len(moduli)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    K = e.relin_num_digits(w)
    rows = [[(_random_keys(moduli, n, L * K, 7000 + 100 * c + 1000 * s), _random_keys(moduli, n, L * K, 8000 + 100 * c + 1000 * s)) for c in range(2)]
            for s in range(steps)]
    imported = [[e.import_relin_keys(w, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka]) for kb, ka in r] for r in rows]
    a0, a1 = rns_poly(611, moduli, n, batch), rns_poly(612, moduli, n, batch)
    rng = np.random.default_rng(5)
    shifts = rng.integers(0, 2 * n, size=(steps, batch), dtype=np.uint32)
    shifts[0, :min(batch, 5)] = [0, 1, n - 1, n, 2 * n - 1][:min(batch, 5)]
    dSh = eng.DeviceBuffer.from_numpy(shifts)
    dA0, dA1 = _up(eng, a0), _up(eng, a1)
    dT0, dT1 = eng.DeviceBuffer(a0.nbytes), eng.DeviceBuffer(a0.nbytes)
    e.blind_rotate([r[0] for r in imported], [r[1] for r in imported], dA0, dA1, dSh, dT0, dT1, batch)
    w0, w1 = rp.blind_rotate(w, a0, a1, shifts, [r[0] for r in rows], [r[1] for r in rows], threads=8)
    assert np.array_equal(dA0.download(a0.shape), w0) and np.array_equal(dA1.download(a0.shape), w1)
    e.check_canonical(dA0, batch); e.check_canonical(dA1, batch)


def test_blind_rotate_rejects_bad_arguments(eng):
    n = 2048; moduli = eng.find_ntt_primes(30, n, 2)
    e = eng.RnsNttEngine(n, moduli); e2 = eng.RnsNttEngine(n, moduli)
    K = e.relin_num_digits(16)
    keys = [_up(eng, k) for k in _random_keys(moduli, n, 2 * K, 1)]
    r = e.import_relin_keys(16, keys, keys); r_other = e2.import_relin_keys(16, keys, keys)
    bufs = [eng.DeviceBuffer(2 * n * 32) for _ in range(4)]
    dSh = eng.DeviceBuffer.from_numpy(np.zeros(1, dtype=np.uint32))
    with pytest.raises(eng.FheError):
        e.blind_rotate_step(r, r_other, bufs[0], bufs[1], dSh, bufs[2], bufs[3], 1)      # rows of another engine
    with pytest.raises(eng.FheError):
        e.blind_rotate_step(r, r, bufs[0], bufs[1], dSh, bufs[0], bufs[3], 1)            # scratch aliases an accumulator
    e.blind_rotate([], [], bufs[0], bufs[1], dSh, bufs[2], bufs[3], 1)                   # zero steps: no-op


def test_blind_rotation_rotates_the_plaintext_on_gpu(eng, oracle):
    """Three blind-rotation steps with secret bits 1, 0, 1 (toy BGV on the host, N = 2048): the decrypted accumulator is
    X^(sum a_i s_i) * m."""
    import bgv_toy
    n, t, w = 2048, 12289, 16                                  # t = 1 (mod 2n) is not needed here: coefficient encoding
    moduli = eng.find_ntt_primes(30, n, 2)
    rp = oracle.RnsPlan(n, moduli)

    def fast_mul(x, y):
        return bgv_toy.from_limb_array(rp.polymul(bgv_toy.to_limb_array(x), bgv_toy.to_limb_array(y), threads=8))

    S = bgv_toy.ToyBGV(n, moduli, t, seed=21, fast_mul=fast_mul)
    rng = random.Random(8)
    m = [rng.randrange(t) for _ in range(n)]
    c0, c1 = S.encrypt(m)
    e = eng.RnsNttEngine(n, moduli)
    dA0, dA1 = _up(eng, bgv_toy.to_limb_array(c0)), _up(eng, bgv_toy.to_limb_array(c1))
    dT0, dT1 = eng.DeviceBuffer(dA0.nbytes), eng.DeviceBuffer(dA0.nbytes)
    total = 0
    for bit, a in ((1, 100), (0, 999), (1, n + 77)):
        r0, r1, K = S.rgsw(bit, w)
        imp = [e.import_relin_keys(w, [_up(eng, bgv_toy.to_limb_array(k)[0]) for k in r[0]], [_up(eng, bgv_toy.to_limb_array(k)[0]) for k in r[1]])
               for r in (r0, r1)]
        dSh = eng.DeviceBuffer.from_numpy(np.array([a], dtype=np.uint32))
        e.blind_rotate_step(imp[0], imp[1], dA0, dA1, dSh, dT0, dT1, 1)
        total += a * bit
    shape = (1, 2, n, 4)
    got = S.decrypt([bgv_toy.from_limb_array(dA0.download(shape)), bgv_toy.from_limb_array(dA1.download(shape))])
    k = total % (2 * n)
    want = [0] * n
    for i, mi in enumerate(m):                                  # X^k * m, negacyclic
        j = i + k
        sign = 1
        while j >= n:
            j -= n; sign = -sign
        want[j] = (sign * mi) % t
    assert got == want


# ------------------------------------------------------------------------------------ N4: samplers, modulus switch, fold
@pytest.mark.parametrize("q,seed,count", [(12289, 1804289383, 4096), (1 << 60, 846930886, 1000), ((1 << 64) - 59, (1 << 64) - 5, 333),
                                          ((1 << 200) + 12289, 99, 70000)])
def test_literal_samplers_match_oracle(eng, oracle, q, seed, count):
    """sample_uniform_kernel / sample_gaussian_kernel (src/polynomial.cu:113-143), literal, incl. the reference's own modulus 2^60
    (src/fhe.cu:13) and a multi-limb modulus (only limbs[0] is used)."""
    d = eng.DeviceBuffer(count * 32)
    eng.sample_uniform_lcg(d, q, seed, count)
    assert np.array_equal(d.download((count, 4)), oracle.sample_uniform_lcg(q, seed, count))
    eng.sample_gaussian_placeholder(d, q, seed, count)
    assert np.array_equal(d.download((count, 4)), oracle.sample_gaussian_placeholder(q, seed, count))
    with pytest.raises(eng.FheError):
        eng.sample_uniform_lcg(d, 1 << 64, seed, count)          # limbs[0] == 0: the reference would divide by zero


@pytest.mark.parametrize("n,spec,batch", [(8192, ("bits", 30, 4), 5), (4096, ("bits", 40, 3), 2), (2048, ("bits", 60, 2), 3), (256, ("bits", 250, 2), 2),
                                          (1024, [12289], 3), (2048, ("bits", 64, 2), 2)])
def test_rns_samplers_match_oracle(eng, oracle, n, spec, batch):
    moduli = _moduli(spec, n); L = len(moduli)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    d = eng.DeviceBuffer(batch * L * n * 32); shape = (batch, L, n, 4)
    for p in (0.5, 0.0, 1.0, 0.3):
        e.sample_ternary(d, p, 1234, batch)
        assert np.array_equal(d.download(shape), rp.sample_ternary(p, 1234, batch))
    e.check_canonical(d, batch)
    for sigma in (3.2, 0.8, 3.2, 19.0):                          # repeated sigma re-uses the cached table
        assert eng.gaussian_cdt(sigma) == oracle.gaussian_cdt(sigma)
        e.sample_gaussian(d, sigma, 99, batch)
        assert np.array_equal(d.download(shape), rp.sample_gaussian(sigma, 99, batch))
    e.check_canonical(d, batch)
    e.sample_uniform(d, 2024, batch)
    assert np.array_equal(d.download(shape), rp.sample_uniform(2024, batch))
    e.check_canonical(d, batch)
    with pytest.raises(eng.FheError):
        e.sample_ternary(d, 1.5, 1, batch)
    if min(moduli) < 2 ** 20:
        with pytest.raises(eng.FheError):
            e.sample_gaussian(d, 2000.0, 1, batch)                # 12 sigma does not fit below q


@pytest.mark.parametrize("old_q,new_q", [("p60", 65537), ((1 << 120) + 451, 257), ((1 << 254) + 79, (1 << 64) - 59), (1000003, 2), (12289, 12289)])
def test_poly_mod_switch_matches_oracle_and_big_integers(eng, oracle, old_q, new_q):
    if old_q == "p60":
        old_q = nm.ntt_primes(60, 4096, 1)[0]
    rng = random.Random(5)
    a = [0, 1, old_q - 1, old_q // 2, old_q // 2 + 1] + [rng.randrange(old_q) for _ in range(4091)]
    arr = oracle.to_limbs(a)
    dA = _up(eng, arr); dR = eng.DeviceBuffer(arr.nbytes)
    eng.poly_mod_switch(dR, dA, old_q, new_q, len(a))
    got = dR.download(arr.shape)
    assert np.array_equal(got, oracle.poly_mod_switch(arr, old_q, new_q))
    assert oracle.from_limbs(got)[:64] == [((x * new_q + old_q // 2) // old_q) % new_q for x in a[:64]]
    with pytest.raises(eng.FheError):
        eng.poly_mod_switch(dR, dA, old_q, 1 << 64, len(a))


def test_decrypt_scaling_through_poly_mod_switch(eng, oracle):
    """FHEContext::decrypt's last step (src/fhe.cu:181-184): m = round(t * x / q) mod t recovers m from x = delta * m + e."""
    q = nm.ntt_primes(60, 4096, 1)[0]; t = 65537; delta = q // t
    rng = random.Random(8)
    m = [rng.randrange(t) for _ in range(4096)]
    x = [(delta * mi + rng.randrange(-1000, 1000)) % q for mi in m]
    arr = oracle.to_limbs(x); dA = _up(eng, arr); dR = eng.DeviceBuffer(arr.nbytes)
    eng.poly_mod_switch(dR, dA, q, t, len(x))
    assert oracle.from_limbs(dR.download(arr.shape)) == m


def test_negacyclic_reduce_matches_oracle(eng, oracle):
    rng = random.Random(4)
    for n, q in [(1024, 12289), (4096, nm.ntt_primes(60, 4096, 1)[0]), (300, (1 << 254) + 79)]:
        d = oracle.to_limbs([rng.randrange(q) for _ in range(2 * n)])
        dD = _up(eng, d)
        eng.negacyclic_reduce(dD, q, n)
        assert np.array_equal(dD.download(d.shape), oracle.negacyclic_reduce(d, q))


# ------------------------------------------------------------------------------------ full-size properties, configs[3] / configs[4]
def test_full_size_properties_config4_ciphertext_multiply(eng, oracle):
    """BASELINE configs[3] per-GPU shape (N = 16384, 6 limbs, 128 ciphertexts): tensor product + relinearisation.  Size-independent
    properties (symmetry, a zero c2 leaves the pair untouched, every batch slot equals a batch-1 call) + oracle spot checks."""
    n, L, batch, w = 16384, 6, 128, 16
    moduli = nm.ntt_primes(30, n, L)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    K = e.relin_num_digits(w)
    kb, ka = _random_keys(moduli, n, L * K, 31000), _random_keys(moduli, n, L * K, 32000)
    rk = e.import_relin_keys(w, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka])
    a0, a1, b0, b1 = (rns_poly(900 + i, moduli, n, batch) for i in range(4))
    dA0, dA1, dB0, dB1 = (_up(eng, x) for x in (a0, a1, b0, b1))
    dC = [eng.DeviceBuffer(a0.nbytes) for _ in range(3)]
    e.ct_multiply(dC[0], dC[1], dC[2], dA0, dA1, dB0, dB1, batch)
    c = [d.download(a0.shape) for d in dC]
    dD = [eng.DeviceBuffer(a0.nbytes) for _ in range(3)]
    e.ct_multiply(dD[0], dD[1], dD[2], dB0, dB1, dA0, dA1, batch)                     # ct(a, b) == ct(b, a)
    for x, d in zip(c, dD):
        assert np.array_equal(d.download(a0.shape), x)
    e.relinearize(rk, dC[0], dC[1], dC[2], batch)
    r0, r1 = dC[0].download(a0.shape), dC[1].download(a0.shape)
    dE0, dE1 = eng.DeviceBuffer(a0.nbytes), eng.DeviceBuffer(a0.nbytes)
    e.ct_multiply_relin(rk, dE0, dE1, dA0, dA1, dB0, dB1, batch)                      # the one-call form (compact workspace) == the two calls
    assert np.array_equal(dE0.download(a0.shape), r0) and np.array_equal(dE1.download(a0.shape), r1)
    dZ = eng.DeviceBuffer(a0.nbytes); dZ.zero()
    e.relinearize(rk, dD[0], dD[1], dZ, batch)                                        # c2 = 0: nothing to switch
    assert np.array_equal(dD[0].download(a0.shape), c[0]) and np.array_equal(dD[1].download(a0.shape), c[1])
    for bi in (0, 77, 127):                                                           # batch slot == batch-1 call == oracle
        s0, s1, s2 = (_up(eng, np.ascontiguousarray(x[bi:bi + 1])) for x in c)
        e.relinearize(rk, s0, s1, s2, 1)
        assert np.array_equal(s0.download((1,) + a0.shape[1:]), r0[bi:bi + 1]) and np.array_equal(s1.download((1,) + a0.shape[1:]), r1[bi:bi + 1])
    bi = 41
    w0, w1, w2 = rp.ct_multiply(*(np.ascontiguousarray(x[bi:bi + 1]) for x in (a0, a1, b0, b1)), threads=8)
    assert all(np.array_equal(x[bi:bi + 1], y) for x, y in zip(c, (w0, w1, w2)))
    o0, o1 = rp.relinearize(w, w0, w1, w2, kb, ka, threads=8)
    assert np.array_equal(r0[bi:bi + 1], o0) and np.array_equal(r1[bi:bi + 1], o1)
    e.check_canonical(dC[0], batch); e.check_canonical(dC[1], batch)


def test_full_size_properties_config5_blind_rotation(eng, oracle):
    """BASELINE configs[4] per-GPU shape (N = 16384, 6 limbs, 128 accumulators, 4 steps): a zero shift is the identity
    (X^0 - 1 = 0), every batch slot equals a batch-1 call, oracle spot check on one accumulator."""
    n, L, batch, w, steps = 16384, 6, 128, 16, 4
    moduli = nm.ntt_primes(30, n, L)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    K = e.relin_num_digits(w)
    rows = [[(_random_keys(moduli, n, L * K, 41000 + 100 * c + 1000 * s), _random_keys(moduli, n, L * K, 42000 + 100 * c + 1000 * s)) for c in range(2)]
            for s in range(steps)]
    imported = [[e.import_relin_keys(w, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka]) for kb, ka in r] for r in rows]
    r0s, r1s = [r[0] for r in imported], [r[1] for r in imported]
    a0, a1 = rns_poly(951, moduli, n, batch), rns_poly(952, moduli, n, batch)
    shifts = np.random.default_rng(9).integers(0, 2 * n, size=(steps, batch), dtype=np.uint32)
    shifts[:, 5] = 0                                                                   # accumulator 5 never rotates
    dSh = eng.DeviceBuffer.from_numpy(shifts)
    dA0, dA1 = _up(eng, a0), _up(eng, a1)
    dT0, dT1 = eng.DeviceBuffer(a0.nbytes), eng.DeviceBuffer(a0.nbytes)
    e.blind_rotate(r0s, r1s, dA0, dA1, dSh, dT0, dT1, batch)
    g0, g1 = dA0.download(a0.shape), dA1.download(a0.shape)
    assert np.array_equal(g0[5], a0[5]) and np.array_equal(g1[5], a1[5])
    one = (1,) + a0.shape[1:]
    for bi in (0, 64, 127):
        s0, s1 = _up(eng, np.ascontiguousarray(a0[bi:bi + 1])), _up(eng, np.ascontiguousarray(a1[bi:bi + 1]))
        t0, t1 = eng.DeviceBuffer(s0.nbytes), eng.DeviceBuffer(s0.nbytes)
        dS1 = eng.DeviceBuffer.from_numpy(np.ascontiguousarray(shifts[:, bi:bi + 1]))
        e.blind_rotate(r0s, r1s, s0, s1, dS1, t0, t1, 1)
        assert np.array_equal(s0.download(one), g0[bi:bi + 1]) and np.array_equal(s1.download(one), g1[bi:bi + 1])
    bi = 99
    w0, w1 = rp.blind_rotate(w, np.ascontiguousarray(a0[bi:bi + 1]), np.ascontiguousarray(a1[bi:bi + 1]), shifts[:, bi:bi + 1],
                             [r[0] for r in rows], [r[1] for r in rows], threads=8)
    assert np.array_equal(g0[bi:bi + 1], w0) and np.array_equal(g1[bi:bi + 1], w1)
    e.check_canonical(dA0, batch); e.check_canonical(dA1, batch)


# ------------------------------------------------------------------------------------ L1: the reference's transform kernels as written
def test_reference_literal_kernels_reproduce_survey_kats_on_gpu(eng, oracle, golden_dir):
    """outputs of the reference's OWN kernel source (ntt_forward_optimized_kernel / ntt_inverse_optimized_kernel
    with the placeholder tables src/ntt.cu:86-97 builds, data x[i] = i + 1, q = 12289) -- reproduced on the GPU through the ABI."""
    import json
    kats = json.load(open(os.path.join(golden_dir, "reference_kats.json")))
    for c in kats["survey_appendix_b"]["literal_kernels_placeholder_tables"]:
        n, q = c["n"], int(c["q"])
        tw = oracle.ref_placeholder_table(n)
        x = oracle.to_limbs(range(1, n + 1))
        dX, dT = _up(eng, x), _up(eng, tw)
        inv0 = oracle.mont_inverse(q)
        eng.ref_forward_kernel_literal(dX, dT, q, inv0, n)
        f = dX.download(x.shape)
        assert oracle.from_limbs(f)[:8] == c["forward_first8"]
        assert np.array_equal(f, oracle.ref_forward_kernel(x, tw, q))
        eng.ref_inverse_kernel_literal(dX, dT, q, inv0, int(c["n_inv"]), n)
        i = dX.download(x.shape)
        assert oracle.from_limbs(i)[:8] == c["then_inverse_first8"]
        assert np.array_equal(i, oracle.ref_inverse_kernel(f, tw, q, int(c["n_inv"])))


@pytest.mark.parametrize("n,q", [(2, 12289), (8, 40961), (64, 12289), (1024, "p60"), (4096, "wide"), (256, 1 << 60)])
def test_reference_literal_kernels_match_oracle(eng, oracle, n, q):
    """Arbitrary tables and data (incl. unreduced values and the even modulus 2^60 the reference really passes): literal, batched."""
    if q == "p60":
        q = nm.ntt_primes(60, 8192, 1)[0]
    elif q == "wide":
        q = nm.ntt_primes(250, 4096, 1)[0]
    rng = random.Random(n)
    batch = 3
    data = oracle.to_limbs([rng.getrandbits(256) if i % 7 == 0 else rng.randrange(q) for i in range(batch * n)])
    tw = oracle.to_limbs([rng.randrange(q) for _ in range(n)])
    n_inv = rng.randrange(q)
    inv0 = oracle.mont_inverse(q)
    dD, dT = _up(eng, data), _up(eng, tw)
    eng.ref_forward_kernel_literal(dD, dT, q, inv0, n, batch)
    f = dD.download(data.shape)
    want_f = np.concatenate([oracle.ref_forward_kernel(np.ascontiguousarray(data[b * n:(b + 1) * n]), tw, q) for b in range(batch)])
    assert np.array_equal(f, want_f)
    eng.ref_inverse_kernel_literal(dD, dT, q, inv0, n_inv, n, batch)
    want_i = np.concatenate([oracle.ref_inverse_kernel(np.ascontiguousarray(want_f[b * n:(b + 1) * n]), tw, q, n_inv) for b in range(batch)])
    assert np.array_equal(dD.download(data.shape), want_i)


def test_reference_literal_kernels_with_real_tables_are_the_cyclic_dft(eng, oracle):
    """SURVEY D3/D4 on the device: with tw[k] = psi^k * R the reference's forward kernel is the cyclic DFT (omega = psi^2) of the
    bit-reversed input and its inverse kernel undoes it -- the reading of the reference under which it is a transform at all."""
    q, n = 12289, 64
    psi = nm.find_psi(n, q); Rm = nm.R % q
    tw = oracle.to_limbs([pow(psi, k, q) * Rm % q for k in range(n)])
    itw = oracle.to_limbs([pow(psi, -k, q) * Rm % q for k in range(n)])
    n_inv_m = pow(n, -1, q) * Rm % q
    rng = random.Random(1)
    x = [rng.randrange(q) for _ in range(n)]
    xb = [x[nm.bitrev(i, 6)] for i in range(n)]
    dX = _up(eng, oracle.to_limbs(xb)); inv0 = oracle.mont_inverse(q)
    eng.ref_forward_kernel_literal(dX, _up(eng, tw), q, inv0, n)
    om = psi * psi % q
    assert oracle.from_limbs(dX.download((n, 4))) == [sum(x[j] * pow(om, j * k, q) for j in range(n)) % q for k in range(n)]
    eng.ref_inverse_kernel_literal(dX, _up(eng, itw), q, inv0, n_inv_m, n)
    assert oracle.from_limbs(dX.download((n, 4))) in (xb, x)


def test_reference_stockham_stage_matches_oracle(eng, oracle):
    """ntt_stockham_kernel (kernels/ntt_kernels.cu:213-243): one out-of-place stage, every stage of n = 256, batch 3."""
    rng = random.Random(2)
    n, q, batch = 256, nm.ntt_primes(60, 4096, 1)[0], 3
    tw = oracle.to_limbs([rng.randrange(q) for _ in range(n)])
    x = oracle.to_limbs([rng.randrange(q) for _ in range(batch * n)])
    inv0 = oracle.mont_inverse(q)
    dT = _up(eng, tw); dA, dB = _up(eng, x), eng.DeviceBuffer(x.nbytes)
    want = x
    for stage in range(8):
        eng.ref_stockham_stage_literal(dB, dA, dT, q, inv0, n, stage, batch)
        want = np.concatenate([oracle.ref_stockham_stage(np.ascontiguousarray(want[b * n:(b + 1) * n]), tw, q, stage) for b in range(batch)])
        assert np.array_equal(dB.download(x.shape), want)
        dA, dB = dB, dA
    with pytest.raises(eng.FheError):
        eng.ref_stockham_stage_literal(dA, dA, dT, q, inv0, n, 0, batch)
    with pytest.raises(eng.FheError):
        eng.ref_stockham_stage_literal(dB, dA, dT, q, inv0, n, 8, batch)


def test_bit_reverse_permutation_and_natural_order_transform(eng, oracle):
    """bit_reverse_kernel's intent (kernels/ntt_kernels.cu:140-161) over log2(n) bits; after it the forward transform's values
    are in natural order: X[k] = sum_j x[j] psi^((2k+1) j)."""
    rng = random.Random(6)
    for n, batch in [(2, 1), (8, 3), (1024, 2), (65536, 1)]:
        x = oracle.to_limbs([rng.getrandbits(200) for _ in range(batch * n)])
        d = _up(eng, x)
        eng.bit_reverse(d, n, batch)
        got = d.download(x.shape).reshape(batch, n, 4)
        bits = n.bit_length() - 1
        perm = np.array([nm.bitrev(i, bits) for i in range(n)])
        assert np.array_equal(got, x.reshape(batch, n, 4)[:, perm])
        eng.bit_reverse(d, n, batch)
        assert np.array_equal(d.download(x.shape), x)                      # an involution
    n, q = 64, 12289
    e = eng.NttEngine(n, q)
    xs = [rng.randrange(q) for _ in range(n)]
    d = _up(eng, oracle.to_limbs(xs))
    e.forward(d); eng.bit_reverse(d, n); eng.capi.sync()
    psi = nm.find_psi(n, q)
    assert oracle.from_limbs(d.download((n, 4))) == [sum(xs[j] * pow(psi, (2 * k + 1) * j, q) for j in range(n)) % q for k in range(n)]


def test_small_and_degree_one_engines_stay_on_the_container_class(eng):
    """The word-sized conversion kernels store through lane pairs of whole waves (store_wave_containers): they require n to be a multiple
    of 256, which holds because word-sized classes exist from n = 2^11 only.  Pin that: small rings and degree-1 bases of word-sized
    primes take the 256-bit container class (and the host guards the kernels with log2 n >= 8 besides)."""
    for n in (8, 64, 256, 1024):
        assert eng.RnsNttEngine(n, _moduli(("bits", 30, 2), n)).width_class == eng.WIDTH_256
    assert eng.RnsNttEngine(None, [12289, 40961]).width_class == eng.WIDTH_256          # degree-1 base of small primes
    assert eng.RnsNttEngine(2048, _moduli(("bits", 30, 2), 2048)).width_class == eng.WIDTH_32


def test_rns_base_without_a_ring(eng, oracle):
    """fhe_rns_base_create (RNSContext, include/rns.cuh:27-66): interleaved [count][num_primes] buffers, arbitrary distinct odd
    primes; rns_add_kernel / rns_mul_kernel literal (src/rns.cu:143-181), conversions exact -- checked against Python integers."""
    primes = [12289, 40961, (1 << 61) - 1, (1 << 127) - 1]                 # no NTT condition; a 127-bit Mersenne prime among them
    e = eng.RnsNttEngine(None, primes)
    assert e.n == 1 and e.width_class == eng.WIDTH_256
    rng = random.Random(10)
    count, L = 777, len(primes)
    Q = 1
    for p in primes:
        Q *= p
    va = [rng.randrange(Q) for _ in range(count)]; vb = [rng.randrange(Q) for _ in range(count)]
    dA, dB = _up(eng, oracle.to_limbs(va)), _up(eng, oracle.to_limbs(vb))
    dRA, dRB, dR = (eng.DeviceBuffer(count * L * 32) for _ in range(3))
    e.to_rns(dRA, dA, count); e.to_rns(dRB, dB, count)
    ra = oracle.from_limbs(dRA.download((count, L, 4)))
    assert ra == [v % p for v in va for p in primes]                       # value-major, prime-minor: the reference's layout
    dV = eng.DeviceBuffer(count * 32)
    e.from_rns(dV, dRA, count)
    assert oracle.from_limbs(dV.download((count, 4))) == va
    e.poly_add(dR, dRA, dRB, count)
    assert oracle.from_limbs(dR.download((count, L, 4))) == [(x + y) % p for x, y in zip(va, vb) for p in primes]
    e.poly_sub(dR, dRA, dRB, count)
    assert oracle.from_limbs(dR.download((count, L, 4))) == [(x - y) % p for x, y in zip(va, vb) for p in primes]
    e.pointwise(dR, dRA, dRB, count)
    assert oracle.from_limbs(dR.download((count, L, 4))) == [(x % p) * (y % p) % p for x, y in zip(va, vb) for p in primes]
    e.mul_mont_literal(dR, dRA, dRB, count)                                # mul_mod_montgomery: carries 2^-256
    assert oracle.from_limbs(dR.download((count, L, 4))) == [nm.mont_mul_ref(x % p, y % p, p) for x, y in zip(va, vb) for p in primes]
    e.multiply(dR, dRA, dRB, count)                                        # the ring Z_q[x]/(x + 1) is Z_q: transforms are the identity
    assert oracle.from_limbs(dR.download((count, L, 4))) == [(x % p) * (y % p) % p for x, y in zip(va, vb) for p in primes]
    with pytest.raises(eng.FheError):
        eng.RnsNttEngine(None, [12289, 12289])                             # not pairwise distinct
    with pytest.raises(eng.FheError):
        eng.RnsNttEngine(None, [12289, 40963])                             # 40963 = 13 * 23 * 137
    narrow = eng.RnsNttEngine(8192, eng.find_ntt_primes(30, 8192, 2))
    with pytest.raises(eng.FheError):
        narrow.mul_mont_literal(dR, dRA, dRB, 1)                           # R = 2^256 products only exist on full-width handles


def test_base_conversion_cache_survives_a_recycled_target_handle(eng, oracle):
    """The conversion matrix is cached per target; a new target engine that happens to reuse a freed handle's address must not
    see the old matrix."""
    n = 2048
    src = nm.ntt_primes(30, n, 3)
    cands = [p for p in nm.ntt_primes(30, n, 12) if p not in src]
    e = eng.RnsNttEngine(n, src); S = oracle.RnsPlan(n, src)
    x = rns_poly(777, src, n, 1); dX = _up(eng, x); dY = eng.DeviceBuffer(2 * n * 32)
    for k in range(4):                                           # create / convert / destroy: allocators hand the same block out again
        dst = cands[2 * k:2 * k + 2]
        t = eng.RnsNttEngine(n, dst)
        e.fast_base_convert(t, dY, dX, 1)
        assert np.array_equal(dY.download((1, 2, n, 4)), S.fast_base_convert(oracle.RnsPlan(n, dst), x))
        del t


@pytest.mark.parametrize("n,spec,batch", [(8192, ("bits", 30, 4), 5), (4096, ("bits", 40, 2), 3), (2048, ("bits", 60, 2), 2), (16384, ("bits", 30, 2), 2),
                                          (16384, ("bits", 40, 2), 1), (256, ("bits", 250, 1), 2), (2048, ("bits", 64, 2), 2), (16384, ("bits", 64, 1), 1)])
@pytest.mark.parametrize("square_kernels", [True, False])
def test_squaring_forms_match_oracle(eng, oracle, monkeypatch, n, spec, batch, square_kernels):
    """multiply(a, a) and ct_multiply((a0, a1), (a0, a1)) take the squaring forms of the kernels (one load and one forward transform per
    operand) when the operand pointers are equal; same bits as the oracle's general product and as the general kernels."""
    if not square_kernels:
        monkeypatch.setenv("FHE_HIP_NO_SQUARE_KERNELS", "1")
    moduli = _moduli(spec, n)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    a0, a1 = rns_poly(1201, moduli, n, batch), rns_poly(1202, moduli, n, batch)
    dA0, dA1 = _up(eng, a0), _up(eng, a1)
    dR = eng.DeviceBuffer(a0.nbytes)
    e.multiply(dR, dA0, dA0, batch)
    assert np.array_equal(dR.download(a0.shape), rp.polymul(a0, a0, threads=8))
    dC = [eng.DeviceBuffer(a0.nbytes) for _ in range(3)]
    e.ct_multiply(dC[0], dC[1], dC[2], dA0, dA1, dA0, dA1, batch)
    want = rp.ct_multiply(a0, a1, a0, a1, threads=8)
    for d, w in zip(dC, want):
        assert np.array_equal(d.download(a0.shape), w)
    assert np.array_equal(dA0.download(a0.shape), a0) and np.array_equal(dA1.download(a0.shape), a1)      # operands preserved
    e.multiply(dA0, dA0, dA0, batch)                                                                         # in-place square
    assert np.array_equal(dA0.download(a0.shape), rp.polymul(a0, a0, threads=8))


@pytest.mark.parametrize("n,spec,w,batch", [(16384, ("bits", 40, 2), 16, 2), (16384, ("bits", 60, 1), 32, 1), (16384, ("bits", 64, 1), 32, 2), (32768, ("bits", 30, 1), 16, 1),
                                            (8192, ("bits", 40, 2), 20, 3), (4096, ("bits", 64, 2), 32, 2), (4096, ("bits", 43, 1), 16, 5),   # two-launch by choice (A/B)
                                            (2048, ("bits", 40, 3), 20, 4), (2048, ("bits", 60, 2), 32, 3)])
@pytest.mark.parametrize("forms", ["default", "no-two-launch", "split-keyswitch", "two-launch"])
def test_tensor_product_without_the_one_launch_kernel(eng, oracle, monkeypatch, n, spec, w, batch, forms):
    """Sizes whose four transformed operands do not fit the register file (8-byte residues at N = 2^14, N = 2^15): the tensor product runs
    as NTT(b0), NTT(b1) into a compact workspace + one launch for the rest (7 transforms), or with FHE_HIP_NO_TWO_LAUNCH_CT=1 as
    multiply + multiply + two-product kernel (11 transforms) -- or, at the smaller sizes of the 8-byte fields where the two-launch form is
    merely the faster choice, as the one-launch kernel; the key switch of the 8-byte residues at N = 2^14 as one workgroup per limb
    with three live arrays, or with FHE_HIP_SPLIT_KEYSWITCH=1 in the split form.  All equal the oracle, alone and inside
    fhe_ct_multiply_relin / fhe_ct_relinearize."""
    if forms == "no-two-launch":
        monkeypatch.setenv("FHE_HIP_NO_TWO_LAUNCH_CT", "1")
    if forms == "split-keyswitch":
        monkeypatch.setenv("FHE_HIP_SPLIT_KEYSWITCH", "1")
    if forms == "two-launch":           # also where the default keeps the one-launch kernel (lazy 64-bit field, full multiply of the full-range one)
        monkeypatch.setenv("FHE_HIP_CT_FORM", "two")
    moduli = _moduli(spec, n); L = len(moduli)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    a0, a1, b0, b1 = (rns_poly(s, moduli, n, batch) for s in (71, 72, 73, 74))
    d = [_up(eng, x) for x in (a0, a1, b0, b1)]
    c = [eng.DeviceBuffer(a0.nbytes) for _ in range(3)]
    for rep in range(2):                                         # the second call reuses the workspace
        e.ct_multiply(c[0], c[1], c[2], d[0], d[1], d[2], d[3], batch)
    t = rp.ct_multiply(a0, a1, b0, b1, threads=8)
    for buf, want in zip(c, t):
        assert np.array_equal(buf.download(a0.shape), want)
    K = e.relin_num_digits(w)
    kb = _random_keys(moduli, n, L * K, 1100); ka = _random_keys(moduli, n, L * K, 1900)
    rk = e.import_relin_keys(w, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka])
    e.ct_multiply_relin(rk, c[0], c[1], d[0], d[1], d[2], d[3], batch)
    w0, w1 = rp.relinearize(w, t[0], t[1], t[2], kb, ka, threads=8)
    assert np.array_equal(c[0].download(a0.shape), w0)
    assert np.array_equal(c[1].download(a0.shape), w1)
    for buf, src in zip(d, (a0, a1, b0, b1)):
        assert np.array_equal(buf.download(a0.shape), src)
    r = [_up(eng, x) for x in t]                                 # the stand-alone key switch (container operands, in place)
    e.relinearize(rk, r[0], r[1], r[2], batch)
    assert np.array_equal(r[0].download(a0.shape), w0)
    assert np.array_equal(r[1].download(a0.shape), w1)
    assert np.array_equal(r[2].download(a0.shape), t[2])


@pytest.mark.parametrize("n,spec,batch", [(8192, ("bits", 30, 4), 7), (4096, ("bits", 40, 2), 3), (2048, ("bits", 60, 2), 2), (256, ("bits", 250, 1), 3), (2048, ("bits", 64, 2), 2)])
def test_multiply_by_one_shared_polynomial(eng, oracle, n, spec, batch):
    """fhe_rns_ntt_multiply_bcast: every element of a batch times ONE polynomial (a key, a plaintext); equals the element-wise products."""
    moduli = _moduli(spec, n)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    a = rns_poly(1301, moduli, n, batch); b1 = rns_poly(1302, moduli, n, 1)
    dA, dB, dR = _up(eng, a), _up(eng, b1), eng.DeviceBuffer(a.nbytes)
    e.multiply_bcast(dR, dA, dB, batch)
    want = rp.polymul(a, np.ascontiguousarray(np.broadcast_to(b1, a.shape)), threads=8)
    assert np.array_equal(dR.download(a.shape), want)
    assert np.array_equal(dA.download(a.shape), a) and np.array_equal(dB.download(b1.shape), b1)
    e.multiply_bcast(dA, dA, dB, batch)                       # in place over the batch operand
    assert np.array_equal(dA.download(a.shape), want)
    with pytest.raises(eng.FheError):
        e.multiply_bcast(dB, dA, dB, 1)                        # the shared operand must survive


@pytest.mark.parametrize("bits,L", [(30, 2), (40, 2), (60, 1), (64, 1), (250, 1)])
def test_check_inputs_switch_rejects_noncanonical_operands(eng, monkeypatch, bits, L):
    """FHE_HIP_CHECK_INPUTS=1 (read at engine creation): the compute entry points scan their operands first and return
    FHE_ERR_NONCANONICAL for a value >= q or (word-sized classes) a non-zero upper word, instead of a silently different product."""
    n = 2048; moduli = nm.ntt_primes(bits, n, L)
    monkeypatch.setenv("FHE_HIP_CHECK_INPUTS", "1")
    e = eng.RnsNttEngine(n, moduli)
    a = rns_poly(5, moduli, n, 1); b = rns_poly(6, moduli, n, 1)
    dA, dB, dR = _up(eng, a), _up(eng, b), eng.DeviceBuffer(a.nbytes)
    e.multiply(dR, dA, dB, 1)                                    # clean operands pass
    bad = a.copy(); q = moduli[L - 1]
    for k in range(4):
        bad[0, L - 1, 7, k] = (q >> (64 * k)) & 0xFFFFFFFFFFFFFFFF      # coefficient == q
    cases = [bad]
    if bits <= 64:
        up = a.copy(); up[0, 0, 100, 2] = 1                     # non-zero upper word: invisible to the word-sized kernels
        cases.append(up)
    for arr in cases:
        dBad = _up(eng, arr)
        for call in (lambda: e.multiply(dR, dBad, dB, 1), lambda: e.multiply(dR, dA, dBad, 1), lambda: e.forward(dBad, 1),
                     lambda: e.ct_multiply(dR, eng.DeviceBuffer(a.nbytes), eng.DeviceBuffer(a.nbytes), dA, dBad, dB, dA, 1)):
            with pytest.raises(eng.FheError) as ei:
                call()
            assert ei.value.code == -6
    monkeypatch.delenv("FHE_HIP_CHECK_INPUTS")
    e2 = eng.RnsNttEngine(n, moduli)                             # without the switch the same call runs (garbage in, garbage out)
    e2.multiply(dR, _up(eng, cases[0]), dB, 1)


@pytest.mark.parametrize("n,spec,w,batch", [(8192, ("bits", 30, 4), 16, 3), (16384, ("bits", 30, 3), 30, 2), (2048, [40961], 8, 9),
                                            (32768, ("bits", 30, 1), 16, 1), (65536, ("bits", 30, 1), 16, 1),      # no fused tensor product / two-pass: composition
                                            (4096, ("bits", 40, 2), 20, 3), (8192, ("bits", 43, 2), 16, 1), (16384, ("bits", 40, 2), 20, 1),
                                            (2048, ("bits", 60, 2), 32, 2), (8192, ("bits", 64, 1), 32, 1), (256, ("bits", 250, 1), 64, 2),
                                            (2048, ("bits", 120, 1), 40, 1)])
@pytest.mark.parametrize("fused", [True, False, "one-launch-keyswitch", "no-four-workgroups", "throughput-kernels"])
def test_ct_multiply_relin_matches_oracle(eng, oracle, monkeypatch, n, spec, w, batch, fused):
    """FHEContext::multiply as the reference declares it (src/fhe.cu:199-224): tensor product + relinearisation in ONE call.  fused: c2
    crosses from the tensor-product kernel to the key-switch kernel in the compact workspace where both kernels exist (else, and with
    FHE_HIP_NO_FUSED_CT_RELIN=1, the two-call composition inside the library); every variant equals oracle ct_multiply + relinearize."""
    # few ciphertexts (4-byte residues) take special forms by default: tensor product over four workgroups per limb polynomial in three launches (else the
    # 16-per-thread kernel), key switch with one workgroup per digit pair + a combining launch; the variants switch them off one at a time and together
    if fused in ("one-launch-keyswitch", "throughput-kernels"):
        monkeypatch.setenv("FHE_HIP_SPLIT_PAIRS_POLYS", "0")
    if fused in ("no-four-workgroups", "throughput-kernels"):
        monkeypatch.setenv("FHE_HIP_COOP_POLYS", "0")
    if fused is False:
        monkeypatch.setenv("FHE_HIP_NO_FUSED_CT_RELIN", "1")
    moduli = _moduli(spec, n); L = len(moduli)
    e = eng.RnsNttEngine(n, moduli); rp = oracle.RnsPlan(n, moduli)
    K = e.relin_num_digits(w)
    kb = _random_keys(moduli, n, L * K, 1100); ka = _random_keys(moduli, n, L * K, 1900)
    rk = e.import_relin_keys(w, [_up(eng, k) for k in kb], [_up(eng, k) for k in ka])
    a0, a1, b0, b1 = (rns_poly(s, moduli, n, batch) for s in (61, 62, 63, 64))
    d = [_up(eng, x) for x in (a0, a1, b0, b1)]
    c0, c1 = eng.DeviceBuffer(a0.nbytes), eng.DeviceBuffer(a0.nbytes)
    for rep in range(2):                                         # the second call reuses the workspace
        e.ct_multiply_relin(rk, c0, c1, d[0], d[1], d[2], d[3], batch)
    t0, t1, t2 = rp.ct_multiply(a0, a1, b0, b1, threads=8)
    w0, w1 = rp.relinearize(w, t0, t1, t2, kb, ka, threads=8)
    assert np.array_equal(c0.download(a0.shape), w0)
    assert np.array_equal(c1.download(a0.shape), w1)
    for buf, src in zip(d, (a0, a1, b0, b1)):
        assert np.array_equal(buf.download(a0.shape), src)
    with pytest.raises(eng.FheError):
        e.ct_multiply_relin(rk, d[0], c1, d[0], d[1], d[2], d[3], batch)     # an output aliases an input


@pytest.mark.parametrize("n,spec,w,batch", [(8192, ("bits", 30, 4), 16, 1), (8192, ("bits", 30, 4), 16, 24), (16384, ("bits", 30, 3), 30, 2), (4096, ("bits", 30, 4), 16, 300),
                                            (4096, ("bits", 40, 3), 20, 2), (16384, ("bits", 40, 2), 20, 2), (2048, ("bits", 60, 2), 32, 3), (2048, ("bits", 64, 2), 32, 2),
                                            (32768, ("bits", 30, 2), 16, 1), (65536, ("bits", 30, 1), 16, 1), (32768, ("bits", 40, 1), 20, 2),
                                            (256, ("bits", 250, 1), 64, 2), (1024, ("bits", 120, 2), 40, 1)])
@pytest.mark.parametrize("form", ["default", "composed"])
def test_reserve_covers_every_entry_point(eng, monkeypatch, n, spec, w, batch, form):
    """fhe_rns_ntt_reserve(h, batch) after the key import: no later call of up to `batch` units grows a library workspace (what a hipGraph capture
    of those calls needs -- growth during capture is refused), on every width class, for few and many ciphertexts, two-pass sizes included."""
    if form == "composed":
        monkeypatch.setenv("FHE_HIP_NO_FUSED_KEYSWITCH", "1"); monkeypatch.setenv("FHE_HIP_NO_FUSED_BLIND_ROTATE", "1"); monkeypatch.setenv("FHE_HIP_NO_FUSED_CT_RELIN", "1")
    moduli = _moduli(spec, n); L = len(moduli)
    e = eng.RnsNttEngine(n, moduli)
    K = e.relin_num_digits(w)
    keys = [_up(eng, k) for k in _random_keys(moduli, n, L * K, 300)]
    rk = e.import_relin_keys(w, keys, keys)
    e.reserve(batch)
    held = e.workspace_bytes()
    assert held > 0
    for nb in sorted({batch, 1, max(1, batch // 2)}):
        x = rns_poly(77, moduli, n, nb)
        d = [_up(eng, x) for _ in range(4)]; o = [eng.DeviceBuffer(x.nbytes) for _ in range(3)]
        e.forward(d[0], nb); e.inverse(d[0], nb)
        e.multiply(o[0], d[0], d[1], nb); e.multiply(o[0], d[0], d[0], nb)
        e.ct_multiply(o[0], o[1], o[2], d[0], d[1], d[2], d[3], nb)
        e.relinearize(rk, o[0], o[1], o[2], nb)
        e.ct_multiply_relin(rk, o[0], o[1], d[0], d[1], d[2], d[3], nb)
        sh = _up(eng, np.arange(3 * nb, dtype=np.uint32).reshape(3, nb) % (2 * n))
        e.blind_rotate([rk] * 3, [rk] * 3, d[0], d[1], sh, o[0], o[1], nb)
        assert e.workspace_bytes() == held, f"a call of {nb} units grew a workspace after reserve({batch})"


def _largest_ntt_primes(bits, n, count):
    """The `count` largest primes below 2^bits with q = 1 (mod 2n): the top of a width class's range."""
    out, step = [], 2 * n
    q = ((1 << bits) // step) * step + 1
    while q >= (1 << bits):
        q -= step
    while len(out) < count:
        if nm.is_prime(q):
            out.append(q)
        q -= step
    return out


@pytest.mark.parametrize("n,bits,L,lazy", [(2048, 250, 2, True), (8192, 250, 1, True), (4096, 122, 2, True), (2048, 122, 1, True), (16384, 250, 1, True),
                                           (2048, 251, 1, False), (2048, 123, 1, False), (2048, 255, 1, False)])
def test_full_width_lazy_tiles_at_the_top_of_their_range(eng, oracle, monkeypatch, n, bits, L, lazy):
    """The full-width tile kernels skip reductions when every modulus leaves six spare bits (q < 2^250 on four limbs, q < 2^122 on two):
    forward butterflies unreduced (values below 23 q), inverse ones below 2q, canonical again before anything is stored.  The LARGEST primes of
    each range, operands at the top of [0, q): the lazy kernels, the canonical ones (FHE_HIP_NO_WIDE_LAZY=1) and the oracle agree bit for bit;
    one bit above the threshold the engine stays on the canonical kernels (same results either way)."""
    moduli = _largest_ntt_primes(bits, n, L)
    assert all(q.bit_length() == bits for q in moduli)
    e = eng.RnsNttEngine(n, moduli)
    monkeypatch.setenv("FHE_HIP_NO_WIDE_LAZY", "1")
    e_canon = eng.RnsNttEngine(n, moduli)
    monkeypatch.delenv("FHE_HIP_NO_WIDE_LAZY")
    assert e.width_class == eng.WIDTH_256
    rp = oracle.RnsPlan(n, moduli)
    batch = 3
    a = rns_poly(11, moduli, n, batch); b = rns_poly(12, moduli, n, batch)
    for l, q in enumerate(moduli):                       # slot 1: every coefficient q - 1; slot 2: q - 1 and 0 alternating against q - 1 and 1
        top = np.array([(q - 1) >> (64 * w) & (2**64 - 1) for w in range(4)], dtype=np.uint64)
        a[1, l, :, :] = top; b[1, l, :, :] = top
        a[2, l, ::2, :] = top; a[2, l, 1::2, :] = 0
        b[2, l, :, :] = top; b[2, l, 1::2, :] = 0; b[2, l, 1::2, 0] = 1
    shape = a.shape
    want_f, want_i, want_m = rp.forward(a, threads=8), rp.inverse(b, threads=8), rp.polymul(a, b, threads=8)
    for engine in (e, e_canon):
        d = _up(eng, a); engine.forward(d, batch)
        assert np.array_equal(d.download(shape), want_f)
        engine.inverse(d, batch)
        assert np.array_equal(d.download(shape), a)
        d = _up(eng, b); engine.inverse(d, batch)
        assert np.array_equal(d.download(shape), want_i)
        dA, dB, dR = _up(eng, a), _up(eng, b), eng.DeviceBuffer(a.nbytes)
        engine.multiply(dR, dA, dB, batch)
        assert np.array_equal(dR.download(shape), want_m)
        engine.multiply(dA, dA, dA, batch)               # square, in place
        assert np.array_equal(dA.download(shape), rp.polymul(a, a, threads=8))
