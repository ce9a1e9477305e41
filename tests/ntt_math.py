"""Independent pure-Python big-integer helpers for the tests (neither oracle nor product code).

Closed forms of the reference primitives, a Miller-Rabin prime search, and the
direct O(n^2) definitions of the negacyclic transform / product that both the oracle and the HIP
engine are checked against.
"""
import random

R_BITS = 256
R = 1 << R_BITS
M256 = R - 1
M64 = (1 << 64) - 1


# ---- closed forms of the leaf primitives (include/bigint.cuh:27-140) ---------------------------
def add_mod_ref(a, b, q):
    s = (a + b) & M256
    t = (s - q) & M256
    return s if (t >> 192) > (s >> 192) else t        # top-limb borrow test (D15)


def sub_mod_ref(a, b, q):
    d = (a - b) & M256
    return (d + q) & M256 if (d >> 192) > (a >> 192) else d


def mont_inverse_ref(q):
    inv, n0 = 1, q & M64
    for _ in range(6):
        inv = (inv * (2 - n0 * inv)) & M64
    return (-inv) & M64


def mont_mul_ref(a, b, q, inv0=None):
    """Limb-wise SOS reduction exactly as written (valid for garbage inv0 / unreduced inputs too)."""
    if inv0 is None:
        inv0 = mont_inverse_ref(q)
    t = a * b                                           # < 2^512
    for i in range(4):
        m = (((t >> (64 * i)) & M64) * inv0) & M64
        t = (t + ((m * q) << (64 * i))) & ((1 << 512) - 1)   # carry out of limb 7 is lost
    u = t >> 256
    d = (u - q) & M256
    return u if (d >> 192) > (u >> 192) else d


def ct_ref(a, b, w, q):
    t = mont_mul_ref(b, w, q)
    return add_mod_ref(a, t, q), sub_mod_ref(a, t, q)


def gs_ref(a, b, w, q):
    return add_mod_ref(a, b, q), mont_mul_ref(sub_mod_ref(a, b, q), w, q)


# ---- primes ---------------------------------------------------------------------------------
def is_prime(n, rounds=24):
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2; s += 1
    rng = random.Random(0xC0FFEE ^ (n & 0xFFFFFFFF))
    bases = list(small) + [rng.randrange(2, n - 1) for _ in range(rounds)]
    for a in bases:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def ntt_primes(bits, n, count):
    """The `count` smallest primes >= 2^(bits-1) with q = 1 (mod 2n)  (SURVEY.md section 8d)."""
    out, step = [], 2 * n
    q = ((1 << (bits - 1)) // step) * step + 1
    while q < (1 << (bits - 1)):
        q += step
    while len(out) < count:
        if is_prime(q):
            out.append(q)
        q += step
    return out


def find_psi(n, q):
    """Same rule as the engine and the oracle: first x^((q-1)/2n), x = 2,3,..., of order exactly 2n."""
    e = (q - 1) // (2 * n)
    x = 2
    while True:
        c = pow(x, e, q)
        if pow(c, n, q) == q - 1:
            return c
        x += 1


def bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1); x >>= 1
    return r


# ---- direct definitions ------------------------------------------------------------------------
def negacyclic_ntt_direct(x, q, psi):
    """X[k] = sum_j x[j] psi^((2*bitrev(k)+1) j)   -- O(n^2)."""
    n = len(x); bits = n.bit_length() - 1
    out = []
    for k in range(n):
        e = 2 * bitrev(k, bits) + 1
        w = pow(psi, e, q)
        acc, p = 0, 1
        for j in range(n):
            acc = (acc + x[j] * p) % q
            p = p * w % q
        out.append(acc)
    return out


def negacyclic_mul_direct(a, b, q):
    n = len(a)
    r = [0] * n
    for i, ai in enumerate(a):
        if ai == 0:
            continue
        for j, bj in enumerate(b):
            k = i + j
            if k < n:
                r[k] = (r[k] + ai * bj) % q
            else:
                r[k - n] = (r[k - n] - ai * bj) % q
    return r


def splitmix64(seed):
    """SplitMix64 stream (the synthetic-input generator of SURVEY.md 8d)."""
    s = seed & M64
    while True:
        s = (s + 0x9E3779B97F4A7C15) & M64
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        yield z ^ (z >> 31)
