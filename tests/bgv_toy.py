"""Test-side toy BGV scheme over an RNS basis (NOT product code, NOT part of the oracle).

Only used to pin the tensor product + relinearisation end to end: Dec(relin(Enc(m1) x Enc(m2))) == m1 * m2,
including the reference test's expectation 15 60 135 240 (tests/test_fhe.cu:270).  Everything here is plain
Python big-integer / numpy arithmetic; polynomial products use the O(n^2) definition.

Conventions match the engine: polynomials are [L][n] residue arrays; a ciphertext is (c0, c1) with
c0 + c1*s = m + t*e (mod Q); relinearisation key (j,k) = (b, a) with b = -a*s + t*e + g_{j,k}*s^2 where g_{j,k} is
2^(k*w) in limb j and 0 in the other limbs (the RNS form of src/fhe.cu:76-111's 2^(i*w)*s^2)."""
import random

import numpy as np

import ntt_math as nm


def _mul(a, b, q):
    return nm.negacyclic_mul_direct([int(x) for x in a], [int(x) for x in b], q)


class ToyBGV:
    def __init__(self, n, moduli, t, seed=1, fast_mul=None):
        """fast_mul(x, y) -> x (*) y on [L][n] python-int arrays (e.g. the oracle's NTT polymul) replaces the O(n^2)
        definition for large n; the small-n tests keep the definition."""
        self.n, self.moduli, self.t, self.L = n, list(moduli), t, len(moduli)
        self.fast_mul = fast_mul
        self.Q = 1
        for q in self.moduli:
            self.Q *= q
        self.rng = random.Random(seed)
        self.s = [self.rng.choice((-1, 0, 1)) for _ in range(n)]

    # ---- helpers on [L][n] python-int arrays -----------------------------------------------------
    def to_rns(self, poly):
        return [[int(c) % q for c in poly] for q in self.moduli]

    def small(self, bound=3):
        return [self.rng.randint(-bound, bound) for _ in range(self.n)]

    def uniform(self):
        return [[self.rng.randrange(q) for _ in range(self.n)] for q in self.moduli]

    def mul(self, x, y):
        if self.fast_mul is not None:
            return self.fast_mul(x, y)
        return [_mul(x[l], y[l], q) for l, q in enumerate(self.moduli)]

    def add(self, x, y):
        return [[(u + v) % q for u, v in zip(x[l], y[l])] for l, q in enumerate(self.moduli)]

    def sub(self, x, y):
        return [[(u - v) % q for u, v in zip(x[l], y[l])] for l, q in enumerate(self.moduli)]

    def crt_centered(self, x):
        out = []
        for i in range(self.n):
            v = 0
            for l, q in enumerate(self.moduli):
                Ql = self.Q // q
                v += x[l][i] * Ql * pow(Ql, -1, q)
            v %= self.Q
            out.append(v - self.Q if v > self.Q // 2 else v)
        return out

    # ---- scheme ------------------------------------------------------------------------------------
    def encrypt(self, m):
        a = self.uniform()
        e = self.small()
        s_r = self.to_rns(self.s)
        c0 = self.sub(self.to_rns([mi + self.t * ei for mi, ei in zip(m, e)]), self.mul(a, s_r))
        return c0, a

    def phase(self, comps):
        """sum_i c_i * s^i, centred mod Q."""
        s_r = self.to_rns(self.s)
        acc = comps[0]
        sp = s_r
        for c in comps[1:]:
            acc = self.add(acc, self.mul(c, sp))
            sp = self.mul(sp, s_r)
        return self.crt_centered(acc)

    def decrypt(self, comps):
        return [v % self.t for v in self.phase(comps)]

    def relin_keys(self, decomp_bits):
        K = (max(q.bit_length() for q in self.moduli) + decomp_bits - 1) // decomp_bits
        s_r = self.to_rns(self.s)
        s2 = self.mul(s_r, s_r)
        keys_b, keys_a = [], []
        for j in range(self.L):
            for k in range(K):
                a = self.uniform()
                e = self.small()
                b = self.sub(self.to_rns([self.t * ei for ei in e]), self.mul(a, s_r))
                g = pow(2, k * decomp_bits, self.moduli[j])
                b[j] = [(u + g * v) % self.moduli[j] for u, v in zip(b[j], s2[j])]
                keys_b.append(b); keys_a.append(a)
        return keys_b, keys_a, K

    def rgsw(self, mu, decomp_bits):
        """RGSW encryption of the small integer mu as two row sets (one per RLWE component), each a pair of key lists in the
        relinearisation-key layout: rows0[jk] = RLWE(0) + (g_jk * mu, 0), rows1[jk] = RLWE(0) + (0, g_jk * mu)."""
        K = (max(q.bit_length() for q in self.moduli) + decomp_bits - 1) // decomp_bits
        s_r = self.to_rns(self.s)
        out = []
        for comp in (0, 1):
            kb, ka = [], []
            for j in range(self.L):
                for k in range(K):
                    a = self.uniform(); e = self.small()
                    b = self.sub(self.to_rns([self.t * ei for ei in e]), self.mul(a, s_r))      # RLWE(0): b + a*s = t*e
                    g = pow(2, k * decomp_bits, self.moduli[j]) * mu % self.moduli[j]
                    tgt = b if comp == 0 else a
                    tgt[j] = [(tgt[j][0] + g) % self.moduli[j]] + tgt[j][1:]                     # + g * mu on the constant coefficient
                    kb.append(b); ka.append(a)
            out.append((kb, ka))
        return out[0], out[1], K

    # ---- slot (batch) encoding: t = 1 (mod 2n) ---------------------------------------------------------
    def _slot_matrix(self):
        """V[i][j] = zeta_i^j mod t at the n odd powers zeta_i = psi^(2i+1) (int64 is enough: t < 2^31)."""
        n, t = self.n, self.t
        assert (t - 1) % (2 * n) == 0 and t < (1 << 31)
        psi = nm.find_psi(n, t)
        pts = np.array([pow(psi, 2 * i + 1, t) for i in range(n)], dtype=np.int64)
        V = np.empty((n, n), dtype=np.int64)
        V[:, 0] = 1
        for j in range(1, n):
            V[:, j] = V[:, j - 1] * pts % t
        return V

    def slot_encode(self, values):
        """Polynomial m with m(zeta_i) = values[i]: m_j = n^-1 * sum_i v_i * zeta_i^-j, and zeta_i^-j = zeta_i^(2n-j)."""
        n, t = self.n, self.t
        V = self._slot_matrix()
        vals = np.array(list(values) + [0] * (n - len(values)), dtype=np.int64) % t
        ninv = pow(n, -1, t)
        m = [int(vals.sum() % t) * ninv % t]
        for j in range(1, n):
            col = (t - V[:, n - j]) % t                     # zeta^(2n-j) = -zeta^(n-j)  (zeta^n = -1)
            m.append(int((vals * col % t).sum() % t) * ninv % t)
        return m

    def slot_decode(self, m):
        t = self.t
        V = self._slot_matrix()
        coeffs = np.array([int(c) % t for c in m], dtype=np.int64)
        return [int((V[i] * coeffs % t).sum() % t) for i in range(self.n)]


def to_limb_array(x):
    """[L][n] python ints -> numpy (1, L, n, 4) container array."""
    L, n = len(x), len(x[0])
    out = np.zeros((1, L, n, 4), np.uint64)
    for l in range(L):
        for i in range(n):
            v = int(x[l][i])
            for k in range(4):
                out[0, l, i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def from_limb_array(a):
    a = np.asarray(a).reshape(a.shape[-3], a.shape[-2], 4)
    return [[int(a[l, i, 0]) | (int(a[l, i, 1]) << 64) | (int(a[l, i, 2]) << 128) | (int(a[l, i, 3]) << 192)
             for i in range(a.shape[1])] for l in range(a.shape[0])]
