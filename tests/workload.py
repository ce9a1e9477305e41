"""Deterministic synthetic polynomials for tests and bench (shape of SURVEY.md section 8d):
coefficients uniform-ish in [0, q_limb) from a SplitMix64 stream, upper limbs zero for word-sized
moduli, layout [batch][L][n] of 32-byte containers (numpy uint64 (..., 4))."""
import numpy as np

GAMMA = np.uint64(0x9E3779B97F4A7C15)


def splitmix64_block(seed, count):
    """count outputs of SplitMix64 started at `seed` (vectorised; wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        s = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + GAMMA * np.arange(1, count + 1, dtype=np.uint64)
        z = s
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rns_poly(seed, moduli, n, batch):
    """[batch][L][n][4] uint64, residues < q_l.  Word-sized moduli: one stream word per coefficient;
    wider moduli: four words reduced with Python ints (small sizes only)."""
    L = len(moduli)
    out = np.zeros((batch, L, n, 4), dtype=np.uint64)
    for b in range(batch):
        for l, q in enumerate(moduli):
            sd = (0x5EED0000 + seed * 1000003 + b * L + l) & 0xFFFFFFFFFFFFFFFF
            if q < (1 << 64):
                out[b, l, :, 0] = splitmix64_block(sd, n) % np.uint64(q)
            else:
                w = splitmix64_block(sd, 4 * n).reshape(n, 4)
                for i in range(n):
                    v = (int(w[i, 0]) | (int(w[i, 1]) << 64) | (int(w[i, 2]) << 128) | (int(w[i, 3]) << 192)) % q
                    for k in range(4):
                        out[b, l, i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out
