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
    wider moduli: four words truncated below q."""
    L = len(moduli)
    out = np.zeros((batch, L, n, 4), dtype=np.uint64)
    for b in range(batch):
        for l, q in enumerate(moduli):
            sd = (0x5EED0000 + seed * 1000003 + b * L + l) & 0xFFFFFFFFFFFFFFFF
            if q < (1 << 64):
                out[b, l, :, 0] = splitmix64_block(sd, n) % np.uint64(q)
            else:
                # wider moduli: four stream words per coefficient, truncated to bit_length(q) - 1 bits
                # (< 2^(bits-1) <= q, so no reduction is needed and the fill stays vectorised)
                w = splitmix64_block(sd, 4 * n).reshape(n, 4)
                keep = q.bit_length() - 1
                for k in range(4):
                    lo = 64 * k
                    if keep <= lo:
                        w[:, k] = 0
                    elif keep < lo + 64:
                        w[:, k] &= np.uint64((1 << (keep - lo)) - 1)
                out[b, l] = w
    return out
