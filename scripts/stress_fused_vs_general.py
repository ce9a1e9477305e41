#!/usr/bin/env python3
"""GPU stress: the fused key-switch / external-product kernels against the general composition (different kernels, same ABI)
over random shapes, many repetitions -- looks for rare synchronisation bugs that a single parity test would miss.
Both sides run on the GPU; the oracle is not involved.   usage: stress_fused_vs_general.py [seconds] [seed]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("gpu-homomorphic-encryption_amd")
from workload import rns_poly  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); cases = 0; launches = 0
while time.time() - t0 < budget:
    log_n = int(rng.choice([11, 12, 13, 14, 14, 14, 15])); n = 1 << log_n
    bits = int(rng.choice([30, 30, 30, 40, 60, 64])) if log_n <= 14 else 30
    L = int(rng.integers(1, 7 if bits == 30 else 4))
    w = int(rng.choice([8, 16, 20, 30])) if bits == 30 else int(rng.choice([16, 20, 32]))
    batch = int(rng.integers(1, 41)) if log_n <= 13 else int(rng.integers(1, 9))
    if cases % 16 == 15:          # a batch large enough for the two-stream pipeline of fhe_ct_multiply_relin (>= 1024 limb polynomials per chunk)
        log_n, n, bits, L, w, batch = 12, 4096, 30, 4, int(rng.choice([16, 30])), int(rng.integers(512, 600))
    moduli = pkg.find_ntt_primes(bits, n, L)
    # the environment switches are read once, at engine creation: one engine per kernel family
    def make(**env):
        for k, v in env.items():
            os.environ[k] = v
        try:
            return pkg.RnsNttEngine(n, moduli)
        finally:
            for k in env:
                os.environ.pop(k, None)
    engs = {"fused": make(), "general": make(FHE_HIP_NO_FUSED_KEYSWITCH="1", FHE_HIP_NO_FUSED_BLIND_ROTATE="1"),
            "fused-single": make(FHE_HIP_NO_PAIRED_TRANSFORMS="1"),
            "fused-alt": make(FHE_HIP_NO_TWO_LAUNCH_CT="1", FHE_HIP_SPLIT_KEYSWITCH="1"),   # the other forms of the N = 2^14 / 2^15 kernels
            # the round-2 forms of what round 3 changed: c2 / accumulators as containers, monomial factor per digit, one stream, throughput kernels for few ciphertexts / accumulators
            "fused-r2": make(FHE_HIP_NO_C2_COMPACTION="1", FHE_HIP_NO_PREROTATION="1", FHE_HIP_CT_RELIN_CHUNKS="1", FHE_HIP_SPLIT_PAIRS_POLYS="0", FHE_HIP_COOP_POLYS="0",
                             **({"FHE_HIP_NO_COMPACT_BLIND_ROTATE": "1"} if cases % 2 else {}))}
    K = engs["fused"].relin_num_digits(w)
    seed = int(rng.integers(1 << 30))
    keys = [[pkg.DeviceBuffer.from_numpy(rns_poly(seed + 31 * i + 997 * h, moduli, n, 1)) for i in range(L * K)] for h in range(4)]
    keysets = {tag: [e.import_relin_keys(w, keys[0], keys[1]), e.import_relin_keys(w, keys[2], keys[3])] for tag, e in engs.items()}
    c = [rns_poly(seed + 5000 + i, moduli, n, batch) for i in range(3)]
    shape = c[0].shape
    # relinearisation, repeated: every repetition must give the same bits
    want = None
    for rep in range(4):
        for tag in ("general", "fused", "fused-single", "fused-alt", "fused-r2"):
            d = [pkg.DeviceBuffer.from_numpy(x) for x in c]
            engs[tag].relinearize(keysets[tag][0], d[0], d[1], d[2], batch); launches += 1
            got = (d[0].download(shape), d[1].download(shape))
            if want is None:
                want = got
            elif not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])):
                print(f"MISMATCH relin {tag} n={n} bits={bits} L={L} w={w} batch={batch} seed={seed} rep={rep}"); sys.exit(1)
    # FHEContext::multiply as one call (compact c0 / c1 / c2 between the kernels) vs tensor product + relinearisation as two calls
    ops = [rns_poly(seed + 7000 + i, moduli, n, batch) for i in range(4)]
    want = None
    for rep in range(2):
        for tag in ("general", "fused", "fused-single", "fused-alt", "fused-r2"):
            d = [pkg.DeviceBuffer.from_numpy(x) for x in ops]
            o = [pkg.DeviceBuffer(ops[0].nbytes) for _ in range(3)]
            if tag == "general":
                engs[tag].ct_multiply(o[0], o[1], o[2], d[0], d[1], d[2], d[3], batch)
                engs[tag].relinearize(keysets[tag][0], o[0], o[1], o[2], batch); launches += 2
            else:
                engs[tag].ct_multiply_relin(keysets[tag][0], o[0], o[1], d[0], d[1], d[2], d[3], batch); launches += 2
            got = (o[0].download(shape), o[1].download(shape))
            if want is None:
                want = got
            elif not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])):
                print(f"MISMATCH ct_multiply_relin {tag} n={n} bits={bits} L={L} w={w} batch={batch} seed={seed} rep={rep}"); sys.exit(1)
    # blind rotation loop: fused ping-pong vs composition
    steps = int(rng.integers(1, 5))
    shifts = rng.integers(0, 2 * n, size=(steps, batch), dtype=np.uint32)
    dSh = pkg.DeviceBuffer.from_numpy(shifts)
    want = None
    for rep in range(3):
        for tag in ("general", "fused", "fused-single", "fused-alt", "fused-r2"):
            a0, a1 = pkg.DeviceBuffer.from_numpy(c[0]), pkg.DeviceBuffer.from_numpy(c[1])
            t0b, t1b = pkg.DeviceBuffer(c[0].nbytes), pkg.DeviceBuffer(c[0].nbytes)
            engs[tag].blind_rotate([keysets[tag][0]] * steps, [keysets[tag][1]] * steps, a0, a1, dSh, t0b, t1b, batch); launches += steps
            got = (a0.download(shape), a1.download(shape))
            if want is None:
                want = got
            elif not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])):
                print(f"MISMATCH blind_rotate {tag} n={n} bits={bits} L={L} w={w} batch={batch} steps={steps} seed={seed} rep={rep}"); sys.exit(1)
    del keysets, engs
    cases += 1
    if cases % 10 == 0:
        print(f"{cases} shapes, {launches} launches, {time.time() - t0:.0f} s", flush=True)
print(f"stress ok: {cases} random shapes, {launches} launches compared bit for bit in {time.time() - t0:.0f} s")
