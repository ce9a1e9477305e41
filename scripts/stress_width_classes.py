#!/usr/bin/env python3
"""GPU stress: the LDS-resident word-sized kernels (FHE_WIDTH_32 / 52 / 64) against the full-width multi-pass kernels
(FHE_HIP_FORCE_WIDTH=256: different kernels, same ABI) over random shapes, every call repeated.  Both sides run on the GPU.
usage: stress_width_classes.py [seconds] [seed]"""
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


def fail(what, **kw):
    print("MISMATCH", what, kw); sys.exit(1)


while time.time() - t0 < budget:
    log_n = int(rng.integers(11, 17)); n = 1 << log_n
    # N = 2^16 on 4-byte residues and N = 2^15 / 2^16 on 8-byte residues are the two-pass sizes (compact workspace between the launches)
    bits = int(rng.choice([30, 30, 40, 60, 64])) if log_n <= 15 else int(rng.choice([30, 30, 40, 64]))
    L = int(rng.integers(1, 5)) if log_n <= 14 else int(rng.integers(1, 3))
    batch = int(rng.integers(1, 13)) if log_n <= 13 else int(rng.integers(1, 4))
    moduli = pkg.find_ntt_primes(bits, n, L)
    os.environ.pop("FHE_HIP_FORCE_WIDTH", None)
    fast = pkg.RnsNttEngine(n, moduli)
    os.environ["FHE_HIP_SMALL_BATCH_POLYS"] = "0"; os.environ["FHE_HIP_COOP_POLYS"] = "0"      # the throughput multiply kernel for these small batches
    fast_tp = pkg.RnsNttEngine(n, moduli)                  # (the default engine takes the four-workgroup form or the latency kernel)
    os.environ["FHE_HIP_SMALL_BATCH_POLYS"] = "1000000"
    fast_lat = pkg.RnsNttEngine(n, moduli)                 # the 16-per-thread latency kernel forced
    os.environ.pop("FHE_HIP_SMALL_BATCH_POLYS", None); os.environ.pop("FHE_HIP_COOP_POLYS", None)
    os.environ["FHE_HIP_FORCE_WIDTH"] = "256"
    wide = pkg.RnsNttEngine(n, moduli)
    os.environ["FHE_HIP_NO_WIDE_LAZY"] = "1"           # its canonical tile kernels (the default engine takes the lazy ones: these moduli leave six spare bits)
    wide_c = pkg.RnsNttEngine(n, moduli)
    os.environ.pop("FHE_HIP_NO_WIDE_LAZY", None)
    os.environ["FHE_HIP_FORCE_WIDTH"] = "128"          # the same class on two 64-bit limbs (R = 2^128)
    wide2 = pkg.RnsNttEngine(n, moduli)
    os.environ["FHE_HIP_NO_WIDE_TILES"] = "1"          # and with every stage as a global-memory pass (no LDS tiles)
    wide2p = pkg.RnsNttEngine(n, moduli)
    os.environ.pop("FHE_HIP_NO_WIDE_TILES", None)
    os.environ.pop("FHE_HIP_FORCE_WIDTH", None)
    assert fast.width_class != pkg.WIDTH_256 and wide.width_class == pkg.WIDTH_256 and wide2.width_class == pkg.WIDTH_256
    seed = int(rng.integers(1 << 30))
    x = [rns_poly(seed + i, moduli, n, batch) for i in range(4)]
    shape = x[0].shape
    info = dict(n=n, bits=bits, L=L, batch=batch, seed=seed)
    ref = {}
    for rep in range(3):
        for eng, tag in ((wide, "wide"), (fast, "fast"), (fast_tp, "fast-throughput-kernel"), (fast_lat, "fast-latency-kernel"), (wide2, "wide-2-limb"), (wide2p, "wide-2-limb-passes"), (wide_c, "wide-canonical-tiles"), (fast, "fast"))[:8 if rep == 0 else 4]:
            d = [pkg.DeviceBuffer.from_numpy(v) for v in x]
            o = [pkg.DeviceBuffer(x[0].nbytes) for _ in range(3)]
            eng.multiply(o[0], d[0], d[1], batch); launches += 1
            got = {"mul": o[0].download(shape)}
            eng.ct_multiply(o[0], o[1], o[2], d[0], d[1], d[2], d[3], batch); launches += 1
            got["ct"] = np.stack([b.download(shape) for b in o])
            eng.forward(d[0], batch); launches += 1
            got["fwd"] = d[0].download(shape)
            eng.inverse(d[0], batch); launches += 1
            if not np.array_equal(d[0].download(shape), x[0]):
                fail("round trip", tag=tag, rep=rep, **info)
            eng.multiply(d[1], d[1], d[1], batch); launches += 1            # in-place squaring
            got["sq"] = d[1].download(shape)
            for k, v in got.items():
                if k not in ref:
                    ref[k] = v
                elif not np.array_equal(ref[k], v):
                    fail(k, tag=tag, rep=rep, **info)
    cases += 1
    if cases % 10 == 0:
        print(f"{cases} shapes, {launches} launches, {time.time() - t0:.0f} s", flush=True)
print(f"stress ok: {cases} random shapes, {launches} launches compared bit for bit in {time.time() - t0:.0f} s")
