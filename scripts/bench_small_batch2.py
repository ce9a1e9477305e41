#!/usr/bin/env python3
"""Small-batch latency of the fused polymul (configs[2]'s batch-1 shape): per-call time at batch 1 .. 256 with each polynomial over four workgroups
(default up to 64 limb polynomials), the 16-per-thread latency kernel (default up to 256) and the throughput kernel, each forced, and the library's default choice.
Writes one JSON line per (batch, kernel) to gpurun_out/small_batch_<tag>.jsonl.   usage: bench_small_batch2.py [tag] [batches, comma separated]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("gpu-homomorphic-encryption_amd")
from workload import rns_poly  # noqa: E402
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
n, L = 8192, 4
moduli = pkg.find_ntt_primes(30, n, L)
out = open(os.path.join(ROOT, "gpurun_out", f"small_batch_{tag}.jsonl"), "w")
def engine(small, coop):
    """small / coop: FHE_HIP_SMALL_BATCH_POLYS / FHE_HIP_COOP_POLYS for this engine (None = the library's defaults)"""
    for k, v in (("FHE_HIP_SMALL_BATCH_POLYS", small), ("FHE_HIP_COOP_POLYS", coop)):
        if v is not None: os.environ[k] = str(v)
    try: return pkg.RnsNttEngine(n, moduli)
    finally:
        os.environ.pop("FHE_HIP_SMALL_BATCH_POLYS", None); os.environ.pop("FHE_HIP_COOP_POLYS", None)
engs = {"four-workgroups-three-launches": engine(None, 1000000), "latency-kernel": engine(1000000, 0), "throughput-kernel": engine(0, 0), "default": engine(None, None)}
BATCHES = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 4, 8, 16, 32, 64, 128, 256]
for B in BATCHES:
    a = pkg.DeviceBuffer.from_numpy(rns_poly(1, moduli, n, B)); b = pkg.DeviceBuffer.from_numpy(rns_poly(2, moduli, n, B)); r = pkg.DeviceBuffer(B * L * n * 32)
    for name, e in engs.items():
        for _ in range(20): e.multiply(r, a, b, B)
        pkg.capi.sync(); t = pkg.Timer(); steps = 400
        t.start(e)
        for _ in range(steps): e.multiply(r, a, b, B)
        t.stop(e); pkg.capi.sync()
        us = t.elapsed_ms() * 1e3 / steps
        line = {"op": "multiply", "n": n, "limbs": L, "batch": B, "kernel": name, "us_per_call": us, "polymul_per_s": B / (us * 1e-6)}
        out.write(json.dumps(line) + "\n"); print(line, flush=True)

# full ciphertext multiply (configs[2]) at small batches: digit pairs of the key switch on separate workgroups (default below 256 limb polynomials) vs the one-launch kernel
def engine2(split, coop):
    os.environ["FHE_HIP_SPLIT_PAIRS_POLYS"] = str(split); os.environ["FHE_HIP_COOP_POLYS"] = str(coop)
    try: return pkg.RnsNttEngine(n, moduli)
    finally:
        os.environ.pop("FHE_HIP_SPLIT_PAIRS_POLYS", None); os.environ.pop("FHE_HIP_COOP_POLYS", None)
engs2 = {"four-workgroup-tensor-product+split-pairs": engine2(1000000, 64), "split-pairs": engine2(1000000, 0), "one-launch": engine2(0, 0), "default": pkg.RnsNttEngine(n, moduli)}
w = 16
for B in BATCHES:
    ops = [pkg.DeviceBuffer.from_numpy(rns_poly(10 + i, moduli, n, B)) for i in range(4)]
    o0, o1 = pkg.DeviceBuffer(B * L * n * 32), pkg.DeviceBuffer(B * L * n * 32)
    for name, e in engs2.items():
        K = e.relin_num_digits(w)
        keys = [[pkg.DeviceBuffer.from_numpy(rns_poly(7000 + 31 * i + 997 * h, moduli, n, 1)) for i in range(L * K)] for h in range(2)]
        rk = e.import_relin_keys(w, keys[0], keys[1])
        for _ in range(10): e.ct_multiply_relin(rk, o0, o1, ops[0], ops[1], ops[2], ops[3], B)
        pkg.capi.sync(); t = pkg.Timer(); steps = 200
        t.start(e)
        for _ in range(steps): e.ct_multiply_relin(rk, o0, o1, ops[0], ops[1], ops[2], ops[3], B)
        t.stop(e); pkg.capi.sync()
        us = t.elapsed_ms() * 1e3 / steps
        line = {"op": "ctrelin", "n": n, "limbs": L, "w": w, "batch": B, "kernel": name, "us_per_call": us, "ct_mul_per_s": B / (us * 1e-6)}
        out.write(json.dumps(line) + "\n"); print(line, flush=True)
