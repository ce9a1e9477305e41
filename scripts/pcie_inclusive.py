#!/usr/bin/env python3
"""PCIe-inclusive rate of the fused polymul for HOST-resident operands (never the bench `value`: the boundary takes device
pointers and the reference's callers keep data on the device; this number is for DESIGN.md section 7).
Pinned host buffers, chunks of the batch double-buffered over two streams: H2D(a, b) -> multiply -> D2H(r) per chunk.
usage: pcie_inclusive.py [batch] [chunk]"""
import importlib
import os
import sys
import time

import numpy as np
import torch                                                   # first: one HIP runtime per process

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("gpu-homomorphic-encryption_amd")
from workload import rns_poly  # noqa: E402

n, L = 8192, 4
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256
moduli = pkg.find_ntt_primes(30, n, L)
S = 32 * n * L
host = [torch.from_numpy(rns_poly(5 + i, moduli, n, batch).view(np.int64)).pin_memory() for i in range(2)]
out = torch.empty_like(host[0]).pin_memory()
streams = [torch.cuda.Stream() for _ in range(2)]
engines = [pkg.RnsNttEngine(n, moduli) for _ in range(2)]
dev = [[torch.empty((chunk, L, n, 4), dtype=torch.int64, device="cuda") for _ in range(3)] for _ in range(2)]
for e, s in zip(engines, streams):
    e.set_stream(s.cuda_stream)


def run():
    for c0 in range(0, batch, chunk):
        k = (c0 // chunk) & 1
        with torch.cuda.stream(streams[k]):
            dev[k][0].copy_(host[0][c0:c0 + chunk], non_blocking=True)
            dev[k][1].copy_(host[1][c0:c0 + chunk], non_blocking=True)
            engines[k].multiply(dev[k][2], dev[k][0], dev[k][1], chunk)
            out[c0:c0 + chunk].copy_(dev[k][2], non_blocking=True)
    torch.cuda.synchronize()


run()
t0 = time.perf_counter(); reps = 3
for _ in range(reps):
    run()
dt = (time.perf_counter() - t0) / reps
print(f"PCIe-inclusive: {batch / dt:.0f} polymul/s, {3 * S * batch / dt / 1e9:.1f} GB/s over the link (2 in + 1 out), batch {batch}, chunks of {chunk}")
# spot check against the device-resident path
d = [pkg.DeviceBuffer.from_numpy(host[i][:chunk].numpy().view(np.uint64)) for i in range(2)]
r = pkg.DeviceBuffer(chunk * S); engines[0].set_stream(None); engines[0].multiply(r, d[0], d[1], chunk)
assert np.array_equal(r.download((chunk, L, n, 4)), out[:chunk].numpy().view(np.uint64)), "host-resident pipeline differs from the device-resident product"
print("results identical to the device-resident path")
