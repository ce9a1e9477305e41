#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DFHE_STAMPS for the F32 / 2^13 instance, e.g. scratch/ab/libfhe_hip_stamps.so copied over
lib/libfhe_hip.so): s_memtime stamps at the phase boundaries of workgroup 0 / wave 0 of ntt16_multiply_kernel at batch 1.
s_memtime counts shader clocks (~2.3 GHz under this load: 51 K cycles = the 22 us of the kernel trace)."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("gpu-homomorphic-encryption_amd")
from workload import rns_poly  # noqa: E402
n, L, B = 8192, 4, 1
moduli = pkg.find_ntt_primes(30, n, L)
e = pkg.RnsNttEngine(n, moduli)
a = pkg.DeviceBuffer.from_numpy(rns_poly(1, moduli, n, B)); b = pkg.DeviceBuffer.from_numpy(rns_poly(2, moduli, n, B)); r = pkg.DeviceBuffer(B * L * n * 32)
lib = pkg.lib()
fn = lib.fhe_debug_stamps16
names = ["start", "loads issued", "operands arrived", "fwd(a) done", "fwd(b) done", "inverse done", "stores issued"]
for it in range(6):
    e.multiply(r, a, b, B); pkg.capi.sync()
    st = (ctypes.c_ulonglong * 16)()
    assert fn(st) == 0
    t = [st[i] for i in range(7)]
    print("run", it, " ".join(f"{names[i]}: +{(t[i] - t[0])} cyc" for i in range(1, 7)))
