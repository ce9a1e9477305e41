#!/usr/bin/env python3
"""CPU baselines of BASELINE.md section 2 with the oracle (a port of the reference path, 256-bit Montgomery):
config 1 (single forward NTT, N = 4096, 1 prime, one thread) and config 2 (polymul N = 8192, 4 limbs; 1 thread and
OpenMP over batch x limb on up to 16 threads).  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as orc
import ntt_math as nm
from workload import rns_poly

def best(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)

out = {}
q = nm.ntt_primes(30, 4096, 1)[0]
p = orc.Plan(4096, q); x = rns_poly(1, [q], 4096, 1)[0, 0]
out["config1_forward_ntt_n4096_1prime_1thread_ms"] = best(lambda: p.forward(x), 20) * 1e3
moduli = nm.ntt_primes(30, 8192, 4); rp = orc.RnsPlan(8192, moduli)
a = rns_poly(2, moduli, 8192, 1); b = rns_poly(3, moduli, 8192, 1)
out["config2_polymul_n8192_4limbs_1thread_ms"] = best(lambda: rp.polymul(a, b, threads=1), 5) * 1e3
threads = max(1, min(orc.max_threads(), len(os.sched_getaffinity(0)), 16))
B = 16 * threads
a = rns_poly(2, moduli, 8192, B); b = rns_poly(3, moduli, 8192, B)
dt = best(lambda: rp.polymul(a, b, threads=threads), 2)
out["config2_polymul_per_s_openmp"] = B / dt; out["threads"] = threads
try:
    out["cpu"] = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    pass
print(json.dumps(out))
