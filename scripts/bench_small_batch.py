#!/usr/bin/env python3
"""Small-batch latency (BASELINE configs[1] / configs[2] at batch 1 .. 256): microseconds per call of the fused polymul and of the
full ciphertext multiply (tensor product + relinearisation), eager (one host launch per kernel) and replayed from a hipGraph that
holds K consecutive calls, next to the launch floor of this box (a one-workgroup kernel through the same two paths).
usage: bench_small_batch.py [out.jsonl]"""
import ctypes
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("gpu-homomorphic-encryption_amd")
from workload import rns_poly  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
vp = ctypes.c_void_p


def ck(rc, what):
    if rc:
        raise RuntimeError(f"{what}: hip error {rc}")


stream = vp()
ck(hip.hipStreamCreate(ctypes.byref(stream)), "hipStreamCreate")


def measure(eng, call, calls_per_graph=20, reps=30):
    """(eager us per call, graph-replay us per call)"""
    eng.set_stream(stream)
    for _ in range(3):
        call()
    pkg.capi.sync()
    t = pkg.Timer(); t.start(eng)
    for _ in range(calls_per_graph * 5):
        call()
    t.stop(eng); pkg.capi.sync()
    eager = t.elapsed_ms() * 1e3 / (calls_per_graph * 5)
    ck(hip.hipStreamBeginCapture(stream, 0), "begin capture")
    for _ in range(calls_per_graph):
        call()
    graph = vp(); ck(hip.hipStreamEndCapture(stream, ctypes.byref(graph)), "end capture")
    ex = vp(); ck(hip.hipGraphInstantiate(ctypes.byref(ex), graph, None, None, ctypes.c_size_t(0)), "instantiate")
    ck(hip.hipGraphLaunch(ex, stream), "graph launch"); pkg.capi.sync()
    t2 = pkg.Timer(); t2.start(eng)
    for _ in range(reps):
        ck(hip.hipGraphLaunch(ex, stream), "graph launch")
    t2.stop(eng); pkg.capi.sync()
    replay = t2.elapsed_ms() * 1e3 / (reps * calls_per_graph)
    hip.hipGraphExecDestroy(ex); hip.hipGraphDestroy(graph)
    return eager, replay


out = []
n, L, bits, w = 8192, 4, 30, 16
moduli = pkg.find_ntt_primes(bits, n, L)
eng = pkg.RnsNttEngine(n, moduli)
K = eng.relin_num_digits(w)
keys = [[pkg.DeviceBuffer.from_numpy(rns_poly(7000 + 31 * i + 997 * h, moduli, n, 1)) for i in range(L * K)] for h in range(2)]
rk = eng.import_relin_keys(w, keys[0], keys[1])
# launch floor: the smallest kernel the library has (element-wise add of ONE 2048-coefficient polynomial: 16 workgroups)
tiny = pkg.RnsNttEngine(2048, pkg.find_ntt_primes(30, 2048, 1))
tb = [pkg.DeviceBuffer.from_numpy(rns_poly(i, pkg.find_ntt_primes(30, 2048, 1), 2048, 1)) for i in range(3)]
fe, fr = measure(tiny, lambda: tiny.poly_add(tb[2], tb[0], tb[1], 1))
out.append({"what": "launch floor (poly_add of one 2048-coefficient polynomial)", "eager_us_per_call": fe, "graph_us_per_call": fr})
print(f"launch floor: eager {fe:.2f} us, graph replay {fr:.2f} us per call", flush=True)
for B in (1, 4, 16, 64, 256):
    bufs = [pkg.DeviceBuffer.from_numpy(rns_poly(10 + i, moduli, n, B)) for i in range(4)]
    outs = [pkg.DeviceBuffer(bufs[0].nbytes) for _ in range(3)]
    e1, r1 = measure(eng, lambda: eng.multiply(outs[0], bufs[0], bufs[1], B))

    def ctrelin():
        eng.ct_multiply(outs[0], outs[1], outs[2], bufs[0], bufs[1], bufs[2], bufs[3], B)
        eng.relinearize(rk, outs[0], outs[1], outs[2], B)
    e2, r2 = measure(eng, ctrelin)
    e3, r3 = measure(eng, lambda: eng.ct_multiply_relin(rk, outs[0], outs[1], bufs[0], bufs[1], bufs[2], bufs[3], B))   # FHEContext::multiply as ONE call
    row = {"batch": B, "n": n, "limbs": L, "prime_bits": bits,
           "ctrelin_one_call": {"eager_us_per_call": e3, "graph_us_per_call": r3, "eager_ct_mul_per_s": B / e3 * 1e6, "graph_ct_mul_per_s": B / r3 * 1e6},
           "multiply": {"eager_us_per_call": e1, "graph_us_per_call": r1, "eager_polymul_per_s": B / e1 * 1e6, "graph_polymul_per_s": B / r1 * 1e6},
           "ctrelin": {"eager_us_per_call": e2, "graph_us_per_call": r2, "eager_ct_mul_per_s": B / e2 * 1e6, "graph_ct_mul_per_s": B / r2 * 1e6,
                       "calls": 2}}
    out.append(row)
    print(f"B={B:4d}  multiply: eager {e1:8.2f} us  graph {r1:8.2f} us   ct_multiply + relinearize (2 calls): eager {e2:8.2f} us  graph {r2:8.2f} us   "
          f"ct_multiply_relin (1 call): eager {e3:8.2f} us  graph {r3:8.2f} us", flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        for r in out:
            f.write(json.dumps(r) + "\n")
