#!/usr/bin/env python3
"""bench.py -- NTT-polymul/sec at N = 8192 with 4 RNS limbs (BASELINE.json configs[1]) on MI355X.

One "step" = one pass of the hot path over one batch of synthetic RNS polynomials resident in HBM:
    r = a (*) b  mod (x^N + 1, q_0..q_{L-1})          fhe_rns_ntt_multiply  (NTTEngine::multiply, src/ntt.cu:49-75)
for `--batch` polynomial pairs per GPU ([batch][L][N] 32-byte containers, 30-bit NTT primes = log_q 120 / 4).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1, the driver's form)

Prints ONE JSON line on rank 0.  `value` = polynomial products per second over all GPUs (weak
scaling: every rank multiplies its own `--batch` pairs; the path has no exchange step, so RCCL is used
only for the barriers and the max-over-ranks of the timings).  `roofline` is measured live with HIP
events on the engine's stream; `cpu_baseline` times the CPU oracle (a port, not the product) on a
bounded sample.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 measured copy)

# ---- the second roof: integer-multiply issue ("report both HBM GB/s and MAD/s so the binding roof is visible") ----
# Peak: 256 CUs x 4 SIMDs, one multiply-class wave instruction (v_mad_u64_u32 / v_mul_hi_u32 / v_mul_lo_u32, and the FP64 FMA class
# of the F52 field, which issues at the same rate) per 4.9 cycles per SIMD at 2.4 GHz -- measured, scratch/ubench.hip, DESIGN 4.1 --
# = 1024 * 64 lanes * 2.4e9 / 4.9 = 32.1 T lane-multiplies/s.
SIMDS, CLOCK_HZ, MUL_ISSUE_CYCLES = 1024, 2.4e9, 4.9
INT_MUL_PEAK = SIMDS * 64 * CLOCK_HZ / MUL_ISSUE_CYCLES
# multiply-class instructions per butterfly / per NTT-domain product, counted in the ISA of each width class:
#   u32 (F32): Montgomery product = v_mad_u64_u32, v_mul_lo_u32, v_mad_u64_u32                                   -> 3 / 3
#   f64 (F52): every FP64 instruction issues at the multiply rate: x*w, fma, w*qinv, x*wq, rint, fma, add + the butterfly's add, sub  -> 9 / 6
#   u64 (F64): Shoup product = 64x64 low (3 x 32-bit multiplies) + 64x64 high (4 multiplies + carries) + low     -> 13 / 16
#   u64 full range (F64X): Montgomery product = 64x64 -> 128 (7), low (3), high (4+)                              -> 18 / 18
#   multi-limb (wide_asm.inc): 2 NW^2 + NW v_mad_u64_u32 for NW = 4 / 8 32-bit words                              -> 36 / 136
# and, where measured, the VALU issue time of a whole butterfly in 2.4 GHz cycles (all instructions, scratch/ubench.hip): 27 / 36 / 100.
MULS = {1: (3, 3, 27.0), 3: (9, 6, 36.0), 2: (13, 16, 100.0), 5: (18, 18, None), 4: (136, 136, None)}


def int_mul_model(width_class, op, n, L, K, br_steps=1, wide_nl=4):
    """(multiply-class lane-instructions, butterflies) ONE unit of `op` executes (one polymul / NTT pair / ct-mul / relin / external product)."""
    per_bfly, per_prod, _ = MULS[width_class]
    if width_class == 4 and wide_nl == 2:
        per_bfly = per_prod = 36
    logn = n.bit_length() - 1
    bfly = (n // 2) * logn                                  # butterflies of one transform of one limb polynomial
    if op == "multiply":
        t, prods = 3 * L, L * n
    elif op == "fwdinv":
        t, prods = 2 * L, 0
    elif op == "ct":
        t, prods = 7 * L, 4 * L * n                         # c0, c2: one product each; c1: two products (one shared reduction on u32)
    elif op == "relin":
        t, prods = L * (L * K + 2), L * (L * K) * 2 * n     # every limb workgroup: L*K digit transforms + 2 inverse; 2 key halves per digit
    elif op == "ctrelin":
        t, prods = 7 * L + L * (L * K + 2), 4 * L * n + L * (L * K) * 2 * n
    elif op == "blindrotate":
        t, prods = L * (2 * L * K + 2), L * (2 * L * K) * 2 * n
    else:
        raise ValueError(op)
    return t * bfly * per_bfly + prods * per_prod, t * bfly


def secondary_roof(width_class, op, n, L, K, units_per_s, wide_nl=4):
    """roofline.secondary: the integer-multiply issue roof beside the HBM one."""
    muls, bflies = int_mul_model(width_class, op, n, L, K, wide_nl=wide_nl)
    achieved = muls * units_per_s
    cyc = MULS[width_class][2]
    return {"bound": "valu-int-mul", "achieved": achieved / 1e9, "peak": INT_MUL_PEAK / 1e9, "unit": "G lane-multiplies/s", "frac": achieved / INT_MUL_PEAK,
            "butterflies_per_s": bflies * units_per_s, "multiply_class_per_butterfly": MULS[width_class][0] if not (width_class == 4 and wide_nl == 2) else 36,
            # all VALU instructions of a butterfly (multiplies + adds + selects), measured cycles per wave-butterfly per SIMD
            "valu_issue_frac": None if cyc is None else bflies * units_per_s * cyc / 64 / (SIMDS * CLOCK_HZ),
            "model": "multiply-class lane-instructions per unit (transforms x butterflies x per-butterfly count + NTT-domain products) x units/s "
                     "against 1024 SIMDs x 64 lanes x 2.4 GHz / 4.9 cycles (v_mad_u64_u32 issue, scratch/ubench.hip)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096,
                    help="polynomial pairs per GPU per step (4096 = the plateau of the batch sweep 1/64/1024/4096; 12 GiB of a, b, r)")
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--limbs", type=int, default=4)
    ap.add_argument("--bits", type=int, default=30, help="bit length of each RNS prime (30 = log_q 120 / 4 limbs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) on a multi-GPU node; gloo only to rehearse the N > 1 path on a one-GPU box")
    ap.add_argument("--device-override", type=int, default=None, help="rehearsal only: put every rank on this device")
    ap.add_argument("--op", choices=["multiply", "fwdinv", "ct", "relin", "ctrelin", "blindrotate"], default="multiply",
                    help="multiply = fused polymul (the headline, configs[1]); fwdinv = forward+inverse NTT pair; "
                         "ct = ciphertext tensor product; relin = key switching of c2 into (c0, c1); "
                         "ctrelin = tensor product + relinearisation (configs[2]: full ciphertext multiply); "
                         "blindrotate = blind-rotation inner loop (configs[4]): --br-steps external products per accumulator and step")
    ap.add_argument("--br-steps", type=int, default=8, help="blindrotate: external products per bench step (one fhe_blind_rotate call)")
    ap.add_argument("--br-keys", type=int, default=4, help="blindrotate: distinct RGSW ciphertexts cycled through by the loop")
    ap.add_argument("--decomp-bits", type=int, default=16, help="relinearisation digit width w (reference default 16)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's contract): every rank processes --batch units; strong: --batch is the TOTAL, split into "
                         "contiguous per-rank blocks (sharding.shard_range)")
    ap.add_argument("--two-calls", action="store_true",
                    help="ctrelin: fhe_ct_multiply followed by fhe_ct_relinearize (c2 through a container buffer) instead of the one-call fhe_ct_multiply_relin")
    ap.add_argument("--shard", choices=["batch", "limb"], default="batch",
                    help="batch (default): every rank owns whole polynomials (all limbs) of its part of the batch; limb: rank r owns the residues "
                         "modulo q_l, l = r (mod G), of every polynomial (ops multiply / fwdinv / ct only; no collective either way)")
    ap.add_argument("--timeout", type=float, default=float(os.environ.get("FHE_BENCH_TIMEOUT", "1500")),
                    help="self-launched ranks (--gpus N without a launcher): kill every rank and exit non-zero after this many seconds")
    ap.add_argument("--no-verify", action="store_true", help="skip the per-rank result checksum / oracle spot check (outside the timed region)")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the child runs of the other BASELINE configurations (profiling scripts pass it: children would add their own counter files)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra legs (op multiply only): batched forward+inverse NTT pairs (the figure the north-star's >= 60 %% target is stated on) "
                         "and the small runs on the other modulus widths")
    return ap.parse_args()


def launch_ranks(args):
    """`bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks as child processes, one per
    GPU, relay rank 0's JSON line, exit non-zero if any rank fails.  The parent never imports torch and never touches
    HIP (a process that has initialised the GPU must not fork / exec workers), it only builds the library once so the
    ranks do not race for the build lock."""
    import socket
    import subprocess
    import tempfile
    pkg = importlib.import_module("gpu-homomorphic-encryption_amd")
    pkg.build_library()                          # make + hipcc only; the library is not loaded here
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tmp = tempfile.mkdtemp(prefix="fhe_bench_")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = open(os.path.join(tmp, f"rank{r}.out"), "w+"); err = open(os.path.join(tmp, f"rank{r}.err"), "w+")
        procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err), out, err))
    failed = None
    deadline = time.time() + args.timeout
    while failed is None and any(p.poll() is None for p, _, _ in procs):
        time.sleep(0.2)
        for r, (p, _, _) in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        if time.time() > deadline:               # a rank that hangs (e.g. in a barrier) instead of exiting must not hang the launcher
            failed = -1
            sys.stderr.write(f"bench.py: ranks still running after --timeout {args.timeout:.0f} s; killing them\n")
    if failed is not None:                       # one rank died: the others would wait in a barrier forever
        time.sleep(2.0 if failed >= 0 else 0.0)
        for p, _, _ in procs:
            if p.poll() is None:
                p.kill()
    rcs = [p.wait() for p, _, _ in procs]
    lines = []
    for r, (p, out, err) in enumerate(procs):
        out.seek(0); err.seek(0)
        o, e = out.read(), err.read()
        out.close(); err.close()
        if r == 0:
            lines = [l for l in o.splitlines() if l.startswith("{")]
        if e.strip() and (rcs[r] != 0 or os.environ.get("FHE_BENCH_VERBOSE")):
            sys.stderr.write(f"--- rank {r} (exit {rcs[r]}) stderr ---\n{e[-4000:]}\n")
    if any(rcs) or len(lines) != 1:
        raise SystemExit(f"bench.py: ranks exited with {rcs}; rank 0 printed {len(lines)} JSON lines")
    line = json.loads(lines[0])
    if line.get("n_gpus") != args.gpus or line.get("ranks_seen") != args.gpus:
        raise SystemExit(f"bench.py: asked for {args.gpus} ranks, the line reports n_gpus={line.get('n_gpus')} ranks_seen={line.get('ranks_seen')}")
    print(lines[0], flush=True)


def fill_device(pkg, buf, seed, moduli, n, batch, chunk=64, limbs=None):
    """Seeded residues, generated on the host in chunks and uploaded (outside any timed region).  limbs: keep only these limbs of
    every polynomial (limb-split sharding: all ranks cut their slices out of the same global polynomials)."""
    import numpy as np
    from workload import rns_poly
    limbs = list(range(len(moduli))) if limbs is None else list(limbs)
    per = len(limbs) * n * 32
    for b0 in range(0, batch, chunk):
        nb = min(chunk, batch - b0)
        arr = rns_poly(seed + b0, moduli, n, nb)
        if len(limbs) != len(moduli):
            arr = np.ascontiguousarray(arr[:, limbs])
        rc = pkg.lib().fhe_hip_memcpy_h2d(buf.ptr + b0 * per, arr.ctypes.data, arr.nbytes)
        if rc:
            raise RuntimeError(pkg.lib().fhe_hip_last_error().decode())


def host_poly(seed, moduli, n, b, chunk=64, limbs=None):
    """The polynomial fill_device(seed, ...) put at batch index b, regenerated on the host: [L][n][4]."""
    import numpy as np
    from workload import rns_poly
    b0 = (b // chunk) * chunk
    p = rns_poly(seed + b0, moduli, n, b - b0 + 1)[b - b0]
    return p if limbs is None else np.ascontiguousarray(p[list(limbs)])


def download_poly(pkg, buf, b, S):
    """One RNS polynomial (S bytes at batch index b) of a device buffer as numpy uint64 [S / 32][4]."""
    import numpy as np
    out = np.empty(S // 8, dtype=np.uint64)
    pkg.capi.sync()
    rc = pkg.lib().fhe_hip_memcpy_d2h(out.ctypes.data, buf.ptr + b * S, S)
    if rc:
        raise RuntimeError(pkg.lib().fhe_hip_last_error().decode())
    return out.reshape(-1, 4)


def verify_shard(pkg, op, moduli, n, B, ins, outs, in_seeds, relin=None):
    """Outside the timed region: SHA-256 over a sample of THIS rank's result polynomials (first, middle, last of the shard)
    and, for the ops the CPU oracle restates, the same hash over the oracle's results for the same operands (downloaded from the
    device, so the check covers exactly what the kernels read).  relin = (w, kb, ka) host key polynomials for --op ctrelin.
    Returns (checksum, oracle_checksum or None); the oracle is the checker here, never the thing measured."""
    import hashlib
    import numpy as np
    L = len(moduli); S = 32 * n * L
    sample = sorted({0, B // 2, B - 1})
    got, want = hashlib.sha256(), hashlib.sha256()
    rp = None
    if op in ("multiply", "fwdinv", "ct", "ctrelin"):
        from oracle import pyoracle as orc
        orc.build()
        rp = orc.RnsPlan(n, moduli)
    for b in sample:
        if op == "fwdinv":
            res = [download_poly(pkg, ins[0], b, S)]               # K forward+inverse pairs in place: back to the operand
            exp = [host_poly(in_seeds[0][0], in_seeds[0][1], n, b, limbs=in_seeds[0][2]).reshape(-1, 4)]
        else:
            res = [download_poly(pkg, o, b, S) for o in (outs[:2] if op == "ctrelin" else outs)] if outs else [download_poly(pkg, i, b, S) for i in ins[:2]]
            exp = None
            if rp is not None:
                ops = [np.ascontiguousarray(download_poly(pkg, i, b, S).reshape(1, L, n, 4)) for i in ins]
                if op == "multiply":
                    exp = [rp.polymul(ops[0], ops[1]).reshape(-1, 4)]
                elif op == "ct":
                    exp = [c.reshape(-1, 4) for c in rp.ct_multiply(ops[0], ops[1], ops[2], ops[3])]
                else:                                              # FHEContext::multiply: tensor product, then relinearisation
                    t0, t1, t2 = rp.ct_multiply(ops[0], ops[1], ops[2], ops[3], threads=4)
                    exp = [c.reshape(-1, 4) for c in rp.relinearize(relin[0], t0, t1, t2, relin[1], relin[2], threads=4)]
        for r in res:
            got.update(np.ascontiguousarray(r).tobytes())
        if exp is not None:
            for e in exp:
                want.update(np.ascontiguousarray(e).tobytes())
    to63 = lambda h: int.from_bytes(h.digest()[:8], "little") >> 1
    return to63(got), (to63(want) if rp is not None else None)


def extra_width_classes(pkg, steps=10, warmup=2):
    """Small fused-polymul runs on the other modulus widths (outside the headline timing), so that every BENCH record shows
    them: 3 x 40-bit (FP64 class), 2 x 60-bit and 2 x 64-bit (64-bit integer classes) and 1 x 128-bit, 1 x 250-bit (full-width class)."""
    out = []
    n = 8192
    for bits, L, B in ((40, 3, 512), (60, 2, 512), (64, 2, 512), (128, 1, 128), (250, 1, 64)):
        try:
            moduli = pkg.find_ntt_primes(bits, n, L)
            eng = pkg.RnsNttEngine(n, moduli)
            S = 32 * n * L
            dA, dB, dR = (pkg.DeviceBuffer(B * S) for _ in range(3))
            fill_device(pkg, dA, 51000 + bits, moduli, n, B); fill_device(pkg, dB, 52000 + bits, moduli, n, B)
            for _ in range(warmup):
                eng.multiply(dR, dA, dB, B)
            t = pkg.Timer(); pkg.capi.sync(); t.start(eng)
            for _ in range(steps):
                eng.multiply(dR, dA, dB, B)
            t.stop(eng); pkg.capi.sync()
            ms = t.elapsed_ms() / steps
            gbs = 3 * S * B / (ms * 1e-3) / 1e9
            rate = B / (ms * 1e-3)
            sec = secondary_roof(int(eng.width_class), "multiply", n, L, 0, rate, wide_nl=2 if bits <= 127 else 4)
            out.append({"prime_bits": bits, "limbs": L, "batch": B, "width_class": int(eng.width_class), "polymul_per_s": rate,
                        "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS,
                        "secondary": {k: sec[k] for k in ("bound", "achieved", "peak", "unit", "frac", "multiply_class_per_butterfly", "valu_issue_frac")}})
            del dA, dB, dR, eng
        except Exception as e:                       # an extra must never take the headline down
            out.append({"prime_bits": bits, "limbs": L, "error": str(e)[:200]})
    return out


def extra_workloads():
    """The other BASELINE configurations on their single-GPU shapes, each a short child run of this script (outside the headline timing), so that the
    driver's record carries them beside the configs[1] line: configs[2] = full ciphertext multiply N = 8192, log_q = 120 (fhe_ct_multiply_relin) and its
    relinearisation alone; configs[3] = the same at N = 16384, 6 limbs, 128 ciphertexts (= 1024 / 8 GPUs); configs[4] = the blind-rotation loop at that shape."""
    import subprocess
    specs = [("configs[2] ctrelin N=8192 4x30-bit w=16", ["--op", "ctrelin", "--batch", "1024"]),
             ("configs[2] relin alone", ["--op", "relin", "--batch", "1024"]),
             ("configs[3] ctrelin N=16384 6x30-bit, 128 ciphertexts per GPU", ["--op", "ctrelin", "--batch", "128", "--n", "16384", "--limbs", "6"]),
             ("configs[3] on 40-bit primes", ["--op", "ctrelin", "--batch", "128", "--n", "16384", "--limbs", "6", "--bits", "40"]),
             ("configs[4] blind-rotation loop N=16384 6x30-bit w=16, 8 steps", ["--op", "blindrotate", "--batch", "128", "--n", "16384", "--limbs", "6"]),
             ("configs[4] on 40-bit primes", ["--op", "blindrotate", "--batch", "128", "--n", "16384", "--limbs", "6", "--bits", "40", "--steps", "3"])]
    out = []
    for name, extra in specs:
        try:
            args = [sys.executable, os.path.abspath(__file__), "--no-extras", "--no-cpu-baseline", "--steps", "10", "--warmup", "3"] + extra
            res = subprocess.run(args, capture_output=True, text=True, timeout=300)
            d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
            r = d["roofline"]
            out.append({"workload": name, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "batch": d["config"]["batch_per_gpu"],
                        "dtype": d["dtype"], "hbm_frac": r["frac"], "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
                        "int_mul_frac": r["secondary"]["frac"], "limiter": r["limiter"], "verified": d.get("verified")})
        except Exception as e:                       # an extra must never take the headline down
            out.append({"workload": name, "error": str(e)[:200]})
    return out


def extra_latency():
    """One call on ONE polynomial / ciphertext / accumulator (the reference's API is one per call, include/ntt.cuh:78-84): microseconds per call of the fused polymul,
    the one-call ciphertext multiply and a blind-rotation step, each a short child run at batch 1 (the library picks its few-ciphertext forms there)."""
    import subprocess
    specs = [("polymul N=8192 4x30-bit, batch 1", ["--op", "multiply", "--batch", "1"], 1),
             ("ciphertext multiply (tensor + relin, w=16) N=8192 4x30-bit, batch 1", ["--op", "ctrelin", "--batch", "1"], 1),
             ("blind-rotation step (external product, w=16) N=8192 4x30-bit, 1 accumulator", ["--op", "blindrotate", "--batch", "1"], None),
             ("blind-rotation step N=16384 6x30-bit (configs[4] size), 1 accumulator", ["--op", "blindrotate", "--batch", "1", "--n", "16384", "--limbs", "6"], None)]
    out = []
    for name, extra, per_step in specs:
        try:
            args = [sys.executable, os.path.abspath(__file__), "--no-extras", "--no-cpu-baseline", "--steps", "200", "--warmup", "20"] + extra
            res = subprocess.run(args, capture_output=True, text=True, timeout=300)
            d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
            # blindrotate: a step of the bench is a loop of several external products; value = external products per second at batch 1
            us = d["ms_per_step"] * 1e3 if per_step else 1e6 / d["value"]
            out.append({"workload": name, "us_per_call": us, "value": d["value"], "unit": d["unit"], "verified": d.get("verified")})
        except Exception as e:
            out.append({"workload": name, "error": str(e)[:200]})
    return out


def cpu_baseline(n, moduli, target_core_seconds=16.0):
    """CPU oracle (oracle/fhe_oracle.c, OpenMP over batch x limb) on a bounded sample of the same workload."""
    from oracle import pyoracle as orc
    from workload import rns_poly
    orc.build()
    # the GPU box gives one GPU a 16-core share of the host; never oversubscribe it
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(orc.max_threads(), avail, int(os.environ.get("FHE_BENCH_CPU_THREADS", "16"))))
    rp = orc.RnsPlan(n, moduli)
    a = rns_poly(7, moduli, n, 1); b = rns_poly(8, moduli, n, 1)
    t0 = time.perf_counter(); rp.polymul(a, b, threads=1); one = time.perf_counter() - t0
    sample = max(cores, int(target_core_seconds / max(one, 1e-6)))
    sample = min(sample, 4096)
    a = rns_poly(7, moduli, n, sample); b = rns_poly(8, moduli, n, sample)
    t0 = time.perf_counter(); rp.polymul(a, b, threads=cores); dt = time.perf_counter() - t0
    out = {"value": sample / dt, "unit": "polymul/s", "cores": int(rp.threads_used), "kind": "port",
           "sample": f"{sample} polymuls of N={n}, L={len(moduli)} (oracle/fhe_oracle.c, 256-bit Montgomery, "
                     f"OpenMP over batch x limb; single-thread {one * 1e3:.1f} ms/polymul)"}
    if max(moduli) < (1 << 62):
        # the same product with 64-bit residues (no 256-bit containers on the CPU side): the fairer number to hold the GPU against
        ns = min(4096, sample * 8)
        a = rns_poly(7, moduli, n, ns); b = rns_poly(8, moduli, n, ns)
        t0 = time.perf_counter(); rp.polymul_narrow(a, b, threads=cores); dt = time.perf_counter() - t0
        out["narrow_port"] = {"value": ns / dt, "unit": "polymul/s", "cores": int(rp.threads_used),
                              "sample": f"{ns} polymuls, word-sized port (64-bit residues, Shoup multiplication, same outputs)"}
    return out


FIELD_OF_CLASS = {1: "F32", 2: "F64", 3: "F52", 5: "F64X"}


def kernel_instance_prefix(kernel, width_class, n):
    """The template-instance prefix rocprofv3 prints for the kernel THIS run launches, e.g. `ntt_multiply_kernel<fhe_dev::F32, 13,`:
    field and log2 n pin the instance (a profile run also launches the other fields' instances in its extra legs)."""
    f = FIELD_OF_CLASS.get(width_class)
    return None if f is None else f"{kernel}<fhe_dev::{f}, {min(n.bit_length() - 1, 15)},"


def pmc_traffic(kernels, width_class, op, n, limbs, bits, batch):
    """HBM bytes per STEP of the kernels this workload launches, from the committed rocprofv3 --pmc passes of the SAME command
    (profiles/<tag>_summary.json, written by scripts/summarize_profile.py: FETCH_SIZE / WRITE_SIZE corrected per load shape,
    profiles/r03_fetch_calibration.txt).  Entries are selected by the exact template instance (field, log2 n) of each kernel the step
    launches -- never by a substring of the name.  None when no profile of this exact workload is committed."""
    import glob
    best = None
    want = [kernel_instance_prefix(k, width_class, n) for k in kernels]
    if None in want:
        return None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
            cfg = d.get("bench_line_under_profiler", {}).get("config", {})
            if (cfg.get("op", "multiply"), cfg.get("n"), cfg.get("limbs"), cfg.get("prime_bits"), cfg.get("batch_per_gpu")) != (op, n, limbs, bits, batch):
                continue
            total, names = 0.0, []
            for w in want:                       # every template instance of the kernel this step launches (e.g. the compact-in / compact-out forms of a loop)
                hits = [(name, k) for name, k in d["kernels"].items() if w in name and ("hbm_bytes_per_step" in k or "hbm_bytes_per_launch" in k)]
                if not hits:
                    raise LookupError(w)
                for name, k in hits:             # per STEP where the summary has it (a step may launch a kernel several times); older summaries: one launch per step
                    total += k.get("hbm_bytes_per_step", k.get("hbm_bytes_per_launch"))
                    names.append(name.split("(")[0])
            best = {"bytes": total, "source": os.path.basename(f), "kernels": names}
        except Exception:
            continue
    return best


LIMB_SPLIT_OPS = ("multiply", "fwdinv", "ct")


def main():
    args = parse()
    if args.shard == "limb" and args.op not in LIMB_SPLIT_OPS:      # refused before any rank or device is touched
        raise SystemExit(f"bench.py: --shard limb cannot run --op {args.op}: relinearisation / key switching / the external product decompose "
                         "EVERY limb of c2 into digits and multiply them into every limb (DESIGN 4.3), so a limb-split rank would need all "
                         "the other ranks' residues -- an all-to-all the path does not have; use --shard batch")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)            # the parent only spawns and relays; each child re-enters main() with RANK / WORLD_SIZE set
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist = None
    use_dist = world > 1 or os.environ.get("FHE_BENCH_FORCE_DIST") == "1"      # the latter: 1-rank RCCL smoke test
    if use_dist:
        import torch                       # first, so the process has ONE libamdhip64 (same SONAME as ours)
    pkg = importlib.import_module("gpu-homomorphic-encryption_amd")
    sharding = importlib.import_module("gpu-homomorphic-encryption_amd.sharding")
    pkg.build_library()                    # no-op when lib/libfhe_hip.so is up to date (it travels with the snapshot); file-locked
    if use_dist:
        if args.device_override is not None:
            os.environ["LOCAL_RANK"] = str(args.device_override); local_rank = args.device_override
        dist = sharding.init_process_group(args.dist_backend)   # RCCL: barriers + max-over-ranks only, no payload collective
    if pkg.device_count() < 1:
        raise SystemExit("bench.py: no HIP device; the engine has no CPU fallback")
    if use_dist:
        rc = pkg.lib().fhe_hip_set_device(local_rank)
        assert rc == 0, pkg.lib().fhe_hip_last_error()
    red_dev = "cuda" if (dist is not None and args.dist_backend == "nccl") else "cpu"

    n, L, B = args.n, args.limbs, args.batch
    if args.scaling == "strong" and args.shard == "batch":   # fixed total work: this rank's contiguous block of the batch
        lo, hi = sharding.shard_range(args.batch, rank, world)
        B = hi - lo
        if B < 1:
            raise SystemExit("bench.py: --scaling strong needs --batch >= number of ranks")
    all_moduli = pkg.find_ntt_primes(args.bits, n, L)
    my_limbs = list(range(L))
    if args.shard == "limb":
        # Limb split (SURVEY 8e secondary partitioning; the reference's docs/ARCHITECTURE.md:499-512, README.md:320 "Distribute RNS
        # components across GPUs"): rank r owns the residues modulo q_l for l = r (mod G) of EVERY polynomial of the batch and builds
        # its engine on that prime subset.  Valid where limbs never meet: transforms, pointwise products, the tensor product.
        my_limbs = sharding.limb_shard(L, rank, world)
        if not my_limbs:
            raise SystemExit(f"bench.py: --shard limb needs --gpus <= --limbs ({world} ranks, {L} limbs: rank {rank} would own nothing)")
    moduli = [all_moduli[l] for l in my_limbs]
    Lr = len(moduli)                                 # limbs this rank computes on
    eng = pkg.RnsNttEngine(n, moduli)
    S = 32 * n * Lr                                  # bytes of one RNS polynomial (this rank's limbs)
    n_in, n_out = {"multiply": (2, 1), "fwdinv": (1, 0), "ct": (4, 3), "relin": (3, 0), "ctrelin": (4, 3), "blindrotate": (2, 2)}[args.op]
    ins = [pkg.DeviceBuffer(B * S) for _ in range(n_in)]
    outs = [pkg.DeviceBuffer(B * S) for _ in range(n_out)]
    # limb split: every rank derives its operands from the SAME global polynomials (seeds do not depend on the rank) and keeps its limbs
    in_seeds = [1000 + 4000 * i + (0 if args.shard == "limb" else rank * 100000) for i in range(n_in)]
    for i, buf in enumerate(ins):
        fill_device(pkg, buf, in_seeds[i], all_moduli, n, B, limbs=my_limbs)
    for buf in outs:
        buf.zero()
    K = 0
    legacy = None                                    # (S-multiples per unit, text): the accounting of earlier rounds where it differs
    relin_host = None
    L_all, L = L, Lr                                 # from here on L = the limbs of the engine this rank runs
    if args.op == "multiply":
        dA, dB = ins; dR = outs[0]
        step = lambda: eng.multiply(dR, dA, dB, B)
        unit, units_per_poly_bytes, kernel = "polymul/s", 3, "ntt_multiply_kernel"
        what = "forward+inverse NTT + pointwise mul (fused polymul)"
    elif args.op == "fwdinv":
        dA = ins[0]
        step = lambda: (eng.forward(dA, B), eng.inverse(dA, B))
        unit, units_per_poly_bytes, kernel = "ntt-pair/s", 4, "ntt_forward_kernel+ntt_inverse_kernel"
        what = "batched forward + inverse NTT pair (in place)"
    elif args.op == "ct":
        step = lambda: eng.ct_multiply(outs[0], outs[1], outs[2], ins[0], ins[1], ins[2], ins[3], B)
        unit, units_per_poly_bytes, kernel = "ct-mul/s", 7, "ntt_ct_multiply_kernel"
        what = "ciphertext tensor product c0=a0b0, c1=a0b1+a1b0, c2=a1b1 (no relinearisation)"
    elif args.op == "blindrotate":
        import numpy as np
        from workload import rns_poly
        K = eng.relin_num_digits(args.decomp_bits)
        rgsw = []
        for g in range(args.br_keys):            # RGSW ciphertext g = two row sets of L*K key pairs each (synthetic uniform residues)
            rows = []
            for c in range(2):
                keys = [[pkg.DeviceBuffer.from_numpy(rns_poly(9000 + 31 * i + 997 * h + 5000 * c + 20000 * g, moduli, n, 1)) for i in range(L * K)]
                        for h in range(2)]
                rows.append(eng.import_relin_keys(args.decomp_bits, keys[0], keys[1]))
                del keys
            rgsw.append(rows)
        R = args.br_steps
        rows0 = [rgsw[s % args.br_keys][0] for s in range(R)]; rows1 = [rgsw[s % args.br_keys][1] for s in range(R)]
        shifts = np.random.default_rng(1234 + rank).integers(0, 2 * n, size=(R, B), dtype=np.uint32)
        dSh = pkg.DeviceBuffer.from_numpy(shifts)
        step = lambda: eng.blind_rotate(rows0, rows1, ins[0], ins[1], dSh, outs[0], outs[1], B)
        # per external product: read the accumulator pair, write the accumulator pair (keys are shared by the whole batch)
        # the CALL reads the accumulator pair once and writes it once (between its steps the pair stays in the library's compact workspace)
        unit, units_per_poly_bytes, kernel = "extprod/s", 4, "ntt_extprod2_kernel"
        legacy = (4 * R, f"4 S per external product (every step in container form, rounds 1-2); a {R}-step fhe_blind_rotate call needs 4 S in all")
        what = (f"blind-rotation inner loop: {R} steps acc += ExtProd((X^a - 1) acc, RGSW_s) per accumulator, w = {args.decomp_bits}, "
                f"{2 * L * K} rows per RGSW, {args.br_keys} RGSW keys cycled")
    else:
        from workload import rns_poly
        K = eng.relin_num_digits(args.decomp_bits)
        keys = [[pkg.DeviceBuffer.from_numpy(rns_poly(7000 + 31 * i + 997 * h, moduli, n, 1)) for i in range(L * K)] for h in range(2)]
        rk = eng.import_relin_keys(args.decomp_bits, keys[0], keys[1])
        if args.op == "relin":
            step = lambda: eng.relinearize(rk, ins[0], ins[1], ins[2], B)
            # read c2 once (re-reads by the L limb workgroups are cache traffic), read + write c0 and c1
            unit, units_per_poly_bytes, kernel = "relin/s", 5, "ntt_keyswitch2_kernel"
            what = f"relinearisation: key switching of c2 into (c0, c1), w = {args.decomp_bits}, {L * K} key levels"
        else:
            if args.two_calls:
                def step():
                    eng.ct_multiply(outs[0], outs[1], outs[2], ins[0], ins[1], ins[2], ins[3], B)
                    eng.relinearize(rk, outs[0], outs[1], outs[2], B)
            else:            # FHEContext::multiply as one ABI call: c2 stays in the library's (compact) workspace
                step = lambda: eng.ct_multiply_relin(rk, outs[0], outs[1], ins[0], ins[1], ins[2], ins[3], B)
            # one call: 4 S in (a0, a1, b0, b1) + 2 S out (c0, c1); c2 and the intermediate c0, c1 never leave the library.  Two calls: 7 S + 5 S.
            unit, units_per_poly_bytes, kernel = "ct-mul/s", (12 if args.two_calls else 6), "ntt_ct_multiply_kernel+ntt_keyswitch2_kernel"
            if not args.two_calls:
                legacy = (12, "12 S = tensor product 7 S + relinearisation 5 S (the two-call minimum, rounds 1-2); fhe_ct_multiply_relin needs 6 S")
            relin_host = (args.decomp_bits, [rns_poly(7000 + 31 * i, moduli, n, 1)[0] for i in range(L * K)],
                          [rns_poly(7000 + 31 * i + 997, moduli, n, 1)[0] for i in range(L * K)])
            what = (f"full ciphertext multiply ({'fhe_ct_multiply + fhe_ct_relinearize' if args.two_calls else 'fhe_ct_multiply_relin'}): "
                    f"tensor product + relinearisation, w = {args.decomp_bits}")

    def barrier():
        pkg.capi.sync()
        if dist is not None:
            dist.barrier()
            pkg.capi.sync()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        timer = pkg.Timer()
        barrier()
        t0 = time.perf_counter()
        timer.start(eng)
        for _ in range(steps):
            fn()
        timer.stop(eng)
        pkg.capi.sync()
        if dist is not None:
            dist.barrier()
        wall = time.perf_counter() - t0
        ev_ms = timer.elapsed_ms()
        wall, ev_ms = sharding.max_over_ranks(dist, [wall, ev_ms], device=red_dev)
        return wall, ev_ms

    wall, ev_ms = timed(step, args.steps, args.warmup)
    ms_per_step = wall * 1e3 / args.steps
    per_unit = args.br_steps if args.op == "blindrotate" else 1
    units_per_step = B * per_unit                    # units one rank processes per step
    total_units = units_per_step * world
    if args.scaling == "strong" and args.shard == "batch":   # ranks hold blocks whose sizes differ by at most one
        total_units = args.batch * per_unit
    if args.shard == "limb":                         # the ranks jointly produce B whole polynomials (each its limbs)
        total_units = B * per_unit
    value = total_units / (wall / args.steps)
    launch_ms = ev_ms / args.steps                   # HIP-event time of one step's launches on the engine stream
    # 3 S per polymul, 4 S per fwd+inv pair, 7 S per tensor product, 5 S per relinearisation; one-call forms: what the CALL must move
    algo_bytes = units_per_poly_bytes * S * B
    achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
    width = {1: "u32", 2: "u64", 3: "f64 (exact integers < 2^53)", 4: "u256", 5: "u64 (full-range)"}[eng.width_class]
    metric = "NTT-polymul/sec (N=8192, 4 RNS limbs) + achieved HBM GB/s vs peak"
    if (args.op, n, L_all) != ("multiply", 8192, 4):
        metric = f"{unit[:-2]}/sec (N={n}, {L_all} RNS limbs) + achieved HBM GB/s vs peak"
    wide_nl = 2 if (eng.width_class == 4 and max(moduli) < (1 << 127)) else 4
    sec = secondary_roof(int(eng.width_class), args.op, n, L, K, units_per_step / (launch_ms * 1e-3), wide_nl=wide_nl)
    streaming = args.op in ("multiply", "fwdinv", "ct") and eng.width_class in (1, 3)
    scaling = "strong" if args.shard == "limb" else args.scaling
    out = {
        "metric": metric,
        "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": width, "data": "synthetic",
        "config": {"workload": f"{'configs[1]: ' if args.op == 'multiply' else ''}{what}, N={n}, {L_all} RNS limbs "
                               f"({args.bits}-bit primes), batch {B} per GPU, 32-byte containers",
                   "op": args.op, "n": n, "limbs": L_all, "prime_bits": args.bits, "batch_per_gpu": B,
                   "parallelism": (f"limb-shard x{world} (rank r: limbs r mod {world})" if args.shard == "limb" else f"batch-shard x{world}")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                     "kernel": kernel, "launch_ms": launch_ms, "algorithmic_bytes_per_launch": algo_bytes,
                     # which roof is nearer: streaming kernels sit on HBM; the key-switch / external-product kernels run 10-26 transforms per
                     # 4-6 S of traffic and the 64-bit / multi-limb classes many multiplies per butterfly -- `secondary` puts a number on that
                     "limiter": "hbm" if streaming else ("valu-int-mul" if sec["frac"] > achieved / HBM_PEAK_GBS else "latency (neither roof: see secondary)"),
                     "secondary": sec},
    }
    if args.shard == "limb":
        out["config"]["limbs_of_rank0"] = my_limbs
        out["roofline"]["algorithmic_bytes_note"] = f"per rank: {units_per_poly_bytes} x 32 x N x {L} limbs of this rank x batch"
    if legacy is not None:                           # earlier rounds credited the step-by-step composition's bytes; kept only so that rounds compare
        lb = legacy[0] * S * B
        out["roofline"]["legacy_accounting"] = {"what": legacy[1], "bytes": lb, "achieved": lb / (launch_ms * 1e-3) / 1e9,
                                                "frac": lb / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    tr = pmc_traffic(kernel.split("+"), int(eng.width_class), args.op, n, L_all, args.bits, B) if args.shard == "batch" else None
    if tr:
        out["roofline"]["traffic"] = tr["bytes"]
        out["roofline"]["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (per-shape factors: profiles/r03_fetch_calibration.txt), "
                                             "profiles/" + tr["source"] + ": " + " + ".join(tr["kernels"]))
    if not args.no_extras and args.op == "multiply":
        w2, e2 = timed(lambda: (eng.forward(dA, B), eng.inverse(dA, B)), args.steps, args.warmup)
        pair_ms = e2 / args.steps
        out["extra_fwd_inv_pairs"] = {"pairs_per_s": (B if args.shard == "limb" else B * world) / (w2 / args.steps), "achieved_GBps": 4 * S * B / (pair_ms * 1e-3) / 1e9,
                                      "frac": 4 * S * B / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # every rank proves its own shard (outside the timed region): result checksum + oracle spot check, gathered to rank 0
    if not args.no_verify:
        csum, want = verify_shard(pkg, args.op, moduli, n, B, ins, outs, [(sd, all_moduli, my_limbs) for sd in in_seeds], relin=relin_host)
        sums = sharding.gather_ints(dist, csum, device=red_dev)
        wants = sharding.gather_ints(dist, -1 if want is None else want, device=red_dev)
        out["ranks_seen"] = len(set(sharding.gather_ints(dist, rank, device=red_dev)))
        out["shard_checksums"] = [f"{v:016x}" for v in sums]
        out["oracle_checksums"] = None if want is None else [f"{v:016x}" for v in wants]
        out["verified"] = None if want is None else (sums == wants)
        if out["verified"] is False:
            if rank == 0:
                print(json.dumps(out), file=sys.stderr, flush=True)
            raise SystemExit("bench.py: a rank's result differs from the CPU oracle (checksums above)")
    else:
        out["ranks_seen"] = len(set(sharding.gather_ints(dist, rank, device=red_dev)))
    out["dist_backend"] = None if dist is None else (args.dist_backend + (" (RCCL over xGMI)" if args.dist_backend == "nccl" else " (CPU rehearsal)"))
    if rank == 0 and world == 1 and not args.no_extras and args.op == "multiply":
        out["extra_width_classes"] = extra_width_classes(pkg)
    if rank == 0 and world == 1 and not args.no_extras and not args.no_extra_workloads and args.op == "multiply" and args.shard == "batch":
        out["extra_workloads"] = extra_workloads()
        out["extra_latency"] = extra_latency()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.op == "multiply":
        out["cpu_baseline"] = cpu_baseline(n, moduli)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
