"""GPU test: the C ABI takes raw device pointers, so torch tensors (device memory) and torch streams plug in directly --
PyTorch is plumbing here, never the product.  Runs in a fresh interpreter that imports torch FIRST, so the process holds
one libamdhip64 (torch's and the engine's share a SONAME)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent(r"""
    import importlib, sys
    import numpy as np
    import torch                                              # first: one HIP runtime per process
    sys.path[:0] = [%(root)r, %(tests)r]
    pkg = importlib.import_module("gpu-homomorphic-encryption_amd")
    from oracle import pyoracle as orc
    from workload import rns_poly
    assert torch.cuda.is_available()
    n, L, batch = 8192, 4, 6
    moduli = pkg.find_ntt_primes(30, n, L)
    eng = pkg.RnsNttEngine(n, moduli)
    a = rns_poly(301, moduli, n, batch); b = rns_poly(302, moduli, n, batch)
    dev = torch.device("cuda", 0)
    tA = torch.from_numpy(a.view(np.int64)).to(dev); tB = torch.from_numpy(b.view(np.int64)).to(dev)
    tR = torch.empty_like(tA)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng.set_stream(side.cuda_stream)                      # enqueue on torch's stream: ordered with torch work on it
        tA2 = tA.clone()                                      # torch kernel on the same stream, before the engine call
        eng.multiply(tR, tA2, tB, batch)                      # tensors are passed as data_ptr()
        total = tR.sum()                                      # torch kernel after it, same stream: sees the product
    side.synchronize()
    got = tR.cpu().numpy().view(np.uint64)
    want = orc.RnsPlan(n, moduli).polymul(a, b, threads=8)
    assert np.array_equal(got, want), "product through torch tensors differs from the oracle"
    assert int(total.item()) == int(want.view(np.int64).sum()), "torch reduction on the same stream did not see the product"
    eng.set_stream(None)
    eng.forward(tA, batch); eng.inverse(tA, batch); pkg.capi.sync()
    assert np.array_equal(tA.cpu().numpy().view(np.uint64), a)
    print("torch interop ok")
""")


@pytest.mark.gpu
def test_engine_runs_on_torch_tensors_and_streams():
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}], capture_output=True, text=True,
                         timeout=900, env=env)
    assert res.returncode == 0 and "torch interop ok" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
