"""N > 1 path on CPU: world_size-2 gloo processes shard a batch, work on their own shard with no
payload exchange, and agree (checksum gather, max-over-ranks timing) with the unsharded run."""
import importlib
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly(pkg):
    sh = importlib.import_module(pkg.__name__ + ".sharding")
    for total in (0, 1, 7, 8, 1024, 1027):
        for world in (1, 2, 3, 8):
            spans = [sh.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sh.shard_range(8, 2, 2)


WORKER = textwrap.dedent("""
    import importlib, os, sys, hashlib, time
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np
    sh = importlib.import_module("gpu-homomorphic-encryption_amd.sharding")
    from oracle import pyoracle as orc          # stands in for the GPU work in this CPU rehearsal
    from workload import rns_poly
    import ntt_math as nm
    dist = sh.init_process_group("gloo")
    rank, world, _ = sh.env_rank_world()
    n, L, batch = 256, 2, 6
    moduli = nm.ntt_primes(30, n, L)
    a = rns_poly(1, moduli, n, batch); b = rns_poly(2, moduli, n, batch)
    lo, hi = sh.shard_range(batch, rank, world)
    rp = orc.RnsPlan(n, moduli)
    dist.barrier(); t0 = time.perf_counter()
    mine = rp.polymul(np.ascontiguousarray(a[lo:hi]), np.ascontiguousarray(b[lo:hi]))
    dist.barrier(); dt = time.perf_counter() - t0
    digest = int.from_bytes(hashlib.sha256(mine.tobytes()).digest()[:7], "little")
    sums = sh.gather_ints(dist, digest)
    tmax = sh.max_over_ranks(dist, [dt, float(rank)])
    assert tmax[1] == world - 1 and tmax[0] >= dt
    if rank == 0:
        full = rp.polymul(a, b)
        want = [int.from_bytes(hashlib.sha256(np.ascontiguousarray(full[s:e]).tobytes()).digest()[:7], "little")
                for s, e in (sh.shard_range(batch, r, world) for r in range(world))]
        assert sums == want, (sums, want)
        print("RANK0 OK", world)
    dist.barrier(); dist.destroy_process_group()
""")


def test_two_rank_gloo_batch_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill(); out, _ = p.communicate()
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "RANK0 OK 2" in outs[0]
