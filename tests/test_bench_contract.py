"""GPU test: bench.py prints ONE JSON line that carries every field of the driver's contract (metric, value, roofline,
cpu_baseline ...) -- run small, in a child process, exactly as the driver runs it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_default_workload_line_has_every_contract_field():
    d = _run("--steps", "3", "--warmup", "1", "--batch", "128")
    assert d["metric"].startswith("NTT-polymul/sec (N=8192, 4 RNS limbs)") and d["unit"] == "polymul/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "u32"
    assert d["value"] > 1e5 and d["ms_per_step"] > 0
    assert abs(d["value"] - 128 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6          # value = units / time
    c = d["config"]
    assert "configs[1]" in c["workload"] and c["n"] == 8192 and c["limbs"] == 4 and c["batch_per_gpu"] == 128 and "model" not in c
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["algorithmic_bytes_per_launch"] == 3 * 32 * 8192 * 4 * 128 and "traffic" in r
    # a committed PMC profile is only ever attached to the workload it was taken on, and then it must be THIS kernel's traffic
    if r["traffic"] is not None:
        assert 0.95 <= r["traffic"] / r["algorithmic_bytes_per_launch"] <= 1.5, r
    sec = r["secondary"]
    assert sec["bound"] == "valu-int-mul" and 0 < sec["frac"] < 1 and abs(sec["frac"] - sec["achieved"] / sec["peak"]) < 1e-9
    assert sec["multiply_class_per_butterfly"] == 3 and r["limiter"] == "hbm"
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["unit"] == "polymul/s" and b["value"] > 0 and b["cores"] >= 1 and "sample" in b
    assert b["narrow_port"]["value"] > b["value"]
    assert d["extra_fwd_inv_pairs"]["pairs_per_s"] > 0
    # every rank proves its shard: result checksum == the oracle's on the sampled polynomials
    assert d["ranks_seen"] == 1 and d["verified"] is True and d["shard_checksums"] == d["oracle_checksums"] and len(d["shard_checksums"]) == 1
    x = {(e["prime_bits"], e["limbs"]): e for e in d["extra_width_classes"]}
    assert set(x) == {(40, 3), (60, 2), (64, 2), (128, 1), (250, 1)} and all("error" not in e and e["polymul_per_s"] > 0 for e in x.values())
    assert all(e["secondary"]["bound"] == "valu-int-mul" and 0 < e["secondary"]["frac"] < 1.2 for e in x.values())
    w = d["extra_workloads"]                  # the other BASELINE configurations, each a short child run
    assert len(w) == 6 and all("error" not in e and e["value"] > 0 and 0 < e["hbm_frac"] < 1 and e["int_mul_frac"] > 0 for e in w), w
    assert w[0]["unit"] == "ct-mul/s" and w[0]["verified"] is True and w[4]["unit"] == "extprod/s"
    lat = d["extra_latency"]                  # one call on one polynomial / ciphertext / accumulator (the few-ciphertext forms)
    assert len(lat) == 4 and all("error" not in e and 1.0 < e["us_per_call"] < 400.0 for e in lat), lat
    assert lat[0]["us_per_call"] < 20.0 and lat[1]["us_per_call"] < 60.0 and lat[2]["us_per_call"] < 60.0, lat     # round 2: 24 / 97 / 103 us
    assert x[(250, 1)]["secondary"]["multiply_class_per_butterfly"] == 136 and x[(128, 1)]["secondary"]["multiply_class_per_butterfly"] == 136
    assert x[(250, 1)]["width_class"] == 4 and x[(40, 3)]["width_class"] == 3


@pytest.mark.gpu
def test_gpus_2_spawns_two_ranks_and_verifies_both_shards():
    """`bench.py --gpus 2` with no launcher around it starts the two ranks itself (here both on device 0 over gloo: the one-GPU
    rehearsal of the driver's multi-GPU run) and reports both ranks' checksums, each equal to the CPU oracle's."""
    d = _run("--gpus", "2", "--dist-backend", "gloo", "--device-override", "0", "--batch", "256", "--steps", "3", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["verified"] is True
    assert len(d["shard_checksums"]) == 2 and d["shard_checksums"] == d["oracle_checksums"]
    assert d["shard_checksums"][0] != d["shard_checksums"][1]            # every rank works on its own data
    assert d["config"]["batch_per_gpu"] == 256 and d["config"]["parallelism"] == "batch-shard x2"
    assert abs(d["value"] - 2 * 256 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d and "gloo" in d["dist_backend"]


@pytest.mark.gpu
def test_one_rank_rccl_group_carries_the_barriers_and_gathers():
    """FHE_BENCH_FORCE_DIST=1: the same RCCL calls the N > 1 run makes (init, barrier, all_reduce MAX, all_gather of the verification
    words on device tensors) with a one-rank group on the one GPU of this box."""
    env = dict(os.environ, FHE_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64", "--no-cpu-baseline", "--no-extras"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["verified"] is True and "nccl" in d["dist_backend"]


def test_gpus_n_without_a_device_fails_loudly():
    """CPU container: the launcher starts N ranks, they find no HIP device, the parent exits non-zero (no silent 1-rank line)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0",
                          "--batch", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    import importlib
    pkg = importlib.import_module("gpu-homomorphic-encryption_amd")
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present; covered by the gpu-marked test")
    assert res.returncode != 0 and not [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert "ranks exited" in res.stderr or "no HIP device" in res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("op,unit", [("ctrelin", "ct-mul/s"), ("blindrotate", "extprod/s"), ("fwdinv", "ntt-pair/s")])
def test_other_workloads_report_their_own_units(op, unit):
    d = _run("--steps", "2", "--warmup", "1", "--batch", "16", "--op", op, "--no-cpu-baseline")
    assert d["unit"] == unit and d["value"] > 0 and d["config"]["op"] == op and "cpu_baseline" not in d
    r = d["roofline"]; S = 32 * 8192 * 4
    assert r["secondary"]["bound"] == "valu-int-mul" and r["secondary"]["frac"] > 0
    # the fraction credits only the bytes the CALL must move; the accounting of earlier rounds is a side field
    want = {"ctrelin": 6 * S * 16, "blindrotate": 4 * S * 16, "fwdinv": 4 * S * 16}[op]
    assert r["algorithmic_bytes_per_launch"] == want and abs(r["frac"] - want / (r["launch_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-9
    if op == "ctrelin":
        assert r["legacy_accounting"]["bytes"] == 12 * S * 16 and d["verified"] is True       # sampled ciphertexts == oracle ct_multiply + relinearize
    if op == "blindrotate":
        assert r["legacy_accounting"]["bytes"] == 4 * 8 * S * 16


@pytest.mark.gpu
def test_limb_shard_rehearsal_two_ranks_union_is_the_product():
    """--gpus 2 --shard limb on the one GPU (gloo): rank 0 computes limbs {0, 2}, rank 1 limbs {1, 3} of the same global polynomials on
    engines built on those prime subsets; each rank's checksum equals the oracle's for ITS limbs, no collective carries payload."""
    d = _run("--gpus", "2", "--shard", "limb", "--dist-backend", "gloo", "--device-override", "0", "--batch", "64", "--steps", "3", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["verified"] is True and d["scaling"] == "strong"
    assert d["shard_checksums"] == d["oracle_checksums"] and d["shard_checksums"][0] != d["shard_checksums"][1]
    assert d["config"]["parallelism"].startswith("limb-shard x2") and d["config"]["limbs_of_rank0"] == [0, 2]
    assert abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6          # the ranks jointly produce 64 whole products per step
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 3 * 32 * 8192 * 2 * 64
