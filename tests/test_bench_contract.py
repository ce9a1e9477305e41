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
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["unit"] == "polymul/s" and b["value"] > 0 and b["cores"] >= 1 and "sample" in b
    assert b["narrow_port"]["value"] > b["value"]
    assert d["extra_fwd_inv_pairs"]["pairs_per_s"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("op,unit", [("ctrelin", "ct-mul/s"), ("blindrotate", "extprod/s"), ("fwdinv", "ntt-pair/s")])
def test_other_workloads_report_their_own_units(op, unit):
    d = _run("--steps", "2", "--warmup", "1", "--batch", "16", "--op", op, "--no-cpu-baseline")
    assert d["unit"] == unit and d["value"] > 0 and d["config"]["op"] == op and "cpu_baseline" not in d
    assert d["roofline"]["limiter"] in ("hbm", "valu-int32-multiply")
