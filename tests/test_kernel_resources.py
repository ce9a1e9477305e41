"""CPU test (hipcc cross-compiles without a GPU): register / LDS budgets of the headline kernel instance (F32, N = 8192).
A silent spill or a lost occupancy step is a performance regression that no parity test sees; the numbers asserted here are
the ones DESIGN.md section 4 argues from (4 workgroups per CU for the fused multiply, 2 for the tensor product and the
key-switch / external-product kernels, no scratch)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpu-homomorphic-encryption_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def resources(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path_factory.mktemp("res")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-DFHE_FIELD=F32", "-DFHE_LOGN=13",
           "-Rpass-analysis=kernel-resource-usage", "-c", "-o", str(out / "x.o"), "lds_inst.hip"]
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    shutil.rmtree(out, ignore_errors=True)
    kernels, cur = {}, None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return kernels


def _all(kernels, needle, expect):
    """Every instantiated variant of a kernel (e.g. the general and the squaring form)."""
    hits = [v for k, v in kernels.items() if needle in k]
    assert len(hits) == expect, (needle, [k for k in kernels if needle in k])
    return hits


def test_streaming_kernels_keep_four_workgroups_per_cu(resources):
    for name, variants in (("ntt_multiply_kernel", 2), ("ntt_forward_kernel", 1), ("ntt_inverse_kernel", 1)):
        for k in _all(resources, name, variants):
            assert k["spill"] == 0 and k.get("scratch", 0) == 0, (name, k)
            assert k["vgprs"] <= 128 and k["occupancy"] >= 4, (name, k)    # 4 waves per SIMD = 4 workgroups of 256 threads per CU
            assert k["lds"] == 33792, (name, k)                             # (8192 + 8192 / 32) * 4 bytes: 4 x 33 KiB <= 160 KiB


def test_compute_bound_kernels_do_not_spill(resources):
    for name, variants, lds in (("ntt_ct_multiply_kernel", 2, 33792), ("ntt_keyswitch2_kernel", 1, 2 * 33792), ("ntt_extprod2_kernel", 1, 2 * 33792)):
        for k in _all(resources, name, variants):
            assert k["spill"] == 0 and k.get("scratch", 0) == 0, (name, k)
            assert k["vgprs"] <= 256 and k["occupancy"] >= 2, (name, k)    # 2 waves per SIMD = 2 workgroups per CU
            assert k["lds"] == lds, (name, k)                               # 2 x 66 KiB <= 160 KiB
