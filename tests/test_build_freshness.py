"""Build robustness (CPU): what makes the library stale, and what must not.

* make's view: every header an object includes is a prerequisite of it -- editing ntt_wide.hip.h or the generated wide_asm.inc
  re-compiles fhe_hip.o (the round-2 Makefile left a stale library behind), editing the LDS transform kernels re-compiles the
  lds_inst objects and leaves fhe_hip.o alone.
* build.py's view: freshness is a content hash in lib/.build_stamp, so a copy of the tree with every mtime reset (the GPU box's
  snapshot) does not rebuild, and a one-byte edit does."""
import importlib
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "gpu-homomorphic-encryption_amd")
CSRC = os.path.join(PKG_DIR, "csrc")


def _make_plan(touched):
    """Objects `make -n` would rebuild if `touched` (a file in csrc/) were newer than everything (-W = pretend it was just modified)."""
    res = subprocess.run(["make", "-C", CSRC, "-n", "-W", touched], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return sorted({tok.split("/")[-1] for line in res.stdout.splitlines() if " -o " in line for tok in [line.split(" -o ", 1)[1].split()[0]]})


@pytest.fixture(scope="module")
def built(pkg):
    pkg.build_library()
    return importlib.import_module(pkg.__name__ + ".build")


@pytest.mark.parametrize("header", ["wide_asm.inc", "ntt_wide.hip.h", "ntt_word.hip.h", "ntt256.hip.h", "ntt_field.hip.h"])
def test_editing_a_header_of_the_host_unit_rebuilds_it(built, header):
    plan = _make_plan(header)
    assert "fhe_hip.o" in plan and "libfhe_hip.so" in plan, plan


def test_editing_the_lds_kernels_rebuilds_every_instance_but_not_the_host_unit(built):
    plan = _make_plan("ntt_lds.hip.h")
    assert "fhe_hip.o" not in plan and "libfhe_hip.so" in plan
    assert sum(1 for o in plan if o.startswith("lds_F")) == 17, plan
    assert "fhe_hip.o" in _make_plan("ntt_field.hip.h") and "lds_F32_13.o" in _make_plan("ntt_field.hip.h")


def test_up_to_date_tree_has_nothing_to_do(built):
    res = subprocess.run(["make", "-C", CSRC, "-n"], capture_output=True, text=True)
    assert res.returncode == 0 and " -o " not in res.stdout, res.stdout[-2000:]
    assert built.is_fresh()
    built.build_library()
    assert built.last_build == {"compiled": [], "seconds": 0.0, "skipped": True}


def test_snapshot_with_reset_mtimes_is_fresh_and_an_edit_is_not(built, tmp_path):
    """Copy the package (sources + lib/) WITHOUT preserving mtimes, then set every mtime to one instant, as a snapshot tool may:
    the stamp still matches, so build_library() returns without calling make.  Appending a byte to wide_asm.inc flips it."""
    import importlib.util
    dst = tmp_path / "gpu-homomorphic-encryption_amd"
    shutil.copytree(PKG_DIR, dst, ignore=shutil.ignore_patterns("obj", "__pycache__", ".pytest_cache"), copy_function=shutil.copyfile)
    os.makedirs(tmp_path / "include"); shutil.copyfile(os.path.join(ROOT, "include", "fhe_hip.h"), tmp_path / "include" / "fhe_hip.h")
    for d, _, files in os.walk(tmp_path):
        for f in files:
            os.utime(os.path.join(d, f), (1_000_000_000, 1_000_000_000))
    spec = importlib.util.spec_from_file_location("build_copy", dst / "build.py")
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    assert b.source_hash() == built.source_hash()
    assert b.is_fresh()
    assert b.build_library() == str(dst / "lib" / "libfhe_hip.so") and b.last_build["skipped"] is True
    with open(dst / "csrc" / "wide_asm.inc", "a") as f:
        f.write("\n")
    assert not b.is_fresh()
