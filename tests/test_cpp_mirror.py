"""Builds tests/cpp/test_fhe_mirror.cpp against include/fhe/*.hpp + libfhe_hip.so with g++ (the host
side of the reference is compiled C++, so its mirror is too) and runs it: host-only on CPU, the full
reference scenarios on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_fhe_mirror.cpp")
OUT_DIR = os.path.join(ROOT, "tests", "cpp", "_build")
EXE = os.path.join(OUT_DIR, "test_fhe_mirror")


def _build(pkg):
    pkg.build_library()
    lib_dir = os.path.dirname(pkg.library_path())
    os.makedirs(OUT_DIR, exist_ok=True)
    deps = [SRC] + [os.path.join(ROOT, "include", "fhe", f) for f in os.listdir(os.path.join(ROOT, "include", "fhe"))]
    deps.append(os.path.join(ROOT, "include", "fhe_hip.h"))
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), SRC, "-L", lib_dir, "-lfhe_hip",
           f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return EXE


def test_cpp_mirror_compiles_and_host_maths_pass(pkg):
    exe = _build(pkg)
    res = subprocess.run([exe, "--host-only"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "host-only: PASSED" in res.stdout


@pytest.mark.gpu
def test_cpp_mirror_reference_scenarios_on_gpu(pkg):
    exe = _build(pkg)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "ALL PASSED" in res.stdout


GRAPH_SRC = os.path.join(ROOT, "tests", "cpp", "test_graph_capture.cpp")
GRAPH_EXE = os.path.join(OUT_DIR, "test_graph_capture")


def _build_graph_test(pkg):
    pkg.build_library()
    lib_dir = os.path.dirname(pkg.library_path())
    os.makedirs(OUT_DIR, exist_ok=True)
    if os.path.exists(GRAPH_EXE) and os.path.getmtime(GRAPH_EXE) >= max(os.path.getmtime(GRAPH_SRC), os.path.getmtime(os.path.join(ROOT, "include", "fhe_hip.h"))):
        return GRAPH_EXE
    cmd = ["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), GRAPH_SRC, "-L", lib_dir, "-lfhe_hip",
           f"-Wl,-rpath,{lib_dir}", "-o", GRAPH_EXE]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return GRAPH_EXE


def test_graph_capture_test_compiles(pkg):
    _build_graph_test(pkg)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(), ("30", "8192", "1"), ("40", "16384", "2")], ids=["n8192x30-batch8", "n8192x30-batch1", "n16384x40-batch2"])
def test_engine_launches_are_graph_capturable(pkg, shape):
    """Tensor product + relinearisation, the one-call multiply (after fhe_rns_ntt_reserve) and a blind-rotation loop captured into hipGraphs on a
    caller-owned stream and replayed; batch 1 takes the few-ciphertext forms, the 40-bit primes the three-array kernels of the 8-byte residues."""
    exe = _build_graph_test(pkg)
    res = subprocess.run([exe, *shape], capture_output=True, text=True, timeout=300)
    print(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "graph capture ok" in res.stdout
