"""Builds lib/libfhe_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import fcntl
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB = os.path.join(_HERE, "lib", "libfhe_hip.so")


def library_path():
    return _LIB


def _sources():
    out = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)]
    out.append(os.path.join(os.path.dirname(_HERE), "include", "fhe_hip.h"))
    return [p for p in out if os.path.isfile(p)]


def build_library(force=False, jobs=None, verbose=False):
    """make -C csrc.  Skips the build when the library is newer than every source (the GPU box has
    the prebuilt .so from the snapshot and need not rebuild)."""
    def fresh():
        return os.path.exists(_LIB) and os.path.getmtime(_LIB) >= max(os.path.getmtime(s) for s in _sources())

    if not force and fresh():
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    with open(os.path.join(os.path.dirname(_LIB), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)       # ranks of one job (or parallel test workers) build once, the others wait
        try:
            if not force and fresh():
                return _LIB
            jobs = jobs or min(8, os.cpu_count() or 1)
            cmd = ["make", "-C", _CSRC, f"-j{jobs}"]
            if force:
                cmd.append("-B")
            res = subprocess.run(cmd, capture_output=not verbose, text=True)
            if res.returncode != 0:
                raise RuntimeError("libfhe_hip.so build failed:\n" + (res.stdout or "") + (res.stderr or ""))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB
