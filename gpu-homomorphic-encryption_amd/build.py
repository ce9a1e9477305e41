"""Builds lib/libfhe_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

Freshness is decided by CONTENT, not by modification times: `lib/.build_stamp` holds a SHA-256 over every source the library is
built from (csrc/*, include/fhe_hip.h) recorded when the library was last built.  A snapshot of the tree with every mtime reset
(the GPU box's copy) therefore does not rebuild -- a clean build is ~7 minutes of 8 cores -- while any edit, including to the
generated csrc/wide_asm.inc, does.  Inside `make` the per-object dependency files (obj/*.d, written by -MMD) decide WHAT is recompiled."""
import fcntl
import hashlib
import os
import subprocess
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB = os.path.join(_HERE, "lib", "libfhe_hip.so")
_STAMP = os.path.join(_HERE, "lib", ".build_stamp")
_SRC_EXT = (".hip", ".h", ".hpp", ".cpp", ".inc")

last_build = None      # {"compiled": [...object names...], "seconds": float, "skipped": bool} of the most recent build_library() call


def library_path():
    return _LIB


def _sources():
    out = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith(_SRC_EXT) or f == "Makefile"]
    out.append(os.path.join(os.path.dirname(_HERE), "include", "fhe_hip.h"))
    return [p for p in out if os.path.isfile(p)]


def source_hash():
    """SHA-256 over (relative name, content) of every source of the library, in a fixed order."""
    h = hashlib.sha256()
    for p in _sources():
        h.update(os.path.relpath(p, _HERE).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def is_fresh():
    """True when the library exists and was built from exactly the sources now on disk."""
    if not (os.path.exists(_LIB) and os.path.exists(_STAMP)):
        return False
    try:
        with open(_STAMP) as f:
            return f.read().split()[0] == source_hash()
    except (OSError, IndexError):
        return False


def build_library(force=False, jobs=None, verbose=False):
    """make -C csrc, unless lib/.build_stamp says the library already corresponds to the sources (the GPU box has the prebuilt .so
    from the snapshot and must not spend its first minutes recompiling).  Records what was compiled in `last_build`."""
    global last_build
    if not force and is_fresh():
        last_build = {"compiled": [], "seconds": 0.0, "skipped": True}
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    with open(os.path.join(os.path.dirname(_LIB), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)       # ranks of one job (or parallel test workers) build once, the others wait
        try:
            if not force and is_fresh():
                last_build = {"compiled": [], "seconds": 0.0, "skipped": True}
                return _LIB
            want = source_hash()
            jobs = jobs or min(8, os.cpu_count() or 1)
            cmd = ["make", "-C", _CSRC, f"-j{jobs}"]
            if force:
                cmd.append("-B")
            t0 = time.time()
            res = subprocess.run(cmd, capture_output=True, text=True)
            dt = time.time() - t0
            if verbose:
                print(res.stdout + res.stderr)
            if res.returncode != 0:
                raise RuntimeError("libfhe_hip.so build failed:\n" + (res.stdout or "") + (res.stderr or ""))
            compiled = [tok for line in res.stdout.splitlines() if " -o " in line for tok in [line.split(" -o ", 1)[1].split()[0]]]
            last_build = {"compiled": [os.path.basename(c) for c in compiled], "seconds": dt, "skipped": False}
            if source_hash() == want:          # nobody edited a source while make ran
                with open(_STAMP, "w") as f:
                    f.write(want + "\n")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB
