#!/bin/bash
# Development aid: recompile ONLY the named objects (e.g. fhe_hip lds_F32_13 lds_F52_14) and relink lib/libfhe_hip.so from whatever
# other objects are already in lib/obj, then mark the library fresh (lib/.build_stamp) so that the GPU box does not spend minutes on
# the stale instances.  Only valid while the launch ABI between objects (lds_launch.h) is unchanged.  ALWAYS finish with a plain
# `make -C gpu-homomorphic-encryption_amd/csrc -j8` before committing a measurement.
set -e
cd "$(dirname "$0")/../gpu-homomorphic-encryption_amd/csrc"
targets=""
for t in "$@"; do targets="$targets ../lib/obj/$t.o"; done
[ -n "$targets" ] && make -j8 $targets
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libfhe_hip.so ../lib/obj/*.o
python3 - <<'PY'
import importlib.util, os
spec = importlib.util.spec_from_file_location("b", os.path.join("..", "build.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
open(b._STAMP, "w").write(b.source_hash() + "\n(dev_relink: not every object was rebuilt)\n")
print("relinked; stamp", b.source_hash()[:16])
PY
