#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration per access shape: runs scratch/fetch_calib (every kernel moves exactly
# BYTES of compulsory HBM traffic per launch) under rocprofv3 --pmc and prints counter / BYTES for each shape.
# usage (GPU box): scripts/fetch_calibration.sh [GiB]   ->  gpurun_out/fetch_calib/summary.txt (copy to profiles/r03_fetch_calibration.txt)
set -u
GIB=${1:-2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fetch_calib
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
[ -x scratch/fetch_calib ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o scratch/fetch_calib scratch/fetch_calib.hip || exit 1
./scratch/fetch_calib $GIB > "$OUT/plain.log" 2>&1 \
&& rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- ./scratch/fetch_calib $GIB > "$OUT/pmc_fetch.log" 2>&1 \
&& rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- ./scratch/fetch_calib $GIB > "$OUT/pmc_write.log" 2>&1
echo "calibration rc=$?"
python3 - "$OUT" "$GIB" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, collections, os, sys
out, gib = sys.argv[1], int(sys.argv[2])
BYTES = gib << 30
print(f"# FETCH_SIZE / WRITE_SIZE calibration on gfx950: every kernel touches each line of a {gib} GiB buffer exactly once (compulsory traffic = {BYTES} B per launch)")
print("# factor = BYTES / (counter x 1024): what the counter must be multiplied by to give bytes for that access shape")
for l in open(os.path.join(out, "plain.log")):
    if "GB/s" in l: print("# rate  " + l.rstrip())
res = {}
for kind, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, f"pmc_{kind}", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == key:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        if k.startswith("__amd"): continue
        if (kind == "fetch") != k.startswith(("rd", "buf")): continue
        kib = sum(v) / len(v)
        print(f"{key:10s} {k:18s} launches {len(v)}  counter {kib:14.1f} KiB  = {kib * 1024 / BYTES:.4f} of BYTES  -> factor {BYTES / (kib * 1024):.4f}")
PY
find "$OUT" -name "*.csv" -size +4M -delete
