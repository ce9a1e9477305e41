#!/bin/bash
# Second SQ counter set for the key-switch / external-product kernels: instruction counts per class and cycles per VALU instruction.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_sq2
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1
grep -o "SQ_[A-Z0-9_]*" "$OUT/avail.txt" | sort -u > "$OUT/sq_names.txt"
CNT1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
CNT2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VALU"
rocprofv3 --pmc $CNT1 --output-format csv -d "$OUT/relin1" -- python3 bench.py --steps 2 --warmup 1 --batch 1024 --op relin --no-cpu-baseline > "$OUT/relin1.log" 2>&1 \
 && rocprofv3 --pmc $CNT2 --output-format csv -d "$OUT/relin2" -- python3 bench.py --steps 2 --warmup 1 --batch 1024 --op relin --no-cpu-baseline > "$OUT/relin2.log" 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob, collections
for op in ("relin1", "relin2"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % op):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "ntt_" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("<")[0].split("::")[-1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(op, k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
find "$OUT" -name "*.csv" -size +4M -delete
