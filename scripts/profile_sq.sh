#!/bin/bash
# SQ counters (one pass, 8 SQ slots) for the fused multiply and the key-switch kernel: LDS bank conflicts and where waves wait.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_sq
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
CNT="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
rocprofv3 --pmc $CNT --output-format csv -d "$OUT/multiply" -- python3 bench.py --steps 2 --warmup 1 --batch 1024 --no-cpu-baseline > "$OUT/multiply.log" 2>&1 \
 && rocprofv3 --pmc $CNT --output-format csv -d "$OUT/ctrelin" -- python3 bench.py --steps 2 --warmup 1 --batch 1024 --op ctrelin --no-cpu-baseline > "$OUT/ctrelin.log" 2>&1 \
 && rocprofv3 --pmc $CNT --output-format csv -d "$OUT/fwdinv" -- python3 bench.py --steps 2 --warmup 1 --batch 1024 --op fwdinv --no-cpu-baseline > "$OUT/fwdinv.log" 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob, collections
for op in ("multiply", "ctrelin", "fwdinv"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % op):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "ntt_" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("<")[0].split("::")[-1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            m = {c: sum(v) / len(v) for c, v in cs.items()}
            wc = m.get("SQ_WAVE_CYCLES", 1)
            print(f"{op:8s} {k:28s} LDS conflict/active = {m.get('SQ_LDS_BANK_CONFLICT',0)/max(1,m.get('SQ_LDS_IDX_ACTIVE',1)):.4f}  "
                  f"wait_any {m.get('SQ_WAIT_ANY',0)/wc:.2f}  wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:.2f}  active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f}  "
                  f"valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f}  lds {m.get('SQ_ACTIVE_INST_LDS',0)/wc:.2f}")
PY
find "$OUT" -name "*.csv" -size +4M -delete
