#!/bin/bash
# Wider SQ counter set (three passes) for one bench.py workload.  usage: scripts/profile_sq_detail.sh <tag> <bench args...>
set -u
TAG=$1; shift
ARGS="$*"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
P1="SQ_WAVE_CYCLES SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_IFETCH"
P2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"
P3="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify $ARGS > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; }
done
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, collections, sys, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "rocclr" in k or "pack_keys" in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print("==", k[:110])
    for c in sorted(cs):
        v = cs[c]
        print(f"   {c:28s} {sum(v)/len(v):14.5g}  (n={len(v)})")
PY
find "$OUT" -name "*.csv" -size +6M -delete
