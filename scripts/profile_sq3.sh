#!/bin/bash
# Where do the waves of a kernel wait?  Five --pmc passes (VALU / LDS / vector memory / instruction cache) over one bench.py workload.
# usage: scripts/profile_sq3.sh <tag> <bench args...>     -> gpurun_out/<tag>/sq3_summary.txt
set -u
TAG=$1; shift
ARGS="$*"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify $ARGS"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
P2="SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL"
P3="SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM"
P4="SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
P5="SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d "$OUT/sq3_$i" -- $B > "$OUT/sq3_$i.log" 2>&1 || echo "pass $i failed (see $OUT/sq3_$i.log)"
done
python3 - "$OUT" <<'PY' | tee "$OUT/sq3_summary.txt"
import csv, glob, collections, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "sq3_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "rocclr" in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    wc = max(1.0, m.get("SQ_WAVE_CYCLES", 1))
    print(f"== {k[:110]}")
    for c in sorted(m):
        print(f"   {c:24s} {m[c]:16.4g}   / wave-cycles {m[c] / wc:8.4f}")
PY
find "$OUT" -name "*.csv" -size +6M -delete
