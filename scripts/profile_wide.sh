#!/bin/bash
# Profiles the full-width (FHE_WIDTH_256) class on the GPU box: kernel trace, FETCH_SIZE / WRITE_SIZE passes and one SQ pass
# for `--op multiply` and `--op fwdinv`.  Usage: scripts/profile_wide.sh <tag> [bench args, default: --bits 64 --limbs 2 --batch 256]
# Output: gpurun_out/<tag>/{multiply,fwdinv}_{trace,pmc_fetch,pmc_write,sq}/ + a text summary gpurun_out/<tag>/summary.txt
set -u
TAG=${1:-r02_u256}
shift || true
ARGS=${*:---bits 64 --limbs 2 --batch 256}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rc=0
for op in multiply fwdinv; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${op}_trace" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --op $op $ARGS > "$OUT/${op}_trace.log" 2>&1 \
  && rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${op}_pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --op $op $ARGS > "$OUT/${op}_pmc_fetch.log" 2>&1 \
  && rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${op}_pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --op $op $ARGS > "$OUT/${op}_pmc_write.log" 2>&1 \
  && rocprofv3 --pmc $SQ --output-format csv -d "$OUT/${op}_sq" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --op $op $ARGS > "$OUT/${op}_sq.log" 2>&1 \
  || { rc=1; break; }
done
echo "profile rc=$rc"
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, collections, json, sys, os
out = sys.argv[1]
for op in ("multiply", "fwdinv"):
    line = None
    p = os.path.join(out, op + "_trace.log")
    if os.path.exists(p):
        for l in open(p):
            if l.startswith("{"): line = json.loads(l)
    if line:
        r = line["roofline"]
        print(f"== {op}: {line['value']:.1f} {line['unit']}  {line['ms_per_step']:.4f} ms/step  algorithmic {r['algorithmic_bytes_per_launch']} B/step  {r['achieved']:.1f} GB/s  frac {r['frac']:.4f}")
    for f in glob.glob(os.path.join(out, op + "_trace", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            print(f"   {r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Percentage']:>6s} %")
    tot = {}
    for kind, key, mul in (("fetch", "FETCH_SIZE", 2048.0), ("write", "WRITE_SIZE", 1024.0)):
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(out, f"{op}_pmc_{kind}", "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == key: acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * mul)
        for k, v in acc.items():
            print(f"   {key:10s} {k[:70]:70s} launches {len(v):4d}  bytes/launch {sum(v)/len(v):.4g}  total {sum(v):.4g}")
            tot[kind] = tot.get(kind, 0) + sum(v)
    if tot and line:
        steps = 3     # --steps 2 --warmup 1
        per_step = sum(tot.values()) / steps
        print(f"   HBM bytes per step (FETCH x2 + WRITE, all kernels): {per_step:.4g} = {per_step / line['roofline']['algorithmic_bytes_per_launch']:.2f} x algorithmic")
    for f in glob.glob(os.path.join(out, op + "_sq", "*", "*_counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            m = {c: sum(v) / len(v) for c, v in cs.items()}
            wc = max(1.0, m.get("SQ_WAVE_CYCLES", 1))
            print(f"   SQ {k[:60]:60s} wait_any {m.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} "
                  f"valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} insts_valu {m.get('SQ_INSTS_VALU',0):.4g} lds_conf/active {m.get('SQ_LDS_BANK_CONFLICT',0)/max(1,m.get('SQ_LDS_IDX_ACTIVE',1)):.3f}")
PY
find "$OUT" -name "*.csv" -size +6M -delete
exit $rc
