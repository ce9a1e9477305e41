#!/bin/bash
# Kernel trace + FETCH_SIZE / WRITE_SIZE + SQ counters for one bench.py workload.  usage: scripts/profile_op.sh <tag> <bench args...>
# Output: gpurun_out/<tag>/{trace,pmc_fetch,pmc_write,sq}/ and gpurun_out/<tag>/summary.txt
set -u
TAG=$1; shift
ARGS="$*"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-verify $ARGS > "$OUT/trace.log" 2>&1 \
&& rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify $ARGS > "$OUT/pmc_fetch.log" 2>&1 \
&& rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify $ARGS > "$OUT/pmc_write.log" 2>&1 \
&& rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify $ARGS > "$OUT/sq.log" 2>&1
echo "profile rc=$?"
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, collections, json, sys, os
out = sys.argv[1]
line = None
for l in open(os.path.join(out, "trace.log")):
    if l.startswith("{"): line = json.loads(l)
if line:
    r = line["roofline"]; c = line["config"]
    print(f"== {c['op']} N={c['n']} L={c['limbs']} {c['prime_bits']}-bit batch {c['batch_per_gpu']}: {line['value']:.1f} {line['unit']}  {line['ms_per_step']:.4f} ms/step  algorithmic {r['algorithmic_bytes_per_launch']} B/step  {r['achieved']:.1f} GB/s  frac {r['frac']:.4f}")
for f in glob.glob(os.path.join(out, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "rocclr" in r["Name"]: continue
        print(f"   {r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Percentage']:>6s} %")
for f in glob.glob(os.path.join(out, "trace", "*", "*_kernel_trace.csv")):
    seen = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k not in seen and "rocclr" not in k:
            seen[k] = r
            print(f"   resources {k[:70]:70s} VGPR {r.get('VGPR_Count')} accum {r.get('Accum_VGPR_Count')} SGPR {r.get('SGPR_Count')} LDS {r.get('LDS_Block_Size')} scratch {r.get('Scratch_Size')} wg {r.get('Workgroup_Size')} grid {r.get('Grid_Size')}")
tot = {}
for kind, key, mul in (("fetch", "FETCH_SIZE", 2048.0), ("write", "WRITE_SIZE", 1024.0)):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, f"pmc_{kind}", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == key and "rocclr" not in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * mul)
    for k, v in acc.items():
        print(f"   {key:10s} {k[:70]:70s} launches {len(v):4d}  bytes/launch {sum(v)/len(v):.4g}")
        tot[kind] = tot.get(kind, 0) + sum(v)
if tot and line:
    per_step = sum(tot.values()) / 3
    print(f"   HBM bytes per step (FETCH x2 + WRITE): {per_step:.4g} = {per_step / line['roofline']['algorithmic_bytes_per_launch']:.2f} x algorithmic")
for f in glob.glob(os.path.join(out, "sq", "*", "*_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "rocclr" in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        wc = max(1.0, m.get("SQ_WAVE_CYCLES", 1))
        print(f"   SQ {k[:60]:60s} wait_any {m.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} "
              f"valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} insts_valu {m.get('SQ_INSTS_VALU',0):.4g} lds_conf/active {m.get('SQ_LDS_BANK_CONFLICT',0)/max(1,m.get('SQ_LDS_IDX_ACTIVE',1)):.3f}")
PY
find "$OUT" -name "*.csv" -size +6M -delete
