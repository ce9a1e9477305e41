#!/bin/bash
# Kernel trace of the one-call ciphertext multiply and of the polymul at batch 1 (N = 8192, 4 x 30-bit, w = 16): which launch takes how long.
# usage (GPU box): scripts/trace_small_ctrelin.sh OUTDIR
cd "$(dirname "$0")/.." || exit 1
OUT=${1:-gpurun_out/trace_small}; rm -rf "$OUT"; mkdir -p "$OUT"
for op in ctrelin multiply; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$op" -- python3 bench.py --no-extra-workloads --no-cpu-baseline --op $op --batch 1 --steps 300 --warmup 20 > "$OUT/$op.log" 2>&1 || exit 1
  python3 - "$OUT/$op" "$op" <<'PY'
import csv, glob, os, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=lambda p: -os.path.getsize(p))[0]
print("==", sys.argv[2], "batch 1")
for r in list(csv.DictReader(open(f)))[:10]:
    print(f'{r["Name"][:90]:92s} calls {r["Calls"]:>6s}  avg {float(r["AverageNs"]) / 1000:7.2f} us')
PY
done
