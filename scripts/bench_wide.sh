#!/bin/bash
# Full-width class (FHE_WIDTH_256) bench lines: 128-bit and 250-bit primes, multiply and forward+inverse.  usage: scripts/bench_wide.sh OUT.jsonl
cd "$(dirname "$0")/.." || exit 1
OUT=${1:-gpurun_out/wide.jsonl}; : > "$OUT"
for bits in 250 120 128; do
  L=2
  for op in multiply fwdinv; do
    python bench.py --no-extra-workloads --steps 5 --warmup 2 --op $op --batch 128 --bits $bits --limbs $L 2>/dev/null | tail -1 >> "$OUT"
  done
done
python3 - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); c = d["config"]
    print(c["workload"][:70], round(d["value"]), d["roofline"]["frac"], d["roofline"].get("secondary", {}).get("frac"))
PY
