#!/bin/bash
# A/B: key switching / external product with digit transforms two at a time vs one at a time (FHE_HIP_NO_PAIRED_TRANSFORMS=1), same box.
TAG=${1:-r01}
OUT=gpurun_out/ab_paired_$TAG.jsonl
: > $OUT
run() { echo "== $*" >&2; timeout -k 10 600 python bench.py --no-cpu-baseline "$@" | grep '^{' >> $OUT || echo "FAILED: $*" >&2; }
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export FHE_HIP_NO_PAIRED_TRANSFORMS=1; else unset FHE_HIP_NO_PAIRED_TRANSFORMS; fi
  run --steps 10 --warmup 2 --op relin --batch 1024
  run --steps 5 --warmup 1 --op blindrotate --batch 1024
  run --steps 10 --warmup 2 --op relin --batch 128 --n 16384 --limbs 6
  run --steps 5 --warmup 1 --op blindrotate --batch 128 --n 16384 --limbs 6
done
python - <<PY
import json
for i, l in enumerate(open("$OUT")):
    d=json.loads(l); c=d["config"]; r=d["roofline"]
    print(f'{"PAIR" if (i // 4) % 2 == 0 else "ONE "} {c["op"]:11s} N={c["n"]:6d} L={c["limbs"]} B={c["batch_per_gpu"]:5d} {d["value"]:12.1f} {d["unit"]:10s} {d["ms_per_step"]:9.4f} ms  {r["achieved"]:8.1f} GB/s  frac {r["frac"]:.3f}')
PY
