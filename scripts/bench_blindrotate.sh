#!/bin/bash
# Blind-rotation inner loop (BASELINE configs[4] shape on one GPU): fused one-launch steps vs the general composition.
TAG=${1:-r01}
OUT=gpurun_out/blindrotate_$TAG.jsonl
: > $OUT
run() { echo "== $*" >&2; timeout -k 10 600 python bench.py --no-cpu-baseline --op blindrotate "$@" | grep '^{' >> $OUT || echo "FAILED: $*" >&2; }
run --steps 5 --warmup 1 --batch 1024 --n 8192  --limbs 4 --bits 30
run --steps 5 --warmup 1 --batch 128  --n 16384 --limbs 6 --bits 30
run --steps 5 --warmup 1 --batch 512  --n 16384 --limbs 6 --bits 30
run --steps 5 --warmup 1 --batch 128  --n 16384 --limbs 6 --bits 30 --decomp-bits 30
run --steps 3 --warmup 1 --batch 128  --n 16384 --limbs 6 --bits 40
FHE_HIP_NO_FUSED_BLIND_ROTATE=1 run --steps 3 --warmup 1 --batch 1024 --n 8192  --limbs 4 --bits 30
FHE_HIP_NO_FUSED_BLIND_ROTATE=1 run --steps 3 --warmup 1 --batch 128  --n 16384 --limbs 6 --bits 30
python - <<PY
import json
for l in open("$OUT"):
    d=json.loads(l); c=d["config"]; r=d["roofline"]
    print(f'{c["op"]:11s} N={c["n"]:6d} L={c["limbs"]} bits={c["prime_bits"]:3d} B={c["batch_per_gpu"]:5d} {d["dtype"][:5]:5s} {d["value"]:12.1f} {d["unit"]:10s} {d["ms_per_step"]:9.4f} ms  {r["achieved"]:8.1f} GB/s  frac {r["frac"]:.3f}')
PY
