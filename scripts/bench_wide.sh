#!/bin/bash
OUT=gpurun_out/matrix_wide.jsonl; : > $OUT
run() { echo "== $*" >&2; timeout -k 10 600 python bench.py --no-cpu-baseline "$@" | grep '^{' >> $OUT || echo "FAILED: $*" >&2; }
run --steps 3 --warmup 1 --op multiply --batch 128 --bits 64 --limbs 2
run --steps 3 --warmup 1 --op multiply --batch 64 --bits 64 --limbs 4 --n 4096
run --steps 3 --warmup 1 --op fwdinv --batch 128 --bits 64 --limbs 2
python - <<PY
import json
for l in open("$OUT"):
    d=json.loads(l); c=d["config"]; r=d["roofline"]
    print(f'{c["op"]:9s} N={c["n"]:6d} L={c["limbs"]} bits={c["prime_bits"]:3d} B={c["batch_per_gpu"]:5d} {d["dtype"][:5]:5s} {d["value"]:12.1f} {d["unit"]:10s} {d["ms_per_step"]:9.4f} ms  {r["achieved"]:8.1f} GB/s  frac {r["frac"]:.3f}')
PY
