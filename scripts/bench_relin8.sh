#!/bin/bash
OUT=gpurun_out/matrix_relin8.jsonl; : > $OUT
run() { echo "== $ENVV $*" >&2; timeout -k 10 600 env $ENVV python bench.py --no-cpu-baseline "$@" | grep '^{' >> $OUT || echo "FAILED: $*" >&2; }
for ENVV in "X=1" "FHE_HIP_NO_FUSED_KEYSWITCH=1"; do
run --steps 5 --warmup 1 --op relin   --batch 128 --n 16384 --limbs 6 --bits 40
run --steps 5 --warmup 1 --op relin   --batch 512 --bits 40 --limbs 3
run --steps 5 --warmup 1 --op relin   --batch 512 --bits 60 --limbs 2
done
ENVV="X=1"
run --steps 5 --warmup 1 --op ctrelin --batch 128 --n 16384 --limbs 6 --bits 40
run --steps 10 --warmup 2 --op relin   --batch 1024
python - <<PY
import json
for l in open("$OUT"):
    d=json.loads(l); c=d["config"]; r=d["roofline"]
    print(f'{c["op"]:9s} N={c["n"]:6d} L={c["limbs"]} bits={c["prime_bits"]:3d} B={c["batch_per_gpu"]:5d} {d["dtype"][:5]:5s} {d["value"]:12.1f} {d["unit"]:10s} {d["ms_per_step"]:9.4f} ms  {r["achieved"]:8.1f} GB/s  frac {r["frac"]:.3f}')
PY
