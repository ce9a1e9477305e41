#!/bin/bash
# The shapes round 3 works on, one JSON line each.  usage: scripts/bench_focus.sh <tag> [group ...]
# groups: ks32 (4-byte key switch / external product), ks52 (8-byte fields at N = 2^14), big (two-pass sizes), small (batch 1 .. 64), wide
TAG=${1:-focus}; shift
GROUPS_=${*:-ks32 ks52 big small wide}
OUT=gpurun_out/focus_$TAG.jsonl
: > $OUT
run() { echo "== $*" >&2; timeout -k 10 600 python bench.py --no-cpu-baseline --no-extras --no-verify "$@" | grep '^{' >> $OUT || echo "FAILED: $*" >&2; }
for g in $GROUPS_; do case $g in
ks32)
  run --steps 10 --warmup 2 --op relin    --batch 1024
  run --steps 10 --warmup 2 --op relin    --batch 1024 --decomp-bits 30
  run --steps 10 --warmup 2 --op ctrelin  --batch 1024
  run --steps 10 --warmup 2 --op ctrelin  --batch 1024 --decomp-bits 30
  run --steps 5  --warmup 1 --op blindrotate --batch 1024
  run --steps 10 --warmup 2 --op ctrelin  --batch 128 --n 16384 --limbs 6 --bits 30
  run --steps 5  --warmup 1 --op blindrotate --batch 128 --n 16384 --limbs 6
  run --steps 5  --warmup 1 --op blindrotate --batch 128 --n 16384 --limbs 6 --decomp-bits 30 ;;
ks52)
  run --steps 5  --warmup 1 --op relin    --batch 128 --n 16384 --limbs 6 --bits 40
  run --steps 5  --warmup 1 --op ctrelin  --batch 128 --n 16384 --limbs 6 --bits 40
  run --steps 5  --warmup 1 --op ctrelin  --batch 128 --n 16384 --limbs 6 --bits 40 --decomp-bits 20
  run --steps 5  --warmup 1 --op blindrotate --batch 128 --n 16384 --limbs 6 --bits 40
  run --steps 5  --warmup 1 --op relin    --batch 512 --bits 40 --limbs 3 --decomp-bits 20
  run --steps 5  --warmup 1 --op relin    --batch 512 --bits 60 --limbs 2 --decomp-bits 32
  run --steps 5  --warmup 1 --op relin    --batch 128 --n 16384 --limbs 3 --bits 60 --decomp-bits 32 ;;
big)
  run --steps 5  --warmup 1 --op fwdinv   --batch 512 --n 65536 --limbs 4 --bits 30
  run --steps 5  --warmup 1 --op multiply --batch 512 --n 65536 --limbs 4 --bits 30
  run --steps 5  --warmup 1 --op fwdinv   --batch 512 --n 32768 --limbs 3 --bits 40
  run --steps 5  --warmup 1 --op multiply --batch 512 --n 32768 --limbs 3 --bits 40
  run --steps 5  --warmup 1 --op multiply --batch 512 --n 32768 --limbs 2 --bits 60
  run --steps 5  --warmup 1 --op ct       --batch 128 --n 65536 --limbs 4 --bits 30 ;;
small)
  run --steps 50 --warmup 5 --op multiply --batch 1
  run --steps 50 --warmup 5 --op multiply --batch 4
  run --steps 50 --warmup 5 --op multiply --batch 16
  run --steps 20 --warmup 3 --op multiply --batch 64
  run --steps 20 --warmup 3 --op ctrelin  --batch 1
  run --steps 20 --warmup 3 --op ctrelin  --batch 16
  run --steps 20 --warmup 3 --op ctrelin  --batch 256 ;;
wide)
  run --steps 3  --warmup 1 --op multiply --batch 256 --bits 128 --limbs 2
  run --steps 3  --warmup 1 --op fwdinv   --batch 256 --bits 128 --limbs 2
  run --steps 3  --warmup 1 --op multiply --batch 128 --bits 250 --limbs 2
  run --steps 3  --warmup 1 --op fwdinv   --batch 128 --bits 250 --limbs 2
  run --steps 3  --warmup 1 --op multiply --batch 256 --bits 100 --limbs 2 ;;
esac; done
python - <<PY
import json
for l in open("$OUT"):
    d=json.loads(l); c=d["config"]; r=d["roofline"]; s=r["secondary"]
    print(f'{c["op"]:11s} N={c["n"]:6d} L={c["limbs"]} bits={c["prime_bits"]:3d} B={c["batch_per_gpu"]:5d} {d["dtype"][:5]:5s} {d["value"]:12.1f} {d["unit"]:10s} {d["ms_per_step"]:9.4f} ms  hbm {r["frac"]:.3f}  int-mul {s["frac"]:.3f}')
PY
