#!/bin/bash
# Runs the bench over the BASELINE configs' single-GPU shapes and the modulus-width variants.
# Output: gpurun_out/matrix_<tag>.jsonl (one JSON line per run)
TAG=${1:-r01}
OUT=gpurun_out/matrix_$TAG.jsonl
: > $OUT
run() { echo "== $*" >&2; timeout -k 10 600 python bench.py --no-cpu-baseline --no-extras --no-verify "$@" | grep '^{' >> $OUT || echo "FAILED: $*" >&2; }
# configs[1]: fused polymul, N = 8192, 4 x 30-bit limbs -- batch sweep
run --steps 10 --warmup 2 --op multiply --batch 4096
run --steps 10 --warmup 2 --op multiply --batch 1024
run --steps 20 --warmup 3 --op multiply --batch 64
run --steps 50 --warmup 5 --op multiply --batch 1
run --steps 10 --warmup 2 --op fwdinv   --batch 4096
# configs[2]: ciphertext multiply (tensor product, relinearisation, both), N = 8192, log_q = 120
run --steps 10 --warmup 2 --op ct       --batch 1024
run --steps 10 --warmup 2 --op relin    --batch 1024
run --steps 10 --warmup 2 --op relin    --batch 1024 --decomp-bits 30
run --steps 10 --warmup 2 --op ctrelin  --batch 1024 --decomp-bits 30
run --steps 10 --warmup 2 --op ctrelin  --batch 256
run --steps 20 --warmup 3 --op ctrelin  --batch 1
run --steps 10 --warmup 2 --op ctrelin  --batch 1024
# configs[3] shape on one GPU: N = 16384, 6 limbs (30-bit and 40-bit bases), 128 ciphertexts = 1024 / 8
run --steps 10 --warmup 2 --op multiply --batch 1024 --n 16384 --limbs 6 --bits 30
run --steps 10 --warmup 2 --op ctrelin  --batch 128 --n 16384 --limbs 6 --bits 30
run --steps 10 --warmup 2 --op ct       --batch 128 --n 16384 --limbs 6 --bits 40
run --steps 5  --warmup 1 --op ctrelin  --batch 128 --n 16384 --limbs 6 --bits 40
run --steps 5  --warmup 1 --op blindrotate --batch 1024
run --steps 5  --warmup 1 --op blindrotate --batch 128 --n 16384 --limbs 6
run --steps 5  --warmup 1 --op blindrotate --batch 128 --n 16384 --limbs 6 --decomp-bits 30
# transforms beyond the LDS range (two-pass, word-sized classes)
run --steps 5  --warmup 1 --op fwdinv   --batch 512 --n 65536 --limbs 4 --bits 30
run --steps 5  --warmup 1 --op multiply --batch 512 --n 65536 --limbs 4 --bits 30
run --steps 5  --warmup 1 --op fwdinv   --batch 512 --n 32768 --limbs 3 --bits 40
run --steps 5  --warmup 1 --op multiply --batch 512 --n 32768 --limbs 3 --bits 40
# modulus-width variants of configs[1]/[2]
run --steps 10 --warmup 2 --op multiply --batch 1024 --bits 40 --limbs 3
run --steps 10 --warmup 2 --op fwdinv   --batch 1024 --bits 40 --limbs 3
run --steps 10 --warmup 2 --op multiply --batch 1024 --bits 60 --limbs 2
run --steps 10 --warmup 2 --op ct       --batch 1024 --bits 60 --limbs 2
run --steps 5  --warmup 1 --op ctrelin  --batch 256 --bits 60 --limbs 2
run --steps 5  --warmup 1 --op relin    --batch 512 --bits 60 --limbs 2 --decomp-bits 32
run --steps 5  --warmup 1 --op blindrotate --batch 512 --bits 60 --limbs 2 --decomp-bits 32
run --steps 10 --warmup 2 --op ct       --batch 1024 --bits 40 --limbs 3
run --steps 5  --warmup 1 --op ctrelin  --batch 1024 --bits 40 --limbs 3 --decomp-bits 20
run --steps 5  --warmup 1 --op relin    --batch 512 --bits 40 --limbs 3 --decomp-bits 20
# configs[3] shape with 8-byte residues: key switch alone, w = 20, 60-bit base; N = 2^15 on the 4-byte residues
run --steps 5  --warmup 1 --op relin    --batch 128 --n 16384 --limbs 6 --bits 40
run --steps 5  --warmup 1 --op ctrelin  --batch 128 --n 16384 --limbs 6 --bits 40 --decomp-bits 20
run --steps 5  --warmup 1 --op blindrotate --batch 128 --n 16384 --limbs 6 --bits 40
run --steps 5  --warmup 1 --op ctrelin  --batch 128 --n 16384 --limbs 3 --bits 60 --decomp-bits 32
run --steps 5  --warmup 1 --op relin    --batch 128 --n 32768 --limbs 4 --bits 30
run --steps 5  --warmup 1 --op ctrelin  --batch 128 --n 32768 --limbs 4 --bits 30
run --steps 5  --warmup 1 --op blindrotate --batch 128 --n 32768 --limbs 4 --bits 30
# 64-bit primes: full-range 64-bit class (FHE_WIDTH_64X)
run --steps 10 --warmup 2 --op multiply --batch 1024 --bits 64 --limbs 2
run --steps 10 --warmup 2 --op multiply --batch 256 --bits 64 --limbs 2
run --steps 10 --warmup 2 --op fwdinv   --batch 1024 --bits 64 --limbs 2
run --steps 10 --warmup 2 --op ct       --batch 1024 --bits 64 --limbs 2
# full-width class (FHE_WIDTH_256): 128-bit and 250-bit primes, and the 64-bit primes forced onto it (the round-1 figure)
run --steps 3  --warmup 1 --op multiply --batch 256 --bits 128 --limbs 2
run --steps 3  --warmup 1 --op fwdinv   --batch 256 --bits 128 --limbs 2
run --steps 3  --warmup 1 --op multiply --batch 128 --bits 250 --limbs 2
run --steps 3  --warmup 1 --op fwdinv   --batch 128 --bits 250 --limbs 2
FHE_HIP_FORCE_WIDTH=256 run --steps 3  --warmup 1 --op multiply --batch 256 --bits 64 --limbs 2
FHE_HIP_FORCE_WIDTH=256 run --steps 3  --warmup 1 --op fwdinv   --batch 256 --bits 64 --limbs 2
python - <<PY
import json
for l in open("$OUT"):
    d=json.loads(l); c=d["config"]; r=d["roofline"]
    print(f'{c["op"]:9s} N={c["n"]:6d} L={c["limbs"]} bits={c["prime_bits"]:3d} B={c["batch_per_gpu"]:5d} {d["dtype"][:5]:5s} {d["value"]:12.1f} {d["unit"]:10s} {d["ms_per_step"]:9.4f} ms  {r["achieved"]:8.1f} GB/s  frac {r["frac"]:.3f}')
PY
