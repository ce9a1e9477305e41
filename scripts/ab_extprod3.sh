#!/bin/bash
# A/B: three-array one-workgroup-per-limb kernels against the split forms (FHE_HIP_SPLIT_KEYSWITCH=1), blind rotation, 8-byte residues, N = 2^14
run() { python bench.py "$@" --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['frac'])"; }
for cfg in "40 6 16" "40 6 20" "60 3 32" "64 3 32"; do set -- $cfg
  echo "blindrotate N=16384 $1-bit x $2, w=$3: joint $(run --op blindrotate --bits $1 --n 16384 --limbs $2 --batch 128 --decomp-bits $3)   split $(FHE_HIP_SPLIT_KEYSWITCH=1 run --op blindrotate --bits $1 --n 16384 --limbs $2 --batch 128 --decomp-bits $3)"
done
