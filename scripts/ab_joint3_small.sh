#!/bin/bash
# A/B: three-array one-workgroup-per-limb kernels (default where use_joint3 says so) against the split forms (FHE_HIP_SPLIT_KEYSWITCH=1), 8-byte residues
run() { python bench.py "$@" --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['frac'])"; }
for n in ${SIZES:-8192 4096 2048}; do
for cfg in "40 3 20" "60 2 32" "64 2 32"; do set -- $cfg
  for op in relin ctrelin blindrotate; do
  echo "$op N=$n $1-bit x $2, w=$3: default $(run --op $op --bits $1 --n $n --limbs $2 --batch 512 --decomp-bits $3)   split $(FHE_HIP_SPLIT_KEYSWITCH=1 run --op $op --bits $1 --n $n --limbs $2 --batch 512 --decomp-bits $3)"
  done
done
done
