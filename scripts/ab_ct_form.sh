#!/bin/bash
# A/B: tensor product of the 8-byte fields as one launch (four live arrays) or two (FHE_HIP_CT_FORM=one|two), alone and inside the one-call multiply
run() { python bench.py "$@" --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['frac'])"; }
for n in ${SIZES:-8192 4096 2048}; do
for cfg in "40 3 20" "60 2 32" "64 2 32"; do set -- $cfg
  for op in ct ctrelin; do
  echo "$op N=$n $1-bit x $2, w=$3: one-launch $(FHE_HIP_CT_FORM=one run --op $op --bits $1 --n $n --limbs $2 --batch 1024 --decomp-bits $3)  two-launch $(FHE_HIP_CT_FORM=two run --op $op --bits $1 --n $n --limbs $2 --batch 1024 --decomp-bits $3)"
  done
done
done
