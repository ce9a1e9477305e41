python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "relin or tensor_product_without" 2>&1 | tail -3 || exit 1
FHE_HIP_SPLIT_KEYSWITCH=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tensor_product_without" 2>&1 | tail -2 || exit 1
for cfg in "40 6 16" "40 6 20" "60 3 32" "64 3 32"; do set -- $cfg
  for op in relin ctrelin; do
    python bench.py --op $op --bits $1 --n 16384 --limbs $2 --batch 128 --decomp-bits $3 --no-cpu-baseline --no-extras > gpurun_out/ksj_$1_$op.json
    FHE_HIP_SPLIT_KEYSWITCH=1 python bench.py --op $op --bits $1 --n 16384 --limbs $2 --batch 128 --decomp-bits $3 --no-cpu-baseline --no-extras > gpurun_out/kss_$1_$op.json
    python - $1 $2 $3 $op <<PY
import json,sys
a=json.loads(open(f"gpurun_out/ksj_{sys.argv[1]}_{sys.argv[4]}.json").read().strip().splitlines()[-1])
b=json.loads(open(f"gpurun_out/kss_{sys.argv[1]}_{sys.argv[4]}.json").read().strip().splitlines()[-1])
print(sys.argv[1:], "joint", round(a["value"]), a["roofline"]["frac"], "split", round(b["value"]), b["roofline"]["frac"], flush=True)
PY
  done
done
