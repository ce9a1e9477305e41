#!/bin/bash
# rocprofv3 --kernel-trace --stats for the non-headline workloads (configs[2..4] shapes); the per-kernel stats are copied to
# gpurun_out/<tag>_ops/<name>_kernel_stats.csv, the bench lines to <name>.json.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_ops
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
prof() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name"; return 1; }
  grep '^{' "$OUT/$name.log" > "$OUT/$name.json"
  cp "$OUT/$name"/*/*_kernel_stats.csv "$OUT/${name}_kernel_stats.csv"
  rm -rf "$OUT/$name"
}
prof ctrelin_n8192 --steps 10 --warmup 2 --op ctrelin --batch 1024 \
 && prof fwdinv_n8192 --steps 10 --warmup 2 --op fwdinv --batch 4096 \
 && prof ctrelin_n16384 --steps 10 --warmup 2 --op ctrelin --batch 128 --n 16384 --limbs 6 \
 && prof blindrotate_n8192 --steps 5 --warmup 1 --op blindrotate --batch 1024 \
 && prof blindrotate_n16384 --steps 5 --warmup 1 --op blindrotate --batch 128 --n 16384 --limbs 6
echo "rc=$?"
head -4 "$OUT"/*_kernel_stats.csv
