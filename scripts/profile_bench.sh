#!/bin/bash
# Profiles bench.py on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats          -> per-kernel durations
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE   -> HBM traffic (separate passes, no tracing)
# Output lands in gpurun_out/<tag>/ ; scripts/summarize_profile.py turns it into profiles/<tag>_*.
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads $EXTRA > "$OUT/bench_trace.log" 2>&1 \
 && rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-workloads $EXTRA > "$OUT/bench_pmc_fetch.log" 2>&1 \
 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-workloads $EXTRA > "$OUT/bench_pmc_write.log" 2>&1
echo "profile rc=$?"
find "$OUT" -name "*.csv" | head -20
# keep the merge-back small: the per-dispatch traces are summarised, large raw files dropped
find "$OUT" -name "*.csv" -size +8M -delete
