#!/bin/bash
# profiles/collect.sh TAG — every rocprofv3 run the numbers in profiles/ and BASELINE.md come from, on the GPU box:
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh r02'
# Outputs under gpurun_out/TAG_*; profiles/summarise.py TAG turns them into the tracked files.
# --pmc passes are separate runs, never combined with tracing (MI355X_MICROARCH.md, HBM section).
set -eo pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --no-cpu-baseline"
step() { echo "[collect] $*"; }

step "kernel trace, default workload (configs[1])"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_pf" -- $B --steps 100 --warmup 10 --no-sweep > "$OUT/${TAG}_trace_pf.json" 2> "$OUT/${TAG}_trace_pf.err"
step "kernel trace, score-only (configs[2])"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_score" -- $B --mode score --particles 1048576 --grid 2048 --steps 50 --warmup 5 > "$OUT/${TAG}_trace_score.json" 2> "$OUT/${TAG}_trace_score.err"
for C in FETCH_SIZE WRITE_SIZE; do
  step "pmc $C: pf default, pf --observed 32, ekf sweep (calibration)"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/${TAG}_pmc_${C}_pf" -- $B --mode pf --steps 8 --warmup 2 --events none > /dev/null 2> "$OUT/${TAG}_pmc_${C}_pf.err"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/${TAG}_pmc_${C}_pfobs32" -- $B --mode pf --observed 32 --steps 8 --warmup 2 --events none > /dev/null 2> "$OUT/${TAG}_pmc_${C}_pfobs32.err"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/${TAG}_pmc_${C}_ekf" -- $B --mode ekf --steps 8 --warmup 2 --events none > /dev/null 2> "$OUT/${TAG}_pmc_${C}_ekf.err"
done
step "copy ceiling of the EKF access shape (pure copy, no arithmetic)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/copy_ceiling "$ROOT/profiles/copy_ceiling.hip"
{ /tmp/copy_ceiling 65536 512; /tmp/copy_ceiling 1048576 1024; } > "$OUT/${TAG}_copy_ceiling.txt" 2>&1
step done
