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
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
step() { echo "[collect] $*"; }

step "kernel trace, default workload (configs[1])"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_pf" -- $B --steps 100 --warmup 10 --no-sweep --event-every 1 > "$OUT/${TAG}_trace_pf.json" 2> "$OUT/${TAG}_trace_pf.err"
step "kernel trace, score-only (configs[2])"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_score" -- $B --mode score --particles 1048576 --grid 2048 --steps 50 --warmup 5 > "$OUT/${TAG}_trace_score.json" 2> "$OUT/${TAG}_trace_score.err"
for C in FETCH_SIZE WRITE_SIZE; do
  step "pmc $C: pf default, pf --observed 32, ekf sweep (calibration)"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/${TAG}_pmc_${C}_pf" -- $B --mode pf --steps 8 --warmup 2 --preroll 0 --events none > /dev/null 2> "$OUT/${TAG}_pmc_${C}_pf.err"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/${TAG}_pmc_${C}_pfobs32" -- $B --mode pf --observed 32 --map-layout rows --steps 8 --warmup 2 --preroll 0 --events none > /dev/null 2> "$OUT/${TAG}_pmc_${C}_pfobs32.err"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/${TAG}_pmc_${C}_ekf" -- $B --mode ekf --steps 8 --warmup 2 --events none > /dev/null 2> "$OUT/${TAG}_pmc_${C}_ekf.err"
done
step "copy ceilings: the EKF's row access shape, and scattered pages (pure copies, no arithmetic)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/copy_ceiling "$ROOT/profiles/copy_ceiling.hip"
{ /tmp/copy_ceiling 65536 512; /tmp/copy_ceiling 1048576 1024; } > "$OUT/${TAG}_copy_ceiling.txt" 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/page_copy_ceiling "$ROOT/profiles/page_copy_ceiling.hip"
{ /tmp/page_copy_ceiling 65536; /tmp/page_copy_ceiling 1048576 67108864; } > "$OUT/${TAG}_page_copy_ceiling.txt" 2>&1
step "hardware counters of the in-filter EKF kernel (one pass per group)"
k=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_WR" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "VALUBusy" "MemUnitStalled" "WriteUnitStalled" "GRBM_GUI_ACTIVE"; do
  k=$((k + 1))
  timeout -k 5 150 rocprofv3 --pmc $G --output-format csv -d "$OUT/${TAG}_ekfpmc_$k" -- $B --mode pf --steps 8 --warmup 2 --preroll 0 --events none --no-sweep > /dev/null 2> "$OUT/${TAG}_ekfpmc_$k.err" || echo "[collect] ekf pmc pass $k ($G) FAILED"
done
step done
