#!/bin/bash
# profiles/collect_r04.sh TAG [BENCH ARGS...] — round 4's evidence runs on the GPU box (default workload = configs[1] on the
# session's default layout, i.e. split; extra arguments go to bench.py, e.g. --map-layout rows):
#   kernel trace + stats of the bench, HBM traffic (FETCH_SIZE / WRITE_SIZE, one --pmc pass each, never combined with tracing:
#   MI355X_MICROARCH.md), and the SQ counters of the dominant kernel in steady state (one pass per group).
# Outputs under gpurun_out/TAG_*; profiles/summarise_r04.py TAG turns them into tracked files under profiles/.
set -o pipefail
TAG=${1:-r04}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs --no-sweep $*"
step() { echo "[collect] $*"; }
step "kernel trace + stats"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- $B --steps 100 --warmup 10 > "$OUT/${TAG}_trace.json" 2> "$OUT/${TAG}_trace.err" || echo "[collect] trace FAILED"
k=0
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_WR" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "VALUBusy" "MemUnitStalled" \
         "WriteUnitStalled" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
  k=$((k + 1))
  step "pmc pass $k: $G"
  timeout -k 5 200 rocprofv3 --pmc $G --output-format csv -d "$OUT/${TAG}_pmc_$k" -- $B --steps 8 --warmup 2 --events none > /dev/null 2> "$OUT/${TAG}_pmc_$k.err" || echo "[collect] pmc pass $k ($G) FAILED"
done
step done
