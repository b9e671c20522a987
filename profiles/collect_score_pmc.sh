#!/bin/bash
# profiles/collect_score_pmc.sh TAG — hardware counters of score_poses_kernel on the score-only microbench
# (BASELINE configs[2]: 1 048 576 poses x 360 beams on a 2048^2 EDT), one rocprofv3 --pmc pass per counter group
# (never combined with tracing).  profiles/summarise_score.py TAG turns the CSVs into profiles/TAG_pmc_score.md.
set -eo pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs --mode score --particles 1048576 --grid 2048 --steps 6 --warmup 2 --events none $SCORE_PMC_EXTRA"
k=0
for G in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "TA_BUSY_avr TA_TA_BUSY_sum" "TA_FLAT_READ_WAVEFRONTS_sum" "TA_TOTAL_WAVEFRONTS_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY" "VALUBusy" "MemUnitStalled" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum" \
         "FETCH_SIZE"; do
  k=$((k + 1))
  echo "[score-pmc] pass $k: $G"
  # a group the hardware cannot collect in one pass makes rocprofv3 abort and then linger: bound every pass
  timeout -k 5 150 rocprofv3 --pmc $G --output-format csv -d "$OUT/${TAG}_scorepmc_$k" -- $B > /dev/null 2> "$OUT/${TAG}_scorepmc_$k.err" || echo "[score-pmc] pass $k FAILED (counters not collectable together?)"
done
echo "[score-pmc] done"
