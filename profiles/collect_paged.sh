#!/bin/bash
# profiles/collect_paged.sh TAG — the paged-map session (bench.py --paged) on the K-observed workload, on the GPU box:
# kernel trace + separate FETCH_SIZE / WRITE_SIZE passes for rows and pages at 500 and 5000 landmarks, 32 observed.
#   gpurun --timeout 900 -- 'bash profiles/collect_paged.sh r02'   then   python profiles/summarise_paged.py r02
set -eo pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs --no-sweep --observed 32"
for L in 500 5000; do
  for P in rows paged; do
    F="--map-layout rows"; [ $P = paged ] && F="--paged"
    echo "[collect_paged] trace L=$L $P"
    timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_pagedtrace_${L}_$P" -- $B --landmarks $L --steps 60 --warmup 10 --events none $F > "$OUT/${TAG}_pagedtrace_${L}_$P.json" 2> "$OUT/${TAG}_pagedtrace_${L}_$P.err"
  done
done
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[collect_paged] pmc $C, 500 landmarks, pages"
  timeout -k 5 200 rocprofv3 --pmc $C --output-format csv -d "$OUT/${TAG}_pagedpmc_${C}" -- $B --paged --steps 12 --warmup 2 --preroll 0 --events none > /dev/null 2> "$OUT/${TAG}_pagedpmc_${C}.err"
done
echo "[collect_paged] done"
