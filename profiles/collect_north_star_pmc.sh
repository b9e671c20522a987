#!/bin/bash
# profiles/collect_north_star_pmc.sh TAG — HBM traffic of the in-filter front kernel at the north-star size (1 048 576 x 1 000)
# and at configs[4]'s per-GPU share (524 288 x 5 000): separate --pmc passes (FETCH_SIZE, WRITE_SIZE), as collect.sh does at
# configs[1].   gpurun --timeout 900 -- 'bash profiles/collect_north_star_pmc.sh r03' ; python profiles/summarise_big_pmc.py r03
set -eo pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs --no-sweep --steps 8 --warmup 2 --preroll 0 --events none"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[big_pmc] $C north star"
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/${TAG}_bigpmc_${C}_ns" -- $B --scaling strong --particles-total 1048576 --landmarks 1000 > /dev/null 2> "$OUT/${TAG}_bigpmc_${C}_ns.err"
  echo "[big_pmc] $C 512k x 5000"
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/${TAG}_bigpmc_${C}_c5" -- $B --particles 524288 --landmarks 5000 > /dev/null 2> "$OUT/${TAG}_bigpmc_${C}_c5.err"
done
echo "[big_pmc] done"
