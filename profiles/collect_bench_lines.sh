#!/bin/bash
# profiles/collect_bench_lines.sh TAG — every bench.py line BASELINE.md §3 / DESIGN.md §8 quote, at HEAD, on the GPU box:
#   gpurun --timeout 1200 -- 'bash profiles/collect_bench_lines.sh r03'     then copy gpurun_out/TAG_bench_*.json into profiles/
# Each line is one `python bench.py ...` (unprofiled); a line that fails is reported and the rest still run, unless it was
# killed at its time limit (then nothing else is started).
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
line() {   # name, flags...
  local name=$1; shift
  echo "[lines] $name: bench.py $*"
  timeout -k 10 280 python3 bench.py "$@" > "$OUT/${TAG}_bench_$name.json" 2> "$OUT/${TAG}_bench_$name.err"
  local rc=$?
  if [ $rc -ge 124 ]; then echo "[lines] $name killed at its limit (rc=$rc): stopping"; exit $rc; fi
  [ $rc -ne 0 ] && { echo "[lines] $name FAILED rc=$rc"; tail -3 "$OUT/${TAG}_bench_$name.err"; }
  return 0
}
Q="--no-cpu-baseline --no-extra-legs"
line default
line default_driver_args --steps 20 --warmup 5
line default_rows $Q --map-layout rows
line default_event_every_frame $Q --event-every 1
line force_collectives $Q --force-collectives
line force_collectives_rows $Q --force-collectives --map-layout rows
line force_collectives_obs32 $Q --force-collectives --observed 32
line force_collectives_obs32_paged $Q --force-collectives --observed 32 --paged
line local_2ranks $Q --gpus 2 --transport local --particles 32768
line obs32_rows $Q --observed 32 --map-layout rows
line obs32_split $Q --observed 32 --map-layout split
line obs32_paged $Q --observed 32 --paged
line obs32_split_pages $Q --observed 32 --map-layout split_pages
line obs32_auto $Q --observed 32
line obs32_ess0.3 $Q --observed 32 --ess 0.3 --map-layout rows --steps 120
line obs32_ess0.1 $Q --observed 32 --ess 0.1 --map-layout rows --steps 120
line obs32_ess0.3_auto $Q --observed 32 --ess 0.3 --steps 120
line ess0.3_auto $Q --ess 0.3 --steps 120
line ess0.3_rows $Q --ess 0.3 --steps 120 --map-layout rows
line obs128_rows $Q --observed 128 --map-layout rows
line obs128_split $Q --observed 128 --map-layout split
line obs128_paged $Q --observed 128 --paged
line obs128_split_pages $Q --observed 128 --map-layout split_pages
line obs128_auto $Q --observed 128
line 5000_obs32_rows $Q --landmarks 5000 --observed 32 --map-layout rows --steps 40
line 5000_obs32_split $Q --landmarks 5000 --observed 32 --map-layout split --steps 40
line 5000_obs32_paged $Q --landmarks 5000 --observed 32 --paged
line 5000_obs32_auto $Q --landmarks 5000 --observed 32
line north_star $Q --scaling strong --particles-total 1048576 --landmarks 1000 --steps 30
line north_star_rows $Q --scaling strong --particles-total 1048576 --landmarks 1000 --steps 30 --map-layout rows
line north_star_obs32_split $Q --scaling strong --particles-total 1048576 --landmarks 1000 --observed 32 --map-layout split --steps 30
line north_star_obs32_paged $Q --scaling strong --particles-total 1048576 --landmarks 1000 --observed 32 --paged --steps 30
line north_star_obs32_auto $Q --scaling strong --particles-total 1048576 --landmarks 1000 --observed 32 --steps 30
line ekf_sweep_1m $Q --mode ekf --particles 1048576 --landmarks 1000 --steps 30
line score_config3 $Q --mode score --particles 1048576 --grid 2048
SLAM_SCORE_PACKED=0 line score_config3_float_grid $Q --mode score --particles 1048576 --grid 2048
line config4_share $Q --particles 1048576 --landmarks 0
line config5_share $Q --particles 524288 --landmarks 5000 --steps 20
line config5_share_rows $Q --particles 524288 --landmarks 5000 --steps 20 --map-layout rows
echo "[lines] done"
