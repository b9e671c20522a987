#!/bin/bash
# profiles/collect_page_sizes.sh TAG — the paged update with 32- (product), 16- and 8-landmark pages on the K-observed workload
# (measurement builds: make -C csrc OUT=../lib_pP EXTRA=-DSLAM_PAGE_LANDMARKS=P, loaded through SLAM_HIP_LIB), next to the pure
# page copies of profiles/page_copy_ceiling.hip.   gpurun --timeout 900 -- 'bash profiles/collect_page_sizes.sh r03'
set -eo pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
PKG=$ROOT/hardware-acceleration-of-lidar-slam_amd
mkdir -p "$OUT"
cd "$ROOT"
B="python3 bench.py --no-cpu-baseline --no-extra-legs --no-sweep --observed 32 --paged --steps 60 --warmup 10"
for P in 32 16 8; do
  LIB=$PKG/lib/libslam_hip.so; [ $P != 32 ] && LIB=$PKG/lib_p$P/libslam_hip.so
  [ -f "$LIB" ] || { echo "[page_sizes] $LIB missing"; continue; }
  echo "[page_sizes] $P-landmark pages: paged tests"
  SLAM_HIP_LIB=$LIB timeout -k 10 300 python3 -m pytest tests/test_gpu_paged.py -q -x -m gpu > "$OUT/${TAG}_pagesize_${P}_tests.log" 2>&1 || { tail -5 "$OUT/${TAG}_pagesize_${P}_tests.log"; echo "[page_sizes] tests FAILED for $P"; continue; }
  for W in "--landmarks 500" "--landmarks 5000" "--particles 1048576 --landmarks 1000"; do
    N=$(echo $W | tr -d ' -' | tr -c 'a-z0-9\n' '_')
    echo "[page_sizes] $P-landmark pages: $W"
    SLAM_HIP_LIB=$LIB timeout -k 10 200 $B $W > "$OUT/${TAG}_pagesize_${P}_$N.json" 2> "$OUT/${TAG}_pagesize_${P}_$N.err" || echo "[page_sizes] bench FAILED ($P, $W)"
  done
done
echo "[page_sizes] done"
