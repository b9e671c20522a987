#!/bin/bash
# profiles/run_gpu_suite.sh TAG [more commands...] — the whole -m gpu suite into gpurun_out/TAG_gputest.log; whatever follows
# runs only if the suite ended by itself (a test failure is fine, a timeout or a kill is not: no GPU step after a hung one).
TAG=${1:-r03}; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/${TAG}_gputest.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/${TAG}_gputest.log
tail -5 gpurun_out/${TAG}_gputest.log
if [ $rc -ge 124 ]; then echo "suite did not end by itself (rc=$rc): stopping"; exit $rc; fi
for c in "$@"; do echo "[suite] $c"; bash -o pipefail -c "$c" || exit $?; done
exit $rc
