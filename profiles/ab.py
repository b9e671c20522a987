#!/usr/bin/env python3
"""profiles/ab.py REPS "BENCH ARGS" NAME=ENV[,ENV...] [NAME=...] — A/B of library builds / environment switches on one box:
runs `python3 bench.py --no-cpu-baseline --no-extra-legs --no-sweep BENCH ARGS` REPS times per variant, interleaved (the
boxes' clock and power states drift by several per cent within a call, so variants are never compared across blocks of
runs), and prints per variant the median and the minimum of ms/step and of the dominant kernel's launch time.
ENV entries are VAR=value; LIB=dir is short for SLAM_HIP_LIB=<package>/dir/libslam_hip.so.  Measurement tooling."""
import json
import os
import statistics
import subprocess
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
pkg = root / "hardware-acceleration-of-lidar-slam_amd"
reps, args = int(sys.argv[1]), sys.argv[2].split()
variants = []
for spec in sys.argv[3:]:
    name, _, envs = spec.partition("=")
    env = dict(os.environ)
    for e in filter(None, envs.split(",")):
        k, _, v = e.partition("=")
        if k == "LIB":
            k, v = "SLAM_HIP_LIB", str(pkg / v / "libslam_hip.so")
        env[k] = v
    variants.append((name, env))
res = {name: [] for name, _ in variants}
for r in range(reps):
    for name, env in variants:
        out = subprocess.run([sys.executable, str(root / "bench.py"), "--no-cpu-baseline", "--no-extra-legs", "--no-sweep", *args],
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[name].append((d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["kernel"]))
        except Exception:
            print(name, "FAILED", out.stderr[-300:], flush=True)
for name, v in res.items():
    if v:
        ms, k = [x[0] for x in v], [x[1] for x in v]
        print(f"{name:14s} ms/step median {statistics.median(ms):.4f} min {min(ms):.4f} | {v[0][2][:26]:26s} median {statistics.median(k) * 1e3:7.1f} us "
              f"min {min(k) * 1e3:7.1f} us  ({len(v)} runs)", flush=True)
