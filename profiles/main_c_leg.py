#!/usr/bin/env python3
"""profiles/main_c_leg.py — bench.py's `cpu_baseline_main_c` leg by itself: the reference's main.c pipeline on the box's host
(the oracle's restatement; the compiled reference when oracle/_ref travelled) beside the engine's drop-in programs on the same
1000 synthetic frames.  One JSON line."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

r = bench.cpu_baseline_main_c()
print(json.dumps({k: r[k] for k in ("reference", "main_cpu_naive_edt", "slam_main", "slam_main_mapper", "drop_in_speedup_vs_main_cpu") if k in r}))
