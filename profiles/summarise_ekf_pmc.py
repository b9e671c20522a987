#!/usr/bin/env python3
"""profiles/summarise_ekf_pmc.py TAG — the per-group counter passes of collect.sh on the in-filter EKF kernel ->
profiles/TAG_pmc_ekf_sq.md (per-launch averages over the steady launches of the dominant EKF kernel)."""
import csv
import glob
import os
import sys
from collections import defaultdict
from pathlib import Path

tag = sys.argv[1]
here = Path(__file__).resolve().parent
src = Path(sys.argv[2]) if len(sys.argv) > 2 else here.parent / "gpurun_out"
vals = defaultdict(list)
kern = None
for d in sorted(glob.glob(str(src / f"{tag}_ekfpmc_*"))):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        continue
    rows = [r for r in csv.DictReader(open(max(fs, key=os.path.getmtime))) if "ekf_update" in r["Kernel_Name"] or "frame_front_kernel" in r["Kernel_Name"]]
    for k, r in enumerate(rows):
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if k == 4:   # the header names the kernel of the launches that are averaged (4..9), not of the stage pass behind them
            kern = r["Kernel_Name"].replace("void slam::", "").replace("slam::", "").replace("(anonymous namespace)::", "").split("(")[0]
md = [f"# {tag}: hardware counters of the in-filter EKF kernel (`{kern}`, cold start: `--preroll 0`), configs[1], one `rocprofv3 --pmc` pass per group", "",
      "| counter | per-launch average (launches 4..9 of `bench.py --steps 8 --warmup 2 --preroll 0 --events none --no-sweep`) |", "|---|---|"]
for name, v in vals.items():
    v = v[4:10] or v   # launches 4..9: steady frames of the timed region (the stage pass behind them runs the two-launch path)
    md.append(f"| {name} | {sum(v) / len(v):.4g} |")
g = {k: sum(v[4:10] or v) / len(v[4:10] or v) for k, v in vals.items()}
if "SQ_WAVE_CYCLES" in g and "SQ_ACTIVE_INST_VALU" in g:
    md += ["", f"SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {g['SQ_ACTIVE_INST_VALU'] / g['SQ_WAVE_CYCLES']:.3f} (share of the wave-cycles in which "
           "a vector ALU instruction executes); SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = "
           f"{g.get('SQ_WAIT_INST_ANY', float('nan')) / g['SQ_WAVE_CYCLES']:.3f}; SQ_WAIT_ANY / SQ_WAVE_CYCLES = "
           f"{g.get('SQ_WAIT_ANY', float('nan')) / g['SQ_WAVE_CYCLES']:.3f}."]
(here / f"{tag}_pmc_ekf_sq.md").write_text("\n".join(md) + "\n")
print("\n".join(md))
