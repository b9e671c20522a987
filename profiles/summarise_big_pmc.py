#!/usr/bin/env python3
"""profiles/summarise_big_pmc.py TAG — gpurun_out/TAG_bigpmc_* (profiles/collect_north_star_pmc.sh) -> profiles/TAG_pmc_big.md and
two more records of profiles/traffic.json: HBM bytes per launch of the in-filter front kernel at 1 048 576 x 1 000 and at
524 288 x 5 000 (FETCH_SIZE doubled as summarise.py's calibration found, WRITE_SIZE exact; counters in KiB)."""
import csv
import glob
import json
import os
import sys
from pathlib import Path

tag = sys.argv[1]
here = Path(__file__).resolve().parent
src = here.parent / "gpurun_out"


def counter(case, name):
    f = max(glob.glob(str(src / f"{tag}_bigpmc_{name}_{case}" / "*" / "*counter_collection.csv")), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if "frame_front_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return [float(r["Counter_Value"]) * 1024 for r in rows]


traffic = json.loads((here / "traffic.json").read_text())
md = [f"# {tag}: HBM traffic of `frame_front_kernel` at the large configurations (separate --pmc passes, `profiles/collect_north_star_pmc.sh`)", ""]
for case, n, L, name in (("ns", 1048576, 1000, "north-star frame 1 048 576 x 1 000"), ("c5", 524288, 5000, "configs[4] per-GPU share 524 288 x 5 000")):
    fetch, write = counter(case, "FETCH_SIZE"), counter(case, "WRITE_SIZE")
    st = slice(3, 9)   # fused launches 3 .. 8 of the 9 (frames 1 .. 9 of --steps 8 --warmup 2): steady frames
    rd = 2.0 * sum(fetch[st]) / len(fetch[st])
    wr = sum(write[st]) / len(write[st])
    alg = 40 * n * L
    Lp = (L + 31) // 32 * 32
    traffic[f"pf:{n}:360:{L}:1024"] = {"bytes_per_launch": rd + wr, "kernel": "frame_front_kernel", "read_bytes": rd, "write_bytes": wr,
                                       "algorithmic_bytes": alg, "source": f"profiles/{tag}_pmc_big.md",
                                       "measured": f"{tag}: builder-run rocprofv3 --pmc passes of bench.py --steps 8 --warmup 2 --preroll 0"}
    md += [f"## {name}", "", "| counter | per-launch values of the fused launches (bytes = raw x 1024) |", "|---|---|",
           "| FETCH_SIZE | " + ", ".join(f"{x / 1e9:.2f} GB" for x in fetch[:10]) + " |",
           "| WRITE_SIZE | " + ", ".join(f"{x / 1e9:.2f} GB" for x in write[:10]) + " |", "",
           f"Steady frames: read = 2 x FETCH_SIZE = {rd / 1e9:.2f} GB, write = {wr / 1e9:.2f} GB (the padded rows: 20 B x {n} x {Lp} = "
           f"{20 * n * Lp / 1e9:.2f} GB), total {(rd + wr) / 1e9:.2f} GB per launch against {alg / 1e9:.2f} GB algorithmic (40 B x n x {L}).", ""]
(here / "traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")
(here / f"{tag}_pmc_big.md").write_text("\n".join(md) + "\n")
print("\n".join(md))
