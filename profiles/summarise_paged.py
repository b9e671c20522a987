#!/usr/bin/env python3
"""profiles/summarise_paged.py TAG — gpurun_out/TAG_paged* (profiles/collect_paged.sh) -> profiles/TAG_paged.md and the
`...:paged:obs32` record of profiles/traffic.json (HBM bytes per launch of ekf_paged_kernel; FETCH_SIZE with the factor 2 the
calibration of summarise.py found)."""
import csv
import glob
import json
import os
import sys
from pathlib import Path

tag = sys.argv[1]
here = Path(__file__).resolve().parent
src = here.parent / "gpurun_out"


def newest(pattern):
    return max(glob.glob(str(src / pattern)), key=os.path.getmtime)


md = [f"# {tag}: landmark maps as copy-on-write pages vs one row per particle — 65 536 particles, 32 landmarks observed per "
      "frame, resampling every frame (`profiles/collect_paged.sh`)", ""]
for L in (500, 5000):
    md += [f"## {L} landmarks: kernels of a frame (rocprofv3 --kernel-trace --stats, average per launch)", "",
           "| kernel | rows: calls | rows: µs | pages: calls | pages: µs |", "|---|---|---|---|---|"]
    stats = {}
    for P in ("rows", "paged"):
        f = newest(f"{tag}_pagedtrace_{L}_{P}/*/*kernel_stats.csv")
        for r in csv.DictReader(open(f)):
            name = r["Name"].replace("void ", "").replace("slam::(anonymous namespace)::", "").split("(")[0]
            if any(k in name for k in ("ekf_", "frame_front", "score_poses", "page_", "free_list", "ancestors_from", "quantise_scan", "logweight")):
                stats.setdefault(name, {})[P] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    for name, v in sorted(stats.items(), key=lambda kv: -max(x[1] * x[0] for x in kv[1].values())):
        if max(x[0] for x in v.values()) < 20:
            continue   # set-up kernels
        cell = lambda P: (f"{v[P][0]} | {v[P][1]:.1f}" if P in v else "— | —")
        md.append(f"| `{name}` | {cell('rows')} | {cell('paged')} |")
    ms = {P: json.loads(open(src / f"{tag}_pagedtrace_{L}_{P}.json").read())["ms_per_step"] for P in ("rows", "paged")}
    md += ["", f"Frame: rows {ms['rows']:.4f} ms, pages {ms['paged']:.4f} ms (under the profiler).", ""]


def counter(name):
    f = newest(f"{tag}_pagedpmc_{name}/*/*counter_collection.csv")
    return [float(r["Counter_Value"]) * 1024 for r in csv.DictReader(open(f)) if "ekf_paged" in r["Kernel_Name"] and r["Counter_Name"] == name]


fetch, write = counter("FETCH_SIZE"), counter("WRITE_SIZE")
steady = slice(4, 14)
rd = 2.0 * sum(fetch[steady]) / len(fetch[steady])
wr = sum(write[steady]) / len(write[steady])
n, K = 65536, 32
md += ["## HBM traffic of `ekf_paged_kernel`, 500 landmarks (separate --pmc passes)", "",
       "| counter | per-launch values (bytes = raw x 1024) |", "|---|---|",
       "| FETCH_SIZE | " + ", ".join(f"{x / 1e6:.1f} MB" for x in fetch[:14]) + " |",
       "| WRITE_SIZE | " + ", ".join(f"{x / 1e6:.1f} MB" for x in write[:14]) + " |", "",
       f"Steady frames: read = 2 x FETCH_SIZE = {rd / 1e6:.1f} MB, write = {wr / 1e6:.1f} MB, total {(rd + wr) / 1e6:.1f} MB per "
       f"launch against {40 * n * K / 1e6:.1f} MB algorithmic (40 B x n x {K} observed) — the row-per-particle update of the same "
       "workload moves 1015 MB (`r02_pmc_ekf.md`)."]
(here / f"{tag}_paged.md").write_text("\n".join(md) + "\n")
tfile = here / "traffic.json"
traffic = json.loads(tfile.read_text())
traffic[f"pf:{n}:360:500:1024:paged:obs32"] = {"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                                                "algorithmic_bytes": 40 * n * K, "kernel": "ekf_paged_kernel",
                                                "source": f"profiles/{tag}_paged.md"}
tfile.write_text(json.dumps(traffic, indent=1) + "\n")
print("\n".join(md))
