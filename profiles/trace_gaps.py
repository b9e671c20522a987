#!/usr/bin/env python3
"""profiles/trace_gaps.py TAG — idle stream time between the kernels of a frame, from the rocprofv3 kernel trace of the
default bench (gpurun_out/TAG_trace_pf, made by profiles/collect.sh with every frame bracketed by HIP events around the EKF
kernel) -> profiles/TAG_trace_gaps.md.  Shows what an event bracket costs the stream."""
import csv
import glob
import statistics
import sys
from pathlib import Path

tag = sys.argv[1]
here = Path(__file__).resolve().parent
src = Path(sys.argv[2]) if len(sys.argv) > 2 else here.parent / "gpurun_out"
f = sorted(glob.glob(str(src / f"{tag}_trace_pf" / "*" / "*kernel_trace.csv")))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
KEYS = ("score_poses", "ekf_update_group", "ekf_update_kernel", "logweight", "quantise_scan", "ancestors_from_scan", "obs_count")


def short(n):
    for k in KEYS:
        if k in n:
            return k + "_kernel" if not k.endswith("kernel") else k
    return n[:40]


idx = [i for i, r in enumerate(rows) if "score_poses" in r["Kernel_Name"]]
gaps, durs, periods = {}, {}, []
for a, b in zip(idx[40:100], idx[41:101]):   # 60 steady frames of the timed region
    seq = rows[a:b + 1]
    periods.append(int(seq[-1]["Start_Timestamp"]) - int(seq[0]["Start_Timestamp"]))
    for x, y in zip(seq[:-1], seq[1:]):
        gaps.setdefault((short(x["Kernel_Name"]), short(y["Kernel_Name"])), []).append(int(y["Start_Timestamp"]) - int(x["End_Timestamp"]))
        durs.setdefault(short(x["Kernel_Name"]), []).append(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
md = [f"# {tag}: stream time between the kernels of a frame (rocprofv3 --kernel-trace of `bench.py --steps 100 --warmup 10 --no-sweep "
      "--event-every 1`: configs[1], the EKF kernel of EVERY frame bracketed by HIP events)", "",
      f"Frame period (median of 60 steady frames): {statistics.median(periods) / 1e3:.1f} µs.", "",
      "| from -> to | idle stream between them, median | frames |", "|---|---|---|"]
for (a, b), v in gaps.items():
    md.append(f"| `{a}` -> `{b}` | {statistics.median(v) / 1e3:.2f} µs | {len(v)} |")
md += ["", "| kernel | duration, median (under the profiler) |", "|---|---|"]
for k, v in durs.items():
    md.append(f"| `{k}` | {statistics.median(v) / 1e3:.1f} µs |")
md += ["", "Only the two boundaries of the event bracket (before and after the EKF kernel) leave the stream idle, about 6 µs each; the "
       "other launches of a frame follow each other without a gap.  Hence `bench.py` brackets every fourth frame by default "
       "(`--event-every`)."]
(here / f"{tag}_trace_gaps.md").write_text("\n".join(md) + "\n")
print("\n".join(md))
