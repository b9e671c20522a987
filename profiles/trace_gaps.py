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
import os
f = max(glob.glob(str(src / f"{tag}_trace_pf" / "*" / "*kernel_trace.csv")), key=os.path.getmtime)   # the newest run
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
KEYS = ("score_poses", "ekf_update_group", "ekf_update_kernel", "logweight", "quantise_scan", "ancestors_from_scan", "obs_count")


def short(n):
    for k in KEYS:
        if k in n:
            return k + "_kernel" if not k.endswith("kernel") else k
    return n[:40]


# one frame = from one front launch (fused: frame_front_kernel; two launches: score_poses_kernel) to the next; the timed region
# of the traced command = frames [preroll + warmup, preroll + warmup + steps) (from the bench line the run printed)
import json
line = json.loads((src / f"{tag}_trace_pf.json").read_text().strip().splitlines()[-1])
first = line["config"].get("preroll_frames", 0) + line["warmup"]
nsteps = line["steps"]
fused = any("frame_front_kernel" in r["Kernel_Name"] for r in rows)
KEYS = (("frame_front",) if fused else ()) + KEYS
idx = [i for i, r in enumerate(rows) if ("frame_front_kernel" in r["Kernel_Name"] if fused else "score_poses" in r["Kernel_Name"])]
idx = idx[first - (2 if fused else 0):]   # fused: the first two frames of a session take the two-launch path
gaps, durs, periods = {}, {}, []
for a, b in zip(idx[:nsteps], idx[1:nsteps + 1]):   # the frames of the timed region
    seq = rows[a:b + 1]
    periods.append(int(seq[-1]["Start_Timestamp"]) - int(seq[0]["Start_Timestamp"]))
    for x, y in zip(seq[:-1], seq[1:]):
        gaps.setdefault((short(x["Kernel_Name"]), short(y["Kernel_Name"])), []).append(int(y["Start_Timestamp"]) - int(x["End_Timestamp"]))
        durs.setdefault(short(x["Kernel_Name"]), []).append(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
md = [f"# {tag}: the kernels of a frame and the stream time between them, TIMED REGION ONLY (rocprofv3 --kernel-trace of `bench.py --steps "
      f"{nsteps} --warmup {line['warmup']} --no-sweep --event-every 1`: configs[1], the dominant kernel of EVERY frame bracketed by HIP events)", "",
      f"Frame period (median of the {len(periods)} timed frames): {statistics.median(periods) / 1e3:.1f} µs; the bench line of the same run: "
      f"{line['ms_per_step'] * 1e3:.1f} µs per step, dominant kernel `{line['roofline']['kernel']}` {line['roofline']['avg_launch_ms'] * 1e3:.1f} µs by HIP events "
      f"({line['roofline']['launches']} launches).", "",
      "| from -> to | idle stream between them, median | frames |", "|---|---|---|"]
for (a, b), v in gaps.items():
    md.append(f"| `{a}` -> `{b}` | {statistics.median(v) / 1e3:.2f} µs | {len(v)} |")
md += ["", "| kernel | launches in the timed region | duration: mean | median | min | max (under the profiler) |", "|---|---|---|---|---|---|"]
for k, v in durs.items():
    md.append(f"| `{k}` | {len(v)} | {statistics.mean(v) / 1e3:.1f} µs | {statistics.median(v) / 1e3:.1f} | {min(v) / 1e3:.1f} | {max(v) / 1e3:.1f} |")
md += ["", "Only the two boundaries of the event bracket (before and after the bracketed kernel) leave the stream idle, about 6 µs each; the "
       "other launches of a frame follow each other without a gap.  Hence `bench.py` brackets every fourth frame by default "
       "(`--event-every`).  The whole-run averages of `rocprofv3 --stats` (`" + tag + "_bench_pf_kernel_stats.csv`) also contain the pre-roll "
       "and warm-up frames, which are slower (`r03_early_frames.md`); the table above is the timed region alone."]
(here / f"{tag}_trace_gaps.md").write_text("\n".join(md) + "\n")
print("\n".join(md))
