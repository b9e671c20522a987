#!/usr/bin/env python3
"""profiles/early_frames.py TAG — the front kernel of a frame, frame by frame from a cold start:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/TAG_early -- python3 bench.py --no-cpu-baseline --no-extra-legs \
        --no-sweep --steps 150 --warmup 0 --preroll 0 --events none
-> profiles/TAG_early_frames.md (why bench.py pre-rolls the filter before its warm-up steps)."""
import csv
import glob
import os
import sys
from pathlib import Path

tag = sys.argv[1]
here = Path(__file__).resolve().parent
src = Path(sys.argv[2]) if len(sys.argv) > 2 else here.parent / "gpurun_out"
f = max(glob.glob(str(src / f"{tag}_early" / "*" / "*kernel_trace.csv")), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
fr = [dur(r) for r in rows if "frame_front_kernel" in r["Kernel_Name"]]
sc = [dur(r) for r in rows if "score_poses" in r["Kernel_Name"]]
avg = lambda v: sum(v) / max(len(v), 1)
md = [f"# {tag}: the front kernel of a frame, frame by frame from a cold start (configs[1])", "",
      "`rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --no-extra-legs --no-sweep --steps 150 --warmup 0 --preroll 0 --events none`:",
      "durations of `frame_front_kernel<2, 4, 4, 8>` (motion + score + landmark update of one frame) in launch order, µs; frames 0 and 1 run",
      "the two-launch path (no resample indices yet / group size not yet known).", "", "```", " ".join(f"{d:.0f}" for d in fr[:150]), "```", "",
      f"Frames 3-10: {avg(fr[1:9]):.0f} µs; 15-30: {avg(fr[13:28]):.0f}; 40-50: {avg(fr[38:48]):.0f}; 90-100: {avg(fr[88:98]):.0f}; 140-148: {avg(fr[138:146]):.0f}.  "
      f"The stand-alone scorer of frame 0: {sc[0]:.0f} µs, of the stage pass after the run: {avg(sc[1:]):.0f} µs.", "",
      "The filter starts from poses spread 5 cm / 0.01 rad around the truth (SURVEY 8d) and takes on the order of a hundred frames to settle to",
      "the spread its motion noise (1 cm, 2 mrad) and its observations sustain; until then neighbouring particles (neighbouring lanes of the",
      "scorer) lie farther apart and a wavefront's gathers fall into more cache lines: the scoring workgroups of the front kernel take longer.",
      "With the sensor-frame landmark update of rounds 1-2 (vector ALU 58-67 % busy) the front kernel felt that in full — 193 µs at frame 3,",
      "175 at 40, 160 at 90, 154 at 140 on the box that made the first version of this file; with the world-frame update the scorer mostly hides",
      "behind the row stores and the series above is what is left.  `bench.py` runs `--preroll 120` untimed frames before its W warm-up steps",
      "(the metric is a steady-state rate); `--preroll 0` starts cold."]
(here / f"{tag}_early_frames.md").write_text("\n".join(md) + "\n")
print("\n".join(md[9:11]))
