#!/usr/bin/env python3
"""profiles/summarise.py TAG [SRC] — turn what profiles/collect.sh TAG left under gpurun_out/ into the tracked files:

    profiles/TAG_bench_pf_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the default bench (in-filter launches only)
    profiles/TAG_bench_score_kernel_stats.csv   the same for the score-only microbench (configs[2])
    profiles/TAG_pmc_ekf.md                     HBM traffic of ekf_update_kernel from separate --pmc passes
    profiles/traffic.json                       HBM bytes per launch, the `traffic` field of bench.py's roofline

Corrections as MI355X_MICROARCH.md (HBM section) prescribes: counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of a coalesced streaming read (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE is exact.  The factor 2 is
re-checked on the `ekf` sweep (identity ancestors: every row read once, buffers far larger than the Infinity Cache, so
the read bytes are known a priori: 20 B x n x padded landmarks).
"""
import csv
import glob
import os
import json
import shutil
import sys
from pathlib import Path

tag = sys.argv[1]
here = Path(__file__).resolve().parent
src = Path(sys.argv[2]) if len(sys.argv) > 2 else here.parent / "gpurun_out"
KERNEL = "ekf_update_kernel"
n, L = 65536, 500
Lp = (L + 31) // 32 * 32


def counter(mode, name):
    f = max(glob.glob(str(src / f"{tag}_pmc_{name}_{mode}" / "*" / "*counter_collection.csv")), key=os.path.getmtime)   # the newest run
    # both out-of-place kernels (ekf_update_kernel / ekf_update_group_kernel) count as "the EKF kernel"
    rows = [r for r in csv.DictReader(open(f)) if ("ekf_update_" in r["Kernel_Name"] or "frame_front_kernel" in r["Kernel_Name"])
            and r["Counter_Name"] == name]
    global last_kernel
    names = [r["Kernel_Name"].replace("void slam::", "").replace("slam::", "").replace("(anonymous namespace)::", "").split("(")[0]
             for r in rows[4:10]]
    last_kernel = max(set(names), key=names.count) if names else KERNEL   # the kernel the steady frames ran
    return [float(r["Counter_Value"]) * 1024 for r in rows]


last_kernel = KERNEL


for mode, out in (("pf", "bench_pf"), ("score", "bench_score")):
    f = sorted(glob.glob(str(src / f"{tag}_trace_{mode}" / "*" / "*kernel_stats.csv")), key=os.path.getmtime)
    if f:
        shutil.copy(f[-1], here / f"{tag}_{out}_kernel_stats.csv")

if (src / f"{tag}_copy_ceiling.txt").exists():
    shutil.copy(src / f"{tag}_copy_ceiling.txt", here / f"{tag}_copy_ceiling.txt")
cal = counter("ekf", "FETCH_SIZE")
factor = 20 * n * Lp / (sum(cal[2:]) / len(cal[2:]))
traffic = {}
md = [f"# {tag}: HBM traffic of `{KERNEL}` from PMC counters (separate --pmc passes, `profiles/collect.sh {tag}`)", "",
      "| run | counter | per-launch values (bytes = raw x 1024), first 12 launches |", "|---|---|---|"]
for mode, flags in (("ekf", "--mode ekf"), ("pf", "--mode pf"), ("pfobs32", "--mode pf --observed 32")):
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        v = counter(mode, name)
        md.append(f"| bench.py {flags} | {name} | " + ", ".join(f"{x / 1e6:.1f} MB" for x in v[:12]) + " |")
md += ["", f"Calibration on the `ekf` sweep (known 20 B x {n} x {Lp} = {20 * n * Lp / 1e6:.1f} MB read per launch, row padding "
       f"included): known / FETCH_SIZE = {factor:.3f} (the guide's factor 2)."]
steps = 10   # --steps 8 --warmup 2: launches 4.. are steady frames; the 12 launches after them are the no-reuse sweep
for mode, key, K in (("pf", f"pf:{n}:360:{L}:1024", L), ("pfobs32", f"pf:{n}:360:{L}:1024:obs32", 32)):
    fetch, write = counter(mode, "FETCH_SIZE"), counter(mode, "WRITE_SIZE")
    steady = slice(4, steps)
    rd = 2.0 * sum(fetch[steady]) / len(fetch[steady])
    wr = sum(write[steady]) / len(write[steady])
    alg = 40 * n * K
    traffic[key] = {"bytes_per_launch": rd + wr, "kernel": last_kernel.split("<")[0], "read_bytes": rd, "write_bytes": wr,
                    "algorithmic_bytes": alg, "fetch_size_calibration_factor": factor, "source": f"profiles/{tag}_pmc_ekf.md",
                    "measured": f"{tag}: builder-run rocprofv3 --pmc passes of bench.py --steps 8 --warmup 2"}
    md.append(f"`{key}`, steady frames: read = 2 x FETCH_SIZE = {rd / 1e6:.1f} MB, write = {wr / 1e6:.1f} MB, total "
              f"{(rd + wr) / 1e6:.1f} MB per launch vs {alg / 1e6:.1f} MB algorithmic (40 B x n x {K} observed).")
md += ["", "The writes are the 20 B x n x 512 of the padded rows (every row is rewritten by the out-of-place update); the reads are "
       "only the rows of the DISTINCT resample ancestors — the offspring of one ancestor are neighbouring particles and "
       "re-read its row from L2.  The launches after the timed region in each list are bench.py's stage pass (the two-launch "
       "path) and its no-reuse sweep (identity ancestors: read = write = 671 MB)."]
(here / "traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")
(here / f"{tag}_pmc_ekf.md").write_text("\n".join(md) + "\n")
print("\n".join(md[-6:]))
