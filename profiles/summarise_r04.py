#!/usr/bin/env python3
"""profiles/summarise_r04.py TAG [KERNEL-SUBSTRING] — the runs of collect_r04.sh -> profiles/TAG_kernel_stats.csv (the stats of
the kernel trace) and profiles/TAG_pmc.md: per-launch averages of every counter over the LAST six launches of the dominant
kernel before the bench's stage pass (steady state: the bench ran its 120-frame pre-roll first), HBM bytes corrected as
MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 32-byte units on gfx950 twice too few: x 2 after the kB scaling;
WRITE_SIZE as is)."""
import csv
import glob
import os
import shutil
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

args = [a for a in sys.argv[1:] if not a.startswith("--traffic-key=")]
traffic_key = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--traffic-key=")), None)
tag = args[0]
want = args[1] if len(args) > 1 else "frame_front_kernel"
here = Path(__file__).resolve().parent
src = here.parent / "gpurun_out"
stats = glob.glob(str(src / f"{tag}_trace" / "*" / "*kernel_stats.csv"))
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), here / f"{tag}_kernel_stats.csv")
vals, kern = defaultdict(list), None
for d in sorted(glob.glob(str(src / f"{tag}_pmc_*"))):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        continue
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        if want in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kern = r["Kernel_Name"].replace("void slam::", "").replace("slam::", "").replace("(anonymous namespace)::", "").split("(")[0]
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=here).stdout.strip()
md = [f"# {tag}: hardware counters of `{kern}` in steady state (`bench.py --steps 8 --warmup 2 --events none --no-sweep`, 120 pre-roll "
      f"frames first; one `rocprofv3 --pmc` pass per group; HEAD {head})", "",
      "| counter | per-launch average over the last six fused launches |", "|---|---|"]
g = {}
for name, v in vals.items():
    v = v[-6:]
    g[name] = sum(v) / len(v)
    md.append(f"| {name} | {g[name]:.5g} |")
if "FETCH_SIZE" in g and "WRITE_SIZE" in g:
    rd, wr = g["FETCH_SIZE"] * 1024 * 2, g["WRITE_SIZE"] * 1024
    md += ["", f"HBM traffic per launch: {rd / 1e6:.1f} MB read (FETCH_SIZE x 1024 x 2, the gfx950 correction) + {wr / 1e6:.1f} MB written "
               f"(WRITE_SIZE x 1024) = {(rd + wr) / 1e9:.4f} GB."]
if "SQ_WAVE_CYCLES" in g:
    for a in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
        if a in g:
            md.append(f"{a} / SQ_WAVE_CYCLES = {g[a] / g['SQ_WAVE_CYCLES']:.3f}")
(here / f"{tag}_pmc.md").write_text("\n".join(md) + "\n")
print("\n".join(md))
if traffic_key and "FETCH_SIZE" in g and "WRITE_SIZE" in g:   # the record bench.py's roofline.traffic comes from
    import json

    tf = here / "traffic.json"
    rec = json.loads(tf.read_text()) if tf.exists() else {}
    rd, wr = g["FETCH_SIZE"] * 1024 * 2, g["WRITE_SIZE"] * 1024
    rec[traffic_key] = {"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "kernel": kern.split("<")[0],
                        "kernel_template": kern, "source": f"profiles/{tag}_pmc.md", "head": head,
                        "bench_args": "--steps 8 --warmup 2 --events none --no-sweep (120 pre-roll frames; the last six fused launches)",
                        "valu_instructions_per_launch": g.get("SQ_INSTS_VALU"), "valu_busy_pct": g.get("VALUBusy")}
    tf.write_text(json.dumps(rec, indent=1) + "\n")
    print(f"traffic.json: {traffic_key} <- {(rd + wr) / 1e9:.4f} GB")
