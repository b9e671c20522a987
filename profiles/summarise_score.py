#!/usr/bin/env python3
"""profiles/summarise_score.py TAG [SRC] [TITLE] — hardware counters of score_poses_kernel (profiles/collect_score_pmc.sh TAG)
-> profiles/TAG_pmc_score.md: per-launch averages, the derived rates, and what they say about the kernel's bound."""
import csv
import glob
import os
import sys
from pathlib import Path

tag = sys.argv[1]
here = Path(__file__).resolve().parent
src = Path(sys.argv[2]) if len(sys.argv) > 2 else here.parent / "gpurun_out"
title = sys.argv[3] if len(sys.argv) > 3 else "1 048 576 poses x 360 beams, 2048^2 EDT (BASELINE configs[2])"
POSES, BEAMS, CUS, CLK = 1048576, 360, 256, 2.4e9

vals = {}
for d in sorted(glob.glob(str(src / f"{tag}_scorepmc_*"))):
    for f in sorted(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)[-1:]:   # the newest run
        rows = [r for r in csv.DictReader(open(f)) if "score_poses_kernel" in r["Kernel_Name"]]
        for name in sorted(set(r["Counter_Name"] for r in rows)):
            v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name]
            vals[name] = sum(v[2:]) / max(len(v[2:]), 1)   # skip the two warm-up launches

g = vals.get
lane_gathers = POSES * BEAMS
wave_instr = g("SQ_INSTS_VMEM_RD", lane_gathers / 64)
md = [f"# {tag}: counters of `score_poses_kernel` — {title}", "",
      "One `rocprofv3 --pmc` pass per counter group (`profiles/collect_score_pmc.sh`), averages per launch.", "",
      "| counter | per launch |", "|---|---|"]
for k in sorted(vals):
    md.append(f"| {k} | {vals[k]:,.1f} |")
md += ["", "Derived:", ""]
if g("SQ_BUSY_CYCLES"):
    cyc = g("SQ_BUSY_CYCLES") / 32            # 32 shader engines
    md.append(f"* kernel length ~ {cyc:,.0f} cycles = {cyc / CLK * 1e3:.3f} ms at 2.4 GHz (SQ_BUSY_CYCLES / 32 shader engines).")
    if g("TA_BUSY_avr"):
        md.append(f"* texture addresser busy {100 * g('TA_BUSY_avr') / cyc:.0f} % of it (TA_BUSY_avr); {g('TA_BUSY_avr') / (wave_instr / CUS):.1f} busy cycles "
                  f"per wave-level gather instruction.")
if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
    md.append(f"* {wave_instr:,.0f} wave-level gathers ({lane_gathers:,} lane gathers): {g('TCP_TOTAL_CACHE_ACCESSES_sum') / wave_instr:.1f} L1 "
              f"accesses per gather instruction (64 = every lane its own access).")
if g("TCP_TCC_READ_REQ_sum") and g("TCP_TOTAL_CACHE_ACCESSES_sum"):
    md.append(f"* L1 (TCP) hit rate {100 * (1 - g('TCP_TCC_READ_REQ_sum') / g('TCP_TOTAL_CACHE_ACCESSES_sum')):.1f} %; "
              f"L2 (TCC) hit rate {100 * g('TCC_HIT_sum', 0) / max(g('TCC_HIT_sum', 0) + g('TCC_MISS_sum', 0), 1):.2f} %; HBM fetch "
              f"{g('FETCH_SIZE', 0) * 1024 * 2 / 1e6:.1f} MB per launch (2 x FETCH_SIZE): the EDT once.")
if g("SQ_INSTS_VALU"):
    md.append(f"* {g('SQ_INSTS_VALU') / wave_instr:.1f} vector-ALU instructions per beam and wavefront; VALUBusy {g('VALUBusy', 0):.0f} %.")
md += ["", "Reading: HBM is idle (the EDT is fetched once and lives in L2 / Infinity Cache), LDS has no bank conflicts (broadcast reads),",
       "the L2 serves every L1 miss.  What is busy is the texture addresser: an uncoalesced dword gather costs it about one cycle",
       "per L1 access, and neighbouring lanes rarely share a cell row (see the accesses per gather instruction above), so the kernel",
       f"runs at the chip's gather rate — {lane_gathers / CUS / max(g('TA_BUSY_avr', 1), 1):.1f} lane-gathers per busy cycle and CU — with the vector ALU second.",
       "Ordering the lanes by pose cell (`bench.py --presort-poses`) removes the L1 misses but not the per-lane cost."]
(here / f"{tag}_pmc_score.md").write_text("\n".join(md) + "\n")
print("\n".join(md[-12:]))
