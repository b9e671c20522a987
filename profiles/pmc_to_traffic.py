#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into profiles/traffic.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_FETCH_SIZE_pf  -- python3 bench.py --mode pf  --steps 6 --warmup 2 --no-cpu-baseline --events none
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_WRITE_SIZE_pf  -- python3 bench.py --mode pf  ...
    (and the same two with --mode ekf: the calibration case, whose read bytes are known a priori)
    python profiles/pmc_to_traffic.py gpurun_out r01

Corrections, as MI355X_MICROARCH.md §HBM prescribes: counters are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a coalesced streaming read (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE is
exact.  The factor 2 is re-checked on the `ekf` sweep (no gather, every row read once — 20 B per particle x
padded landmark — buffers far larger than the Infinity Cache, so HBM read bytes are known a priori).
"""
import csv
import glob
import json
import sys
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
here = Path(__file__).resolve().parent
KERNEL = "ekf_update_kernel"


def counter(mode, name):
    f = glob.glob(str(src / f"pmc_{name}_{mode}" / "*" / "*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name]
    vals = [float(r["Counter_Value"]) * 1024 for r in rows]
    grid = int(rows[-1]["Grid_Size"])
    return vals, grid


n, L = 65536, 500
Lp = (L + 31) // 32 * 32          # rows are padded to 32 floats per plane and whole 128-landmark batches are moved,
alg_side = 20 * n * L             # padding included: the sweep really reads and writes 20 B x n x Lp
cal, _ = counter("ekf", "FETCH_SIZE")
factor = 20 * n * Lp / (sum(cal[2:]) / len(cal[2:]))
fetch, _ = counter("pf", "FETCH_SIZE")
write, _ = counter("pf", "WRITE_SIZE")
steady = slice(4, None)   # skip the frames before the particle cloud has settled
rd = 2.0 * sum(fetch[steady]) / len(fetch[steady])
wr = sum(write[steady]) / len(write[steady])
out = {f"pf:{n}:360:{L}:1024": {KERNEL: rd + wr, "read_bytes": rd, "write_bytes": wr,
                                "algorithmic_bytes": 2 * alg_side, "fetch_size_calibration_factor": factor,
                                "source": f"profiles/{tag}_pmc_ekf.md"}}
(here / "traffic.json").write_text(json.dumps(out, indent=1) + "\n")
md = [f"# {tag}: HBM traffic of `{KERNEL}` from PMC counters (separate --pmc passes)", "",
      "| run | counter | per-launch values (bytes, raw x 1024) |", "|---|---|---|"]
for mode in ("ekf", "pf"):
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        v, _ = counter(mode, name)
        md.append(f"| bench.py --mode {mode} | {name} | " + ", ".join(f"{x / 1e6:.1f} MB" for x in v[:10]) + " |")
md += ["", f"Calibration on the `ekf` sweep (known 20 B x {n} x {Lp} = {20 * n * Lp / 1e6:.1f} MB read per launch, row "
       f"padding included): known / FETCH_SIZE = {factor:.3f} (the guide's factor 2).",
       f"`pf` mode, steady frames: read = 2 x FETCH_SIZE = {rd / 1e6:.1f} MB, write = {wr / 1e6:.1f} MB, total "
       f"{(rd + wr) / 1e6:.1f} MB per launch vs {2 * alg_side / 1e6:.1f} MB algorithmic "
       f"({100 * ((rd + wr) / (2 * alg_side) - 1):+.1f} %): the writes are the 20 B x n x {Lp} of the padded rows; the "
       f"reads are only the rows of the DISTINCT ancestors — the offspring of one ancestor are neighbouring "
       f"particles and re-read its row from L2."]
(here / f"{tag}_pmc_ekf.md").write_text("\n".join(md) + "\n")
print("\n".join(md[-2:]))
