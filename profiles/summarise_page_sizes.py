#!/usr/bin/env python3
"""profiles/summarise_page_sizes.py TAG — what profiles/collect_page_sizes.sh TAG left under gpurun_out/ ->
profiles/TAG_page_sizes.md (the paged update with 32-, 16- and 8-landmark pages, next to pure copies of such pages)."""
import json
import shutil
import sys
from pathlib import Path

tag = sys.argv[1]
here = Path(__file__).resolve().parent
src = Path(sys.argv[2]) if len(sys.argv) > 2 else here.parent / "gpurun_out"
work = [("landmarks500", "65 536 x 500"), ("landmarks5000", "65 536 x 5 000"), ("particles1048576landmarks1000", "1 048 576 x 1 000")]
md = [f"# {tag}: the paged landmark update with 32- (product), 16- and 8-landmark pages, 32 landmarks observed per frame, every "
      "frame resampled", "",
      "`bench.py --paged --observed 32 --steps 60 --warmup 10` on measurement builds of the library (`make -C csrc OUT=../lib_pP "
      "EXTRA=-DSLAM_PAGE_LANDMARKS=P`, loaded through `SLAM_HIP_LIB`; `tests/test_gpu_paged.py` green on each).  HIP-event "
      "average of the update kernel (`ekf_paged_lds_kernel`) and the whole frame.", "",
      "| workload | 32-landmark pages (640 B): update, frame | 16 (320 B) | 8 (160 B) |", "|---|---|---|---|"]
for key, name in work:
    cells = []
    for P in (32, 16, 8):
        f = src / f"{tag}_pagesize_{P}_{key}.json"
        try:
            d = json.loads(f.read_text())
            cells.append(f"{d['roofline']['avg_launch_ms'] * 1e3:.1f} µs, {d['ms_per_step']:.4f} ms")
        except Exception:
            cells.append("—")
    md.append(f"| {name} | " + " | ".join(cells) + " |")
md += ["", "Touched pages per particle and frame on this workload (32 nearest of Morton-ordered landmarks): 5.0 / 7.0 / 10.6 at 500 "
       "landmarks, 4.0 / 5.3 / 8.3 at 5 000 — page bytes read + written per particle 6.4 / 4.5 / 3.4 KB at 500 landmarks (1.28 KB of them "
       "algorithmic).  The flat page table is 16 / 32 / 63 entries at 500 landmarks and 157 / 313 / 625 at 5 000 (copied per particle "
       "and frame), which is what the small pages lose to there.", "",
       "Smaller pages move fewer bytes and are not faster: scattered small pages copy at a lower rate (below), so 32 stays.", ""]
pc = src / f"{tag}_page_copy_ceiling.txt"
if pc.exists():
    shutil.copy(pc, here / f"{tag}_page_copy_ceiling.txt")
    md += [f"## Pure copies of scattered pages (`profiles/page_copy_ceiling.hip`, `{tag}_page_copy_ceiling.txt`)", "", "```"]
    md += [ln.rstrip() for ln in pc.read_text().splitlines() if "scattered" in ln or ln.startswith("particles")]
    md += ["```", "",
           "A page table that names random pages of the pool, T pages read and T written per particle, nothing else: at 65 536 "
           "particles 5 x 640 B copy in 89 µs, 7 x 320 B in 70 µs, 11 x 160 B in 78 µs (37.7 / 29.3 / 32.8 µs when 16 neighbours share their "
           "source pages); at 1 048 576 particles 1.20 / 1.19 / 1.47 ms.  160-byte pages reach 2.9 TB/s where 640-byte pages reach 4.7: the "
           "bytes saved are lost to the rate.  The product kernel (71-74 µs at 65 536 x 500 with about half the ancestors distinct) sits "
           "inside the band of the pure copy of its own page size."]
(here / f"{tag}_page_sizes.md").write_text("\n".join(md) + "\n")
print("\n".join(md[:14]))
