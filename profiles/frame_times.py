"""profiles/frame_times.py [layout] [repeats] — per-frame wall times (host clock, a synchronisation per frame) of the configs[1]
frame with the 32 nearest landmarks observed, for one map layout: looks for frames that take far longer than the rest
(measurement tooling; the session calls are those of bench.py's end_to_end_obs32 leg)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

layout = sys.argv[1] if len(sys.argv) > 1 else "auto"
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nosync = len(sys.argv) > 3 and sys.argv[3] in ("nosync", "stages")   # like bench.py's leg: the host runs ahead, one synchronisation per region
settle = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0               # seconds of host sleep between session set-up and the first (warm-up) frame
stages = len(sys.argv) > 3 and sys.argv[3] == "stages"                # ... with every stage bracketed by HIP events: which stage holds a stall
pkg = load_package()
dev = torch.device("cuda", 0)
eng = pkg.Engine(0)
n, L, Lp, beams = 65536, 500, 512, 360
rng = np.random.default_rng(4321)
lm = bench.make_landmarks(L, rng)
fr = bench.make_frames(60, beams, lm, rng, 32)
pixel = np.float32(20.48 / 1024)
occ = bench.occupancy(1024, float(pixel), -4.24, -10.24)
d_occ = torch.from_numpy(occ).to(dev)
d_edt = torch.empty((1024, 1024), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
eng.edt_dev(d_occ, 1024, 1024, 1024, 10.0, d_edt)
eng.grid_set_dev(0, d_edt, pkg.grid_meta(1024, 1024, 1024, pixel, np.float32(-4.24), np.float32(-10.24)))
d_scan = torch.from_numpy(np.stack([np.stack([f["bx"], f["by"]]) for f in fr])).to(dev)
tabs = bench.obs_tables(torch, fr, L, dev)
for rep in range(repeats):
    ses = pkg.PfSession(eng, n, L, sigma=bench.SIGMA, meas_var=bench.MEAS_VAR, score_gain=bench.SCORE_GAIN, seed=1234, map_layout=layout)
    g = torch.Generator(device="cpu").manual_seed(1234)
    p0 = bench.true_pose(0)
    ses.set_poses(*[(p0[k] + s * torch.randn(n, generator=g)).numpy() for k, s in ((0, 0.05), (1, 0.05), (2, 0.01))])
    m0 = torch.zeros((n, 5, Lp), dtype=torch.float32, device=dev)
    bench.fill_maps(torch, m0, lm, L, dev, n)
    torch.cuda.synchronize()
    ses.set_map_dev(m0, 5 * Lp, Lp)
    eng.sync()
    del m0
    times, host = [], []
    if nosync:
        warm, t0 = 12, 0.0
        for k in range(52):
            if k == 0 and settle > 0:   # before the warm-up frames, so that the timed frames follow warm ones directly
                torch.cuda.synchronize()
                time.sleep(settle)
            if k == warm:
                torch.cuda.synchronize()
                if stages:
                    eng.profile_enable(*range(eng.PROF_COUNT))
                    for kk in range(eng.PROF_COUNT):
                        eng.profile_read(kk)
                t0 = time.perf_counter()
            eng.scan_set_dev(d_scan[k, 0], d_scan[k, 1], beams)
            eng.obs_set_dev(tabs[k, 0], tabs[k, 1], L)
            ses.step(0, fr[k]["dp"], True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if stages:
            tot = {eng.PROF_NAMES[kk]: eng.profile_read(kk) for kk in range(eng.PROF_COUNT)}
            eng.profile_enable()
            print(f"rep {rep} stages (total ms over 40 frames): " + ", ".join(f"{k} {v[0]:.3f}" for k, v in tot.items() if v[1]) +
                  f"; sum {sum(v[0] for v in tot.values()):.3f}, wall {1e3 * (t2 - t0):.3f}")
        print(f"rep {rep} layout {layout} nosync: {1e3 * (t2 - t0) / 40:.4f} ms per frame (host issued the 40 frames in {1e3 * (t1 - t0):.2f} ms), "
              f"ended on {'pages' if ses.is_paged() else 'rows'}, conversions {ses.conversions() if hasattr(ses, 'conversions') else '?'}")
        ses.close()
        continue
    for k in range(len(fr)):
        t0 = time.perf_counter()
        eng.scan_set_dev(d_scan[k, 0], d_scan[k, 1], beams)
        eng.obs_set_dev(tabs[k, 0], tabs[k, 1], L)
        ses.step(0, fr[k]["dp"], True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t0))
        host.append(1e3 * (t1 - t0))
    print(f"rep {rep} layout {layout} ended on {'pages' if ses.is_paged() else 'rows'}: median {np.median(times):.3f} ms, "
          f"frames over 1 ms: {[(k, round(t, 2), round(h, 2)) for k, (t, h) in enumerate(zip(times, host)) if t > 1.0]}")
    ses.close()
eng.close()
