#!/usr/bin/env python3
"""make_golden — capture golden vectors from the compiled, UNMODIFIED reference.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference); its outputs under
tests/golden/ are DATA (inputs + the reference's outputs), never reference source text.

    python oracle/make_golden.py

What it writes (SURVEY.md §8c "Fixtures to commit"):
  tests/golden/datasets.json        generator command lines + sha256 of the CSVs they produce
                                    (gen_dataset is libm-free, so the GPU box regenerates them)
  tests/golden/parity_pose.txt      `pose = ...` lines of Subsystem_1/main.c, 1000 frames
  tests/golden/parity_map.csv       its map_output.csv
  tests/golden/loop_pose.txt        same for Subsystem_1/main_accelerated.c, 3480 frames
  tests/golden/loop_map.csv
  tests/golden/hall_pose.txt        Subsystem_1/main.c on the "hall" set (out-of-bounds beams on ~390 frames)
  tests/golden/hall_map.csv
  tests/golden/frames_head.csv      first 3 text frames of the parity set (parser fixture)
  tests/golden/functions.npz        per-function inputs/outputs: angle table, scan clean-up,
                                    transform, local map, raster, EDT (both reference variants),
                                    single-pose scores (the res={0,0,0} trick), full FastMatch calls
  tests/golden/edt_standalone.npz   square-grid cases run through the STAND-ALONE scatter EDT file
                                    Submodule_2/Accelereated_Euclidean_Distance_Transform.c:1,36
                                    (`--only standalone` rewrites just this file)
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402

GOLD = ROOT / "tests" / "golden"
NB = 1079

DATASETS = {
    "parity": ["1000", str(NB), "-2.351831", "0.004363", "1"],
    "loop": ["3480", str(NB), "-2.351831", "0.004363", "2", "0.004", "0.0018"],
    # 34 x 20 m hall whose far end starts beyond the 24 m usable range: ~390 of the 1000 frames match with
    # beams outside the grid, i.e. the hit-scratch quirk shapes the map that the run builds
    "hall": ["1000", str(NB), "-2.351831", "0.004363", "3", "0.012", "0.0002", "1"],
}


def sha256(path: Path) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


class Ref:
    """ctypes view of oracle/_ref/libref_{main,accel}.so (see oracle/ref/ref_wrap.c)."""

    def __init__(self, which: str):
        L = C.CDLL(str(oracle.REF / f"libref_{which}.so"))
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        for n in ("ref_ranges", "ref_angles", "ref_scan_x", "ref_scan_y", "ref_scan_tx", "ref_scan_ty", "ref_map_x",
                  "ref_map_y", "ref_map_pose", "ref_local_x", "ref_local_y", "ref_fm_pose", "ref_fm_hits"):
            getattr(L, n).restype = fp
        for n in ("ref_scan_size", "ref_map_size", "ref_local_size", "ref_fm_hits_size"):
            getattr(L, n).restype = ip
        for n in ("ref_metric", "ref_pixel_size", "ref_top_left"):
            getattr(L, n).restype = fp
            getattr(L, n).argtypes = [C.c_int]
        for n in ("ref_grid", "ref_grid_size"):
            getattr(L, n).restype = ip
            getattr(L, n).argtypes = [C.c_int]
        L.ref_read_frame.argtypes = [C.c_char_p, C.c_int]
        L.ref_read_scan.argtypes = [C.c_int]
        L.ref_read_scan.restype = C.c_int
        L.ref_transform.argtypes = [fp]
        L.ref_initialise.argtypes = [fp]
        L.ref_extract_local_map.argtypes = [C.c_float]
        L.ref_occupancy_grid.argtypes = [C.c_float, C.c_float]
        L.ref_edt.argtypes = [C.c_int]
        L.ref_fastmatch.argtypes = [C.c_int, fp, fp]
        self.L = L

    @staticmethod
    def f3(v):
        return (C.c_float * 3)(*[float(x) for x in v])

    def arr(self, ptr, n, dtype=np.float32):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).copy()

    def grid(self, which):
        ld = 400 if which else 200
        g = np.ctypeslib.as_array(self.L.ref_grid(which), shape=(ld, ld))
        m = np.ctypeslib.as_array(self.L.ref_metric(which), shape=(ld, ld))
        return g, m

    def grid_meta(self, which):
        gs = self.arr(self.L.ref_grid_size(which), 2, np.int32)
        tl = self.arr(self.L.ref_top_left(which), 2)
        px = self.arr(self.L.ref_pixel_size(which), 1)[0]
        return dict(rows=int(gs[0]), cols=int(gs[1]), ld=400 if which else 200, pixel=px, min_x=tl[0], min_y=tl[1])

    def score_single(self, which, pose):
        """Single-pose trick: res = {0,0,0} makes all 27 candidates the same pose."""
        self.L.ref_fastmatch(which, self.f3(pose), self.f3([0, 0, 0]))
        n = self.L.ref_fm_hits_size()[0]
        hits = self.arr(self.L.ref_fm_hits(), n)
        s = np.float32(0)
        for h in hits:   # the reference's in-order float accumulation (main.c:516)
            s = np.float32(s + h)
        return s, n, hits


def whole_program(tmp: Path, out: dict):
    for name, args in DATASETS.items():
        csv = tmp / f"{name}.csv"
        oracle.run_tool("gen_dataset", csv, *args)
        out[name] = {"gen_args": args, "sha256": sha256(csv), "bytes": csv.stat().st_size}
    env = dict(os.environ)
    for name, exe in (("parity", "main_ref"), ("loop", "main_accel_ref"), ("hall", "main_ref")):
        env["ORACLE_DATASET"] = str(tmp / f"{name}.csv")
        env["ORACLE_MAP_OUT"] = str(GOLD / f"{name}_map.csv")
        r = subprocess.run([str(oracle.REF / exe)], env=env, check=True, capture_output=True, text=True)
        poses = [ln for ln in r.stdout.splitlines() if ln.startswith("pose =")]
        (GOLD / f"{name}_pose.txt").write_text("\n".join(poses) + "\n")
        out[name]["reference_program"] = {"main_ref": "Subsystem_1/main.c", "main_accel_ref": "Subsystem_1/main_accelerated.c"}[exe]
        out[name]["pose_lines"] = len(poses)
    with open(tmp / "parity.csv") as f:
        (GOLD / "frames_head.csv").write_text("".join(f.readline() for _ in range(3)))


def per_function(tmp: Path) -> dict:
    rm, ra = Ref("main"), Ref("accel")
    csv = str(tmp / "parity.csv").encode()
    g: dict[str, np.ndarray] = {}
    rng = np.random.default_rng(20240607)

    # ---- A1/A2: frame parse, angle table, scan clean-up (frames 0, 41, 400)
    rm.L.ref_set_lidar()
    ra.L.ref_set_lidar()
    g["angles"] = rm.arr(rm.L.ref_angles(), NB)
    for fr in (0, 41, 400):
        rm.L.ref_read_frame(csv, fr)
        g[f"ranges_{fr}"] = rm.arr(rm.L.ref_ranges(), NB)
        n = rm.L.ref_read_scan(24)
        g[f"scan_x_{fr}"] = rm.arr(rm.L.ref_scan_x(), n)
        g[f"scan_y_{fr}"] = rm.arr(rm.L.ref_scan_y(), n)

    # ---- A3/A4/A5/A6 on a realistic state: map := scan 0 at the origin; scan := frame 41 at a guess
    def setup(ref: Ref):
        ref.L.ref_read_frame(csv, 0)
        ref.L.ref_read_scan(24)
        ref.L.ref_transform(ref.f3([0, 0, 0]))
        ref.L.ref_initialise(ref.f3([0, 0, 0]))
        ref.L.ref_read_frame(csv, 41)
        n = ref.L.ref_read_scan(24)
        ref.L.ref_transform(ref.f3([0.15, 0.004, -0.024]))
        ref.L.ref_extract_local_map(1.0)
        ref.L.ref_occupancy_grid(0.2, 0.1)
        return n

    n41 = setup(rm)
    setup(ra)
    g["state_pose"] = np.array([0.15, 0.004, -0.024], np.float32)
    g["state_tx"] = rm.arr(rm.L.ref_scan_tx(), n41)
    g["state_ty"] = rm.arr(rm.L.ref_scan_ty(), n41)
    nm = rm.L.ref_map_size()[0]
    g["state_map_x"] = rm.arr(rm.L.ref_map_x(), nm)
    g["state_map_y"] = rm.arr(rm.L.ref_map_y(), nm)
    nl = rm.L.ref_local_size()[0]
    g["state_local_x"] = rm.arr(rm.L.ref_local_x(), nl)
    g["state_local_y"] = rm.arr(rm.L.ref_local_y(), nl)
    for w in (0, 1):
        mm, am = rm.grid_meta(w), ra.grid_meta(w)
        assert mm == am, (mm, am)
        g[f"state_meta_{w}"] = np.array([mm["rows"], mm["cols"], mm["ld"]], np.int32)
        g[f"state_metaf_{w}"] = np.array([mm["pixel"], mm["min_x"], mm["min_y"]], np.float32)
        occ, met = rm.grid(w)
        _, met_a = ra.grid(w)
        r, c = mm["rows"], mm["cols"]
        assert np.array_equal(met[:r, :c].view(np.uint32), met_a[:r, :c].view(np.uint32)), "main.c vs main_accelerated.c EDT differ"
        g[f"state_occ_{w}"] = occ[:r, :c].astype(np.int8)
        g[f"state_edt_{w}"] = met[:r, :c].copy()

    # ---- A6 stand-alone EDT cases, written straight into the reference's grids
    def edt_case(tag, which, rows, cols, occ_rc, use_main=True):
        for ref, vname in ((rm, "main"), (ra, "accel")):
            if vname == "main" and not use_main:
                continue
            occ, met = ref.grid(which)
            occ[:] = 0
            occ[:rows, :cols] = occ_rc
            met[:] = -1.0   # sentinel: cells outside rows x cols must stay untouched (Q7)
            gs = ref.L.ref_grid_size(which)
            gs[0], gs[1] = rows, cols
            ref.L.ref_edt(which)
            res = met[:rows, :cols].copy()
            assert np.all(met[rows:, :] == -1.0) and np.all(met[:, cols:] == -1.0)
            if f"edt_{tag}_out" in g:
                assert np.array_equal(g[f"edt_{tag}_out"].view(np.uint32), res.view(np.uint32)), tag
            g[f"edt_{tag}_out"] = res
        g[f"edt_{tag}_occ"] = occ_rc.astype(np.int8)
        g[f"edt_{tag}_which"] = np.array([which], np.int32)

    edt_case("empty", 0, 23, 31, np.zeros((23, 31), np.int32))
    one = np.zeros((40, 57), np.int32); one[17, 44] = 1
    edt_case("single", 0, 40, 57, one)
    edt_case("sparse_nonsquare", 0, 61, 83, (rng.random((61, 83)) < 0.02).astype(np.int32))
    edt_case("dense", 0, 50, 37, (rng.random((50, 37)) < 0.6).astype(np.int32))
    edt_case("full", 0, 18, 25, np.ones((18, 25), np.int32))
    edt_case("tall_fine", 1, 231, 97, (rng.random((231, 97)) < 0.01).astype(np.int32))
    edt_case("max_coarse", 0, 200, 200, (rng.random((200, 200)) < 0.01).astype(np.int32), use_main=False)
    edt_case("max_fine", 1, 400, 400, (rng.random((400, 400)) < 0.004).astype(np.int32), use_main=False)

    # ---- A7: restore the realistic state (EDT cases clobbered the grids), then score poses
    setup(rm)
    sx = rm.arr(rm.L.ref_scan_x(), n41)
    sy = rm.arr(rm.L.ref_scan_y(), n41)
    assert np.array_equal(sx, g["scan_x_41"])
    K = 48
    base = np.array([0.164, 0.0043, -0.0246], np.float32)
    poses = base + (rng.standard_normal((K, 3)) * np.array([0.08, 0.08, 0.02])).astype(np.float32)
    poses[0] = base
    poses[1] = base + np.array([30.0, 0, 0], np.float32)      # everything out of bounds -> score 0, count 0
    poses[2] = base + np.array([2.5, -1.5, 0.5], np.float32)  # partially out of bounds
    poses = poses.astype(np.float32)
    g["score_poses"] = poses
    ct, st = oracle.libm_cos_sin(poses[:, 2])
    g["score_ct"], g["score_st"] = ct, st
    for w in (0, 1):
        sc, cn, hh = [], [], []
        for p in poses:
            s, n, h = rm.score_single(w, p)
            sc.append(s); cn.append(n); hh.append(np.pad(h, (0, NB - n)))
        g[f"score_val_{w}"] = np.array(sc, np.float32)
        g[f"score_cnt_{w}"] = np.array(cn, np.int32)
        g[f"score_hits_{w}"] = np.array(hh, np.float32)

    # full lattice calls with the reference's two resolutions (main.c:832-833) + odd ones
    calls = []
    for w, guess, res in [
        (0, base, [0.05, 0.05, 0.008727]),
        (1, base, [0.05, 0.05, 0.008727]),
        (1, base, [0.025, 0.025, 0.004363]),
        (1, base + np.array([0.11, -0.07, 0.013], np.float32), [0.025, 0.025, 0.004363]),
        (0, base + np.array([-0.3, 0.2, -0.05], np.float32), [0.05, 0.05, 0.008727]),
        (1, base + np.array([2.5, -1.5, 0.5], np.float32), [0.05, 0.9, 0.008727]),   # res[1] must be ignored
        (0, base, [0.0, 0.0, 0.0]),
    ]:
        guess = np.asarray(guess, np.float32)
        res = np.asarray(res, np.float32)
        rm.L.ref_fastmatch(w, rm.f3(guess), rm.f3(res))
        out_pose = rm.arr(rm.L.ref_fm_pose(), 3)
        nbest = rm.L.ref_fm_hits_size()[0]
        hits_after = rm.arr(rm.L.ref_fm_hits(), NB)
        # how many of those entries are live = in-bounds count of the LAST candidate (Q2)
        last = np.array([np.float32(guess[0] + res[0]), np.float32(guess[1] + res[0]), np.float32(guess[2] + res[2])], np.float32)
        _, nlast, hl = rm.score_single(w, last)
        assert np.array_equal(hl.view(np.uint32), hits_after[:nlast].view(np.uint32)), "Q2 check"
        calls.append((w, guess, res, out_pose, nbest, nlast, hits_after))
    g["fm_which"] = np.array([c[0] for c in calls], np.int32)
    g["fm_guess"] = np.array([c[1] for c in calls], np.float32)
    g["fm_res"] = np.array([c[2] for c in calls], np.float32)
    g["fm_pose"] = np.array([c[3] for c in calls], np.float32)
    g["fm_nbest"] = np.array([c[4] for c in calls], np.int32)
    g["fm_nlast"] = np.array([c[5] for c in calls], np.int32)
    g["fm_hits"] = np.array([np.where(np.arange(NB) < c[5], c[6], 0) for c in calls], np.float32)
    return g


def standalone_edt() -> dict:
    """Square grids through Submodule_2/Accelereated_Euclidean_Distance_Transform.c (its (width, height) order is
    only meaningful when both are equal).  Cells outside the n x n square must stay untouched."""
    L = C.CDLL(str(oracle.REF / "libref_edt_standalone.so"))
    L.ref_sa_grid.restype = C.POINTER(C.c_int)
    L.ref_sa_metric.restype = C.POINTER(C.c_float)
    L.ref_sa_grid.argtypes = L.ref_sa_metric.argtypes = [C.c_int]
    L.ref_sa_edt.argtypes = [C.c_int, C.c_int, C.c_int]
    rng = np.random.default_rng(20261004)
    g: dict[str, np.ndarray] = {}
    cases = [("sq64_sparse", 0, 64, 0.02), ("sq120_dense", 0, 120, 0.3), ("sq200_max", 0, 200, 0.008),
             ("sq37_single", 0, 37, None), ("sq250_fine", 1, 250, 0.01), ("sq400_max", 1, 400, 0.004), ("sq16_empty", 1, 16, 0.0)]
    for tag, which, n, dens in cases:
        ld = 400 if which else 200
        occ = np.ctypeslib.as_array(L.ref_sa_grid(which), shape=(ld, ld))
        met = np.ctypeslib.as_array(L.ref_sa_metric(which), shape=(ld, ld))
        occ[:] = 0
        if dens is None:
            occ[n // 3, n - 5] = 1
        else:
            occ[:n, :n] = rng.random((n, n)) < dens
        met[:] = -1.0
        L.ref_sa_edt(which, n, n)
        assert np.all(met[n:, :] == -1.0) and np.all(met[:, n:] == -1.0)
        g[f"{tag}_occ"] = occ[:n, :n].astype(np.int8)
        g[f"{tag}_out"] = met[:n, :n].copy()
        g[f"{tag}_which"] = np.array([which], np.int32)
    return g


def main():
    if not Path("/root/reference/Subsystem_1/main.c").exists():
        sys.exit("make_golden: /root/reference is not present; golden vectors can only be made in the build container")
    oracle.build(ref=True)
    GOLD.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(GOLD / "edt_standalone.npz", **standalone_edt())
    if sys.argv[1:] == ["--only", "standalone"]:
        print("golden vectors written to", GOLD / "edt_standalone.npz")
        return
    info: dict = {}
    with tempfile.TemporaryDirectory() as d:
        tmp = Path(d)
        whole_program(tmp, info)
        g = per_function(tmp)
    np.savez_compressed(GOLD / "functions.npz", **g)
    (GOLD / "datasets.json").write_text(json.dumps(info, indent=1) + "\n")
    print("golden vectors written to", GOLD)
    for p in sorted(GOLD.iterdir()):
        print(f"  {p.name:24s} {p.stat().st_size:9d} B")


if __name__ == "__main__":
    main()
