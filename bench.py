#!/usr/bin/env python3
"""bench.py — particle-updates/s of the particle-filter frame loop on synthetic 360-beam scans.

    python bench.py [--gpus N --steps K --warmup W]
        N = 1: this process.  N > 1 without a launcher: bench.py starts `python -m torch.distributed.run --nnodes=1
        --nproc-per-node N --master-addr 127.0.0.1 ...` on itself BEFORE anything touches the GPU and relays rank 0's line.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W           (the driver's form: one rank per GPU, RCCL over xGMI)
    python bench.py --gpus N --transport local [--device-index 0]
        rehearsal: N ranks as threads of this process sharing ONE card through the in-process transport
        (slam_comm_create_local) — the whole sharded code path except RCCL itself; the number it prints is not a
        scaling figure (the ranks share a GPU).

A "step" is one whole frame of the hot path over one batch of synthetic input — ONE call into the C session
(slam_pf_step): motion sample -> scan-match score (the reference's FastMatch inner loop, Subsystem_1/main.c:459-518, for
every particle) -> per-particle x per-landmark 2x2 EKF -> weight normalisation -> systematic resample (gathers fused into
the next frame's motion / EKF kernels) -> exchange between GPUs.  Workload = BASELINE.json configs[1] per GPU: 65536
particles, 360 beams, 500 landmarks all observed, 1024 x 1024 EDT grid; weak scaling (per-GPU work fixed) unless
--scaling strong.  Particles, maps, EDT and scan are resident in HBM when the timed region starts, and so is the sensor
data of every frame (--host-sensor sends its 8.9 KB over PCIe frame by frame instead).
value = N_total_particles * K / max-over-ranks wall time.

Other workloads (never the default; used to fill BASELINE.md):
    --mode score   scan-match-only microbench (configs[2]: --particles 1048576 --grid 2048)
    --mode ekf     EKF sweep only (north-star roofline case: --particles 1048576 --landmarks 1000)
    --scaling strong --particles-total 1048576 --landmarks 1000
                   the north-star target: N ranks split the SAME problem (what ">= 6x at 8 GPUs" is defined on)
    --observed 32 [--map-layout rows|pages|auto]   the K-nearest end-to-end variant of SURVEY 8(d)

One JSON line on stdout (rank 0).  Beyond the contract's fields:
  roofline      dominant kernel of the timed region: launch duration from HIP events on the kernel's own stream
                (slam_profile_*).  `achieved` names its basis and the kernel it belongs to (`achieved_basis`,
                `achieved_kernel`): HBM traffic from a PMC record of this workload and kernel (profiles/traffic.json)
                over the live duration; else the no-reuse sweep; else algorithmic bytes.  `logical_rate_gbs` is always
                SURVEY 8(d)'s algorithmic bytes over the live duration.
  stage_avg_ms  every stage of a frame through the ABI's per-stage timers (a short pass after the timed region)
  north_star, copy_ceiling, end_to_end_obs32   (N = 1, default workload) bounded extra legs after the timed region:
                the EKF sweep at 1 048 576 x 1 000, a pure copy of the same shape, and the configs[1] frame with the 32
                nearest landmarks observed on rows and on pages; `headline_round2_method`: configs[1] measured the way
                BENCH_r01 / BENCH_r02 were (cold filter, 5 + 20 frames, every frame bracketed), as two launches and fused.
                `sharded_rehearsal_2_ranks_one_card`: the sharded session (two ranks as threads on this card, in-process
                transport) with its per-stage timers, run as a child process.  They never touch `value`.
  cpu_baseline  the CPU port of the same frame loop (oracle/, one thread) on a bounded sample; cpu_baseline_threads:
                its per-particle stages on up to 16 host threads; cpu_baseline_main_c: the reference's own pipeline
                (main.c rows A1-A8) on this host's CPU — the compiled reference when oracle/_ref/ travelled here, and
                the oracle's restatement — beside the engine's drop-in programs on the same scans.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import socket
import subprocess
import sys
import tempfile
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

# synthetic world: 15 x 11 m room with two boxes, in the reference's pose convention
ROOM = (-3.0, -5.5, 12.0, 5.5)
BOXES = [(5.0, 2.0, 7.0, 3.5), (2.0, -4.0, 3.0, -3.0)]
SIGMA, MEAS_VAR, SCORE_GAIN = (0.01, 0.01, 0.002), 0.02 ** 2 * 4, 0.02


def raycast(px, py, ang):
    """Range of rays from (px,py) at world angles `ang` against the room (from inside) and the boxes."""
    dx, dy = np.cos(ang), np.sin(ang)
    best = np.full(ang.shape, 1e30)
    for (x0, y0, x1, y1) in [ROOM] + BOXES:
        for ex in (x0, x1):
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (ex - px) / dx
            yy = py + t * dy
            ok = (t > 1e-9) & (yy >= y0) & (yy <= y1) & (t < best)
            best = np.where(ok, t, best)
        for ey in (y0, y1):
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (ey - py) / dy
            xx = px + t * dx
            ok = (t > 1e-9) & (xx >= x0) & (xx <= x1) & (t < best)
            best = np.where(ok, t, best)
    return best


def occupancy(grid, pixel, min_x, min_y):
    """Walls of the room and boxes rasterised 3 cells thick -> ~1 % occupied."""
    occ = np.zeros((grid, grid), np.int32)

    def cells(v, lo):
        return int(round((v - lo) / pixel))

    for (x0, y0, x1, y1) in [ROOM] + BOXES:
        c0, c1, r0, r1 = cells(x0, min_x), cells(x1, min_x), cells(y0, min_y), cells(y1, min_y)
        for r in (r0, r1):
            occ[max(r - 1, 0): r + 2, max(c0, 0): c1 + 1] = 1
        for c in (c0, c1):
            occ[max(r0, 0): r1 + 1, max(c - 1, 0): c + 2] = 1
    return occ


def morton(points, bits=10):
    """Z-order key of 2-D points inside the room (ids assigned along a space-filling sweep)."""
    q = np.clip(((points - [ROOM[0], ROOM[1]]) / [ROOM[2] - ROOM[0], ROOM[3] - ROOM[1]] * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = np.zeros(len(points), np.int64)
    for b in range(bits):
        key |= ((q[:, 0] >> b) & 1) << (2 * b) | ((q[:, 1] >> b) & 1) << (2 * b + 1)
    return key


def true_pose(f):
    """Robot truth at frame f: 4 mm and 0.6 mrad per frame on an arc (theta in the reference's sign)."""
    th = -0.0006 * f
    R = 0.004 / 0.0006
    return np.array([R * np.sin(0.0006 * f), R * (1 - np.cos(0.0006 * f)), th])


def sensor_frame(points, pose):
    """H (m - t), H = [[ct,-st],[st,ct]]: inverse of the reference's world = R^T p + t (main.c:115-116)."""
    ct, st = np.cos(pose[2]), np.sin(pose[2])
    d = points - pose[:2]
    return np.stack([ct * d[:, 0] - st * d[:, 1], st * d[:, 0] + ct * d[:, 1]], 1)


def make_frames(nframes, nbeams, landmarks, rng, observed=0):
    """observed = K > 0: only the K landmarks nearest to the sensor are seen in a frame (SURVEY §8d's end-to-end
    variant); 0: every landmark is seen in every frame (the roofline sweep, the default workload)."""
    frames = []
    ang = -np.pi + 2 * np.pi * np.arange(nbeams) / nbeams
    prev = true_pose(0)
    for f in range(1, nframes + 1):
        pose = true_pose(f)
        r = raycast(pose[0], pose[1], ang - pose[2]) + rng.uniform(-0.005, 0.005, nbeams)
        bx, by = (r * np.cos(ang)).astype(np.float32), (r * np.sin(ang)).astype(np.float32)
        z = sensor_frame(landmarks, pose) + rng.normal(0, 0.02, landmarks.shape)
        ids = rng.permutation(len(landmarks)).astype(np.int32)
        if 0 < observed < len(landmarks):
            near = np.argsort(np.hypot(z[:, 0], z[:, 1]))[:observed]
            ids = ids[np.isin(ids, near)]
        frames.append(dict(bx=bx, by=by, dp=(pose - prev).astype(np.float32), ids=ids,
                           zx=z[ids, 0].astype(np.float32), zy=z[ids, 1].astype(np.float32)))
        prev = pose
    return frames


def make_landmarks(L, rng):
    lm = np.stack([rng.uniform(ROOM[0] + 0.5, ROOM[2] - 0.5, L), rng.uniform(ROOM[1] + 0.5, ROOM[3] - 0.5, L)], 1)
    if L:   # landmark ids in discovery order along a sweep of the room: neighbours in space are neighbours in the map rows
        lm = lm[np.argsort(morton(lm), kind="stable")]
    return lm


def preroll_frames(args):
    """Frames the filter runs (untimed) before the W warm-up steps, so that the timed steps see a filter in steady state."""
    return max(0, args.preroll) if args.mode == "pf" else 0


def build_inputs(args):
    """Everything synthetic, as numpy (identical on every rank; made once per process)."""
    rng = np.random.default_rng(4321)
    L = 0 if args.mode == "score" else args.landmarks
    landmarks = make_landmarks(L, rng)
    pixel = np.float32(20.48 / args.grid)
    min_x, min_y = np.float32(-4.24), np.float32(-10.24)
    occ = occupancy(args.grid, float(pixel), float(min_x), float(min_y))
    nframes = preroll_frames(args) + args.steps + args.warmup + 12   # + the short per-stage timing pass after the timed region
    frames = make_frames(nframes, args.beams, landmarks, rng, args.observed)
    return dict(L=L, landmarks=landmarks, pixel=pixel, min_x=min_x, min_y=min_y, occ=occ, frames=frames)


# ------------------------------------------------------------------------------------------------ CPU baselines
def host_cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, occ, meta_t, frames, landmarks, budget_s=12.0):
    """The CPU port (oracle/, TEST INFRASTRUCTURE used here only as the reported baseline): the same
    frame loop, single thread, on a bounded sample of the workload."""
    import oracle

    rows = cols = args.grid
    edt = oracle.edt(occ, rows, cols, 10.0, "window")
    m = oracle.meta(rows, cols, cols, *meta_t)
    n = 1024
    rng = np.random.default_rng(99)
    p0 = true_pose(0)
    x = (p0[0] + rng.normal(0, 0.05, n)).astype(np.float32)
    y = (p0[1] + rng.normal(0, 0.05, n)).astype(np.float32)
    th = (p0[2] + rng.normal(0, 0.01, n)).astype(np.float32)
    L = len(landmarks)
    mp = np.zeros((n, 5, max(L, 1)), np.float32)   # one row per particle, like the engine
    if L:
        mp[:, 0] = landmarks[:, 0] + rng.normal(0, 0.1, (n, L))
        mp[:, 1] = landmarks[:, 1] + rng.normal(0, 0.1, (n, L))
        mp[:, 2] = 0.05
        mp[:, 4] = 0.05
    anc = None
    done = 0
    t0 = time.perf_counter()
    while True:
        fr = frames[done % len(frames)]
        x, y, th = oracle.motion_sample(x, y, th, anc, n, 0, fr["dp"], SIGMA, 1234, done)
        score, _ = oracle.score_poses_det(m, edt, fr["bx"], fr["by"], x, y, th)
        ll = None
        if L and args.mode != "score":
            mp, ll = oracle.ekf_update(mp, x, y, th, anc, fr["ids"], fr["zx"], fr["zy"], MEAS_VAR)
        logw, mx = oracle.logweight(score, ll, SCORE_GAIN)
        wq, _ = oracle.quantise_weights(logw, mx)
        anc = oracle.resample(wq, 1234, done)
        done += 1
        el = time.perf_counter() - t0
        if el > budget_s and done >= 3:
            break
    return {"value": n * done / el, "unit": "particle-updates/s", "cores": 1, "kind": "port",
            "host_cpu": host_cpu_model(), "host_cores_total": os.cpu_count(),
            "sample": f"{n} particles x {done} frames of the same workload ({args.beams} beams, {L} landmarks, "
                      f"{args.grid}^2 EDT), oracle/ C port, 1 thread, {el:.1f} s"}


def cpu_baseline_threads(args, occ, meta_t, frames, landmarks, budget_s=8.0):
    """The same CPU port with the per-particle stages (motion, score, EKF) spread over host threads, particles in
    contiguous chunks — SURVEY 8(d)'s optional "all cores" line, labelled as such.  The GPU box gives one GPU's job 16
    cores' worth of CPU, so 16 threads at most.  ctypes releases the interpreter lock inside the C functions."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    import oracle

    T = max(1, min(16, os.cpu_count() or 1))
    rows = cols = args.grid
    edt = oracle.edt(occ, rows, cols, 10.0, "window")
    m = oracle.meta(rows, cols, cols, *meta_t)
    per = 512
    n = per * T
    rng = np.random.default_rng(99)
    p0 = true_pose(0)
    pose = [(p0[k] + rng.normal(0, s, n)).astype(np.float32) for k, s in ((0, 0.05), (1, 0.05), (2, 0.01))]
    L = len(landmarks)
    use_ekf = bool(L) and args.mode != "score"
    mp = np.zeros((n, 5, max(L, 1)), np.float32)
    if L:
        mp[:, 0] = landmarks[:, 0] + rng.normal(0, 0.1, (n, L))
        mp[:, 1] = landmarks[:, 1] + rng.normal(0, 0.1, (n, L))
        mp[:, 2] = 0.05
        mp[:, 4] = 0.05
    mp2 = np.empty_like(mp)
    new = [np.empty(n, np.float32) for _ in range(3)]
    score, ll = np.empty(n, np.float32), np.zeros(n, np.float32)
    lib = oracle.lib()
    anc = None
    done = 0

    def chunk(c, fr, frame):
        sl = slice(c * per, (c + 1) * per)
        a = None if anc is None else anc[sl]
        src = pose if a is not None else [p[sl] for p in pose]   # ancestors index the whole population
        x, y, th = oracle.motion_sample(src[0], src[1], src[2], a, per, c * per, fr["dp"], SIGMA, 1234, frame)
        for k, v in enumerate((x, y, th)):
            new[k][sl] = v
        score[sl] = oracle.score_poses_det(m, edt, fr["bx"], fr["by"], x, y, th)[0]
        if use_ekf:
            ids = np.ascontiguousarray(fr["ids"], np.int32)
            lib.orc_ekf_update(mp if a is not None else mp[sl], mp2[sl], 5 * L, L, L, x, y, th,
                               a.ctypes.data_as(C.c_void_p) if a is not None else None, per, ids,
                               np.ascontiguousarray(fr["zx"], np.float32), np.ascontiguousarray(fr["zy"], np.float32), len(ids),
                               MEAS_VAR, ll[sl])

    with ThreadPoolExecutor(T) as pool:
        t0 = time.perf_counter()
        while True:
            fr = frames[done % len(frames)]
            list(pool.map(lambda c: chunk(c, fr, done), range(T)))
            pose = [v.copy() for v in new]
            if use_ekf:
                mp, mp2 = mp2, mp
            elif L and anc is not None:
                mp = mp[anc]
            logw, mx = oracle.logweight(score, ll if use_ekf else None, SCORE_GAIN)
            wq, _ = oracle.quantise_weights(logw, mx)
            anc = np.ascontiguousarray(oracle.resample(wq, 1234, done), np.int32)
            done += 1
            el = time.perf_counter() - t0
            if el > budget_s and done >= 3:
                break
    return {"value": n * done / el, "unit": "particle-updates/s", "cores": T, "kind": "port",
            "sample": f"{n} particles x {done} frames of the same workload, oracle/ C port, per-particle stages on {T} host "
                      f"threads (weights and resample on one), {el:.1f} s"}


def cpu_baseline_main_c():
    """SURVEY 8(d)'s CPU baseline of the reference's OWN pipeline (main.c rows A1-A8: parse, clean-up, transform, local
    map, raster, EDT, 27-pose matcher x 10, key-frame test, map append) on this host, one thread, on the 1000-frame
    synthetic parity set (regenerated here by oracle/gen_dataset, libm-free, 1079 beams) — beside the engine's two
    drop-in programs on the same scans in the same run.  `main_cpu` is the oracle's restatement (kind "port") with the
    reference's naive EDT (main.c:223-269) and with the scatter EDT (main_accelerated.c:215-283); `reference` is the
    compiled, unmodified main.c when oracle/_ref/main_ref travelled to this box.  One scan = 270 pose evaluations
    (2 matcher calls x 5 passes x 27 candidates)."""
    import oracle

    out = {"dataset": "synthetic parity set, 1000 frames x 1079 beams (oracle/gen_dataset)", "cores": 1,
           "host_cpu": host_cpu_model(), "pose_evals_per_scan": 270}
    pat = re.compile(r"frames (\d+)\s+wall ([\d.]+) s(?:\s+edt ([\d.]+) s / (\d+) calls\s+match ([\d.]+) s / (\d+) calls)?")
    with tempfile.TemporaryDirectory() as td:
        csv, mp = Path(td) / "parity.csv", Path(td) / "map.csv"
        info = json.loads((ROOT / "tests" / "golden" / "datasets.json").read_text())["parity"]
        oracle.run_tool("gen_dataset", csv, *info["gen_args"])

        def parse(stderr):
            m = pat.search(stderr)
            if not m:
                return None
            frames, wall = int(m.group(1)), float(m.group(2))
            r = {"frames": frames, "wall_s": wall, "scans_per_s": (frames - 1) / wall, "pose_evals_per_s": 270 * (frames - 1) / wall}
            if m.group(3):
                r["edt_ms_per_call"] = 1e3 * float(m.group(3)) / max(int(m.group(4)), 1)
                r["match_ms_per_call"] = 1e3 * float(m.group(5)) / max(int(m.group(6)), 1)
            return r

        for variant, name in ((0, "naive"), (1, "scatter")):
            r = oracle.run_tool("main_cpu", csv, 1000, 1079, variant, mp, capture_output=True, text=True)
            out[f"main_cpu_{name}_edt"] = parse(r.stderr)
        naive, scat = out["main_cpu_naive_edt"], out["main_cpu_scatter_edt"]
        out.update({"kind": "port", "scans_per_s": naive["scans_per_s"], "pose_evals_per_s": naive["pose_evals_per_s"],
                    "edt_ms_per_call_naive": naive["edt_ms_per_call"], "edt_ms_per_call_scatter": scat["edt_ms_per_call"]})
        ref = ROOT / "oracle" / "_ref" / "main_ref"
        if ref.exists():   # the reference itself, compiled in the build container from the sources where they lie
            t0 = time.perf_counter()
            r = subprocess.run([str(ref)], env=dict(os.environ, ORACLE_DATASET=str(csv), ORACLE_MAP_OUT=str(mp)),
                               capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if r.returncode == 0:
                out["reference"] = {"kind": "reference", "program": "Subsystem_1/main.c (gcc -O2, unmodified)", "frames": 1000,
                                    "wall_s": wall, "scans_per_s": 999 / wall, "pose_evals_per_s": 270 * 999 / wall}
        exe = ROOT / "hardware-acceleration-of-lidar-slam_amd" / "lib" / "slam_main"
        for flag, name in ((None, "slam_main"), ("--mapper", "slam_main_mapper")):
            cmd = [str(exe)] + ([flag] if flag else []) + [str(csv), "1000", "1079", str(mp)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            out[name] = parse(r.stderr) if r.returncode == 0 else {"error": r.stderr.strip()[-200:]}
        if out.get("slam_main") and "scans_per_s" in out["slam_main"]:
            out["drop_in_speedup_vs_main_cpu"] = out["slam_main"]["scans_per_s"] / naive["scans_per_s"]
    return out


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block (for native libraries that print to stdout)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


# ------------------------------------------------------------------------------------------------ ranks
class SingleCtx:
    """One rank in this process; torch.distributed only when torchrun started several of us."""
    threads = False

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dev_index = local_rank if args.device_index is None else args.device_index
        self.transport = "rccl"
        self.dist = None

    def init(self, torch, dev):
        if self.world > 1 and self.dist is None:   # (a second leg of the same process keeps the process group)
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            with stdout_to_stderr():
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=dev)
                dist.barrier()   # RCCL initialises lazily: do it (and print its banner) here, not in the timed region
            self.dist = dist
        self.torch, self.dev = torch, dev

    def make_comm(self, pkg, eng, force):
        if self.world == 1 and not force:
            return None
        # rendezvous token of the engine's own RCCL communicator: made by rank 0, handed out through the process group
        # torchrun set up (used for nothing else but this, the barriers and the final timing reduction)
        torch = self.torch
        uid = torch.zeros(pkg.COMM_ID_BYTES, dtype=torch.uint8, device=self.dev)
        if self.rank == 0:
            uid = torch.tensor(list(pkg.comm_unique_id()), dtype=torch.uint8, device=self.dev)
        if self.world > 1:
            self.dist.broadcast(uid, src=0)
        with stdout_to_stderr():   # RCCL prints its version banner to stdout; stdout carries ONE JSON line
            return pkg.Comm.rccl(eng, self.rank, self.world, bytes(uid.cpu().tolist()))

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, v):
        if self.world == 1:
            return float(v)
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def finish(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


class ThreadShared:
    def __init__(self, pkg, world):
        self.group = pkg.LocalGroup(world)
        self.bar = threading.Barrier(world)
        self.vals = [0.0] * world
        self.lock = threading.Lock()
        self.failed = False


class ThreadCtx:
    """One rank = one thread of this process; every rank on the same card, exchanges through slam_comm_create_local."""
    threads = True
    transport = "local (in-process, one card shared by all ranks)"

    def __init__(self, args, rank, shared):
        self.world, self.rank, self.shared = args.gpus, rank, shared
        self.dev_index = 0 if args.device_index is None else args.device_index

    def init(self, torch, dev):
        self.torch, self.dev = torch, dev

    def make_comm(self, pkg, eng, force):
        return pkg.Comm.local(eng, self.shared.group, self.rank)

    def barrier(self):
        self.torch.cuda.synchronize()
        self.shared.bar.wait(timeout=300)
        self.torch.cuda.synchronize()

    def max_over_ranks(self, v):
        self.shared.vals[self.rank] = float(v)
        self.shared.bar.wait(timeout=300)
        m = max(self.shared.vals)
        self.shared.bar.wait(timeout=300)
        return m

    def finish(self):
        pass


# ------------------------------------------------------------------------------------------------ one rank
SETTLE_S = 0.25


def settle(torch, seconds=None):
    """Between set-up (engine and session creation, allocations) and the first frame: drain the device and let the host
    sleep.  The driver answers an allocation of pinned host memory (hipHostMalloc: the engine makes a few when it is
    created) some 10-50 ms later by holding the process's queues for 65-80 ms — stage timers normal, host issue times
    normal, the stream simply stands still (profiles/r03_stall_trigger.txt, r03_stall_frames.txt);
    without the pause that can land in the frames that follow."""
    torch.cuda.synchronize()
    time.sleep(SETTLE_S if seconds is None else seconds)


def fill_maps(torch, m0, landmarks, L, dev, n):
    """Plausible maps [n][5][Lp]: every landmark near its true place with a loose covariance."""
    lm = torch.from_numpy(landmarks.astype(np.float32)).to(dev)
    for i0 in range(0, n, 65536):                     # in chunks: the temporaries of a 1M x 1k map are 4 GB each
        i1 = min(i0 + 65536, n)
        m0[i0:i1, 0, :L] = lm[:, 0] + 0.1 * torch.randn((i1 - i0, L), device=dev)
        m0[i0:i1, 1, :L] = lm[:, 1] + 0.1 * torch.randn((i1 - i0, L), device=dev)
    m0[:n, 2, :L] = 0.05
    m0[:n, 3, :L] = 0.0
    m0[:n, 4, :L] = 0.05


def obs_tables(torch, frames, L, dev):
    """Observation tables indexed by landmark, one per frame (NaN = not observed)."""
    tab = np.full((len(frames), 2, L), np.nan, np.float32)
    for k, f in enumerate(frames):
        tab[k, 0, f["ids"]] = f["zx"]
        tab[k, 1, f["ids"]] = f["zy"]
    return torch.from_numpy(tab).to(dev)


def run_rank(args, ctx, inp, last=True):
    """last = False: another leg follows in this process (the rendezvous of the ranks stays up)."""
    import torch

    from __graft_entry__ import load_package

    pkg = load_package()
    world, rank = ctx.world, ctx.rank
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the engine has no CPU fallback")
    dev_index = ctx.dev_index
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ctx.init(torch, dev)

    eng = pkg.Engine(dev_index)
    eng.ekf_form_set(args.ekf_form)
    L, landmarks, frames = inp["L"], inp["landmarks"], inp["frames"]
    pixel, min_x, min_y, occ = inp["pixel"], inp["min_x"], inp["min_y"], inp["occ"]

    d_occ = torch.from_numpy(occ).to(dev)
    d_edt = torch.empty((args.grid, args.grid), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    eng.edt_dev(d_occ, args.grid, args.grid, args.grid, 10.0, d_edt)
    meta = pkg.grid_meta(args.grid, args.grid, args.grid, pixel, min_x, min_y)
    eng.grid_set_dev(0, d_edt, meta)

    n = args.particles
    comm = ctx.make_comm(pkg, eng, args.force_collectives)
    pf = pkg.PfSession(eng, n, L, sigma=SIGMA, meas_var=MEAS_VAR, score_gain=SCORE_GAIN, seed=1234, comm=comm,
                       resample_ess_frac=args.ess, map_layout=args.map_layout)
    Lp = (L + 31) // 32 * 32

    def views():
        """(pose [3][n], current map [rows][5][Lp] or None, spare map, pending gather index or None) as torch tensors."""
        v = pf.device_view()
        t = {k: (torch.as_tensor(v[k], device=dev) if v[k] is not None else None) for k in ("pose", "map", "map_spare", "anc")}
        return t["pose"], t["map"], t["map_spare"], t["anc"]

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    p0 = true_pose(0)
    init = [(p0[k] + s * torch.randn(n, generator=g)).numpy() for k, s in ((0, 0.05), (1, 0.05), (2, 0.01))]
    if args.presort_poses:
        cell = 4.0 * float(pixel)
        key = np.lexsort((np.floor(init[0] / cell), np.floor(init[1] / cell), np.floor(init[2] / (cell / 8.0))))
        init = [a[key] for a in init]
    pf.set_poses(*init)
    if L:
        rows_now = views()[1]
        # a session on pages has no rows to write into: the same rows are made in a scratch tensor and handed over once
        m0 = rows_now if rows_now is not None else torch.empty((n, 5, Lp), dtype=torch.float32, device=dev)
        fill_maps(torch, m0, landmarks, L, dev, n)
        if rows_now is None:
            torch.cuda.synchronize()
            pf.set_map_dev(m0, 5 * Lp, Lp)
            eng.sync()
        del m0, rows_now
    score_t = torch.zeros(n, dtype=torch.float32, device=dev)
    count_t = torch.zeros(n, dtype=torch.int32, device=dev)
    loglik_t = torch.zeros(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    # sensor data of every frame resident in HBM before the timed region (bench contract); --host-sensor
    # uploads it frame by frame through the host-buffer entry points instead (8.9 KB per frame over PCIe)
    d_scan = torch.from_numpy(np.stack([np.stack([f["bx"], f["by"]]) for f in frames])).to(dev)        # [F][2][B]
    d_z = obs_tables(torch, frames, L, dev) if L else None                                              # [F][2][L]
    # per-frame views of the resident sensor data, made once (a tensor slice costs microseconds of host time)
    scan_v = [(d_scan[k, 0], d_scan[k, 1]) for k in range(len(frames))]
    obs_v = [(d_z[k, 0], d_z[k, 1]) for k in range(len(frames))] if L else None

    sweep = {}

    off = preroll_frames(args)   # frames[0 .. off) are the pre-roll: step k of the bench is frame off + k

    def one_step(k):
        k += off
        fr = frames[k]
        if args.host_sensor:
            eng.scan_upload(fr["bx"], fr["by"])
            obs, obs_dev = ((fr["ids"], fr["zx"], fr["zy"]) if L else None), None
        else:
            eng.scan_set_dev(scan_v[k][0], scan_v[k][1], args.beams)
            obs, obs_dev = None, (obs_v[k] if L else None)
        if args.mode == "pf":
            if obs_dev:
                eng.obs_set_dev(*obs_dev, L)
            elif obs:
                eng.obs_upload(*obs, L)
            pf.step(0, fr["dp"], L > 0)
            return
        if not sweep:
            p, a, b, _ = views()
            sweep.update(pose=p, maps=(a, b))
        p = sweep["pose"]
        if args.mode == "score":
            eng.score_poses_dev(0, p[0], p[1], p[2], n, score_t, count_t)
        else:   # ekf sweep: out of place, ping-pong between the two map buffers
            if obs_dev:
                eng.obs_set_dev(*obs_dev, L)
            else:
                eng.obs_upload(*obs, L)
            eng.ekf_update_dev(sweep["maps"][k & 1], sweep["maps"][1 - (k & 1)], 5 * Lp, Lp, L, p[0], p[1], p[2], None, n,
                               MEAS_VAR, loglik_t)

    settle(torch)
    # Pre-roll: the filter starts from poses spread 5 cm / 0.01 rad around the truth and needs on the order of a
    # hundred frames to settle to the spread its motion noise and its observations sustain; until then neighbouring particles
    # lie farther apart, the scorer's gathers of a wavefront fall into more cache lines and a frame takes up to 10 % longer
    # (front kernel 156-161 us over the first 30 frames, 147 at 40-50, 143-145 from 90 on: profiles/r03_early_frames.md;
    # with the sensor-frame update of rounds 1-2 it was 193 us at frame 3 and 154 at 140).  The metric is a
    # steady-state rate, so these frames run before the W warm-up steps, untimed; --preroll 0 starts cold.
    for k in range(-off, 0):
        one_step(k)
    for k in range(args.warmup):
        one_step(k)
    ctx.barrier()
    dominant = eng.PROF_SCORE if args.mode == "score" or L == 0 else eng.PROF_EKF
    timed = {"dominant": (dominant,), "all": (eng.PROF_SCORE, eng.PROF_EKF), "none": ()}[args.events]
    eng.profile_enable(*timed)
    for kk in range(eng.PROF_COUNT):
        eng.profile_read(kk)
    migrated = 0
    forms0 = eng.ekf_form_counts() + eng.ekf_inplace_form_counts()
    fused0 = eng.frame_fusion_count()
    # A HIP-event bracket around one kernel costs the STREAM ~12 us (rocprofv3 timeline of this command: ~6 us of idle
    # stream before and after the bracketed kernel, none between the other launches), so only every `--event-every`-th frame
    # of the timed region carries it: the kernel's duration is still measured live, inside the timed region, on the kernel's
    # own stream, while the measurement itself costs the frame rate under 1.5 % instead of ~10 %.  `roofline.launches` says how
    # many launches were bracketed; --event-every 1 brackets every frame.
    # (default: every 8th frame of runs of 80 steps or more — 25 samples in the default 200 —, every 4th of shorter ones — 5 samples
    # in the driver's 20 steps.  profiles/short_run.sh, ms per frame on one box: 200 steps 0.1077, without brackets 0.1070; 20 steps
    # with a bracket in every 2nd frame 0.1176, in 2 of the 20 0.1160, without 0.1144: a region that short pays about 7 % for its
    # two ends — an empty queue behind the synchronisation in front, the synchronisation behind — whatever is bracketed)
    every = max(1, args.event_every if args.event_every > 0 else (8 if args.steps >= 80 else (4 if args.steps >= 12 else 2)))
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        if every > 1:
            eng.profile_enable(*(timed if (k - args.warmup) % every == 0 else ()))
        one_step(k)
        migrated += pf.rows_received() if args.mode == "pf" else 0
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    eng.profile_enable()
    forms1 = eng.ekf_form_counts() + eng.ekf_inplace_form_counts()
    forms = tuple(b - a for a, b in zip(forms0, forms1))   # EKF launches of the timed region, by kernel (out of place x2, in place x2)
    fused_n = eng.frame_fusion_count() - fused0             # ... of which went out fused with the motion + score launch
    layout_in_region = pf.layout()   # "rows", "pages", "split" or "split_pages": what the timed frames ran on
    paged_in_region = layout_in_region in ("pages", "split_pages")
    split_in_region = layout_in_region in ("split", "split_pages")
    elapsed = ctx.max_over_ranks(elapsed)
    migrated = ctx.max_over_ranks(migrated / max(args.steps, 1))

    score_ms, score_n = eng.profile_read(eng.PROF_SCORE)
    ekf_ms, ekf_n = eng.profile_read(eng.PROF_EKF)
    # A start/stop event bracket around ONE kernel also contains the stream's marker handling; an empty bracket
    # measures it (~5-7 us).  Both are reported; the kernel duration used for the roofline is bracket - empty
    # bracket, which is what rocprofv3's kernel trace of the same process shows (profiles/README.md).
    bracket_overhead_ms = eng.profile_bracket_overhead()

    def kernel_ms(total_ms, launches):
        return max(total_ms / max(launches, 1) - bracket_overhead_ms, 0.0)

    # every stage of a frame through the per-stage timers: a short extra pass, outside the timing (the trajectory simply
    # continues; the brackets cost stream time, which is why the timed region carries only the dominant kernel's)
    eng.profile_enable(*range(eng.PROF_COUNT))
    eng.frame_fusion_set(False)   # the stages one by one: score and landmark update as two launches (same bits)
    extra = min(10, args.steps)
    for k in range(extra):
        one_step(args.warmup + args.steps + k)
    eng.profile_enable()
    eng.frame_fusion_set(True)
    stage_avg_ms = {}
    for kk in range(eng.PROF_COUNT):
        ms, cnt = eng.profile_read(kk)
        # several ranks: the slowest rank's average and the busiest rank's launch count (a stage like `unpack` runs only on
        # ranks that received rows in these few frames; every rank takes part in the reduction, in the same order)
        avg = ctx.max_over_ranks(kernel_ms(ms, cnt) if cnt else 0.0) if world > 1 else kernel_ms(ms, cnt)
        per_frame = ctx.max_over_ranks(cnt / max(extra, 1)) if world > 1 else cnt / max(extra, 1)
        if per_frame > 0:
            stage_avg_ms[eng.PROF_NAMES[kk]] = {"avg_ms": avg, "per_frame": per_frame}
        if kk == eng.PROF_SCORE and not score_n:
            score_ms, score_n = ms, cnt
        if kk == eng.PROF_EKF and not ekf_n:
            ekf_ms, ekf_n = ms, cnt
    n_total = n * world
    value = n_total * args.steps / elapsed

    # how much of the population the resample kept distinct (the rows of repeated ancestors come out of L2 or stay in
    # registers, which is what makes the in-filter EKF faster than a sweep): one torch.unique, outside the timed region
    distinct_frac = None
    torch.cuda.synchronize()
    if args.mode == "pf" and views()[3] is not None:
        distinct_frac = torch.unique(views()[3]).numel() / n

    # the same EKF kernel WITHOUT ancestor sharing: identity ancestors on the same buffers, after the timed region.
    # Every row is read from HBM once and written once: the streaming figure of the kernel.
    L_obs = args.observed if 0 < args.observed < L else L
    score_bytes = (12 + 4 * args.beams + 4) * n              # pose read + one EDT gather per beam + score write
    ekf_bytes = 40 * n * L_obs                               # 20 B read + 20 B written per (particle, OBSERVED landmark)
    no_reuse = None
    # classes in use (split layout): what the covariance part of a frame costs is proportional to it
    classes_in_use = None
    if layout_in_region == "split":
        classes_in_use = int(torch.as_tensor(pf.split_view()["live_count"], device=dev)[0])
    sweep_rows = views()[1:3]
    if L and args.mode != "score" and not args.no_sweep and sweep_rows[0] is None and not paged_in_region:
        # a split session has no rows: the sweep of the ROW kernel (the 20 B + 20 B streaming figure the north star's
        # read-only fraction is defined on) runs on scratch rows of the same shape, when they fit beside the session
        free_b, _ = torch.cuda.mem_get_info(dev)
        if free_b > 2.1 * n * 5 * Lp * 4:
            ma = torch.zeros((n, 5, Lp), dtype=torch.float32, device=dev)
            mb = torch.empty((n, 5, Lp), dtype=torch.float32, device=dev)
            fill_maps(torch, ma, landmarks, L, dev, n)
            torch.cuda.synchronize()
            sweep_rows = (ma, mb)
            del ma, mb
    if L and args.mode != "score" and not args.no_sweep and sweep_rows[0] is not None:
        eng.profile_enable(eng.PROF_EKF)
        eng.profile_read(eng.PROF_EKF)
        p = views()[0]
        ma, mb = sweep_rows
        base = args.warmup + args.steps
        for k in range(12):
            eng.obs_set_dev(*obs_v[(off + base + k) % len(frames)], L)
            eng.ekf_update_dev((ma, mb)[k & 1], (mb, ma)[k & 1], 5 * Lp, Lp, L, p[0], p[1], p[2], None, n, MEAS_VAR, loglik_t)
        eng.profile_enable()
        nr_ms, nr_n = eng.profile_read(eng.PROF_EKF)
        t_nr = kernel_ms(nr_ms, nr_n)
        if t_nr > 0:
            # an out-of-place launch rewrites EVERY row whatever was observed: its HBM bytes are 40 B x n x Lp
            moved = ekf_bytes if L_obs == L else 40 * n * Lp
            no_reuse = {"kernel": "ekf_update_kernel (identity ancestors: no row is shared), same buffers and observations",
                        "achieved": moved / (t_nr * 1e-3) / 1e9, "frac": moved / (t_nr * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "avg_launch_ms": t_nr, "launches": int(nr_n), "bytes_per_launch": moved,
                        "algorithmic_bytes_per_launch": ekf_bytes,
                        "read_only_frac": 20 * n * L_obs / (t_nr * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del p, ma, mb
    del sweep_rows

    # the engine picks among kernels that give the same bits; name the one that ran
    if paged_in_region:
        ekf_name = "ekf_paged_kernel"
    elif forms[2] + forms[3] > forms[0] + forms[1]:
        ekf_name = "ekf_sparse_kernel" if forms[3] > forms[2] else "ekf_update_kernel (in place)"
    elif fused_n > forms[1] // 2:
        ekf_name = "frame_front_kernel"   # motion + score and the grouped landmark update in one launch
    else:
        ekf_name = "ekf_split_kernel" if split_in_region else ("ekf_update_group_kernel" if forms[1] > forms[0] else "ekf_update_kernel")
    if ekf_n and (ekf_ms / ekf_n >= score_ms / max(score_n, 1) or args.mode == "ekf"):
        kern, raw_ms, dur_ms, alg = ekf_name, ekf_ms / ekf_n, kernel_ms(ekf_ms, ekf_n), ekf_bytes
    else:
        kern, raw_ms, dur_ms, alg = ("score_poses_kernel", score_ms / max(score_n, 1), kernel_ms(score_ms, score_n), score_bytes)
    logical = alg / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
    # HBM bytes per launch from rocprofv3 --pmc runs of this same command (profiles/collect_pmc.sh): used only when the
    # record is for this workload AND this kernel
    traffic, traffic_src, valu = None, None, None
    tfile = ROOT / "profiles" / "traffic.json"
    if tfile.exists():
        key = (f"{args.mode}:{n}:{args.beams}:{L}:{args.grid}" + (":paged" if paged_in_region else "") + (":split" if split_in_region else "")
               + (f":obs{L_obs}" if L_obs != L else "") + (f":ess{args.ess}" if 0 < args.ess < 1 else ""))
        rec = json.loads(tfile.read_text()).get(key, {})
        rk = rec.get("kernel", kern).split("<")[0]   # (ekf_paged_kernel names the paged update in either of its forms)
        if rk == kern.split(" ")[0] or (rk.startswith("ekf_paged") and kern.startswith("ekf_paged")):
            traffic, traffic_src = rec.get("bytes_per_launch"), rec.get("source")
            if rec.get("valu_instructions_per_launch"):   # the same record's vector-ALU counters (static, builder-run)
                valu = {"instructions_per_launch": rec["valu_instructions_per_launch"], "busy_pct": rec.get("valu_busy_pct"),
                        "issue_time_ms_on_1024_simds_at_2p4_ghz": rec["valu_instructions_per_launch"] * 4 / 1024 / 2.4e9 * 1e3,
                        "source": rec.get("source"), "kernel_template": rec.get("kernel_template"), "head": rec.get("head")}
    # What the launch must move, from THIS run's own figures (checkable without a profiler): an out-of-place update writes
    # every row it owns in full and reads every distinct ancestor's row once (the offspring share it through registers / L2);
    # rows hold 20 B per landmark, split rows 8 B plus 24 B per landmark and covariance class in use (read by the particles'
    # update out of L2, read and rewritten once by cov_update_kernel).  A static PMC record that is far off this model was
    # taken on another filter state (or another kernel) and is flagged.
    traffic_model = None
    gated = 0 < args.ess < 1
    # (a gated session on ROWS updates the frames that keep their population in place: no model for that mix; on the split layout
    # such a frame goes through the identity index out of place — every row written, every row read)
    if (args.mode == "pf" and L and not paged_in_region and distinct_frac is not None and (not gated or split_in_region)
            and kern.startswith(("frame_front", "ekf_update", "ekf_split"))):
        per = 8 if split_in_region else 20
        written = per * n * Lp
        f_res = pf.frames_resampled() / max(args.warmup + args.steps + off + extra, 1) if gated else 1.0   # share of frames that resampled
        f_res = min(max(f_res, 0.0), 1.0)
        read = (f_res * distinct_frac + (1.0 - f_res)) * per * n * Lp + (24 * Lp * (classes_in_use or 1) if split_in_region else 0)
        traffic_model = {"bytes_written": written, "bytes_read": read, "bytes": written + read,
                         "basis": f"{per} B x n x Lp written + distinct_ancestor_frac x {per} B x n x Lp read"
                                  + (" (frames that kept their population: every row read)" if gated else "")
                                  + (" + 24 B x Lp x classes in use" if split_in_region else ""),
                         "distinct_ancestor_frac": distinct_frac, "classes_in_use": classes_in_use,
                         "frames_resampled_frac": f_res if gated else None}
        if traffic:
            traffic_model["record_over_model"] = traffic / (written + read)
            traffic_model["record_suspect"] = bool(abs(traffic / (written + read) - 1.0) > 0.15)
    if traffic and dur_ms > 0:
        achieved, a_kernel = traffic / (dur_ms * 1e-3) / 1e9, kern
        basis = f"hbm_traffic (PMC record of this workload and kernel, {traffic_src}) / this run's launch time"
    elif traffic_model and dur_ms > 0:
        achieved, a_kernel = traffic_model["bytes"] / (dur_ms * 1e-3) / 1e9, kern
        basis = ("traffic model of this run (no PMC record for this workload and kernel): " + traffic_model["basis"]
                 + " / this run's launch time")
    elif kern.startswith(("ekf_update", "ekf_split", "frame_front")) and no_reuse and args.mode == "pf":
        achieved, a_kernel = no_reuse["achieved"], no_reuse["kernel"]
        basis = "no_reuse sweep (no PMC record for this workload and kernel): HBM bytes of a launch without shared rows / its launch time"
    else:
        achieved, a_kernel, basis = logical, kern, "algorithmic bytes / launch time in the timed region"
    if no_reuse:
        ro_frac, ro_basis = no_reuse["read_only_frac"], "no-reuse sweep"
    elif (kern.startswith("ekf") or kern.startswith("frame_front")) and dur_ms > 0:
        ro_frac, ro_basis = 20 * n * L_obs / (dur_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "in-filter launches of the timed region"
    else:
        ro_frac, ro_basis = None, None
    out = {
        "metric": "particle-updates/sec (N_particles x scans/s) on 360-beam lidar",
        "value": value, "unit": "particle-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": {"pf": ("BASELINE configs[1]" if (n, L, args.beams, args.grid) == (65536, 500, 360, 1024)
                                       else "particle filter") + ": full frame (motion, scan-match score, EKF, weights, resample)",
                                "score": "scan-match score only", "ekf": "EKF sweep only"}[args.mode],
                   "mode": args.mode, "particles_per_gpu": n, "particles_total": n_total, "beams": args.beams,
                   "landmarks": L, "landmarks_observed_per_frame": L_obs,
                   "edt_grid": f"{args.grid}x{args.grid}",
                   "parallelism": f"particle-shard x{world}" + (" (multi-GPU code path forced)" if args.force_collectives else ""),
                   "transport": ctx.transport if (world > 1 or args.force_collectives) else None,
                   "scaling": args.scaling,
                   "rows_received_per_frame_max_rank": migrated if world > 1 else 0,
                   "distinct_ancestor_frac": distinct_frac,
                   "resample_ess_frac": args.ess,
                   "preroll_frames": off,
                   "map_layout": {"requested": args.map_layout,
                                  "in_timed_region": {"pages": "pages (copy-on-write, 32 landmarks)", "rows": "rows",
                                                      "split": "split (means per particle, covariances per covariance class)",
                                                      "split_pages": "split pages (means on copy-on-write pages of 32 landmarks, "
                                                                     "covariances per covariance class)"}[layout_in_region],
                                  "covariance_classes_in_use": classes_in_use},
                   "frames_resampled": (pf.frames_resampled() if 0 < args.ess < 1 else None)},
        "roofline": {"bound": "hbm", "kernel": kern, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_model": traffic_model, "valu": valu,
                     "achieved_basis": basis, "achieved_kernel": a_kernel,
                     "algorithmic_bytes_per_launch": alg, "logical_rate_gbs": logical,
                     "logical_frac": logical / HBM_PEAK_GBS,
                     "no_reuse": no_reuse,
                     # the north star's own definition: 20 B x n x L_observed / t / peak
                     "read_only_frac": ro_frac, "read_only_basis": ro_basis,
                     "note": ("in a running filter the rows of repeated resample ancestors are re-read from L2 or kept in "
                              "registers: `logical_rate_gbs` (SURVEY 8d's 40 B per particle and observed landmark / launch time) is "
                              "then not an HBM rate; `achieved` is one, see `achieved_basis` / `achieved_kernel`; `no_reuse` is the "
                              "row kernel streaming every row from HBM" if kern.startswith("ekf_update") else
                              "ONE launch holds the frame's motion sample + scan-match score (scoring workgroups: gathers out of L2) and "
                              "its landmark update (grouped row kernel: HBM writes), interleaved; `avg_launch_ms` is that launch, "
                              "`algorithmic_bytes_per_launch` the update's 40 B per particle and observed landmark; `stage_avg_ms` has "
                              "the two as separate launches (slam_frame_fusion_set(0)); `no_reuse` is the row kernel alone streaming "
                              "every row from HBM"
                              + ("; SPLIT layout: the update reads and writes the MEANS of a row only (8 B per particle and landmark "
                                 "each way), the covariance planes exist once per covariance class and are rewritten by "
                                 "cov_update_kernel (stage `pages`), see `traffic_model`; with 2.5 x fewer bytes to move the launch is "
                                 "bound by its vector instructions and the drain of its stores, not by HBM (`valu`, "
                                 "profiles/r04_split_tuning.md): `frac` is what the remaining bytes amount to" if split_in_region else "")
                              if kern.startswith("frame_front") else
                              "paged maps: the update reads the touched pages of the ancestors (shared pages out of L2) and writes "
                              "fresh pages; `logical_rate_gbs` is SURVEY 8d's 40 B per particle and observed landmark / launch time"
                              if kern.startswith("ekf") else
                              "EDT gathers are served by L2 / Infinity Cache: logical-byte rate, not HBM traffic"),
                     "avg_launch_ms": dur_ms, "avg_event_bracket_ms": raw_ms,
                     "event_bracket_overhead_ms": bracket_overhead_ms,
                     "launches": int(score_n if kern.startswith("score") else ekf_n),
                     "event_sampling": f"HIP-event bracket in every {every}{'st' if every == 1 else 'th'} frame of the timed region "
                                       f"({args.steps} frames); a bracket idles the stream ~12 us",
                     "ekf_launches_by_kernel": {"ekf_update_kernel": forms[0], "ekf_update_group_kernel": forms[1] - fused_n,
                                                "frame_front_kernel (score + grouped update)": fused_n,
                                                "ekf_update_kernel(in place)": forms[2], "ekf_sparse_kernel": forms[3]},
                     "other_kernel_avg_ms": {"score_poses_kernel": kernel_ms(score_ms, score_n),
                                             ekf_name: kernel_ms(ekf_ms, ekf_n)}},
        "stage_avg_ms": stage_avg_ms,
    }
    torch.cuda.synchronize()
    pf.close()
    if comm:
        comm.close()
    del d_scan, d_z, scan_v, obs_v, sweep
    torch.cuda.empty_cache()

    legs_ok = rank == 0 and world == 1 and not ctx.threads
    if legs_ok and args.mode == "pf" and not args.no_extra_legs and not args.force_collectives:
        try:
            out.update(extra_legs(args, torch, pkg, eng, dev, inp, kernel_ms))
        except Exception as ex:   # a leg that fails must not cost the headline line
            out["extra_legs_error"] = f"{type(ex).__name__}: {ex}"
    if legs_ok and not args.no_cpu_baseline:
        meta_t = (float(pixel), float(min_x), float(min_y))
        out["cpu_baseline"] = cpu_baseline(args, occ, meta_t, frames, landmarks)
        out["cpu_baseline_threads"] = cpu_baseline_threads(args, occ, meta_t, frames, landmarks)
        try:
            out["cpu_baseline_main_c"] = cpu_baseline_main_c()
        except Exception as ex:
            out["cpu_baseline_main_c"] = {"error": f"{type(ex).__name__}: {ex}"}
    elif rank == 0:
        out["cpu_baseline"] = None
    torch.cuda.synchronize()
    if last:
        ctx.finish()
    eng.close()
    if legs_ok and args.mode == "pf" and not args.no_extra_legs and not args.force_collectives:
        out["sharded_rehearsal_2_ranks_one_card"] = sharded_rehearsal(args)
    return out


def sharded_rehearsal(args):
    """The sharded code path on THIS box, in the driver's record: `bench.py --gpus 2 --transport local` as a child process —
    two ranks of the C session as threads sharing this card through the in-process transport, every exchange step of a frame
    (all-reduce, all-gathers, exchange plan, rows packed, sent and unpacked) and the per-stage timers.  Not a scaling figure:
    the ranks share one GPU and the transport is device copies behind a host rendezvous; RCCL between ranks needs more than
    one GPU."""
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--transport", "local", "--steps", "40", "--warmup", "10",
           "--particles", "32768", "--no-cpu-baseline", "--no-extra-legs", "--no-sweep", "--settle-ms", str(args.settle_ms)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-300:]}
        d = json.loads(line[-1])
        return {"command": "bench.py " + " ".join(cmd[2:]), "n_gpus": d["n_gpus"], "transport": d["config"]["transport"],
                "particles_total": d["config"]["particles_total"], "ms_per_step": d["ms_per_step"],
                "rows_received_per_frame_max_rank": d["config"]["rows_received_per_frame_max_rank"],
                "stage_avg_ms": d["stage_avg_ms"], "note": "rehearsal of the sharded path on one card, not a scaling figure"}
    except Exception as ex:
        return {"error": f"{type(ex).__name__}: {ex}"}


def extra_legs(args, torch, pkg, eng, dev, inp, kernel_ms):
    """Bounded legs after the headline's timed region (N = 1): the north-star sweep, the copy ceiling of the same shape on
    the same box, and the configs[1] frame with the 32 nearest landmarks observed on rows and on pages."""
    res = {}
    # ---- (a) + (b): 1 048 576 x 1 000, every landmark observed, identity ancestors: 20 GB read + 20 GB written per launch
    n, L, Lp = 1048576, 1000, 1024
    free_b, _ = torch.cuda.mem_get_info(dev)
    if free_b > 2.2 * n * 5 * Lp * 4:
        rng = np.random.default_rng(7)
        lm = make_landmarks(L, rng)
        a = torch.empty((n, 5, Lp), dtype=torch.float32, device=dev)
        b = torch.empty((n, 5, Lp), dtype=torch.float32, device=dev)
        a[:, :, L:] = 0.0
        fill_maps(torch, a, lm, L, dev, n)
        p0 = true_pose(0)
        pose = torch.stack([p0[k] + s * torch.randn(n, device=dev) for k, s in ((0, 0.05), (1, 0.05), (2, 0.01))]).float().contiguous()
        fr = make_frames(12, args.beams, lm, rng, 0)
        tabs = obs_tables(torch, fr, L, dev)
        ll = torch.empty(n, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        eng.ekf_form_set(0)
        for k in range(2):   # warm-up
            eng.obs_set_dev(tabs[k, 0], tabs[k, 1], L)
            eng.ekf_update_dev((a, b)[k & 1], (b, a)[k & 1], 5 * Lp, Lp, L, pose[0], pose[1], pose[2], None, n, MEAS_VAR, ll)
        eng.profile_enable(eng.PROF_EKF)
        eng.profile_read(eng.PROF_EKF)
        for k in range(12):
            eng.obs_set_dev(tabs[k, 0], tabs[k, 1], L)
            eng.ekf_update_dev((a, b)[k & 1], (b, a)[k & 1], 5 * Lp, Lp, L, pose[0], pose[1], pose[2], None, n, MEAS_VAR, ll)
        eng.profile_enable()
        ms, cnt = eng.profile_read(eng.PROF_EKF)
        t = kernel_ms(ms, cnt)
        eng.ekf_form_set(args.ekf_form)
        res["north_star"] = {"what": "EKF sweep (--mode ekf form): ekf_update_kernel, identity ancestors, every landmark observed",
                             "particles": n, "landmarks": L, "launches": int(cnt), "avg_launch_ms": t,
                             "no_reuse_frac": 40 * n * L / (t * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "read_only_frac": 20 * n * L / (t * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "target_read_only_frac": 0.40}
        tc = eng.profile_copy_ceiling(a, b, n, Lp, 8)
        tc_small = eng.profile_copy_ceiling(a, b, 65536, 512, 20)
        res["copy_ceiling"] = {"what": "pure copy with the update's access shape on this box (slam_profile_copy_ceiling): "
                                       "20 B read + 20 B written per (row, column), no arithmetic",
                               "rows": n, "columns": Lp, "ms_per_copy": tc,
                               "read_write_gbs": 40 * n * Lp / (tc * 1e-3) / 1e9,
                               "copy_ceiling_read_only_frac": 20 * n * Lp / (tc * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "at_65536x512": {"ms_per_copy": tc_small,
                                                "copy_ceiling_read_only_frac": 20 * 65536 * 512 / (tc_small * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        res["north_star"]["fraction_of_copy_ceiling"] = (tc * L / Lp) / t   # the copy moves 1 024 columns, the sweep's algorithmic bytes are 1 000
        del a, b, pose, tabs, ll
        torch.cuda.empty_cache()
    else:
        res["north_star"] = {"skipped": f"needs 43 GB of HBM, {free_b / 1e9:.0f} GB free"}

    # ---- (c) configs[1] with the 32 nearest landmarks observed, every frame resampled: rows against pages
    n, L, Lp = 65536, 500, 512
    rng = np.random.default_rng(4321)
    lm = make_landmarks(L, rng)
    steps, warm, chunk = 40, 12 + preroll_frames(args), 10   # the same pre-roll as the main path
    fr = make_frames(steps + warm, args.beams, lm, rng, 32)
    d_scan = torch.from_numpy(np.stack([np.stack([f["bx"], f["by"]]) for f in fr])).to(dev)
    tabs = obs_tables(torch, fr, L, dev)
    e2e = {"what": "configs[1] (65536 x 500, 360 beams, 1024^2 EDT) with the 32 nearest landmarks observed per frame, every "
                   "frame resampled: whole frames on the C session; `*_ms` = median over chunks of 10 frames (one "
                   "synchronisation per chunk; `*_ms_chunks` lists them)", "steps": steps}
    for layout in ("rows", "split", "pages", "split_pages", "auto"):
        ses = pkg.PfSession(eng, n, L, sigma=SIGMA, meas_var=MEAS_VAR, score_gain=SCORE_GAIN, seed=1234, map_layout=layout)
        g = torch.Generator(device="cpu").manual_seed(1234)
        p0 = true_pose(0)
        ses.set_poses(*[(p0[k] + s * torch.randn(n, generator=g)).numpy() for k, s in ((0, 0.05), (1, 0.05), (2, 0.01))])
        m0 = torch.zeros((n, 5, Lp), dtype=torch.float32, device=dev)
        fill_maps(torch, m0, lm, L, dev, n)
        torch.cuda.synchronize()
        ses.set_map_dev(m0, 5 * Lp, Lp)
        eng.sync()
        del m0
        chunk_ms = []
        settle(torch)
        for k in range(steps + warm):
            if k >= warm and (k - warm) % chunk == 0:
                torch.cuda.synchronize()
                if k > warm:
                    chunk_ms.append(1e3 * (time.perf_counter() - t0) / chunk)
                t0 = time.perf_counter()
            eng.scan_set_dev(d_scan[k, 0], d_scan[k, 1], args.beams)
            eng.obs_set_dev(tabs[k, 0], tabs[k, 1], L)
            ses.step(0, fr[k]["dp"], True)
        torch.cuda.synchronize()
        chunk_ms.append(1e3 * (time.perf_counter() - t0) / chunk)
        e2e[f"{layout}_ms"] = float(np.median(chunk_ms))
        e2e[f"{layout}_ms_chunks"] = [round(c, 4) for c in chunk_ms]
        e2e[f"{layout}_ended_on"] = ses.layout()
        ses.close()
    res["end_to_end_obs32"] = e2e

    # ---- (d) the headline workload measured the way rounds 1 and 2 measured it, for comparison with their driver records:
    # a cold filter (no pre-roll), 5 warm-up + 20 timed frames, score and landmark update as two launches, the update of EVERY
    # frame bracketed by HIP events.  The difference to `value` is method (steady state, sampled brackets) and the fused front.
    fr = make_frames(25, args.beams, lm, rng, 0)
    d_scan = torch.from_numpy(np.stack([np.stack([f["bx"], f["by"]]) for f in fr])).to(dev)
    tabs = obs_tables(torch, fr, L, dev)
    old = {}
    for name, fuse, lay in (("two_launches_cold_every_frame_bracketed", False, "rows"), ("fused_front_cold_every_frame_bracketed", True, "rows"),
                            ("default_layout_cold_every_frame_bracketed", True, args.map_layout)):
        ses = pkg.PfSession(eng, n, L, sigma=SIGMA, meas_var=MEAS_VAR, score_gain=SCORE_GAIN, seed=1234, map_layout=lay)
        g = torch.Generator(device="cpu").manual_seed(1234)
        ses.set_poses(*[(p0[k] + s * torch.randn(n, generator=g)).numpy() for k, s in ((0, 0.05), (1, 0.05), (2, 0.01))])
        m0 = torch.zeros((n, 5, Lp), dtype=torch.float32, device=dev)
        fill_maps(torch, m0, lm, L, dev, n)
        torch.cuda.synchronize()
        ses.set_map_dev(m0, 5 * Lp, Lp)
        eng.sync()
        del m0
        eng.frame_fusion_set(fuse)
        settle(torch)
        for k in range(25):
            if k == 5:
                torch.cuda.synchronize()
                eng.profile_enable(eng.PROF_EKF)
                eng.profile_read(eng.PROF_EKF)
                t0 = time.perf_counter()
            eng.scan_set_dev(d_scan[k, 0], d_scan[k, 1], args.beams)
            eng.obs_set_dev(tabs[k, 0], tabs[k, 1], L)
            ses.step(0, fr[k]["dp"], True)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 20
        eng.profile_enable()
        kms, kn = eng.profile_read(eng.PROF_EKF)
        old[name] = {"ms_per_step": ms, "particle_updates_per_s": n / (ms * 1e-3), "bracketed_kernel_avg_ms": kernel_ms(kms, kn),
                     "map_layout": ses.layout()}
        ses.close()
    eng.frame_fusion_set(True)
    old["what"] = ("configs[1] as BENCH_r01 / BENCH_r02 measured it (--steps 20 --warmup 5 from a cold start, every frame's dominant "
                   "launch bracketed): on rows with the front as two launches (round 2's frame) and fused (round 3's), and on this "
                   "run's --map-layout (the product's default)")
    res["headline_round2_method"] = old
    # the like-for-like figure for round-on-round comparisons: one method (cold start, 5 + 20 frames, every frame bracketed)
    res["value_cold_method"] = old["default_layout_cold_every_frame_bracketed"]["particle_updates_per_s"]
    res["value_cold_method_what"] = ("particle-updates/s of configs[1] measured by rounds 1-2's method (see headline_round2_method); "
                                     "`value` is the steady-state rate behind a pre-roll with sampled brackets")
    return res


def wants_strong_leg(args):
    """The north star's scaling target (>= 6x at 8 GPUs against 1) is defined on ONE problem, 1 048 576 particles x 1 000
    landmarks, split over the ranks — not on the weak-scaling headline.  A default run (the driver's `bench.py --gpus N`)
    therefore measures that problem too, as a second leg at every N, 1 included: the curve falls out of the legs' values."""
    return (args.mode == "pf" and args.scaling == "weak" and not args.no_extra_legs and not args.force_collectives and args.ess == 0.0
            and args.observed == 0 and (args.particles, args.landmarks, args.beams, args.grid) == (65536, 500, 360, 1024)
            and 1048576 % args.gpus == 0)


def strong_leg(args, ctx):
    """The second leg: `--scaling strong --particles-total 1048576 --landmarks 1000` on the same ranks."""
    import copy

    a = copy.copy(args)
    a.scaling, a.particles_total, a.particles, a.landmarks = "strong", 1048576, 1048576 // args.gpus, 1000
    a.steps, a.warmup, a.preroll = min(args.steps, 30), 5, 30
    a.no_extra_legs = a.no_cpu_baseline = a.no_sweep = True
    a.event_every = 2
    r = run_rank(a, ctx, build_inputs(a))
    if r is None:
        return None
    keep = {k: r[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "stage_avg_ms")}
    keep["what"] = ("the north-star problem, 1 048 576 particles x 1 000 landmarks (all observed), 360 beams, 1024^2 EDT, split over the "
                    "ranks: whole frames on the C session after 30 pre-roll + 5 warm-up frames; speed-up at N GPUs = this value at N / "
                    "this value at 1")
    keep["particles_per_gpu"] = r["config"]["particles_per_gpu"]
    keep["map_layout"] = r["config"]["map_layout"]["in_timed_region"]
    keep["rows_received_per_frame_max_rank"] = r["config"]["rows_received_per_frame_max_rank"]
    keep["dominant_kernel"] = {k: r["roofline"][k] for k in ("kernel", "avg_launch_ms", "launches", "traffic_model")}
    return keep


# ------------------------------------------------------------------------------------------------ launching
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed frames (SURVEY 8d: 200 timed frames after 20 warm-up)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=["pf", "score", "ekf"], default="pf")
    ap.add_argument("--particles", type=int, default=65536, help="per GPU (weak scaling, the default)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --particles per GPU (default; BASELINE configs[1] per GPU).  strong: --particles-total is "
                         "split over the ranks (the north star's 8-vs-1-GPU target is defined on the same total problem)")
    ap.add_argument("--particles-total", type=int, default=None, help="total population for --scaling strong")
    ap.add_argument("--beams", type=int, default=360)
    ap.add_argument("--landmarks", type=int, default=500)
    ap.add_argument("--grid", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the bounded legs after the timed region (north-star sweep, copy ceiling, 32-observed frames)")
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip the no-reuse EKF sweep after the timed region (so that a kernel trace of this run holds in-filter launches only)")
    ap.add_argument("--transport", choices=["rccl", "local"], default="rccl",
                    help="rccl (default): one process per GPU, RCCL over xGMI.  local: rehearsal, all ranks are threads of this "
                         "process on ONE card (--device-index, default 0), exchanging through slam_comm_create_local")
    ap.add_argument("--device-index", type=int, default=None, help="force every rank onto this GPU (--transport local)")
    ap.add_argument("--host-sensor", action="store_true", help="upload scan + observations from the host every frame")
    ap.add_argument("--preroll", type=int, default=120,
                    help="--mode pf: untimed frames the filter runs before the W warm-up steps, so that the timed steps see a "
                         "filter in steady state (0 = start cold from the 5 cm / 0.01 rad initial spread)")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="host sleep between set-up and the warm-up frames (see settle(); 0 = none)")
    ap.add_argument("--event-every", type=int, default=0,
                    help="bracket the timed kernels with HIP events in every N-th frame of the timed region (1 = every frame; "
                         "default 0 = every 8th frame, every 4th when --steps < 80, every 2nd when --steps < 12)")
    ap.add_argument("--events", choices=["dominant", "all", "none"], default="dominant",
                    help="kernels bracketed by HIP events inside the timed region (a pair costs a few us of stream time)")
    ap.add_argument("--observed", type=int, default=0,
                    help="landmarks seen per frame: 0 = all (default, the roofline workload), K = the K nearest")
    ap.add_argument("--ess", type=float, default=0.0,
                    help="ESS-gated resampling: resample only in frames whose effective sample size is below ESS * N "
                         "(0 = every frame, the default and the headline workload)")
    ap.add_argument("--presort-poses", action="store_true",
                    help="experiment, --mode score: upload the poses grouped by 4-pixel / matching-heading cells")
    ap.add_argument("--ekf-form", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="out-of-place EKF kernel: -1 the engine chooses (default), 0 one wavefront per particle, 1 / 2 per 4 / 2 particles")
    ap.add_argument("--map-layout", choices=["auto", "rows", "pages", "split", "split_pages"], default="auto",
                    help="slam_pf_config.map_layout: auto (default: the session chooses and may change while it runs: split for dense "
                         "frames on one GPU, pages for sparse ones, rows when sharded or gated), rows, pages, split (means per "
                         "particle, covariances per covariance class)")
    ap.add_argument("--paged", action="store_true", help="= --map-layout pages")
    ap.add_argument("--force-collectives", action="store_true",
                    help="diagnostics, --gpus 1 only: run the multi-GPU code path (every RCCL collective, the sharded index "
                         "kernels, the plan read-back) on a one-rank group to price its control overhead")
    args = ap.parse_args()
    global SETTLE_S
    SETTLE_S = max(0.0, args.settle_ms * 1e-3)
    if args.paged:
        args.map_layout = "pages"
    if args.mode != "pf":
        args.map_layout = "rows"   # the sweeps run on the session's row buffers
    return args


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.transport == "local" and world_env is not None and int(world_env) > 1:
        sys.exit("bench.py: --transport local runs its ranks as threads of ONE process; do not start it under torchrun")
    if args.scaling == "strong":
        if not args.particles_total or args.particles_total % args.gpus:
            sys.exit("bench.py: --scaling strong needs --particles-total divisible by the number of ranks")
        args.particles = args.particles_total // args.gpus

    if args.gpus > 1 and args.transport == "rccl" and world_env is None:
        # No launcher around us: become one.  Nothing in this process has touched the GPU yet (numpy only), the ranks are
        # fresh child processes; their rank 0 prints the JSON line on the stdout we share.
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    if args.transport == "local" and args.gpus > 1:
        from __graft_entry__ import load_package

        import torch

        pkg = load_package()
        inp = build_inputs(args)
        torch.cuda.init()   # the HIP runtime comes up ONCE, here, not in N threads at the same time
        torch.cuda.set_device(0 if args.device_index is None else args.device_index)
        torch.cuda.synchronize()
        shared = ThreadShared(pkg, args.gpus)
        results, errors = [None] * args.gpus, []

        def work(r):
            try:
                tctx = ThreadCtx(args, r, shared)
                results[r] = run_rank(args, tctx, inp, last=not wants_strong_leg(args))
                if wants_strong_leg(args):
                    leg = strong_leg(args, tctx)
                    if r == 0:
                        results[0]["north_star_strong"] = leg
            except BaseException as ex:   # a rank that dies must not leave the others inside a rendezvous
                errors.append((r, ex))
                shared.bar.abort()

        th = [threading.Thread(target=work, args=(r,), name=f"rank{r}") for r in range(args.gpus)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        shared.group.close()
        if errors:
            for r, ex in errors:
                print(f"bench.py: rank {r} failed: {type(ex).__name__}: {ex}", file=sys.stderr)
            sys.exit(1)
        print(json.dumps(results[0]))
        return

    ctx = SingleCtx(args)
    if ctx.world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ctx.world}")
    out = run_rank(args, ctx, build_inputs(args), last=not wants_strong_leg(args))
    if wants_strong_leg(args):
        try:
            leg = strong_leg(args, ctx)
        except Exception as ex:   # a leg that fails must not cost the headline line
            leg = {"error": f"{type(ex).__name__}: {ex}"}
            if ctx.world > 1:
                # several ranks: the peers may be anywhere (inside a collective of the leg, or past it) — no further collective
                # is safe.  Rank 0 hands over the headline it already has; every rank that got here leaves without the
                # process-group teardown (which is collective); ranks still waiting give up at their own time limit.
                print(f"bench.py: rank {ctx.rank}: north_star_strong failed: {leg['error']}", file=sys.stderr, flush=True)
                if ctx.rank == 0:
                    out["north_star_strong"] = leg
                    print(json.dumps(out), flush=True)
                sys.stdout.flush()
                os._exit(0 if ctx.rank == 0 else 1)
        if ctx.rank == 0:
            out["north_star_strong"] = leg
    if ctx.rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
