#!/usr/bin/env python3
"""bench.py — particle-updates/s of the particle-filter frame loop on synthetic 360-beam scans.

    python bench.py [--gpus N --steps K --warmup W]                 (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W    (N > 1, one rank per GPU, RCCL)

A "step" is one whole frame of the hot path over one batch of synthetic input: motion sample ->
scan-match score (the reference's FastMatch inner loop, Subsystem_1/main.c:459-518, for every
particle) -> per-particle x per-landmark 2x2 EKF over ALL landmarks -> weight normalisation ->
systematic resample (gathers fused into the next frame's motion/EKF kernels) -> migration between
GPUs.  Workload = BASELINE.json configs[1] per GPU: 65536 particles, 360 beams, 500 landmarks,
1024 x 1024 EDT grid; weak scaling (per-GPU work fixed).  Particles, maps, EDT and scan are resident
in HBM when the timed region starts, and so is the sensor data of every frame (360 beams + 500
observations = 8.9 KB per frame; --host-sensor sends it over PCIe frame by frame instead).
value = N_total_particles * K / max-over-ranks wall time.

Other workloads (never the default; used to fill BASELINE.md):
    --mode score   scan-match-only microbench (configs[2]: --particles 1048576 --grid 2048)
    --mode ekf     EKF sweep only (north-star roofline case: --particles 1048576 --landmarks 1000)
    --scaling strong --particles-total 1048576 --landmarks 1000
                   the north-star target: N ranks split the SAME problem (what ">= 6x at 8 GPUs" is defined on)

One JSON line on stdout (rank 0), including
  roofline     — dominant kernel: launch duration from HIP events recorded on the kernel's own stream inside the
                 timed region (slam_profile_* in the C ABI).  `achieved` is the HBM traffic rate when the PMC traffic of
                 this workload is on file (profiles/traffic.json), else the algorithmic-byte rate; `logical_rate_gbs` is
                 always the algorithmic-byte rate (40 B per particle and OBSERVED landmark); `no_reuse` is a
                 short sweep of the same kernel on the same buffers with identity ancestors, run after the timed region
                 (nothing comes out of L2 there: the streaming figure); `read_only_frac` is the north star's definition
                 (20 B x n x L_observed / t / peak) on that sweep
  cpu_baseline — the CPU port of the same frame loop (oracle/, single thread) timed on this host
                 on a bounded sample (rank 0, N = 1 only); cpu_baseline_threads: the same port with its per-particle
                 stages on up to 16 host threads (SURVEY 8d's optional "all cores" line), about 8 s more
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

# synthetic world: 15 x 11 m room with two boxes, in the reference's pose convention
ROOM = (-3.0, -5.5, 12.0, 5.5)
BOXES = [(5.0, 2.0, 7.0, 3.5), (2.0, -4.0, 3.0, -3.0)]


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
        x, y, th = oracle.motion_sample(x, y, th, anc, n, 0, fr["dp"], args.sigma, 1234, done)
        score, _ = oracle.score_poses_det(m, edt, fr["bx"], fr["by"], x, y, th)
        ll = None
        if L and args.mode != "score":
            mp, ll = oracle.ekf_update(mp, x, y, th, anc, fr["ids"], fr["zx"], fr["zy"], args.meas_var)
        logw, mx = oracle.logweight(score, ll, args.score_gain)
        wq, _ = oracle.quantise_weights(logw, mx)
        anc = oracle.resample(wq, 1234, done)
        done += 1
        el = time.perf_counter() - t0
        if el > budget_s and done >= 3:
            break
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n * done / el, "unit": "particle-updates/s", "cores": 1, "kind": "port",
            "host_cpu": model, "host_cores_total": os.cpu_count(),
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
        x, y, th = oracle.motion_sample(src[0], src[1], src[2], a, per, c * per, fr["dp"], args.sigma, 1234, frame)
        for k, v in enumerate((x, y, th)):
            new[k][sl] = v
        score[sl] = oracle.score_poses_det(m, edt, fr["bx"], fr["by"], x, y, th)[0]
        if use_ekf:
            ids = np.ascontiguousarray(fr["ids"], np.int32)
            lib.orc_ekf_update(mp if a is not None else mp[sl], mp2[sl], 5 * L, L, L, x, y, th,
                               a.ctypes.data_as(C.c_void_p) if a is not None else None, per, ids,
                               np.ascontiguousarray(fr["zx"], np.float32), np.ascontiguousarray(fr["zy"], np.float32), len(ids),
                               args.meas_var, ll[sl])

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
            logw, mx = oracle.logweight(score, ll if use_ekf else None, args.score_gain)
            wq, _ = oracle.quantise_weights(logw, mx)
            anc = np.ascontiguousarray(oracle.resample(wq, 1234, done), np.int32)
            done += 1
            el = time.perf_counter() - t0
            if el > budget_s and done >= 3:
                break
    return {"value": n * done / el, "unit": "particle-updates/s", "cores": T, "kind": "port",
            "sample": f"{n} particles x {done} frames of the same workload, oracle/ C port, per-particle stages on {T} host "
                      f"threads (weights and resample on one), {el:.1f} s"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
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
    ap.add_argument("--driver", choices=["c", "py"], default="c",
                    help="c (default): the C session slam_pf_* — one call per frame, several GPUs: the engine issues every RCCL "
                         "exchange itself.  py: pf.py on the stage entry points with torch.distributed (rehearsal with gloo)")
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip the no-reuse EKF sweep after the timed region (so that a kernel trace of this run holds in-filter launches only)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, production) | gloo (functional rehearsal)")
    ap.add_argument("--device-index", type=int, default=None, help="force every rank onto this GPU (rehearsal only)")
    ap.add_argument("--host-sensor", action="store_true", help="upload scan + observations from the host every frame")
    ap.add_argument("--events", choices=["dominant", "all", "none"], default="dominant",
                    help="kernels bracketed by HIP events inside the timed region (a pair costs a few us of stream time)")
    ap.add_argument("--observed", type=int, default=0,
                    help="landmarks seen per frame: 0 = all (default, the roofline workload), K = the K nearest")
    ap.add_argument("--ess", type=float, default=0.0,
                    help="ESS-gated resampling: resample only in frames whose effective sample size is below ESS * N "
                         "(0 = every frame, the default and the headline workload); --driver c")
    ap.add_argument("--presort-poses", action="store_true",
                    help="experiment, --mode score: upload the poses grouped by 4-pixel / matching-heading cells (what a "
                         "spatial ordering of the lanes would buy the scorer)")
    ap.add_argument("--ekf-form", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="out-of-place EKF kernel: -1 the engine chooses (default), 0 one wavefront per particle, 1 / 2 per 4 / 2 particles")
    ap.add_argument("--paged", action="store_true",
                    help="C session, one GPU: landmark maps as copy-on-write pages (slam_pf_paged_set) instead of one row per "
                         "particle: for frames that observe few of many landmarks (--observed K); no no-reuse sweep")
    ap.add_argument("--force-collectives", action="store_true",
                    help="diagnostics, --gpus 1 only: run the multi-GPU code path (every RCCL collective, the sharded index "
                         "kernels, the plan read-back) on a one-rank group to price its control overhead")
    ap.add_argument("--stats", action="store_true",
                    help="diagnostics: print the number of distinct resample ancestors per frame (synchronises; not for timing)")
    args = ap.parse_args()
    args.sigma = (0.01, 0.01, 0.002)
    args.meas_var = 0.02 ** 2 * 4
    args.score_gain = 0.02

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch N>1 with torch.distributed.run")
    if args.scaling == "strong":
        if not args.particles_total or args.particles_total % world:
            sys.exit("bench.py: --scaling strong needs --particles-total divisible by the number of ranks")
        args.particles = args.particles_total // world

    import torch
    import torch.distributed as dist

    from __graft_entry__ import load_package

    pkg = load_package()
    from hardware_acceleration_of_lidar_slam_amd.pf import HipOps, ParticleFilter

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the engine has no CPU fallback")
    dev_index = local_rank if args.device_index is None else args.device_index
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world == 1 and args.force_collectives and args.driver == "py":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(args.dist_backend, rank=0, world_size=1, **({"device_id": dev} if args.dist_backend == "nccl" else {}))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                dist.barrier()   # RCCL initialises lazily: do it (and print its banner) here, not in the timed region
            else:
                dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    eng = pkg.Engine(dev_index)
    eng.ekf_form_set(args.ekf_form)
    if args.paged:
        if args.driver != "c" or world > 1 or args.force_collectives or args.mode != "pf":
            sys.exit("bench.py: --paged is for the C session on one GPU, --mode pf")
        eng.pf_paged_set(True)
        args.no_sweep = True
    use_c = args.driver == "c"
    if use_c and args.dist_backend != "nccl":
        sys.exit("bench.py: --driver c exchanges over RCCL; use --driver py for a gloo rehearsal")
    if use_c and args.stats:
        sys.exit("bench.py: --stats needs --driver py (the C session keeps its weights to itself)")
    ops = None
    if not use_c:
        ops = HipOps(eng)
        ops.bind_stream()   # the py driver shares torch's current stream; the C session runs on the engine's own

    # ---- synthetic inputs (identical on every rank)
    rng = np.random.default_rng(4321)
    L = 0 if args.mode == "score" else args.landmarks
    landmarks = np.stack([rng.uniform(ROOM[0] + 0.5, ROOM[2] - 0.5, L), rng.uniform(ROOM[1] + 0.5, ROOM[3] - 0.5, L)], 1)
    if L:   # landmark ids in discovery order along a sweep of the room: neighbours in space are neighbours in the map rows
        landmarks = landmarks[np.argsort(morton(landmarks), kind="stable")]
    pixel = np.float32(20.48 / args.grid)
    min_x, min_y = np.float32(-4.24), np.float32(-10.24)
    occ = occupancy(args.grid, float(pixel), float(min_x), float(min_y))
    nframes = args.steps + args.warmup + 10   # + the short per-kernel timing pass after the timed region
    frames = make_frames(nframes, args.beams, landmarks, rng, args.observed)

    d_occ = torch.from_numpy(occ).to(dev)
    d_edt = torch.empty((args.grid, args.grid), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    eng.edt_dev(d_occ, args.grid, args.grid, args.grid, 10.0, d_edt)
    meta = pkg.grid_meta(args.grid, args.grid, args.grid, pixel, min_x, min_y)
    eng.grid_set_dev(0, d_edt, meta)

    n = args.particles
    multi_path = world > 1 or args.force_collectives
    comm = None
    if use_c:
        if multi_path:
            # rendezvous token of the engine's own RCCL communicator: made by rank 0, handed out through the process
            # group torchrun set up (used for nothing else but this, the barriers and the final timing reduction)
            uid = torch.zeros(pkg.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
            if rank == 0:
                uid = torch.tensor(list(pkg.comm_unique_id()), dtype=torch.uint8, device=dev)
            if world > 1:
                dist.broadcast(uid, src=0)
            with stdout_to_stderr():   # RCCL prints its version banner to stdout; stdout carries ONE JSON line
                comm = pkg.Comm.rccl(eng, rank, world, bytes(uid.cpu().tolist()))
        pf = pkg.PfSession(eng, n, L, sigma=args.sigma, meas_var=args.meas_var, score_gain=args.score_gain, seed=1234, comm=comm,
                           resample_ess_frac=args.ess)
        Lp = (L + 31) // 32 * 32
    else:
        pf = ParticleFilter(ops, n, L, device=dev, rank=rank, world=world, seed=1234, sigma=args.sigma,
                            meas_var=args.meas_var, score_gain=args.score_gain, grid_slot=0,
                            force_collectives=args.force_collectives and world == 1)
        Lp = pf.Lp

    def views():
        """(pose [3][n], current map [rows][5][Lp], spare map, pending gather index or None) as torch tensors."""
        if use_c:
            v = pf.device_view()
            t = {k: (torch.as_tensor(v[k], device=dev) if v[k] is not None else None) for k in ("pose", "map", "map_spare", "anc")}
            return t["pose"], t["map"], t["map_spare"], t["anc"]
        return pf.pose[pf.cur], (pf.map[pf.cur] if L else None), (pf.map[1 - pf.cur] if L else None), pf.src_idx

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    p0 = true_pose(0)
    init = [(p0[k] + s * torch.randn(n, generator=g)).numpy() for k, s in ((0, 0.05), (1, 0.05), (2, 0.01))]
    if args.presort_poses:
        cell = 4.0 * float(pixel)
        key = np.lexsort((np.floor(init[0] / cell), np.floor(init[1] / cell), np.floor(init[2] / (cell / 8.0))))
        init = [a[key] for a in init]
    pf.set_poses(*init)
    if L:
        lm = torch.from_numpy(landmarks.astype(np.float32)).to(dev)
        paged = use_c and pf.is_paged()
        # paged maps have no rows to write into: the same rows are made in a scratch tensor and handed over once
        m0 = torch.empty((n, 5, Lp), dtype=torch.float32, device=dev) if paged else views()[1]   # [particle][plane][Lp]
        for i0 in range(0, n, 65536):                     # in chunks: the temporaries of a 1M x 1k map are 4 GB each
            i1 = min(i0 + 65536, n)
            m0[i0:i1, 0, :L] = lm[:, 0] + 0.1 * torch.randn((i1 - i0, L), device=dev)
            m0[i0:i1, 1, :L] = lm[:, 1] + 0.1 * torch.randn((i1 - i0, L), device=dev)
        m0[:n, 2, :L] = 0.05
        m0[:n, 3, :L] = 0.0
        m0[:n, 4, :L] = 0.05
        if paged:
            torch.cuda.synchronize()
            pf.set_map_dev(m0, 5 * Lp, Lp)
            eng.sync()
            del m0
    score_t = torch.zeros(n, dtype=torch.float32, device=dev)
    count_t = torch.zeros(n, dtype=torch.int32, device=dev)
    loglik_t = torch.zeros(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    # sensor data of every frame resident in HBM before the timed region (bench contract); --host-sensor
    # uploads it frame by frame through the host-buffer entry points instead (8.9 KB per frame over PCIe)
    d_scan = torch.from_numpy(np.stack([np.stack([f["bx"], f["by"]]) for f in frames])).to(dev)        # [F][2][B]
    if L:   # observation tables indexed by landmark (NaN = not observed; here every landmark is observed)
        tab = np.full((len(frames), 2, L), np.nan, np.float32)
        for k, f in enumerate(frames):
            tab[k, 0, f["ids"]] = f["zx"]
            tab[k, 1, f["ids"]] = f["zy"]
        d_z = torch.from_numpy(tab).to(dev)                                                              # [F][2][L]

    # per-frame views of the resident sensor data, made once (a tensor slice costs microseconds of host time)
    scan_v = [(d_scan[k, 0], d_scan[k, 1]) for k in range(len(frames))]
    obs_v = [(d_z[k, 0], d_z[k, 1]) for k in range(len(frames))] if L else None

    sweep_pose, sweep_maps = None, None

    def one_step(k):
        nonlocal sweep_pose, sweep_maps
        fr = frames[k]
        if args.host_sensor:
            eng.scan_upload(fr["bx"], fr["by"])
            obs, obs_dev = ((fr["ids"], fr["zx"], fr["zy"]) if L else None), None
        else:
            eng.scan_set_dev(scan_v[k][0], scan_v[k][1], args.beams)
            obs, obs_dev = None, (obs_v[k] if L else None)
        if args.mode == "pf" and use_c:
            if obs_dev:
                eng.obs_set_dev(*obs_dev, L)
            elif obs:
                eng.obs_upload(*obs, L)
            pf.step(0, fr["dp"], L > 0)
        elif args.mode == "pf":
            pf.step(fr["dp"], obs, obs_dev)
            if args.stats:
                w = torch.exp((pf.logw - pf.logw.max()).double())
                print(f"[stats] rank {rank} frame {k}: distinct ancestors {torch.unique(pf.src_idx).numel()} of {n}, "
                      f"ESS {float(w.sum() ** 2 / (w * w).sum()):.1f}, received rows {pf.migrated_last}", file=sys.stderr)
        else:
            if sweep_pose is None:
                sweep_pose, a, b, _ = views()
                sweep_maps = (a, b)
            p = sweep_pose
            if args.mode == "score":
                eng.score_poses_dev(0, p[0], p[1], p[2], n, score_t, count_t)
            else:   # ekf sweep: out of place, ping-pong between the two map buffers
                if obs_dev:
                    eng.obs_set_dev(*obs_dev, L)
                else:
                    eng.obs_upload(*obs, L)
                eng.ekf_update_dev(sweep_maps[k & 1], sweep_maps[1 - (k & 1)], 5 * Lp, Lp, L, p[0], p[1], p[2], None, n,
                                   args.meas_var, loglik_t)

    def rows_received():
        return (pf.rows_received() if use_c else pf.migrated_last) if args.mode == "pf" else 0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        one_step(k)
    barrier()
    dominant = eng.PROF_SCORE if args.mode == "score" or L == 0 else eng.PROF_EKF
    timed = {"dominant": (dominant,), "all": (eng.PROF_SCORE, eng.PROF_EKF), "none": ()}[args.events]
    eng.profile_enable(*timed)
    for kk in (eng.PROF_SCORE, eng.PROF_EKF):
        eng.profile_read(kk)
    migrated = 0
    forms0 = eng.ekf_form_counts() + eng.ekf_inplace_form_counts()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        one_step(k)
        migrated += rows_received()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.profile_enable()
    forms1 = eng.ekf_form_counts() + eng.ekf_inplace_form_counts()
    forms = tuple(b - a for a, b in zip(forms0, forms1))   # EKF launches of the timed region, by kernel (out of place x2, in place x2)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        t[0] = migrated / max(args.steps, 1)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        migrated = float(t[0])

    score_ms, score_n = eng.profile_read(eng.PROF_SCORE)
    ekf_ms, ekf_n = eng.profile_read(eng.PROF_EKF)
    # A start/stop event bracket around ONE kernel also contains the stream's marker handling; an empty bracket
    # measures it (~5-7 us).  Both are reported; the kernel duration used for the roofline is bracket - empty
    # bracket, which is what rocprofv3's kernel trace of the same process shows (profiles/README.md).
    bracket_overhead_ms = eng.profile_bracket_overhead()
    if args.events != "all":   # the kernels not timed inside the region: a short extra pass, outside the timing
        eng.profile_enable(eng.PROF_SCORE, eng.PROF_EKF)
        for k in range(min(10, args.steps)):
            one_step(args.warmup + args.steps + k)   # the trajectory simply continues (no wrap-around)
        eng.profile_enable()
        s2, n2 = eng.profile_read(eng.PROF_SCORE)
        e2, m2 = eng.profile_read(eng.PROF_EKF)
        if not score_n:
            score_ms, score_n = s2, n2
        if not ekf_n:
            ekf_ms, ekf_n = e2, m2
    n_total = n * world
    value = n_total * args.steps / elapsed

    def kernel_ms(total_ms, launches):
        return max(total_ms / max(launches, 1) - bracket_overhead_ms, 0.0)

    # how much of the population the resample kept distinct (the rows of repeated ancestors come out of L2, which is
    # what makes the in-filter EKF faster than a sweep): one torch.unique, outside the timed region
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
    if L and args.mode != "score" and not args.no_sweep:
        eng.profile_enable(eng.PROF_EKF)
        eng.profile_read(eng.PROF_EKF)
        p, ma, mb, _ = views()
        base = args.warmup + args.steps
        for k in range(12):
            eng.obs_set_dev(*obs_v[(base + k) % len(frames)], L)
            eng.ekf_update_dev((ma, mb)[k & 1], (mb, ma)[k & 1], 5 * Lp, Lp, L, p[0], p[1], p[2], None, n,
                               args.meas_var, loglik_t)
        eng.profile_enable()
        nr_ms, nr_n = eng.profile_read(eng.PROF_EKF)
        t_nr = kernel_ms(nr_ms, nr_n)
        if t_nr > 0:
            no_reuse = {"achieved": ekf_bytes / (t_nr * 1e-3) / 1e9, "frac": ekf_bytes / (t_nr * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "avg_launch_ms": t_nr, "launches": int(nr_n), "bytes_per_launch": ekf_bytes,
                        "what": "ekf_update_kernel, identity ancestors (no row is shared), same buffers and observations"}

    # the engine picks one of two out-of-place EKF kernels that give the same bits; name the one that ran
    ekf_name = "ekf_update_group_kernel" if forms[1] > forms[0] else "ekf_update_kernel"
    if args.paged:
        ekf_name = "ekf_paged_kernel"
    if ekf_n and (ekf_ms >= score_ms or args.mode == "ekf"):
        kern, raw_ms, dur_ms, alg = ekf_name, ekf_ms / ekf_n, kernel_ms(ekf_ms, ekf_n), ekf_bytes
    else:
        kern, raw_ms, dur_ms, alg = ("score_poses_kernel", score_ms / max(score_n, 1), kernel_ms(score_ms, score_n),
                                     score_bytes)
    logical = alg / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
    traffic, traffic_src = None, None
    tfile = ROOT / "profiles" / "traffic.json"   # HBM bytes per launch from rocprofv3 --pmc runs of this same command
    if tfile.exists():
        key = (f"{args.mode}:{n}:{args.beams}:{L}:{args.grid}" + (":paged" if args.paged else "") + (f":obs{L_obs}" if L_obs != L else "")
               + (f":ess{args.ess}" if 0 < args.ess < 1 else ""))   # a gated run has its own traffic (none on file: falls back)
        rec = json.loads(tfile.read_text()).get(key, {})
        traffic, traffic_src = rec.get("ekf_update_kernel" if kern.startswith("ekf") else kern), rec.get("source")
    # `achieved`: the rate at which HBM itself was driven when the PMC traffic of this workload is on file; else the
    # no-reuse sweep of the same kernel (a filter's EKF re-reads shared ancestor rows from L2, so its algorithmic-byte
    # rate is not an HBM rate and can exceed the peak); the scorer's EDT gathers are cache-resident by design, its
    # logical rate is reported against the HBM peak because BASELINE.json asks for it.
    if traffic and dur_ms > 0:
        achieved, basis = traffic / (dur_ms * 1e-3) / 1e9, f"hbm_traffic (PMC, {traffic_src}) / launch time in the timed region"
    elif kern.startswith("ekf") and no_reuse and args.mode == "pf":
        achieved, basis = no_reuse["achieved"], "no_reuse sweep (no PMC traffic on file for this workload)"
    else:
        achieved, basis = logical, "algorithmic bytes / launch time in the timed region"
    t_ro = no_reuse["avg_launch_ms"] if no_reuse else (dur_ms if kern.startswith("ekf") else 0.0)
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
                   "edt_grid": f"{args.grid}x{args.grid}", "parallelism": f"particle-shard x{world}" + (" (multi-GPU code path forced)" if args.force_collectives else ""),
                   "rows_received_per_frame_max_rank": migrated if world > 1 else 0,
                   "distinct_ancestor_frac": distinct_frac,
                   "resample_ess_frac": args.ess if use_c else 0.0,
                   "map_layout": "pages (copy-on-write, 32 landmarks)" if args.paged else "rows",
                   "frames_resampled": (pf.frames_resampled() if use_c and 0 < args.ess < 1 else None)},
        "roofline": {"bound": "hbm", "kernel": kern, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "achieved_basis": basis,
                     "algorithmic_bytes_per_launch": alg, "logical_rate_gbs": logical,
                     "logical_frac": logical / HBM_PEAK_GBS,
                     "no_reuse": no_reuse,
                     # the north star's own definition: 20 B x n x L_observed / t / peak, t from the no-reuse sweep
                     "read_only_frac": (20 * n * L_obs / (t_ro * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_ro > 0 and L else None,
                     "note": ("in a running filter the rows of repeated resample ancestors are re-read from L2: "
                              "`logical_rate_gbs` (SURVEY 8d's 40 B per particle and observed landmark / launch time) is then "
                              "not an HBM rate; `achieved` is, see `achieved_basis`; `no_reuse` is the kernel streaming "
                              "every row from HBM" if kern.startswith("ekf_update") else
                              "paged maps: the update reads the touched pages of the ancestors (shared pages out of L2) and writes "
                              "fresh pages; `logical_rate_gbs` is SURVEY 8d's 40 B per particle and observed landmark / launch time"
                              if kern.startswith("ekf") else
                              "EDT gathers are served by L2 / Infinity Cache: logical-byte rate, not HBM traffic"),
                     "avg_launch_ms": dur_ms, "avg_event_bracket_ms": raw_ms,
                     "event_bracket_overhead_ms": bracket_overhead_ms,
                     "launches": int(ekf_n if kern.startswith("ekf") else score_n),
                     "ekf_launches_by_kernel": {"ekf_update_kernel": forms[0], "ekf_update_group_kernel": forms[1],
                                                "ekf_update_kernel(in place)": forms[2], "ekf_sparse_kernel": forms[3]},
                     "other_kernel_avg_ms": {"score_poses_kernel": kernel_ms(score_ms, score_n),
                                             "ekf_update_kernel": kernel_ms(ekf_ms, ekf_n)}},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, occ, (float(pixel), float(min_x), float(min_y)), frames, landmarks)
        out["cpu_baseline_threads"] = cpu_baseline_threads(args, occ, (float(pixel), float(min_x), float(min_y)), frames, landmarks)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    torch.cuda.synchronize()
    if use_c:
        pf.close()
        if comm:
            comm.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
