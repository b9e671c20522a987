"""Worker of the world_size-2 gloo test (tests/test_pf_sharding_gloo.py).  TEST INFRASTRUCTURE."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def make_world(seed=11, rows=120, cols=140, nbeams=90, L=6):
    import oracle

    rng = np.random.default_rng(seed)
    occ = np.zeros((rows, cols), np.int32)
    occ[5, 5:cols - 5] = occ[rows - 6, 5:cols - 5] = 1
    occ[5:rows - 5, 5] = occ[5:rows - 5, cols - 6] = 1
    occ[40:60, 70:80] = 1
    edt = oracle.edt(occ, rows, cols, 10.0, "window")
    meta = oracle.meta(rows, cols, cols, 0.1, -7.0, -6.0)
    ang = np.linspace(-np.pi, np.pi, nbeams, endpoint=False)
    rad = rng.uniform(1.0, 4.5, nbeams)
    bx, by = (rad * np.cos(ang)).astype(np.float32), (rad * np.sin(ang)).astype(np.float32)
    lm = rng.uniform(-3, 3, (L, 2)).astype(np.float32)
    return meta, edt, bx, by, lm


def init_state(n_total, L, lm, seed=5):
    rng = np.random.default_rng(seed)
    x = (0.3 * rng.standard_normal(n_total)).astype(np.float32)
    y = (0.3 * rng.standard_normal(n_total)).astype(np.float32)
    th = (0.05 * rng.standard_normal(n_total)).astype(np.float32)
    # the second half of the population starts far off: its weights collapse and its slots get
    # refilled from the first half, i.e. from the OTHER rank when sharded over two
    x[n_total // 2:] += 2.5
    mp = np.zeros((5, L, n_total), np.float32)
    mp[0] = lm[:, 0:1] + 0.1 * rng.standard_normal((L, n_total))
    mp[1] = lm[:, 1:2] + 0.1 * rng.standard_normal((L, n_total))
    mp[2] = 0.05; mp[4] = 0.05
    if L:
        mp[2, L - 1] = -1.0   # one landmark not seen yet
    return x, y, th, np.ascontiguousarray(mp.transpose(2, 0, 1))   # one row per particle: [n][5][L]


def observations(lm, frame):
    ids = np.arange(len(lm), dtype=np.int32)
    if frame % 2:
        ids = ids[::2].copy()   # some frames observe only a subset: exercises the copy-through path
    z = lm[ids] + 0.01 * np.float32(frame)
    return ids, z[:, 0].copy(), z[:, 1].copy()


def run_filter(rank, world, n_total, L, frames, group=None):
    from __graft_entry__ import load_package
    from _oracle_ops import OracleOps

    load_package()
    from _pf_rehearsal import ParticleFilter

    meta, edt, bx, by, lm = make_world(L=L)
    ops = OracleOps(meta, edt, bx, by)
    n = n_total // world
    pf = ParticleFilter(ops, n, L, device="cpu", rank=rank, world=world, group=group, seed=77,
                        sigma=(0.02, 0.02, 0.004), meas_var=0.02, score_gain=0.05 if L else 1.0)
    x, y, th, mp = init_state(n_total, L, lm)
    sl = slice(rank * n, (rank + 1) * n)
    pf.set_poses(x[sl], y[sl], th[sl])
    if L:
        pf.set_map(mp[sl])
    migrated = []
    for f in range(frames):
        pf.step([0.01, -0.005, 0.002], observations(lm, f) if L else None)
        migrated.append(pf.migrated_last)
    out = {"pose": pf.poses().clone().numpy(), "logw": pf.logw.clone().numpy(), "migrated": migrated,
           "best": pf.best_particle()}
    if L:
        out["map"] = pf.maps().clone().numpy()
    return out


def worker(rank, world, port, n_total, L, frames, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = run_filter(rank, world, n_total, L, frames)
        np.savez(Path(outdir) / f"rank{rank}.npz", pose=out["pose"], logw=out["logw"], migrated=np.array(out["migrated"]),
                 best=np.array(out["best"]), **({"map": out["map"]} if L else {}))
    finally:
        dist.destroy_process_group()


def worker_gpu(rank, world, port, n_total, L, frames, outdir):
    """Two ranks sharing ONE MI355X (cuda:0), gloo for the exchange (staged through the host by pf.py):
    the real HIP stages + the real sharding/migration logic, without needing a second card."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package

        pkg = load_package()
        from _pf_rehearsal import HipOps, ParticleFilter

        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        meta, edt, bx, by, lm = make_world(L=L)
        eng = pkg.Engine(0)
        d_edt = torch.from_numpy(edt).to(dev)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.grid_set_dev(0, d_edt, pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
        eng.scan_upload(bx, by)
        n = n_total // world
        pf = ParticleFilter(HipOps(eng), n, L, device=dev, rank=rank, world=world, seed=77, sigma=(0.02, 0.02, 0.004),
                            meas_var=0.02, score_gain=0.05 if L else 1.0, grid_slot=0)
        x, y, th, mp = init_state(n_total, L, lm)
        sl = slice(rank * n, (rank + 1) * n)
        pf.set_poses(x[sl], y[sl], th[sl])
        if L:
            pf.set_map(mp[sl])
        migrated = []
        for f in range(frames):
            pf.step([0.01, -0.005, 0.002], observations(lm, f) if L else None)
            migrated.append(pf.migrated_last)
        torch.cuda.synchronize()
        out = {"pose": pf.poses().cpu().numpy(), "logw": pf.logw.cpu().numpy(), "migrated": np.array(migrated),
               "best": np.array(pf.best_particle())}
        if L:
            out["map"] = pf.maps().cpu().numpy()
        np.savez(Path(outdir) / f"rank{rank}.npz", **out)
        eng.close()
    finally:
        dist.destroy_process_group()
