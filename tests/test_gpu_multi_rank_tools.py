"""What a first run on several GPUs will need, rehearsed on one card: bench.py's own launcher and its one-card
rehearsal of N ranks, the failure path of the communicators, the per-stage timers, the copy-ceiling probe.
RCCL with more than one rank cannot run here (it refuses two ranks on one device): the RCCL halves of these paths are
exercised on a one-rank communicator only."""
import json
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

from __graft_entry__ import ROOT, load_package

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _bench(*flags, timeout=600):
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *flags], capture_output=True, text=True, timeout=timeout)


def test_bench_two_ranks_on_one_card_local_transport():
    """python bench.py --gpus 2 --transport local: two ranks as threads sharing this card, the sharded C session with every
    exchange step; ONE JSON line, rows really travel."""
    r = _bench("--gpus", "2", "--transport", "local", "--steps", "5", "--warmup", "3", "--no-cpu-baseline", "--particles", "16384")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["particles_total"] == 32768 and out["scaling"] == "weak"
    assert out["config"]["rows_received_per_frame_max_rank"] > 0
    assert out["config"]["transport"].startswith("local")
    assert out["value"] > 0 and out["cpu_baseline"] is None
    for stage in ("score", "ekf", "weights", "scan", "ancestors", "plan", "pack", "unpack", "collectives"):
        assert stage in out["stage_avg_ms"], (stage, out["stage_avg_ms"])


def test_bench_two_ranks_paged_local_transport():
    """... the same with the maps on pages and 32 landmarks observed (bench.py --paged used to refuse several ranks)."""
    r = _bench("--gpus", "2", "--transport", "local", "--steps", "5", "--warmup", "3", "--no-cpu-baseline", "--particles", "8192",
               "--paged", "--observed", "32")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    # at 8 192 particles per rank the paged update and the scorer take about as long as each other: either may be the dominant kernel
    assert out["n_gpus"] == 2 and out["roofline"]["kernel"] in ("ekf_paged_kernel", "score_poses_kernel")
    assert out["config"]["map_layout"]["in_timed_region"].startswith("pages")


def test_bench_launches_its_own_ranks():
    """A bare `python bench.py --gpus 2` (no torchrun around it) starts torch.distributed.run on itself before it touches
    the GPU.  This box has ONE GPU, so the second rank finds no device and the launch must fail — loudly, through the
    launcher, without a JSON line."""
    r = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "torch.distributed" in r.stderr or "ChildFailedError" in r.stderr or "elastic" in r.stderr, r.stderr[-1500:]


def test_a_rank_that_gives_up_releases_the_others():
    """In-process group of 3: rank 1 aborts its communicator at frame 3 (slam_comm_abort).  The other ranks' next collective
    fails with SLAM_ERR_COMM at once — not after the 120 s rendezvous limit — and every later call on their
    communicators fails the same way."""
    from test_gpu_configs import _run_c_session_ranks

    pkg = load_package()
    t0 = time.perf_counter()
    with pytest.raises((pkg.SlamError, RuntimeError)) as ei:
        _run_c_session_ranks(3, 3072, 6, 6, fail_rank=1, fail_frame=3)
    assert time.perf_counter() - t0 < 30
    if isinstance(ei.value, pkg.SlamError):
        assert ei.value.status == -6   # SLAM_ERR_COMM


def test_abort_on_a_one_rank_rccl_communicator():
    """ncclCommAbort through the ABI: after slam_comm_abort the session's next frame returns SLAM_ERR_COMM, destroy works."""
    import _shard_worker as W

    pkg = load_package()
    meta, edt, bx, by, lm = W.make_world(L=6)
    eng = pkg.Engine(0)
    eng.grid_set_dev(0, torch.from_numpy(edt).to(DEV), pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
    eng.scan_upload(bx, by)
    comm = pkg.Comm.rccl(eng, 0, 1, pkg.comm_unique_id())
    ses = pkg.PfSession(eng, 2048, 0, comm=comm)
    for _ in range(3):
        ses.step(0, [0.01, 0.0, 0.0], False)
    ses.best()
    comm.abort()
    with pytest.raises(pkg.SlamError) as ei:
        ses.step(0, [0.01, 0.0, 0.0], False)
    assert ei.value.status == -6
    ses.close()
    comm.close()
    eng.close()


def test_stage_timers_cover_a_single_gpu_frame_and_copy_ceiling():
    """slam_profile_*: every stage of a single-GPU frame reports launches; the copy probe runs at a sane rate."""
    import _shard_worker as W

    pkg = load_package()
    L = 200
    meta, edt, bx, by, lm = W.make_world(L=L)
    eng = pkg.Engine(0)
    eng.grid_set_dev(0, torch.from_numpy(edt).to(DEV), pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
    eng.scan_upload(bx, by)
    for layout, stages in (("rows", ("score", "ekf", "weights", "scan", "ancestors")),
                           ("pages", ("score", "ekf", "weights", "scan", "ancestors", "pages"))):
        ses = pkg.PfSession(eng, 4096, L, map_layout=layout)
        x, y, th, mp = W.init_state(4096, L, lm)
        ses.set_poses(x, y, th)
        ses.set_map(mp)
        eng.profile_enable(*range(eng.PROF_COUNT))
        for f in range(4):
            eng.obs_upload(*W.observations(lm, f), L)
            ses.step(0, [0.01, 0.0, 0.0], True)
        eng.profile_enable()
        got = {eng.PROF_NAMES[k]: eng.profile_read(k) for k in range(eng.PROF_COUNT)}
        for s in stages:
            assert got[s][1] == 4 and 0 < got[s][0] < 100.0, (layout, s, got[s])
        for s in ("plan", "pack", "unpack", "collectives"):
            assert got[s][1] == 0
        ses.close()
    a = torch.zeros((8192, 5, 512), device=DEV)
    b = torch.empty_like(a)
    torch.cuda.synchronize()
    ms = eng.profile_copy_ceiling(a, b, 8192, 512, 10)
    assert torch.equal(a, b)
    rate = 2 * a.numel() * 4 / (ms * 1e-3) / 1e12
    assert 0.5 < rate < 40.0, rate                      # TB/s, read + write (84 MB each way: out of the Infinity Cache)
    with pytest.raises(pkg.SlamError):
        eng.profile_copy_ceiling(a, a, 8192, 512, 1)
    eng.close()
