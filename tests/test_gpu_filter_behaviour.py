"""Behaviour of the particle-filter stages as a FILTER (rows A9-A12 have no reference counterpart, so bit-exactness against
this repository's CPU specification says nothing about whether that specification is a sound FastSLAM): a robot driving
along an arc through the synthetic room of bench.py, scans ray-cast from the true pose, landmark observations with 2 cm of
noise, maps that start out empty.  The C-level session must keep the pose and learn the landmarks."""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("slam_bench_for_tests", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def _run(ess, paged=False, n=8192, L=60, frames=40, grid=1024, beams=360):
    B = _bench()
    pkg = load_package()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2026)
    room = B.ROOM
    lm = np.stack([rng.uniform(room[0] + 0.5, room[2] - 0.5, L), rng.uniform(room[1] + 0.5, room[3] - 0.5, L)], 1)
    pixel, min_x, min_y = np.float32(20.48 / grid), np.float32(-4.24), np.float32(-10.24)
    occ = B.occupancy(grid, float(pixel), float(min_x), float(min_y))
    fr = B.make_frames(frames, beams, lm, rng)
    eng = pkg.Engine(0)
    eng.pf_paged_set(paged)
    d_occ = torch.from_numpy(occ).to(dev)
    d_edt = torch.empty((grid, grid), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    eng.edt_dev(d_occ, grid, grid, grid, 10.0, d_edt)
    eng.grid_set_dev(0, d_edt, pkg.grid_meta(grid, grid, grid, pixel, min_x, min_y))
    ses = pkg.PfSession(eng, n, L, seed=11, sigma=(0.01, 0.01, 0.002), meas_var=0.02 ** 2 * 4, score_gain=0.02,
                        resample_ess_frac=ess)
    p0 = B.true_pose(0)
    ses.reset(p0.astype(np.float32))                                   # empty maps (P_xx = -1: not seen yet)
    ses.set_poses(*[(p0[k] + s * rng.standard_normal(n)).astype(np.float32) for k, s in ((0, 0.05), (1, 0.05), (2, 0.01))])
    err_xy, err_th = [], []
    for f, frame in enumerate(fr, start=1):
        eng.scan_upload(frame["bx"], frame["by"])
        eng.obs_upload(frame["ids"], frame["zx"], frame["zy"], L)
        ses.step(0, frame["dp"], True)
        truth = B.true_pose(f)
        est = ses.mean(float(truth[2]))
        err_xy.append(float(np.hypot(est[0] - truth[0], est[1] - truth[1])))
        err_th.append(abs(float(est[2] - truth[2])))
    maps = ses.maps()                                                  # [n][5][L], the population after the last resample
    resampled = ses.frames_resampled()
    ses.close()
    eng.close()
    lm_est = maps[:, 0:2, :].mean(axis=0).T                            # posterior mean of every landmark
    lm_err = np.hypot(lm_est[:, 0] - lm[:, 0], lm_est[:, 1] - lm[:, 1])
    return dict(err_xy=np.array(err_xy), err_th=np.array(err_th), lm_err=lm_err, pxx=maps[:, 2, :].mean(axis=0),
                pyy=maps[:, 4, :].mean(axis=0), resampled=resampled, frames=frames)


@pytest.mark.parametrize("ess,paged", [(0.0, False), (0.1, False), (0.0, True)])
def test_filter_keeps_the_pose_and_learns_the_landmarks(ess, paged):
    r = _run(ess, paged)
    print(f"ess={ess}: pose error last 10 frames max {r['err_xy'][-10:].max():.4f} m / {r['err_th'][-10:].max():.5f} rad, "
          f"over all frames {r['err_xy'].max():.4f} m; landmark error mean {r['lm_err'].mean():.4f} max {r['lm_err'].max():.4f} m; "
          f"P_xx mean {r['pxx'].mean():.2e}; frames resampled {r['resampled']} of {r['frames']}")
    # the start is known to 5 cm / 0.01 rad (1 sigma); pixels are 2 cm; observations carry 2 cm of noise
    # (measured: 6-7 mm / 0.9 mrad; landmarks 4 mm mean, 11 mm max)
    assert r["err_xy"].max() < 0.02 and r["err_th"].max() < 0.004
    assert r["lm_err"].max() < 0.03 and r["lm_err"].mean() < 0.01
    # every landmark has been seen 40 times; k sightings of a static point with measurement covariance R I (first
    # sighting: P = R) leave P = R / k whatever the poses were: the Kalman law, independent of this repository's CPU port
    rk = 0.02 ** 2 * 4 / r["frames"]
    assert np.allclose(r["pxx"], rk, rtol=5e-3) and np.allclose(r["pyy"], rk, rtol=5e-3)
    if ess:
        assert 0 < r["resampled"] < r["frames"] - 1       # the gate kept some populations and resampled others
