"""The CPU specification of the particle-filter stages (oracle/slam_oracle_pf.c, rows A9-A12: no reference counterpart)
checked as a FILTER, not against itself: on a synthetic drive through bench.py's room it must keep the pose, learn the
landmarks, and its landmark variances must follow the Kalman law for k independent sightings of a static point, P = R / k.
The GPU session is held to the same on a larger population in tests/test_gpu_filter_behaviour.py."""
import importlib.util
import sys
from pathlib import Path

import numpy as np

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


def test_specification_is_a_working_filter(orc):
    B = _bench()
    n, L, frames, grid, beams = 1024, 16, 25, 512, 180
    rng = np.random.default_rng(7)
    room = B.ROOM
    lm = np.stack([rng.uniform(room[0] + 0.5, room[2] - 0.5, L), rng.uniform(room[1] + 0.5, room[3] - 0.5, L)], 1)
    pixel, min_x, min_y = np.float32(20.48 / grid), np.float32(-4.24), np.float32(-10.24)
    occ = B.occupancy(grid, float(pixel), float(min_x), float(min_y))
    edt = orc.edt(occ, grid, grid, 10.0, "window")
    meta = orc.meta(grid, grid, grid, float(pixel), float(min_x), float(min_y))
    fr = B.make_frames(frames, beams, lm, rng)
    sigma, mvar, gain, seed = (0.01, 0.01, 0.002), 0.02 ** 2 * 4, 0.04, 5
    p0 = B.true_pose(0)
    x, y, th = [(p0[k] + s * rng.standard_normal(n)).astype(np.float32) for k, s in ((0, 0.05), (1, 0.05), (2, 0.01))]
    mp = np.zeros((n, 5, L), np.float32)
    mp[:, 2, :] = -1.0                                    # nothing seen yet
    anc, err = None, []
    for f, frame in enumerate(fr, start=1):
        x, y, th = orc.motion_sample(x, y, th, anc, n, 0, frame["dp"], sigma, seed, f - 1)
        score, _ = orc.score_poses_det(meta, edt, frame["bx"], frame["by"], x, y, th)
        mp, ll = orc.ekf_update(mp, x, y, th, anc, frame["ids"], frame["zx"], frame["zy"], mvar)
        logw, m = orc.logweight(score, ll, gain)
        wq, _ = orc.quantise_weights(logw, m)
        anc = orc.resample(wq, seed, f - 1)
        truth = B.true_pose(f)
        w = wq.astype(np.float64) / wq.astype(np.float64).sum()
        err.append((float(np.hypot((w * x).sum() - truth[0], (w * y).sum() - truth[1])), abs(float((w * th).sum() - truth[2]))))
    err = np.array(err)
    post = mp[anc]                                        # the population after the last resample
    lm_est = post[:, 0:2, :].mean(axis=0).T
    lm_err = np.hypot(lm_est[:, 0] - lm[:, 0], lm_est[:, 1] - lm[:, 1])
    print(f"pose error max {err[:, 0].max():.4f} m / {err[:, 1].max():.5f} rad; landmark error mean {lm_err.mean():.4f} "
          f"max {lm_err.max():.4f} m; P_xx {post[:, 2, :].mean():.3e} (R / k = {mvar / frames:.3e})")
    # start known to 5 cm / 0.01 rad (1 sigma), 4 cm pixels, 2 cm observation noise
    assert err[:, 0].max() < 0.04 and err[:, 1].max() < 0.01 and err[-10:, 0].max() < 0.03
    assert lm_err.max() < 0.06 and lm_err.mean() < 0.03
    # k sightings of a static landmark with measurement covariance R I (first sighting: P = R): P = R / k, whatever the pose
    for plane in (2, 4):
        assert np.allclose(post[:, plane, :], mvar / frames, rtol=2e-3)
    assert np.abs(post[:, 3, :]).max() < 1e-3 * mvar / frames
