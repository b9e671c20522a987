"""bench.py's host-side bookkeeping (no GPU): argument defaults, the frames it synthesises for pre-roll + warm-up + timed
steps + the stage pass, the synthetic world of SURVEY 8(d)."""
import sys

import numpy as np

import bench


def _args(*argv):
    old = sys.argv
    sys.argv = ["bench.py", *argv]
    try:
        return bench.parse_args()
    finally:
        sys.argv = old


def test_defaults_follow_survey_8d():
    a = _args()
    assert (a.gpus, a.steps, a.warmup, a.particles, a.beams, a.landmarks, a.grid) == (1, 200, 20, 65536, 360, 500, 1024)
    assert a.mode == "pf" and a.scaling == "weak" and a.map_layout == "auto" and a.preroll == 120
    assert bench.preroll_frames(a) == 120
    assert bench.preroll_frames(_args("--mode", "score")) == 0 and bench.preroll_frames(_args("--preroll", "0")) == 0


def test_frames_cover_preroll_warmup_steps_and_stage_pass():
    a = _args("--steps", "7", "--warmup", "3", "--preroll", "5", "--landmarks", "40", "--grid", "256", "--observed", "8")
    inp = bench.build_inputs(a)
    assert len(inp["frames"]) == 5 + 7 + 3 + 12
    f = inp["frames"][0]
    assert f["bx"].shape == (360,) and f["bx"].dtype == np.float32 and f["dp"].shape == (3,)
    assert len(f["ids"]) == 8 and len(set(f["ids"].tolist())) == 8 and f["zx"].shape == (8,)
    occ = inp["occ"]
    assert occ.shape == (256, 256) and 0.005 < occ.mean() < 0.05          # ~1-2 % occupied
    b = bench.build_inputs(a)                                              # deterministic
    assert all(np.array_equal(x["bx"], y["bx"]) and np.array_equal(x["ids"], y["ids"]) for x, y in zip(inp["frames"], b["frames"]))


def test_landmark_ids_follow_a_space_filling_sweep():
    rng = np.random.default_rng(1)
    lm = bench.make_landmarks(512, rng)
    key = bench.morton(lm)
    assert np.all(np.diff(key) >= 0)
    d = np.hypot(*(lm[1:] - lm[:-1]).T)
    assert np.median(d) < 1.0                                              # neighbours in the row are neighbours in the room
