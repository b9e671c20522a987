"""slam_pf_config.map_layout = SLAM_MAP_AUTO: a session whose observation density changes while it runs moves its maps
between rows and pages and stays bit-identical to a session pinned to rows (and to one pinned to pages) all the way."""
import numpy as np
import pytest
import torch

from __graft_entry__ import load_package
from conftest import bits

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _obs(frame, lm, dense_from, dense_to):
    L = len(lm)
    if dense_from <= frame < dense_to:
        ids = np.arange(L, dtype=np.int32)                               # every landmark
    else:
        ids = ((np.arange(24) + 31 * frame) % L).astype(np.int32)        # 24 neighbours of 400: well under a quarter
        ids = np.unique(ids).astype(np.int32)
    z = lm[ids] + 0.01 * np.float32(frame % 7)
    return ids, z[:, 0].copy(), z[:, 1].copy()


def _run(layout, n, L, frames, ess=0.0, paged_set=False):
    import _shard_worker as W

    pkg = load_package()
    meta, edt, bx, by, lm = W.make_world(L=L)
    x, y, th, mp = W.init_state(n, L, lm)
    eng = pkg.Engine(0)
    eng.pf_paged_set(paged_set)
    eng.grid_set_dev(0, torch.from_numpy(edt).to(DEV), pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
    eng.scan_upload(bx, by)
    ses = pkg.PfSession(eng, n, L, seed=77, sigma=(0.02, 0.02, 0.004), meas_var=0.02, score_gain=0.05, resample_ess_frac=ess,
                        map_layout=layout)
    ses.set_poses(x, y, th)
    ses.set_map(mp)
    out = {"best": [], "paged": [], "maps": {}}
    for f in range(frames):
        eng.obs_upload(*_obs(f, lm, 14, 30), L)
        ses.step(0, [0.01, -0.005, 0.002], True)
        out["best"].append(ses.best())
        out["paged"].append(ses.is_paged())
        if f in (5, 13, 20, 29, frames - 1):
            out["maps"][f] = ses.maps()
    out["pose"], out["changes"] = ses.poses(), ses.layout_changes()
    ses.close()
    eng.close()
    return out


@pytest.mark.parametrize("ess", [0.0, 0.3])
def test_auto_layout_follows_the_observation_density_and_keeps_the_bits(ess):
    frames = 44
    rows = _run("rows", 4096, 400, frames, ess)
    auto = _run("auto", 4096, 400, frames, ess)
    pages = _run("pages", 4096, 400, frames, ess)
    assert not any(rows["paged"]) and all(pages["paged"]) and rows["changes"] == 0 and pages["changes"] == 0
    # sparse frames 0..13 -> pages after three counts; dense frames 14..29 -> back to rows; sparse again -> pages again
    # (after the first eight frames the count is taken every 8th frame, so the later moves take a few dozen frames)
    assert not auto["paged"][0] and auto["paged"][6] and auto["paged"][13], auto["paged"]
    assert auto["changes"] >= 2 and not all(auto["paged"][14:40]), (auto["changes"], auto["paged"])
    for other in (auto, pages):
        assert np.array_equal(bits(other["pose"]), bits(rows["pose"]))
        for f, m in rows["maps"].items():
            assert np.array_equal(bits(other["maps"][f]), bits(m)), f
        for a, b in zip(other["best"], rows["best"]):
            assert a[2] == b[2] and a[1] == b[1] and np.array_equal(bits(a[0]), bits(b[0]))


def test_paged_set_pins_auto_sessions_to_pages_and_small_maps_stay_on_rows():
    pinned = _run("auto", 1024, 400, 20, paged_set=True)
    assert all(pinned["paged"]) and pinned["changes"] == 0
    small = _run("auto", 1024, 24, 20)            # one page per particle: nothing to gain, AUTO stays on rows
    assert not any(small["paged"])


@pytest.mark.parametrize("world", [3])
def test_auto_layout_in_a_sharded_session(world):
    """Ranks of a sharded session decide on their own (the counts arrive without synchronisation); a migrating particle's
    record does not depend on the layout of either side, so the population still equals one rank on rows."""
    from test_gpu_configs import _run_c_session_ranks

    n_total, L, frames = 3072, 100, 16
    one = _run_c_session_ranks(1, n_total, L, frames, transport=None)[0]
    many = _run_c_session_ranks(world, n_total, L, frames, layout="auto", sparse_obs=True)
    ref = _run_c_session_ranks(1, n_total, L, frames, transport=None, sparse_obs=True)[0]
    assert not np.array_equal(bits(ref["map"]), bits(one["map"]))           # the sparse frames are a different run
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in many], axis=1)), bits(ref["pose"]))
    assert np.array_equal(bits(np.concatenate([p["map"] for p in many], axis=0)), bits(ref["map"]))
    assert any(p["paged_end"] for p in many)


def test_sharded_auto_with_a_map_read_between_every_two_frames():
    """A map getter between two frames completes the exchange early: the rows of remote ancestors then already sit in the
    staging tail when SLAM_MAP_AUTO moves the maps at the start of the next frame, and the pending gather index names them.
    Both moves (rows -> pages on sparse frames, pages -> rows on dense ones) must take the tail along: the population equals
    one rank on rows, frame by frame."""
    from test_gpu_configs import _run_c_session_ranks

    n_total, L, frames, world = 3072, 100, 30, 3
    kw = dict(sparse_obs=True, dense_from=12, maps_every_frame=True)
    ref = _run_c_session_ranks(1, n_total, L, frames, transport=None, **kw)[0]
    many = _run_c_session_ranks(world, n_total, L, frames, layout="auto", **kw)
    assert set(ref["layouts"]) == {"rows"}
    moved = [p["layouts"] for p in many]
    # (round 4: a sharded session's AUTO moves between split and split pages like a single-GPU one)
    assert any("split_pages" in m[:12] for m in moved) and any("split_pages" in m and m[-1] == "split" for m in moved), moved
    assert any(sum(p["rows"]) > 0 for p in many), "nothing migrated: the staging tail was never used"
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in many], axis=1)), bits(ref["pose"]))
    assert np.array_equal(bits(np.concatenate([p["map"] for p in many], axis=0)), bits(ref["map"]))
    for f in range(frames):   # every 97th particle of the population, every frame
        got = np.concatenate([p["frame_maps"][f] for p in many], axis=0)
        assert np.array_equal(bits(got), bits(ref["frame_maps"][f])), f
