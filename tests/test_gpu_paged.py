"""Landmark maps as copy-on-write pages (slam_pf_paged_set, csrc/paged_kernels.hip) against the row-per-particle session:
the same poses, maps, heaviest particle and posterior mean, bit for bit, frame after frame — page tables shared by the
offspring of an ancestor, fresh pages for the touched ones, the free list rebuilt every frame, frames without
observations, first sightings out of an empty map, the resample gate."""
import numpy as np
import pytest
import torch

from __graft_entry__ import load_package
from conftest import bits

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _run(paged, n, L, frames, ess, obs_of, start):
    import _shard_worker as W

    pkg = load_package()
    meta, edt, bx, by, lm = W.make_world(L=max(L, 1))
    lm = lm[:L]
    x, y, th, mp = W.init_state(n, L, lm)
    eng = pkg.Engine(0)
    eng.grid_set_dev(0, torch.from_numpy(edt).to(DEV), pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
    eng.scan_upload(bx, by)
    ses = pkg.PfSession(eng, n, L, seed=77, sigma=(0.02, 0.02, 0.004), meas_var=0.02, score_gain=0.05, resample_ess_frac=ess,
                        map_layout="pages" if paged else "rows")
    assert ses.is_paged() == bool(paged)
    if start == "empty":
        ses.reset([0.0, 0.0, 0.0])
        ses.set_poses(x, y, th)
    else:
        ses.set_poses(x, y, th)
        ses.set_map(mp)
    out = {"best": [], "mean": []}
    for f in range(frames):
        obs = obs_of(f, lm)
        if obs is not None:
            eng.obs_upload(*obs, L)
        ses.step(0, [0.01, -0.005, 0.002], obs is not None)
        out["best"].append(ses.best())
        out["mean"].append(ses.mean(0.05))
        if f == frames // 2:
            out["map_mid"] = ses.maps()
    out["pose"], out["map"], out["resampled"] = ses.poses(), ses.maps(), ses.frames_resampled()
    ses.close()
    eng.close()
    return out


def _some(k, stride=1):
    def f(frame, lm):
        if frame == 2:
            return None                                   # a frame without observations: the tables just follow
        r = np.random.default_rng(1000 + frame)
        L = len(lm)
        ids = np.sort(r.permutation(L)[:min(k, L)])[::stride].astype(np.int32)
        z = lm[ids] + 0.01 * np.float32(frame)
        return ids, z[:, 0].copy(), z[:, 1].copy()
    return f


def _block(k):
    def f(frame, lm):                                     # k neighbouring landmarks, the block moves from frame to frame
        L = len(lm)
        ids = ((np.arange(k) + 37 * frame) % L).astype(np.int32)
        ids = np.unique(ids).astype(np.int32)
        z = lm[ids] + 0.01 * np.float32(frame)
        return ids, z[:, 0].copy(), z[:, 1].copy()
    return f


@pytest.mark.parametrize("n,L,obs,start,ess", [
    (3000, 6, _some(6), "map", 0.0),          # one page per particle
    (3000, 40, _some(40, 2), "map", 0.0),     # two pages, every second landmark
    (2048, 200, _some(12), "empty", 0.0),     # sparse observations on an empty map: first sightings, one shared start page
    (4096, 500, _block(32), "map", 0.0),      # the K-nearest pattern: a block of neighbours, 16 pages per particle
    (4096, 500, _block(32), "map", 0.2),      # ... with the resample gate
    (1000, 129, _some(129), "map", 0.0),      # every landmark, a last page with one landmark in it
    (257, 1000, _some(300), "empty", 0.5),    # many touched pages (odd counts), long tables
])
def test_paged_session_equals_row_session(n, L, obs, start, ess):
    frames = 8
    rows = _run(False, n, L, frames, ess, obs, start)
    pages = _run(True, n, L, frames, ess, obs, start)
    assert np.array_equal(bits(pages["pose"]), bits(rows["pose"]))
    assert np.array_equal(bits(pages["map_mid"]), bits(rows["map_mid"]))
    assert np.array_equal(bits(pages["map"]), bits(rows["map"]))
    assert pages["resampled"] == rows["resampled"]
    for a, b in zip(pages["best"], rows["best"]):
        assert a[2] == b[2] and a[1] == b[1] and np.array_equal(bits(a[0]), bits(b[0]))
    for a, b in zip(pages["mean"], rows["mean"]):
        assert np.array_equal(bits(a), bits(b))
    if ess:
        assert 0 < rows["resampled"] < frames - 1, rows["resampled"]


def test_paged_session_many_frames_of_free_list_turnover():
    """60 frames: the free list is used up and made anew many times over (12 of 200 landmarks seen per frame: ~5 of 7
    pages touched; the list lasts one or two frames), pages are shared, dropped and recycled — still the row session's bits."""
    rows = _run(False, 2048, 200, 60, 0.0, _some(12), "empty")
    pages = _run(True, 2048, 200, 60, 0.0, _some(12), "empty")
    assert np.array_equal(bits(pages["pose"]), bits(rows["pose"]))
    assert np.array_equal(bits(pages["map_mid"]), bits(rows["map_mid"]))
    assert np.array_equal(bits(pages["map"]), bits(rows["map"]))
    for a, b in zip(pages["best"], rows["best"]):
        assert a[2] == b[2] and a[1] == b[1]


@pytest.mark.parametrize("world,n_total,L,ess", [(2, 4096, 6, 0.0), (4, 4096, 40, 0.0), (3, 3000, 6, 0.0), (4, 4096, 6, 0.5),
                                                 (8, 16384, 100, 0.0)])
def test_sharded_paged_session_equals_one_rank_on_rows(world, n_total, L, ess):
    """Paged maps in a SHARDED session (ranks = threads on this card, in-process transport): a migrating particle travels
    with all its pages and lands on fresh pages behind a staging table row.  Poses, maps and the heaviest particle equal
    the single-GPU row session's, bit for bit; rows really travel."""
    from test_gpu_configs import _run_c_session_ranks

    frames = 9
    one = _run_c_session_ranks(1, n_total, L, frames, transport=None, ess=ess)[0]
    many = _run_c_session_ranks(world, n_total, L, frames, ess=ess, paged=True)
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in many], axis=1)), bits(one["pose"]))
    assert np.array_equal(bits(np.concatenate([p["map"] for p in many], axis=0)), bits(one["map"]))
    for p in many:
        assert p["best"][2] == one["best"][2] and p["best"][1] == one["best"][1]
        assert np.array_equal(bits(p["mean"]), bits(one["mean"]))
    if not ess:
        assert max(max(p["rows"]) for p in many) > 10


def test_paged_session_set_map_dev_and_views():
    """slam_pf_set_map_dev (rows on the device -> pages), slam_pf_get_map_host back; a paged session shows no rows."""
    pkg = load_package()
    eng = pkg.Engine(0)
    eng.pf_paged_set(True)
    n, L, Lp = 700, 70, 96
    ses = pkg.PfSession(eng, n, L)
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((n, 5, Lp)).astype(np.float32)
    d = torch.from_numpy(rows).to(DEV)
    torch.cuda.synchronize()
    ses.set_map_dev(d, 5 * Lp, Lp)
    assert np.array_equal(bits(ses.maps()), bits(rows[:, :, :L]))
    wide0 = torch.full((n, 5 * Lp + 37), 7.0, device=DEV)
    wide0[:, :5 * Lp] = d.reshape(n, 5 * Lp)
    torch.cuda.synchronize()
    ses.set_map_dev(wide0, 5 * Lp + 37, Lp)
    assert np.array_equal(bits(ses.maps()), bits(rows[:, :, :L]))
    v = ses.device_view()
    assert v["map"] is None and v["map_spare"] is None and v["pose"] is not None
    ses.close()
    eng.pf_paged_set(False)
    ses = pkg.PfSession(eng, n, L, map_layout="rows")   # rows: the same entry point copies into the row buffer
    ses.set_map_dev(d, 5 * Lp, Lp)
    assert np.array_equal(bits(ses.maps()), bits(rows[:, :, :L]))
    wide = torch.full((n, 5 * Lp + 37), 7.0, device=DEV)          # a row stride wider than 5 planes
    wide[:, :5 * Lp] = d.reshape(n, 5 * Lp)
    torch.cuda.synchronize()
    ses.set_map_dev(wide, 5 * Lp + 37, Lp)
    assert np.array_equal(bits(ses.maps()), bits(rows[:, :, :L]))
    assert ses.device_view()["map"] is not None
    ses.close()
    eng.close()
