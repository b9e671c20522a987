"""The SPLIT layout of the landmark maps (csrc/split_kernels.hip, ekf_split_body in csrc/pf_kernels.hip): means per particle,
covariances per covariance class.  The layout is not part of the specification: a split session must give the bits of a row
session — poses, maps, heaviest particle, frame after frame — whatever the classes look like (one for the whole population;
one per particle; anything between), for every row length (whole passes, tails, rows shorter than a batch), fused with the
scorer or as a launch of its own, on frames that observe everything, something or nothing.  PARITY UNPINNED all the same: the
reference has no landmarks; the split kernel is compared with the CPU specification directly in
tests/test_gpu_frame_front_at_size.py.

The bookkeeping is checked from the outside through slam_pf_split_device_view: neighbouring particles with bit-identical
covariance planes share a class, the list of classes in use only shrinks and always covers the classes the particles name,
and the class rows equal what the particles of a row session carry.
"""
import numpy as np
import pytest
import torch

from __graft_entry__ import load_package
from conftest import bits

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _tensor(a):
    return torch.as_tensor(a, device=DEV)


def _maps(n, L, lm, cov, seed=17):
    """cov: "shared" (every particle the same covariances), "own" (every particle its own, some landmarks unseen),
    "families" (runs of 1 .. 40 neighbouring particles share theirs)."""
    import _shard_worker as W

    x, y, th, mp = W.init_state(n, L, lm)
    rng = np.random.default_rng(seed)
    if cov == "shared":
        mp[:, 2], mp[:, 3], mp[:, 4] = 0.05, 0.01, 0.04
        mp[:, 2, L // 3] = -1.0
    else:
        a = (0.02 + 0.2 * rng.random((n, L))).astype(np.float32)
        c = (0.02 + 0.2 * rng.random((n, L))).astype(np.float32)
        b = ((rng.random((n, L)) - 0.5) * np.sqrt(a * c)).astype(np.float32)
        a[rng.random((n, L)) < 0.05] = -1.0
        if cov == "families":
            head = np.zeros(n, np.int64)
            i = 0
            while i < n:
                k = int(rng.integers(1, 41))
                head[i:i + k] = i
                i += k
            a, b, c = a[head], b[head], c[head]
        mp[:, 2], mp[:, 3], mp[:, 4] = a, b, c
    return x, y, th, mp


def _run(layout, n, L, frames, cov="own", fused=True, form=-1, nbeams=None, skip_obs=(), inspect=False, sparse=0):
    import _shard_worker as W

    pkg = load_package()
    meta, edt, bx, by, lm = W.make_world(L=L)
    x, y, th, mp = _maps(n, L, lm, cov)
    eng = pkg.Engine(0)
    eng.frame_fusion_set(fused)
    eng.ekf_form_set(form)
    eng.grid_set_dev(0, torch.from_numpy(edt).to(DEV), pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
    eng.scan_upload(bx[:nbeams], by[:nbeams]) if nbeams else eng.scan_upload(bx, by)
    ses = pkg.PfSession(eng, n, L, seed=91, sigma=(0.02, 0.02, 0.004), meas_var=0.02, score_gain=0.05, map_layout=layout)
    ses.set_poses(x, y, th)
    ses.set_map(mp)
    out = {"best": [], "maps": {}, "layout": [], "live": [], "classes": []}
    if inspect:
        v = ses.split_view()
        cls = _tensor(v["cls"])[:n].cpu().numpy()
        same = np.all(bits(mp[1:, 2:5]) == bits(mp[:-1, 2:5]), axis=(1, 2))
        want = np.concatenate([[0], np.cumsum(~same)])
        assert np.array_equal(cls, want), "classes = runs of neighbouring particles with identical covariance planes"
        assert int(_tensor(v["live_count"])[0]) == want[-1] + 1
        got_cov = _tensor(v["cov"])[torch.from_numpy(cls).to(DEV).long()][:, :, :L].cpu().numpy()
        assert np.array_equal(bits(got_cov), bits(mp[:, 2:5]))
        assert np.array_equal(bits(_tensor(v["mean"])[:n, :, :L].cpu().numpy()), bits(mp[:, 0:2]))
    rng = np.random.default_rng(5)
    for f in range(frames):
        ids = np.sort(rng.choice(L, size=(L if f % 3 else max(L // 3, 1)), replace=False)).astype(np.int32)   # all, or a third
        if sparse and f % 4:   # `sparse` neighbours somewhere in the map on three frames of four
            ids = np.unique((np.arange(sparse) + 37 * f) % L).astype(np.int32)
        z = lm[ids] + np.float32(0.01) * np.float32(f % 5)
        eng.obs_upload(ids, z[:, 0].copy(), z[:, 1].copy(), L)
        ses.step(0, [0.01, -0.005, 0.002], f not in skip_obs)
        out["best"].append(ses.best())
        out["layout"].append(ses.layout())
        if inspect:
            eng.sync()
            v = ses.split_view()
            nlive = int(_tensor(v["live_count"])[0])
            live = set(_tensor(v["live"])[:nlive].cpu().numpy().tolist())
            named = set(np.unique(_tensor(v["cls"])[:n].cpu().numpy()).tolist())
            assert len(live) == nlive and named <= live, f"frame {f}: the list of classes in use misses {sorted(named - live)[:5]}"
            out["live"].append(nlive)
            out["classes"].append(len(named))
        if f in (1, frames - 1):
            sel = np.unique(np.concatenate([np.arange(0, n, max(n // 257, 1)), [n - 1]])).astype(np.int32)
            out["maps"][f] = ses.map_rows(sel)
    out["pose"] = ses.poses()
    out["all_maps"] = ses.maps() if n * L <= 4_000_000 else None
    out["fused_launches"] = eng.frame_fusion_count()
    ses.close()
    eng.close()
    return out


def _same(a, b):
    assert np.array_equal(bits(a["pose"]), bits(b["pose"]))
    for f, m in b["maps"].items():
        assert np.array_equal(bits(a["maps"][f]), bits(m)), f
    if b["all_maps"] is not None:
        assert np.array_equal(bits(a["all_maps"]), bits(b["all_maps"]))
    for p, q in zip(a["best"], b["best"]):
        assert p[2] == q[2] and p[1] == q[1] and np.array_equal(bits(p[0]), bits(q[0]))


@pytest.mark.parametrize("n,L,cov,form,nbeams", [
    (16384, 300, "own", -1, None), (16384, 300, "shared", -1, None), (16384, 300, "families", 1, None),
    (5000, 513, "families", -1, None), (6001, 700, "own", 2, None), (140000, 200, "families", -1, None),
    (3073, 130, "own", -1, 37), (4099, 129, "shared", 1, 1), (4096, 100, "families", -1, None), (2048, 300, "own", -1, None),
    (3000, 31, "families", -1, None), (1500, 5000, "families", -1, None)])
def test_split_gives_the_bits_of_rows(n, L, cov, form, nbeams):
    frames = 7
    rows = _run("rows", n, L, frames, cov, True, form, nbeams)
    for fused in (True, False):
        split = _run("split", n, L, frames, cov, fused, form, nbeams)
        assert set(split["layout"]) == {"split"}
        assert fused or split["fused_launches"] == 0
        _same(split, rows)


def test_split_fused_front_runs_where_the_row_session_fuses():
    rows = _run("rows", 16384, 300, 5, "shared")
    split = _run("split", 16384, 300, 5, "shared")
    assert rows["fused_launches"] == 4 and split["fused_launches"] == 4


def test_frames_without_observations_and_auto():
    """Frames without a landmark update move means and classes with their particles (split_gather_kernel); AUTO keeps a
    single-GPU session that resamples every frame on the split layout while its frames are dense."""
    rows = _run("rows", 8192, 260, 9, "families", skip_obs=(0, 3, 4, 8))
    split = _run("split", 8192, 260, 9, "families", skip_obs=(0, 3, 4, 8), inspect=True)
    auto = _run("auto", 8192, 260, 9, "families", skip_obs=(0, 3, 4, 8))
    assert set(auto["layout"]) == {"split"}
    _same(split, rows)
    _same(auto, rows)


@pytest.mark.parametrize("cov", ["shared", "families", "own"])
def test_classes_in_use_only_shrink(cov):
    out = _run("split", 12000, 200, 10, cov, inspect=True)
    live, classes = out["live"], out["classes"]
    assert all(a >= b for a, b in zip(live, live[1:])), live
    assert all(nl >= nc for nl, nc in zip(live, classes))
    if cov == "shared":
        assert live == [1] * 10
    else:
        assert live[0] > 100 and live[-1] < live[0]


def test_split_map_io_and_views():
    """Maps in and out through every entry point while the session is split: set_map (host), set_map_dev (any strides), maps(),
    map_rows(); slam_pf_device_view shows no rows; a reset leaves one class of landmarks not seen yet."""
    import _shard_worker as W

    pkg = load_package()
    n, L = 3000, 70
    Lp = (L + 31) // 32 * 32
    meta, edt, bx, by, lm = W.make_world(L=L)
    x, y, th, mp = _maps(n, L, lm, "families")
    eng = pkg.Engine(0)
    eng.grid_set_dev(0, torch.from_numpy(edt).to(DEV), pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
    eng.scan_upload(bx, by)
    ses = pkg.PfSession(eng, n, L, seed=3, sigma=(0.02, 0.02, 0.004), meas_var=0.02, score_gain=0.05, map_layout="split")
    assert ses.layout() == "split" and not ses.is_paged()
    v = ses.device_view()
    assert v["map"] is None and v["map_spare"] is None and v["pose"] is not None
    sv = ses.split_view()
    assert int(_tensor(sv["live_count"])[0]) == 1 and bool((_tensor(sv["cov"])[0, 0, :L] == -1).all())   # after create: nothing seen
    ses.set_poses(x, y, th)
    ses.set_map(mp)
    assert np.array_equal(bits(ses.maps()), bits(mp))
    sel = np.array([5, 5, 0, n - 1, 1234, 77], np.int32)
    assert np.array_equal(bits(ses.map_rows(sel)), bits(mp[sel]))
    wide = torch.full((n, 5 * Lp + 37), 7.0, device=DEV)
    d = torch.zeros((n, 5, Lp), device=DEV)
    d[:, :, :L] = torch.from_numpy(mp).to(DEV)
    wide[:, :5 * Lp] = d.reshape(n, 5 * Lp)
    torch.cuda.synchronize()
    ses.set_map_dev(wide, 5 * Lp + 37, Lp)
    assert np.array_equal(bits(ses.maps()), bits(mp))
    for f in range(3):
        eng.obs_upload(*W.observations(lm, f), L)
        ses.step(0, [0.01, -0.005, 0.002], True)
    assert np.array_equal(bits(ses.map_rows(sel)), bits(ses.maps()[sel]))      # pending gather applied in both
    with pytest.raises(pkg.SlamError):
        ses.set_map(mp)                                                         # a gather is pending
    ses.reset([0.0, 0.0, 0.0])
    m = ses.maps()
    assert bool((m[:, 2] == -1).all()) and bool((m[:, [0, 1, 3, 4]] == 0).all())
    ses.close()
    eng.close()


@pytest.mark.parametrize("world,n_total,L,recv_capacity,frames", [(2, 4096, 6, 0, 12), (3, 3000, 40, 0, 12), (4, 8192, 130, 0, 12),
                                                                  (8, 32768, 40, 0, 12), (4, 4096, 70, -600, 30), (2, 16384, 140, 0, 10),
                                                                  (3, 24576, 300, -900, 12)])
def test_sharded_split_ranks_on_one_card_equal_one_rank_on_rows(world, n_total, L, recv_capacity, frames, monkeypatch):
    """A sharded session on the split layout: covariance classes are local to a rank, a migrating particle travels as the same
    record as on rows (its means and its class's covariances) and becomes a class of its own where it arrives, numbered from a
    free list on the device.  2 / 3 / 4 / 8 ranks sharing this card against ONE rank on rows: poses, maps, heaviest particle,
    bit for bit.  Negative capacities: a list of class numbers hands out at most that many (SLAM_SPLIT_CLASS_ROOM), fewer than
    rows arrive over the run, so new lists are made from the stamps on the way.  Populations of >= 3 072 particles per rank with
    rows longer than 128 take the fused front: the groups fed from local rows go out with the score, the rest behind the exchange."""
    from test_gpu_configs import _run_c_session_ranks

    if recv_capacity < 0:
        monkeypatch.setenv("SLAM_SPLIT_CLASS_ROOM", str(-recv_capacity))
        room, recv_capacity = -recv_capacity, 0
    else:
        room = 0
    one = _run_c_session_ranks(1, n_total, L, frames, transport=None)[0]
    many = _run_c_session_ranks(world, n_total, L, frames, layout="split", recv_capacity=recv_capacity)
    assert all(set(p["layouts"]) == {"split"} for p in many)
    assert sum(sum(p["rows"]) for p in many) > 0, "nothing migrated"
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in many], axis=1)), bits(one["pose"]))
    assert np.array_equal(bits(np.concatenate([p["map"] for p in many], axis=0)), bits(one["map"]))
    for p in many:
        assert p["best"][2] == one["best"][2] and p["best"][1] == one["best"][1]
        assert np.array_equal(bits(p["best"][0]), bits(one["best"][0]))
    if n_total // world >= 3072 and L > 128:   # shapes the fused front takes: the update went out in two launches per frame
        assert all(p["fused"] >= frames - 3 for p in many), [p["fused"] for p in many]
    if room:   # more rows arrived than a list of class numbers hands out: new lists were made on the way
        assert max(sum(p["rows"]) for p in many) > room, [sum(p["rows"]) for p in many]


def test_sharded_auto_is_split_and_sharded_split_with_map_reads():
    """AUTO keeps the ranks of a sharded session on the split layout too; a map getter between two frames completes the
    exchange early, after which a move of the maps has to take the staging tail along (rows <-> split through the same
    conversions as one GPU)."""
    from test_gpu_configs import _run_c_session_ranks

    n_total, L, frames, world = 3072, 100, 14, 3
    ref = _run_c_session_ranks(1, n_total, L, frames, transport=None, maps_every_frame=True)[0]
    many = _run_c_session_ranks(world, n_total, L, frames, layout="auto", maps_every_frame=True)
    assert all(set(p["layouts"]) == {"split"} for p in many), [p["layouts"] for p in many]
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in many], axis=1)), bits(ref["pose"]))
    assert np.array_equal(bits(np.concatenate([p["map"] for p in many], axis=0)), bits(ref["map"]))
    for f in range(frames):
        got = np.concatenate([p["frame_maps"][f] for p in many], axis=0)
        assert np.array_equal(bits(got), bits(ref["frame_maps"][f])), f


@pytest.mark.parametrize("n,L,cov,sparse,skip", [
    (8192, 400, "own", 12, ()), (8192, 400, "shared", 12, (3, 4)), (5000, 513, "families", 40, (6,)), (3000, 1100, "families", 5, ()),
    (4097, 31, "own", 3, (2,)), (20000, 96, "families", 0, ()),
])
def test_split_pages_give_the_bits_of_rows(n, L, cov, sparse, skip):
    """SLAM_MAP_SPLIT_PAGES: the means on copy-on-write pages of two planes, the covariances per class.  Frames that observe
    a few neighbours (the list form of the paged update), a third or all landmarks (its page-wide form) and nothing (tables
    and classes follow their particles): the bits of a row session."""
    kw = dict(cov=cov, sparse=sparse, skip_obs=skip)
    pages = _run("split_pages", n, L, 11, **kw)
    assert set(pages["layout"]) == {"split_pages"}
    _same(pages, _run("rows", n, L, 11, **kw))


@pytest.mark.parametrize("ess", [0.3, 0.05])
def test_split_in_a_gated_session(ess):
    """ESS-gated resampling on the split layout: a frame that keeps its population updates through the identity index (the
    split update has no in-place form), the weights carry over; dense and sparse frames; the bits of a gated row session."""
    from test_gpu_auto_layout import _run as auto_run

    rows = auto_run("rows", 4096, 400, 30, ess=ess)
    split = auto_run("split", 4096, 400, 30, ess=ess)
    assert not any(split["paged"]) and split["changes"] == 0
    assert np.array_equal(bits(split["pose"]), bits(rows["pose"]))
    for f, m in rows["maps"].items():
        assert np.array_equal(bits(split["maps"][f]), bits(m)), f
    for a, b in zip(split["best"], rows["best"]):
        assert a[2] == b[2] and a[1] == b[1] and np.array_equal(bits(a[0]), bits(b[0]))


def test_sharded_split_in_a_gated_session():
    """... and sharded: 3 ranks on one card, gated, split layout, against one gated rank on rows."""
    from test_gpu_configs import _run_c_session_ranks

    n_total, L, frames = 3072, 100, 14
    ref = _run_c_session_ranks(1, n_total, L, frames, transport=None, ess=0.4)[0]
    many = _run_c_session_ranks(3, n_total, L, frames, layout="split", ess=0.4)
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in many], axis=1)), bits(ref["pose"]))
    assert np.array_equal(bits(np.concatenate([p["map"] for p in many], axis=0)), bits(ref["map"]))


def test_split_pages_in_a_gated_session():
    """An explicit SLAM_MAP_SPLIT_PAGES session with ESS-gated resampling (AUTO would keep a gated session on rows): frames that
    keep their population update through the identity index; the bits of a gated row session."""
    from test_gpu_auto_layout import _run as auto_run

    rows = auto_run("rows", 4096, 400, 30, ess=0.3)
    pages = auto_run("split_pages", 4096, 400, 30, ess=0.3)
    assert all(pages["paged"]) and pages["changes"] == 0
    assert np.array_equal(bits(pages["pose"]), bits(rows["pose"]))
    for f, m in rows["maps"].items():
        assert np.array_equal(bits(pages["maps"][f]), bits(m)), f
    for a, b in zip(pages["best"], rows["best"]):
        assert a[2] == b[2] and a[1] == b[1] and np.array_equal(bits(a[0]), bits(b[0]))


def test_split_pages_map_io():
    """set_map / map getters / reset on split pages."""
    import _shard_worker as W

    pkg = load_package()
    n, L = 3000, 200
    meta, edt, bx, by, lm = W.make_world(L=L)
    x, y, th, mp = _maps(n, L, lm, "families")
    eng = pkg.Engine(0)
    ses = pkg.PfSession(eng, n, L, seed=3, sigma=(0.02, 0.02, 0.004), meas_var=0.02, score_gain=0.05, map_layout="split_pages")
    assert ses.layout() == "split_pages" and ses.is_paged()
    fresh = ses.maps()
    assert np.all(fresh[:, 2] == -1.0) and np.all(fresh[:, [0, 1, 3, 4]] == 0.0)
    ses.set_map(mp)
    assert np.array_equal(bits(ses.maps()), bits(mp))
    sel = np.array([5, 5, 2999, 0, 17], np.int32)
    assert np.array_equal(bits(ses.map_rows(sel)), bits(mp[sel]))
    assert ses.split_view()["mean"] is None and ses.paged_view()["planes"] == 2   # the means are on pages of two planes
    ses.reset([0.0, 0.0, 0.0])
    assert np.array_equal(bits(ses.maps()), bits(fresh))
    ses.close()
    eng.close()


@pytest.mark.parametrize("world,n_total,L,frames", [(2, 4096, 70, 12), (3, 3072, 100, 14), (4, 8192, 130, 10), (2, 16384, 200, 8), (8, 8192, 40, 20)])
def test_sharded_split_pages_ranks_on_one_card_equal_one_rank_on_rows(world, n_total, L, frames, monkeypatch):
    """Sharded sessions on split pages: a migrating particle is packed from its mean pages and its class's covariance rows and
    unpacked onto fresh mean pages as a class of its own (class numbers recycled from short lists); frames that observe a few
    landmarks; map reads between the frames (they complete the exchange early); against one rank on rows, frame by frame.  The
    last case has ranks large enough for the fused front of sharded SPLIT sessions, which a session on split pages must not
    take (it works on mean rows)."""
    from test_gpu_configs import _run_c_session_ranks

    monkeypatch.setenv("SLAM_SPLIT_CLASS_ROOM", "64")
    kw = dict(sparse_obs=True, maps_every_frame=True)
    ref = _run_c_session_ranks(1, n_total, L, frames, transport=None, **kw)[0]
    many = _run_c_session_ranks(world, n_total, L, frames, layout="split_pages", **kw)
    assert all(set(p["layouts"]) == {"split_pages"} for p in many), [p["layouts"] for p in many]
    assert any(sum(p["rows"]) > 0 for p in many), "nothing migrated"
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in many], axis=1)), bits(ref["pose"]))
    assert np.array_equal(bits(np.concatenate([p["map"] for p in many], axis=0)), bits(ref["map"]))
    for f in range(frames):
        got = np.concatenate([p["frame_maps"][f] for p in many], axis=0)
        assert np.array_equal(bits(got), bits(ref["frame_maps"][f])), f


def test_auto_moves_between_split_and_pages_and_keeps_the_bits():
    """The AUTO run of tests/test_gpu_auto_layout.py, looked at from this side: dense frames on split, sparse ones on split
    pages (the means move onto pages, the classes stay)."""
    from test_gpu_auto_layout import _run as auto_run

    pkg = load_package()
    seen = []
    step = pkg.PfSession.step

    def spy(self, *a, **k):
        step(self, *a, **k)
        seen.append(self.layout())

    pkg.PfSession.step = spy
    try:
        auto = auto_run("auto", 4096, 400, 44)
    finally:
        pkg.PfSession.step = step
    assert seen[0] == "split" and "split_pages" in seen[:14] and "split" in seen[14:40], seen
    assert "rows" not in seen and "pages" not in seen
    rows = auto_run("rows", 4096, 400, 44)
    assert np.array_equal(bits(auto["pose"]), bits(rows["pose"]))
    for f, m in rows["maps"].items():
        assert np.array_equal(bits(auto["maps"][f]), bits(m)), f
