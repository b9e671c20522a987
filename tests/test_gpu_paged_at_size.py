"""Paged landmark maps (csrc/paged_kernels.hip) against the CPU specification at the sizes their numbers are quoted on:
65 536 x 500, 65 536 x 5 000 and 1 048 576 x 1 000 particles x landmarks with 32 landmarks observed per frame, and one
small population checked in full.  The comparison partner is oracle/slam_oracle_pf.c (orc_ekf_update) — NOT the row
session: PARITY UNPINNED all the same, the reference has no particles or landmarks.

Per frame, for >= 4 096 sampled slots (all slots of the small population): the ancestor is read from
slam_pf_device_view().anc BEFORE the step, its landmarks are assembled from the page pool through the page table
(slam_pf_paged_device_view; torch indexing, none of the engine's own page -> row kernels), the slot's landmarks after the
step likewise, and both the landmark values and the log-likelihood must equal the specification's bit for bit.
Whole-pool properties after every frame:
  * every page a current table names carries the frame's stamp;
  * every page handed out this frame is named by exactly one table (a page named by two tables was not written), and —
    where a snapshot of the pool fits — no other page changed;
  * the rest of the free list names no page a table names, and a free list made anew holds exactly the pages that did not
    carry the previous frame's stamp: named + free = pool.
Every run makes its free list anew at least once (asserted).

Round 4: the same on SPLIT PAGES (SLAM_MAP_SPLIT_PAGES: the means on pages of two planes in the session's two mean buffers,
the covariances in the rows of the particle's covariance class): the prior is put together from the mean pages through the
table and from the class row through the particle's class number, both read before the step; the pool properties are those
of the mean pages.
"""
import numpy as np
import pytest
import torch

from __graft_entry__ import load_package
from conftest import bits

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
MEAS_VAR = 0.0016


def _tensor(a, dtype=None):
    t = torch.as_tensor(a, device=DEV)
    return t if dtype is None else t.view(dtype)


def _rows_through_tables(pv, slots, sv=None):
    """[len(slots)][5][nb * page] landmarks of table rows `slots`, assembled with torch from pool and table (split pages:
    means from the two halves of the pool, covariances from the class rows of `sv` = slam_pf_split_device_view)."""
    table = _tensor(pv["table"])
    pages = table[slots.long()].long()                       # [S][nb]
    if pv["planes"] == 5:
        r = _tensor(pv["pool"])[pages]                       # [S][nb][5][page]
        return r.permute(0, 2, 1, 3).reshape(len(slots), 5, -1)
    H = pv["half_pages"]
    lo, hi = (_tensor(b) for b in pv["pool"])
    m = torch.where((pages < H)[:, :, None, None], lo[pages.clamp(max=H - 1)], hi[(pages - H).clamp(min=0)])   # [S][nb][2][page]
    m = m.permute(0, 2, 1, 3).reshape(len(slots), 2, -1)
    cov = _tensor(sv["cov"])[_tensor(sv["cls"])[slots.long()].long()]                                           # [S][3][Lp]
    return torch.cat([m, cov], dim=1)


def _observations(frame, L, K, rng):
    """Even frames: K neighbouring landmarks (the K-nearest pattern, 1-2 pages); odd frames: K landmarks scattered over
    the whole map (many pages: these frames exhaust the free list)."""
    if frame % 2 == 0:
        ids = (np.arange(K) + 97 * frame) % L
    else:
        ids = rng.permutation(L)[:K]
    ids = np.unique(ids).astype(np.int32)
    zx, zy = rng.normal(0, 2, len(ids)).astype(np.float32), rng.normal(0, 2, len(ids)).astype(np.float32)
    return ids, zx, zy


def _paged_frames(orc, n, L, K, frames, nsample, snapshot, layout="pages", family=1):
    import _shard_worker as W

    pkg = load_package()
    meta, edt, bx, by, _ = W.make_world(L=1)
    eng = pkg.Engine(0)
    eng.grid_set_dev(0, torch.from_numpy(edt).to(DEV), pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
    eng.scan_upload(bx, by)
    ses = pkg.PfSession(eng, n, L, seed=91, sigma=(0.02, 0.02, 0.004), meas_var=MEAS_VAR, score_gain=0.05, map_layout=layout)
    assert ses.is_paged() and ses.layout() == layout
    split = layout == "split_pages"
    rng = np.random.default_rng(n + L)
    ses.set_poses(*((s * rng.standard_normal(n)).astype(np.float32) for s in (0.3, 0.3, 0.05)))
    Lp = (L + 31) // 32 * 32
    g = torch.Generator(device=DEV).manual_seed(L)
    m0 = torch.full((n, 5, Lp), -555.0, device=DEV)
    for i0 in range(0, n, 32768):                                             # means N(0, 3), covariances A A^T + 0.02 I,
        i1 = min(i0 + 32768, n)                                               # every 10th landmark "not seen yet"
        m0[i0:i1, 0:2, :L] = 3.0 * torch.randn((i1 - i0, 2, L), device=DEV, generator=g)
        a = 0.3 * torch.randn(((i1 - i0 + family - 1) // family, 4, L), device=DEV, generator=g)
        a = a.repeat_interleave(family, dim=0)[:i1 - i0]    # `family` neighbouring particles share their covariances
        m0[i0:i1, 2, :L] = a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1] + 0.02
        m0[i0:i1, 3, :L] = a[:, 0] * a[:, 2] + a[:, 1] * a[:, 3]
        m0[i0:i1, 4, :L] = a[:, 2] * a[:, 2] + a[:, 3] * a[:, 3] + 0.02
        m0[i0:i1, 2, 3:L:10] = -1.0
        del a
    torch.cuda.synchronize()
    ses.set_map_dev(m0, 5 * Lp, Lp)
    eng.sync()
    del m0
    torch.cuda.empty_cache()

    pv = ses.paged_view()
    P, nb, page = pv["npages"], pv["pages_per_particle"], pv["page_landmarks"]
    assert page * nb == Lp and P == 2 * n * nb and pv["planes"] == (2 if split else 5)

    def pool_bits(pv):   # the whole pool as one [P][planes * page] int32 tensor (a copy)
        bufs = pv["pool"] if split else [pv["pool"]]
        return torch.cat([_tensor(b).reshape(-1, pv["planes"] * page).view(torch.int32) for b in bufs]).clone()
    renewals = 0
    for f in range(frames):
        s = np.arange(n) if nsample >= n else np.unique(np.concatenate(
            [rng.integers(0, n, nsample), [0, 1, n - 2, n - 1], np.arange(n // 2, n // 2 + 64)]))
        d_s = torch.from_numpy(s).to(DEV)
        v = ses.device_view()
        pv = ses.paged_view()
        sv = ses.split_view() if split else None
        assert sv is None or sv["mean"] is None
        src = d_s if v["anc"] is None else _tensor(v["anc"])[d_s.long()]      # the ancestors, read BEFORE the step
        prior = _rows_through_tables(pv, src, sv).cpu().numpy()
        stamp_before = pv["stamp_now"]
        live_before = int((_tensor(pv["stamp"]) == int(stamp_before)).sum())
        snap = pool_bits(pv) if snapshot else None
        ids, zx, zy = _observations(f, L, K, rng)
        eng.obs_upload(ids, zx, zy, L)
        ses.step(0, [0.01, -0.005, 0.002], True)
        eng.sync()

        # ---- the sampled slots against the specification
        v = ses.device_view()
        pv = ses.paged_view()
        sv = ses.split_view() if split else None
        pose = _tensor(v["pose"])[:, d_s.long()].cpu().numpy()
        got = _rows_through_tables(pv, d_s, sv).cpu().numpy()
        got_ll = _tensor(v["loglik"])[d_s.long()].cpu().numpy()
        want = np.full_like(prior, -777.0)
        want_ll = np.empty(len(s), np.float32)
        orc.lib().orc_ekf_update(np.ascontiguousarray(prior), want, 5 * Lp, Lp, L, pose[0].copy(), pose[1].copy(), pose[2].copy(),
                                 None, len(s), ids, zx, zy, len(ids), MEAS_VAR, want_ll)
        assert np.array_equal(bits(got[:, :, :L]), bits(want[:, :, :L])), f"frame {f}: landmark values"
        assert np.array_equal(bits(got_ll), bits(want_ll)), f"frame {f}: log-likelihoods"
        assert np.array_equal(bits(got[:, :, L:]), bits(prior[:, :, L:]))       # the tail of the last page travels unchanged

        # ---- whole-pool properties
        table = _tensor(pv["table"])[:n]
        stamp = _tensor(pv["stamp"])
        now = int(pv["stamp_now"])
        assert pv["stamp_now"] == stamp_before + 1
        assert bool((stamp[table.reshape(-1).long()] == now).all()), "a page named by a current table without the frame's stamp"
        counts = torch.bincount(table.reshape(-1), minlength=P)
        free, used, renewed, base = (int(x) for x in _tensor(pv["state"]).cpu().numpy())
        T = len(np.unique(ids // page))
        assert used - base == n * T
        fresh = _tensor(pv["freelist"])[base:used].long()
        assert bool((counts[fresh] == 1).all()), "a page written this frame is named by no table or by several"
        assert int((stamp == now).sum()) == int((counts > 0).sum())             # stamped = named, nothing else
        assert bool((counts[_tensor(pv["freelist"])[used:free].long()] == 0).all()), "the free list names a page in use"
        if renewed:
            renewals += 1
            assert free == P - live_before, "named + free != pool"
        if snap is not None:
            written = torch.zeros(P, dtype=torch.bool, device=DEV)
            written[fresh] = True
            assert bool(torch.equal(pool_bits(pv)[~written], snap[~written])), "a page not handed out this frame changed"
            del snap, written
        del counts, fresh
    assert renewals >= 1, "no frame made the free list anew"
    ses.close()
    eng.close()
    torch.cuda.empty_cache()


LAYOUTS = pytest.mark.parametrize("layout", ["pages", "split_pages"])


@LAYOUTS
def test_paged_small_population_every_particle_vs_specification(orc, layout):
    """2 048 x 200, 12 observed: every slot of every frame against the specification, pool snapshot compared (split pages:
    every particle starts in a covariance class of its own)."""
    _paged_frames(orc, 2048, 200, 12, frames=10, nsample=1 << 30, snapshot=True, layout=layout)


@LAYOUTS
def test_paged_64k_x_500_obs32(orc, layout):
    """BASELINE configs[1] with 32 observed landmarks (the size bench.py --observed 32 is quoted on)."""
    _paged_frames(orc, 65536, 500, 32, frames=6, nsample=4096, snapshot=True, layout=layout, family=16)


@LAYOUTS
def test_paged_64k_x_5000_obs32(orc, layout):
    """65 536 x 5 000 (157 pages per particle, a 20-million-page pool), 32 observed."""
    _paged_frames(orc, 65536, 5000, 32, frames=12, nsample=4096, snapshot=False, layout=layout, family=16)


@LAYOUTS
def test_paged_north_star_1m_x_1000_obs32(orc, layout):
    """The north-star size on pages: 1 048 576 x 1 000 (a 67-million-page pool, 43 GB; split pages: 17 GB of mean pages),
    32 observed."""
    _paged_frames(orc, 1048576, 1000, 32, frames=4, nsample=4096, snapshot=False, layout=layout, family=256)


def test_map_rows_of_chosen_particles_and_frame_outputs():
    """slam_pf_get_map_rows_host on both layouts = the same rows of slam_pf_get_map_host; slam_pf_device_view's score /
    logw / loglik: logw = loglik - gain * score in float32."""
    import _shard_worker as W

    pkg = load_package()
    n, L = 3000, 70
    meta, edt, bx, by, lm = W.make_world(L=L)
    x, y, th, mp = W.init_state(n, L, lm)
    sel = np.array([5, 5, 0, n - 1, 1234, 77], np.int32)
    for layout in ("rows", "pages"):
        eng = pkg.Engine(0)
        eng.grid_set_dev(0, torch.from_numpy(edt).to(DEV), pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
        eng.scan_upload(bx, by)
        ses = pkg.PfSession(eng, n, L, seed=3, sigma=(0.02, 0.02, 0.004), meas_var=0.02, score_gain=0.05, map_layout=layout)
        assert ses.is_paged() == (layout == "pages")
        ses.set_poses(x, y, th)
        ses.set_map(mp)
        assert ses.device_view()["logw"] is None
        assert np.array_equal(bits(ses.map_rows(sel)), bits(mp[sel]))
        for f in range(3):
            eng.obs_upload(*W.observations(lm, f), L)
            ses.step(0, [0.01, -0.005, 0.002], True)
        assert np.array_equal(bits(ses.map_rows(sel)), bits(ses.maps()[sel]))      # pending gather applied in both
        v = ses.device_view()
        score, logw, ll = (torch.as_tensor(v[k], device=DEV).cpu().numpy() for k in ("score", "logw", "loglik"))
        assert np.array_equal(bits(logw), bits(ll - score * np.float32(0.05)))
        ses.step(0, [0.01, -0.005, 0.002], False)
        assert ses.device_view()["loglik"] is None                                 # a frame without observations
        with pytest.raises(pkg.SlamError):
            ses.map_rows([n])
        ses.close()
        eng.close()


def test_one_session_per_engine():
    """The stages keep per-population state inside the engine (resample gate, carried weights, exchange plan): a second
    live session on the same engine is refused instead of silently sharing it; after the first is closed it is welcome."""
    pkg = load_package()
    eng = pkg.Engine(0)
    a = pkg.PfSession(eng, 512, 0, resample_ess_frac=0.5)
    with pytest.raises(pkg.SlamError) as err:
        pkg.PfSession(eng, 256, 0)
    assert err.value.status == -4 and "one session per engine" in str(err.value)
    a.close()
    b = pkg.PfSession(eng, 256, 0)
    b.close()
    eng.close()
