"""frame_front_kernel (csrc/pf_kernels.hip: motion sample + scan-match score + grouped landmark update in ONE launch — the
kernel every headline number of bench.py runs) against the CPU specification DIRECTLY, at the shapes those numbers are
quoted on: 65 536 x 500 (BASELINE configs[1], all landmarks observed and the 32 nearest), 1 048 576 x 1 000 (the north-star
workload) and 524 288 x 5 000 (configs[4]'s per-GPU share) — bench.py's own world (1024 x 1024 EDT, 360 beams, its room,
landmarks, frames and initial population).

The comparison partner is oracle/slam_oracle_pf.c, not another kernel.  Its scoring half restates the reference's FastMatch
inner loop (Subsystem_1/main.c:459-518: rotate, offset, roundf, bounds test, in-order float sum); the motion sample, the
landmark update and the log-likelihood have no reference counterpart: PARITY UNPINNED for those.

Both layouts that launch it are pinned: rows (20 B per particle and landmark) and split (means per particle, covariances per
covariance class: csrc/split_kernels.hip), the latter with a few thousand classes so that class hand-over, class rows and the
classes' own update (cov_update_kernel) are all in the comparison.

Per frame, on a session with fusion on: the pending gather index, the source poses and — for >= 4 096 sampled slots —
the ancestor's map row are read from slam_pf_device_view BEFORE the step; after it
  * the new pose of EVERY slot            == orc_motion_sample(source pose of its ancestor, slot id, frame),
  * score and in-bounds count (sampled)   == orc_score_poses_det on that pose,
  * landmark row and log-likelihood       == orc_ekf_update(ancestor's row, that pose, the frame's observations),
bit for bit; slam_frame_fusion_count must have advanced (every frame but the first, which has no gather index yet) and
slam_frame_front_last says which instantiation ran.
"""
import types

import numpy as np
import pytest
import torch

from __graft_entry__ import load_package
from conftest import bits

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
GRID, BEAMS, SEED = 1024, 360, 1234


def _tensor(a):
    return torch.as_tensor(a, device=DEV)


def _rows_of(ses, layout, slots):
    """[len(slots)][5][Lp] landmark rows of the CURRENT particles `slots` (before the pending gather), read from the session's
    own buffers with torch: the row buffer, or — split layout — the particle's means beside its class's covariance planes."""
    if layout == "rows":
        return _tensor(ses.device_view()["map"])[slots.long()]
    sv = ses.split_view()
    cls = _tensor(sv["cls"])[slots.long()].long()
    return torch.cat([_tensor(sv["mean"])[slots.long()], _tensor(sv["cov"])[cls]], dim=1)


def _front_frames(orc, n, L, observed, frames, nsample, nscore, expect, layout="rows"):
    import bench as B

    pkg = load_package()
    rng = np.random.default_rng(4321)
    landmarks = B.make_landmarks(L, rng)
    pixel, min_x, min_y = np.float32(20.48 / GRID), np.float32(-4.24), np.float32(-10.24)
    occ = B.occupancy(GRID, float(pixel), float(min_x), float(min_y))
    fr = B.make_frames(frames, BEAMS, landmarks, rng, observed)

    eng = pkg.Engine(0)
    d_edt = torch.empty((GRID, GRID), dtype=torch.float32, device=DEV)
    eng.edt_dev(torch.from_numpy(occ).to(DEV), GRID, GRID, GRID, 10.0, d_edt)
    eng.grid_set_dev(0, d_edt, pkg.grid_meta(GRID, GRID, GRID, pixel, min_x, min_y))
    eng.sync()
    edt = d_edt.cpu().numpy()                       # the EDT itself is pinned elsewhere (tests/test_gpu_scanmatch.py)
    ometa = orc.meta(GRID, GRID, GRID, float(pixel), float(min_x), float(min_y))

    ses = pkg.PfSession(eng, n, L, sigma=B.SIGMA, meas_var=B.MEAS_VAR, score_gain=B.SCORE_GAIN, seed=SEED, map_layout=layout)
    assert ses.layout() == layout
    Lp = (L + 31) // 32 * 32
    g = torch.Generator(device="cpu").manual_seed(SEED)
    p0 = B.true_pose(0)
    ses.set_poses(*[(p0[k] + s * torch.randn(n, generator=g)).numpy() for k, s in ((0, 0.05), (1, 0.05), (2, 0.01))])
    m0 = _tensor(ses.device_view()["map"]) if layout == "rows" else torch.empty((n, 5, Lp), dtype=torch.float32, device=DEV)
    torch.manual_seed(7)
    B.fill_maps(torch, m0, landmarks, L, DEV, n)
    # unlike bench.py: covariances that differ — from particle to particle on rows, from family to family (runs of 1 .. 64
    # neighbouring particles) on the split layout, whose classes they become — and some landmarks nobody has seen yet
    fam = None
    if layout != "rows":
        fam = torch.zeros(n, dtype=torch.long)
        i, frng = 0, np.random.default_rng(3)
        while i < n:
            k = int(frng.integers(1, 65))
            fam[i:i + k] = i
            i += k
        fam = fam.to(DEV)
    for i0 in range(0, n, 65536):
        i1 = min(i0 + 65536, n)
        a = 0.2 * torch.randn((i1 - i0, 4, L), device=DEV)
        m0[i0:i1, 2, :L] = a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1] + 0.02
        m0[i0:i1, 3, :L] = a[:, 0] * a[:, 2] + a[:, 1] * a[:, 3]
        m0[i0:i1, 4, :L] = a[:, 2] * a[:, 2] + a[:, 3] * a[:, 3] + 0.02
        m0[i0:i1, 2, 5:L:17] = -1.0
        del a
    if fam is not None:
        for i0 in range(0, n, 65536):                                          # heads precede their families: in order
            i1 = min(i0 + 65536, n)
            m0[i0:i1, 2:5] = m0[fam[i0:i1], 2:5]
        torch.cuda.synchronize()
        ses.set_map_dev(m0, 5 * Lp, Lp)
        eng.sync()
        assert int(_tensor(ses.split_view()["live_count"])[0]) == len(torch.unique(fam))
    torch.cuda.synchronize()
    del m0, fam
    torch.cuda.empty_cache()

    srng = np.random.default_rng(n + L)
    fused_before = eng.frame_fusion_count()
    kernels = set()
    for f in range(frames):
        s = np.unique(np.concatenate([srng.integers(0, n, nsample), [0, 1, 2, 3, n - 4, n - 3, n - 2, n - 1],
                                      np.arange(n // 2, n // 2 + 64)]))
        d_s = torch.from_numpy(s).to(DEV)
        v = ses.device_view()
        anc = None if v["anc"] is None else _tensor(v["anc"]).cpu().numpy()
        src_pose = _tensor(v["pose"]).cpu().numpy()
        src_rows = d_s if anc is None else torch.from_numpy(anc[s]).to(DEV)
        prior = _rows_of(ses, layout, src_rows).cpu().numpy()             # [S][5][Lp], BEFORE the step
        ids, zx, zy = fr[f]["ids"], fr[f]["zx"], fr[f]["zy"]
        eng.scan_upload(fr[f]["bx"], fr[f]["by"])
        eng.obs_upload(ids, zx, zy, L)
        ses.step(0, fr[f]["dp"], True)
        eng.sync()
        now = eng.frame_fusion_count()
        assert now - fused_before == (1 if f else 0), f"frame {f}: fused launches {now - fused_before}"
        fused_before = now
        if f:
            kernels.add(eng.frame_front_last())

        v = ses.device_view()
        pose = _tensor(v["pose"]).cpu().numpy()
        # ---- motion sample, every slot
        x, y, th = orc.motion_sample(src_pose[0], src_pose[1], src_pose[2], anc, n, 0, fr[f]["dp"], np.array(B.SIGMA, np.float32),
                                     SEED, f)
        assert np.array_equal(bits(pose), bits(np.stack([x, y, th]))), f"frame {f}: poses"
        # ---- scan-match score (main.c:459-518), a sample
        sc = s if nscore >= len(s) else np.unique(np.concatenate([s[:: max(len(s) // 1024, 1)], srng.integers(0, n, nscore)]))
        d_sc = torch.from_numpy(sc).to(DEV).long()
        want_score, want_count = orc.score_poses_det(ometa, edt, fr[f]["bx"], fr[f]["by"], x[sc], y[sc], th[sc])
        assert np.array_equal(_tensor(v["count"])[d_sc].cpu().numpy(), want_count), f"frame {f}: in-bounds counts"
        assert np.array_equal(bits(_tensor(v["score"])[d_sc].cpu().numpy()), bits(want_score)), f"frame {f}: scores"
        # ---- landmark update + log-likelihood, the sampled slots
        got = _rows_of(ses, layout, d_s).cpu().numpy()
        got_ll = _tensor(v["loglik"])[d_s.long()].cpu().numpy()
        want = np.full_like(prior, -777.0)
        want_ll = np.empty(len(s), np.float32)
        orc.lib().orc_ekf_update(np.ascontiguousarray(prior), want, 5 * Lp, Lp, L, x[s].copy(), y[s].copy(), th[s].copy(), None,
                                 len(s), np.ascontiguousarray(ids, np.int32), zx, zy, len(ids), B.MEAS_VAR, want_ll)
        assert np.array_equal(bits(got[:, :, :L]), bits(want[:, :, :L])), f"frame {f}: landmark values"
        assert np.array_equal(bits(got_ll), bits(want_ll)), f"frame {f}: log-likelihoods"
        # logw = loglik - gain * score, in float32 (what the resample consumes)
        lw = _tensor(v["logw"])[d_sc].cpu().numpy()
        ll_sc = _tensor(v["loglik"])[d_sc].cpu().numpy()
        assert np.array_equal(bits(lw), bits(ll_sc - want_score * np.float32(B.SCORE_GAIN))), f"frame {f}: log-weights"
    ses.close()
    eng.close()
    torch.cuda.empty_cache()
    assert kernels <= set(expect), f"fused instantiations that ran: {kernels}, expected among {expect}"
    return kernels


LAYOUTS = ["rows", "split"]   # split: what bench.py's default (SLAM_MAP_AUTO) runs since round 4
# (particles per updating wavefront, lanes per pose) of the fused launches: rows 4 once the resample stage has reported few
# distinct ancestors (2 before, and for large frames), split 8 (4 before)
EXPECT_64K = {"rows": [(4, 4), (2, 4)], "split": [(8, 4), (4, 4)]}
EXPECT_BIG = {"rows": [(2, 1)], "split": [(8, 1), (4, 1)]}


@pytest.mark.parametrize("layout", LAYOUTS)
def test_front_64k_x_500_all_observed(orc, layout):
    """BASELINE configs[1]: the shape of bench.py's headline (frame_front_kernel<2, 4, 4, 8> once the resample stage has
    reported few distinct ancestors, <2, 2, 4, 8> before that)."""
    ran = _front_frames(orc, 65536, 500, 0, frames=5, nsample=4096, nscore=1 << 30, expect=EXPECT_64K[layout], layout=layout)
    assert EXPECT_64K[layout][0] in ran, ran


@pytest.mark.parametrize("layout", LAYOUTS)
def test_front_64k_x_500_obs32(orc, layout):
    """configs[1] with the 32 nearest landmarks observed (bench.py --observed 32 --map-layout rows / split)."""
    _front_frames(orc, 65536, 500, 32, frames=4, nsample=4096, nscore=1 << 30, expect=EXPECT_64K[layout], layout=layout)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_front_north_star_1m_x_1000(orc, layout):
    """The north-star workload, 1 048 576 x 1 000: frame_front_kernel<2, 2, 1, 16> (one lane per pose, 2 particles per
    updating wavefront)."""
    ran = _front_frames(orc, 1048576, 1000, 0, frames=4, nsample=4096, nscore=32768, expect=EXPECT_BIG[layout], layout=layout)
    assert ran and ran <= set(EXPECT_BIG[layout]), ran


@pytest.mark.parametrize("layout", LAYOUTS)
def test_front_512k_x_5000(orc, layout):
    """configs[4]'s per-GPU share, 524 288 x 5 000 (105 GB of rows; 74 GB split)."""
    ran = _front_frames(orc, 524288, 5000, 0, frames=3, nsample=4096, nscore=32768, expect=EXPECT_BIG[layout], layout=layout)
    assert ran and ran <= set(EXPECT_BIG[layout]), ran
