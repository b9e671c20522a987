"""The front of a single-GPU frame on rows as ONE launch (frame_front_kernel: motion sample + scan-match score and the grouped
out-of-place landmark update, interleaved workgroups) against the two launches it replaces: the same bits — poses, scores'
consequences (weights -> resample indices -> heaviest particle), maps — frame after frame, for both lane mappings of the
scorer (4 lanes per pose below 131 072 particles, 1 above) and both group sizes of the update.  HIP against HIP, at small sizes:
the fused kernel itself is pinned against the CPU specification in tests/test_gpu_frame_front_at_size.py, at the shapes its
numbers are quoted on (the session-vs-oracle tests of test_gpu_pf.py use populations and rows too small for the fused launch,
and the full-size tests of test_gpu_configs.py call the stage entry point, which never launches it)."""
import numpy as np
import pytest
import torch

from __graft_entry__ import load_package
from conftest import bits

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _run(fused, n, L, frames, form=-1, nbeams=None, random_cov=False):
    import _shard_worker as W

    pkg = load_package()
    meta, edt, bx, by, lm = W.make_world(L=L)
    x, y, th, mp = W.init_state(n, L, lm)
    if random_cov:   # every particle its own covariances (symmetric positive definite) and its own unseen landmarks
        rng0 = np.random.default_rng(17)
        a = (0.02 + 0.2 * rng0.random((n, L))).astype(np.float32)
        c = (0.02 + 0.2 * rng0.random((n, L))).astype(np.float32)
        mp[:, 2], mp[:, 4] = a, c
        mp[:, 3] = ((rng0.random((n, L)) - 0.5) * np.sqrt(a * c)).astype(np.float32)
        mp[:, 2][rng0.random((n, L)) < 0.05] = -1.0
    eng = pkg.Engine(0)
    eng.frame_fusion_set(fused)
    eng.ekf_form_set(form)
    eng.grid_set_dev(0, torch.from_numpy(edt).to(DEV), pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
    eng.scan_upload(bx[:nbeams], by[:nbeams]) if nbeams else eng.scan_upload(bx, by)
    ses = pkg.PfSession(eng, n, L, seed=91, sigma=(0.02, 0.02, 0.004), meas_var=0.02, score_gain=0.05, map_layout="rows")
    ses.set_poses(x, y, th)
    ses.set_map(mp)
    del mp
    out = {"best": [], "maps": {}}
    rng = np.random.default_rng(5)
    for f in range(frames):
        ids = np.sort(rng.choice(L, size=(L if f % 3 else max(L // 3, 1)), replace=False)).astype(np.int32)   # all, or a third
        z = lm[ids] + np.float32(0.01) * np.float32(f % 5)
        eng.obs_upload(ids, z[:, 0].copy(), z[:, 1].copy(), L)
        ses.step(0, [0.01, -0.005, 0.002], True)
        out["best"].append(ses.best())
        if f in (1, frames - 1):
            sel = np.unique(np.concatenate([np.arange(0, n, max(n // 257, 1)), [n - 1]])).astype(np.int32)
            out["maps"][f] = ses.map_rows(sel) if hasattr(ses, "map_rows") else ses.maps()[sel]
    out["pose"] = ses.poses()
    out["fused_launches"] = eng.frame_fusion_count()
    out["forms"] = eng.ekf_form_counts()
    ses.close()
    eng.close()
    return out


@pytest.mark.parametrize("n,L,form,nbeams", [(16384, 300, -1, None), (16384, 300, 2, None), (16384, 300, 1, None), (5000, 513, -1, None),
                                              (140000, 200, -1, None), (3073, 130, -1, 37), (4099, 129, 1, 1)])
def test_fused_front_gives_the_bits_of_the_two_launches(n, L, form, nbeams):
    frames = 7
    two = _run(False, n, L, frames, form, nbeams)
    one = _run(True, n, L, frames, form, nbeams)
    assert two["fused_launches"] == 0
    assert one["fused_launches"] == frames - 1, one["fused_launches"]   # the first frame has no resample indices yet
    assert np.array_equal(bits(one["pose"]), bits(two["pose"]))
    for f, m in two["maps"].items():
        assert np.array_equal(bits(one["maps"][f]), bits(m)), f
    for a, b in zip(one["best"], two["best"]):
        assert a[2] == b[2] and a[1] == b[1] and np.array_equal(bits(a[0]), bits(b[0]))


def test_shapes_the_fused_front_does_not_take():
    """Short rows (<= 128 landmarks), few particles (the one-wavefront-per-pose scorer) and the one-wavefront-per-particle
    update (ekf form 0) keep the two launches."""
    assert _run(True, 4096, 100, 4)["fused_launches"] == 0
    assert _run(True, 2048, 300, 4)["fused_launches"] == 0
    assert _run(True, 16384, 300, 4, form=0)["fused_launches"] == 0


@pytest.mark.parametrize("n,L", [(16384, 300), (6000, 700)])
def test_hoisted_grouped_update_equals_the_update_of_one_particle_at_a_time(n, L):
    """The grouped kernels work out the covariance part of the update once per source row (ekf_prepare); the kernel with one
    wavefront per particle (ekf form 0) does everything per particle.  Same bits, with covariances and unseen landmarks that
    differ from particle to particle."""
    frames = 6
    per_particle = _run(False, n, L, frames, form=0, random_cov=True)
    for fused, form in ((True, -1), (False, 1), (False, 2)):
        grouped = _run(fused, n, L, frames, form=form, random_cov=True)
        assert np.array_equal(bits(grouped["pose"]), bits(per_particle["pose"]))
        for f, m in per_particle["maps"].items():
            assert np.array_equal(bits(grouped["maps"][f]), bits(m)), (fused, form, f)
        for a, b in zip(grouped["best"], per_particle["best"]):
            assert a[2] == b[2] and a[1] == b[1] and np.array_equal(bits(a[0]), bits(b[0]))
