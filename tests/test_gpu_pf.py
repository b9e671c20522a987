"""GPU parity of the particle-filter stages against the CPU specification.

PARITY UNPINNED with respect to the reference (it has no such stages): the
expected values come from oracle/slam_oracle_pf.c, this build's own specification, whose
known-answer tests are in tests/test_oracle_pf.py.  The bar is still bit-exactness — every stage
uses only correctly-rounded float32 operations, integer arithmetic and the specified det_ functions —
and the resample indices in particular are exact for any sharding.
"""
import numpy as np
import pytest
import torch

from __graft_entry__ import load_package
from conftest import bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def eng():
    pkg = load_package()
    e = pkg.Engine(0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    yield e
    torch.cuda.synchronize()
    e.close()


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def test_ieee_division_and_sqrt_are_correctly_rounded(eng, orc):
    """The EKF relies on hipcc's default correctly-rounded fp32 divide/sqrt; exercise them through the
    EKF (1/det) and the motion sample (sqrt) on awkward operands by comparing with the CPU."""
    n = 1 << 16
    rng = np.random.default_rng(1)
    z = np.zeros(n, np.float32)
    x, y, th = (torch.empty(n, device=DEV) for _ in range(3))
    sig = [1e-3, 7.7, 123.456]
    eng.motion_sample_dev((dev(z), dev(z), dev(z)), None, (x, y, th), n, 12345678901, [0.1, 0.2, 0.3], sig, 0xABCDEF0123, 9)
    wx, wy, wt = orc.motion_sample(z, z, z, None, n, 12345678901, [0.1, 0.2, 0.3], sig, 0xABCDEF0123, 9)
    assert np.array_equal(bits(host(x)), bits(wx)) and np.array_equal(bits(host(y)), bits(wy))
    assert np.array_equal(bits(host(th)), bits(wt))


@pytest.mark.parametrize("n,first_id,with_anc", [(1000, 0, False), (65536, 1 << 33, True), (1, 5, False), (777, 64, True)])
def test_motion_sample(eng, orc, n, first_id, with_anc):
    rng = np.random.default_rng(n)
    m = n + 13
    sx, sy, st = (rng.standard_normal(m).astype(np.float32) for _ in range(3))
    anc = np.sort(rng.integers(0, m, n)).astype(np.int32) if with_anc else None
    dp, sig = [0.004, -0.001, 0.0006], [0.05, 0.05, 0.01]
    x, y, th = (torch.empty(n, device=DEV) for _ in range(3))
    eng.motion_sample_dev((dev(sx), dev(sy), dev(st)), dev(anc) if with_anc else None, (x, y, th), n, first_id, dp, sig, 4242, 3)
    wx, wy, wt = orc.motion_sample(sx, sy, st, anc, n, first_id, dp, sig, 4242, 3)
    assert np.array_equal(bits(host(x)), bits(wx))
    assert np.array_equal(bits(host(y)), bits(wy))
    assert np.array_equal(bits(host(th)), bits(wt))


def _rand_map(rng, L, rows, Lp=None):
    """Random maps in the engine's layout, one row per particle: [rows][5][Lp]; columns >= L are padding (-555)."""
    Lp = L if Lp is None else Lp
    mp = np.full((rows, 5, Lp), -555.0, np.float32)
    mp[:, 0:2, :L] = rng.normal(0, 3, (rows, 2, L))
    A = rng.normal(0, 0.3, (rows, L, 2, 2))
    P = A @ np.swapaxes(A, -1, -2) + 0.02 * np.eye(2)
    mp[:, 2, :L], mp[:, 3, :L], mp[:, 4, :L] = P[..., 0, 0], P[..., 0, 1], P[..., 1, 1]
    mp[:, 2, rng.integers(0, L, max(L // 10, 1))] = -1.0   # a few landmarks not seen yet
    return mp


@pytest.mark.parametrize("n,L,Lp,nobs,with_anc", [
    (5000, 40, 64, 40, False),       # everything observed, one partly filled batch
    (5000, 40, 43, 7, True),         # subset observed + fused gather: copy-through of the other 33; rows not 128-B aligned
    (300, 500, 512, 500, True),      # BASELINE config 2 landmark count: 4 batches of 128, the last one partial
    (300, 700, 704, 641, True),      # accumulators reused across batches, partial last batch, 59 copied through
    (1, 3, 3, 3, False), (4097, 33, 64, 33, True), (256, 10, 32, 0, True), (130, 128, 128, 128, False),
    (77, 129, 160, 129, True), (64, 300, 320, 1, True), (50, 257, 257, 200, True), (9, 1, 1, 1, False),
])
@pytest.mark.parametrize("form", [0, 1, 2])
def test_ekf_update(eng, orc, n, L, Lp, nobs, with_anc, form):
    import ctypes as C
    eng.ekf_form_set(form)   # every out-of-place kernel (one wavefront per particle / per 4 / per 2 particles) gives these bits
    rng = np.random.default_rng(n * 31 + L)
    rows = n + 37 if with_anc else n
    mp = _rand_map(rng, L, rows, Lp)
    x, y, th = (rng.normal(0, 1, n).astype(np.float32) for _ in range(3))
    anc = np.sort(rng.integers(0, rows, n)).astype(np.int32) if with_anc else None
    ids = rng.permutation(L)[:nobs].astype(np.int32)
    zx, zy = rng.normal(0, 2, nobs).astype(np.float32), rng.normal(0, 2, nobs).astype(np.float32)
    d_in = dev(mp)
    d_out = torch.full((rows, 5, Lp), -777.0, device=DEV)
    ll = torch.empty(n, device=DEV)
    eng.obs_upload(ids, zx, zy, L)
    eng.ekf_update_dev(d_in, d_out, 5 * Lp, Lp, L, dev(x), dev(y), dev(th), dev(anc) if with_anc else None, n, 0.015, ll)
    want = np.full((rows, 5, Lp), -777.0, np.float32)
    wl = np.empty(n, np.float32)
    orc.lib().orc_ekf_update(mp, want, 5 * Lp, Lp, L, x, y, th, anc.ctypes.data_as(C.c_void_p) if with_anc else None, n,
                             ids, zx, zy, nobs, 0.015, wl)
    got = host(d_out)
    assert np.array_equal(bits(got[:, :, :L]), bits(want[:, :, :L]))
    assert np.all(got[n:] == -777.0)                      # nothing written to rows >= n
    pad = got[:n, :, L:]                                  # row padding: left alone, or carried over from the source row
    assert np.all((pad == -777.0) | (pad == -555.0))
    assert np.array_equal(bits(host(ll)), bits(wl))
    eng.ekf_form_set(-1)


def test_ekf_update_randomised_shapes(eng, orc):
    """40 seeded random cases of the EKF update against the CPU specification: particle counts around the workgroup
    size, landmark counts around the 128-landmark batches, plane strides from tight to generous (so that the
    unpredicated, the tail and the mixed paths all occur), observation fractions from none to all, unseen
    landmarks, with and without the fused gather, in place and out of place."""
    import ctypes as C
    rng = np.random.default_rng(20261004)
    for case in range(40):
        n = int(rng.choice([1, 3, 4, 5, 63, 64, 65, 255, 257, 1000]))
        L = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513]))
        Lp = L + int(rng.choice([0, 1, 5, 31, 127, 200]))
        in_place = bool(rng.random() < 0.25)
        with_anc = (not in_place) and bool(rng.random() < 0.6)
        eng.ekf_form_set(case % 3)
        eng.ekf_inplace_form_set(case // 3 % 2)   # in place: whole rows / the observed landmarks from the compact list
        rows = n + (int(rng.integers(0, 50)) if with_anc else 0)
        mp = _rand_map(rng, L, rows, Lp)
        x, y, th = (rng.normal(0, 1, n).astype(np.float32) for _ in range(3))
        anc = np.sort(rng.integers(0, rows, n)).astype(np.int32) if with_anc else None
        nobs = int(rng.choice([0, 1, L // 3, L]))
        ids = rng.permutation(L)[:nobs].astype(np.int32)
        zx, zy = rng.normal(0, 2, nobs).astype(np.float32), rng.normal(0, 2, nobs).astype(np.float32)
        eng.obs_upload(ids, zx, zy, L)
        d_in = dev(mp)
        d_out = d_in if in_place else torch.full((rows, 5, Lp), -777.0, device=DEV)
        ll = torch.empty(n, device=DEV)
        eng.ekf_update_dev(d_in, d_out, 5 * Lp, Lp, L, dev(x), dev(y), dev(th), dev(anc) if with_anc else None, n, 0.015, ll)
        want = mp.copy() if in_place else np.full((rows, 5, Lp), -777.0, np.float32)
        wl = np.empty(n, np.float32)
        src = want if in_place else mp
        orc.lib().orc_ekf_update(src, want, 5 * Lp, Lp, L, x, y, th, anc.ctypes.data_as(C.c_void_p) if with_anc else None, n,
                                 ids, zx, zy, nobs, 0.015, wl)
        got = host(d_out)
        tag = f"case {case}: n={n} L={L} Lp={Lp} nobs={nobs} anc={with_anc} in_place={in_place}"
        assert np.array_equal(bits(got[:n, :, :L]), bits(want[:n, :, :L])), tag
        assert np.array_equal(bits(host(ll)), bits(wl)), tag
        if not in_place:
            assert np.all(got[n:] == -777.0), tag
        else:
            assert np.array_equal(bits(got[:, :, L:]), bits(mp[:, :, L:])), tag    # in place never touches the padding
    eng.ekf_form_set(-1)
    eng.ekf_inplace_form_set(-1)


@pytest.mark.parametrize("form", [0, 1, -1])
def test_ekf_in_place_forms(eng, orc, form):
    """The two in-place kernels (whole rows in batches of 128 landmarks; the observed landmarks only, from the compact
    list) against the CPU specification: landmark counts up to configs[4]'s 5000, observation counts from none to all —
    with several observations per log-likelihood accumulator (l mod 128), more than 128 observations (several passes of
    the list kernel), unseen landmarks, observations given as a host list and as a device table.  form -1: whatever
    the engine picks from the previous list's count must give the same bits."""
    import ctypes as C
    rng = np.random.default_rng(77 + form)
    eng.ekf_inplace_form_set(form)
    before = eng.ekf_inplace_form_counts()
    ncases = 0
    for L, nobs_list in [(1, [0, 1]), (127, [5, 127]), (500, [0, 1, 32, 125, 300, 500]), (1000, [32, 129, 1000]), (5000, [32, 700, 5000])]:
        for nobs in nobs_list:
            for n in (3, 260):
                Lp = L + int(rng.choice([0, 12]))
                mp = _rand_map(rng, L, n, Lp)
                x, y, th = (rng.normal(0, 1, n).astype(np.float32) for _ in range(3))
                if rng.random() < 0.5:   # clustered: landmarks of ONE accumulator (l mod 128), as many as there are
                    ids = int(rng.integers(0, 128)) + 128 * rng.permutation((L + 127) // 128)[:nobs]
                    ids = ids[ids < L]
                else:
                    ids = rng.permutation(L)[:nobs]
                ids = ids.astype(np.int32)
                rng.shuffle(ids)
                k = len(ids)
                zx, zy = rng.normal(0, 2, k).astype(np.float32), rng.normal(0, 2, k).astype(np.float32)
                if ncases % 2:
                    eng.obs_upload(ids, zx, zy, L)
                    tabs = None
                else:
                    tx, ty = np.full(L, np.nan, np.float32), np.full(L, np.nan, np.float32)
                    tx[ids], ty[ids] = zx, zy
                    tabs = (dev(tx), dev(ty))
                    eng.obs_set_dev(tabs[0], tabs[1], L)
                d = dev(mp)
                ll = torch.empty(n, device=DEV)
                eng.ekf_update_dev(d, d, 5 * Lp, Lp, L, dev(x), dev(y), dev(th), None, n, 0.015, ll)
                want, wl = mp.copy(), np.empty(n, np.float32)
                orc.lib().orc_ekf_update(want, want, 5 * Lp, Lp, L, x, y, th, None, n, ids, zx, zy, k, 0.015, wl)
                tag = f"form {form}: n={n} L={L} Lp={Lp} nobs={k}"
                assert np.array_equal(bits(host(d)), bits(want)), tag   # padding included: never touched
                assert np.array_equal(bits(host(ll)), bits(wl)), tag
                ncases += 1
    after = eng.ekf_inplace_form_counts()
    eng.ekf_inplace_form_set(-1)
    d0, d1 = after[0] - before[0], after[1] - before[1]
    assert d0 + d1 == ncases
    if form == 0:
        assert d1 == 0
    elif form == 1:
        assert d0 == 0
    else:
        assert d0 > 0 and d1 > 0   # the feedback chose each of them at some point


@pytest.mark.parametrize("form", [0, 1])
def test_ekf_in_place_sees_a_rewritten_device_table(eng, orc, form):
    """slam_obs_set_dev adopts the caller's arrays: when the caller rewrites them between two launches (same pointers,
    no second slam_obs_set_dev), the next update must use the new contents — the compact list is never reused for tables
    the engine does not own."""
    rng = np.random.default_rng(31 + form)
    n, L = 300, 200
    mp = _rand_map(rng, L, n)
    x, y, th = (rng.normal(0, 1, n).astype(np.float32) for _ in range(3))
    tx, ty = torch.full((L,), float("nan"), device=DEV), torch.full((L,), float("nan"), device=DEV)
    eng.ekf_inplace_form_set(form)
    eng.obs_set_dev(tx, ty, L)
    d = dev(mp)
    ll = torch.empty(n, device=DEV)
    want = mp.copy()
    for rnd in range(3):
        ids = np.sort(rng.permutation(L)[:7 + 5 * rnd]).astype(np.int32)
        zx, zy = rng.normal(0, 2, len(ids)).astype(np.float32), rng.normal(0, 2, len(ids)).astype(np.float32)
        tx.fill_(float("nan"))
        ty.fill_(float("nan"))
        tx[torch.from_numpy(ids.astype(np.int64)).to(DEV)] = dev(zx)
        ty[torch.from_numpy(ids.astype(np.int64)).to(DEV)] = dev(zy)
        torch.cuda.synchronize()
        eng.ekf_update_dev(d, d, 5 * L, L, L, dev(x), dev(y), dev(th), None, n, 0.02, ll)
        want, wl = orc.ekf_update(want, x, y, th, None, ids, zx, zy, 0.02)
        assert np.array_equal(bits(host(d)), bits(want)), rnd
        assert np.array_equal(bits(host(ll)), bits(wl)), rnd
    eng.ekf_inplace_form_set(-1)


def test_ekf_in_place_and_argument_checks(eng, orc):
    pkg = load_package()
    rng = np.random.default_rng(2)
    n, L = 1000, 12
    mp = _rand_map(rng, L, n)
    x, y, th = (rng.normal(0, 1, n).astype(np.float32) for _ in range(3))
    ids = np.arange(L, dtype=np.int32)
    zx, zy = rng.normal(0, 2, L).astype(np.float32), rng.normal(0, 2, L).astype(np.float32)
    d = dev(mp)
    ll = torch.empty(n, device=DEV)
    eng.obs_upload(ids, zx, zy, L)
    eng.ekf_update_dev(d, d, 5 * L, L, L, dev(x), dev(y), dev(th), None, n, 0.02, ll)   # in place, no gather
    want, wl = orc.ekf_update(mp, x, y, th, None, ids, zx, zy, 0.02)
    assert np.array_equal(bits(host(d)), bits(want)) and np.array_equal(bits(host(ll)), bits(wl))
    # in place with only some landmarks observed: the others stay as they are
    eng.obs_upload(ids[:5], zx[:5], zy[:5], L)
    before = host(d).copy()
    eng.ekf_update_dev(d, d, 5 * L, L, L, dev(x), dev(y), dev(th), None, n, 0.02, ll)
    want2, wl2 = orc.ekf_update(before, x, y, th, None, ids[:5], zx[:5], zy[:5], 0.02)
    assert np.array_equal(bits(host(d)), bits(want2)) and np.array_equal(bits(host(ll)), bits(wl2))
    eng.obs_upload(ids, zx, zy, L)
    with pytest.raises(pkg.SlamError):   # gather in place is a race: rejected
        eng.ekf_update_dev(d, d, 5 * L, L, L, dev(x), dev(y), dev(th), dev(np.zeros(n, np.int32)), n, 0.02, ll)
    with pytest.raises(pkg.SlamError):   # planes overlapping: plane_stride < nlandmarks
        eng.ekf_update_dev(d, d, 5 * L, L - 2, L, dev(x), dev(y), dev(th), None, n, 0.02, ll)
    with pytest.raises(pkg.SlamError):   # rows overlapping: row_stride < 5 * plane_stride
        eng.ekf_update_dev(d, d, 5 * L - 2, L, L, dev(x), dev(y), dev(th), None, n, 0.02, ll)
    with pytest.raises(pkg.SlamError):   # a NaN measurement in the list
        eng.obs_upload(ids[:2], np.array([np.nan, 1.0], np.float32), zy[:2], L)
    with pytest.raises(pkg.SlamError):   # duplicate landmark ids
        eng.obs_upload(np.array([1, 1], np.int32), zx[:2], zy[:2], L)
    with pytest.raises(pkg.SlamError):   # id out of range
        eng.obs_upload(np.array([L], np.int32), zx[:1], zy[:1], L)
    with pytest.raises(pkg.SlamError) as ei:   # observation list made for another landmark count
        eng.ekf_update_dev(d, d, 5 * (L + 2), L + 2, L + 1, dev(x), dev(y), dev(th), None, n, 0.02, ll)
    assert ei.value.status == -4


@pytest.mark.parametrize("n", [1, 63, 64, 65, 2048, 2049, 100_000, 1_048_576 + 3])
def test_weights_and_prefix_sum(eng, orc, n):
    rng = np.random.default_rng(n)
    score = rng.uniform(0, 400, n).astype(np.float32)
    ll = rng.normal(-50, 20, n).astype(np.float32)
    logw, d_max = torch.empty(n, device=DEV), torch.empty(1, device=DEV)
    eng.logweight_dev(dev(score), dev(ll), 0.37, n, logw, d_max)
    wl, wm = orc.logweight(score, ll, 0.37)
    assert np.array_equal(bits(host(logw)), bits(wl)) and host(d_max)[0] == wm
    wq, d_sum = torch.empty(n, dtype=torch.int64, device=DEV), torch.empty(1, dtype=torch.int64, device=DEV)
    eng.quantise_weights_dev(logw, d_max, n, wq, d_sum)
    q, s = orc.quantise_weights(wl, wm)
    assert np.array_equal(host(wq).view(np.uint64), q) and int(host(d_sum)[0]) == s
    assert q.max() == 1 << 32
    cdf = torch.empty(n, dtype=torch.int64, device=DEV)
    eng.prefix_sum_dev(wq, n, cdf)
    assert np.array_equal(host(cdf).view(np.uint64), np.cumsum(q, dtype=np.uint64))
    # score-only and loglik-only forms
    eng.logweight_dev(dev(score), None, 0.37, n, logw, d_max)
    assert np.array_equal(bits(host(logw)), bits(orc.logweight(score, None, 0.37)[0]))
    eng.logweight_dev(None, dev(ll), 0.0, n, logw, d_max)
    assert np.array_equal(bits(host(logw)), bits(ll))


@pytest.mark.parametrize("n,single_gpu", [(1, True), (2047, True), (2048, False), (2049, True), (65536, True), (1_000_003, False)])
def test_fused_quantise_scan_offsets(eng, orc, n, single_gpu):
    """Frame-loop form (weights never stored): first[] must equal the staged specification."""
    rng = np.random.default_rng(n + 5)
    score = rng.uniform(0, 300, n).astype(np.float32)
    ll = rng.normal(-40, 15, n).astype(np.float32)
    seed, frame = 0xFEEDFACE12345, 11
    wl, wm = orc.logweight(score, ll, 0.21)
    q, ssum = orc.quantise_weights(wl, wm)
    cdf = orc.prefix_sum(q)
    logw = torch.empty(n, device=DEV)
    first = torch.empty(n, dtype=torch.int32, device=DEV)
    if single_gpu:   # no maximum, no totals on the device: everything derived inside the kernels
        eng.logweight_dev(dev(score), dev(ll), 0.21, n, logw, None)
        eng.quantise_scan_dev(logw, None, n, None)
        eng.offspring_from_scan_dev(n, None, None, seed, frame, n, first)
        want = orc.offspring_offsets(cdf, 0, ssum, orc.comb_offset(seed, frame, ssum), n)
    else:            # this shard is the middle one of three: base and grand total come from "other GPUs"
        d_max, d_sum = torch.empty(1, device=DEV), torch.empty(1, dtype=torch.int64, device=DEV)
        eng.logweight_dev(dev(score), dev(ll), 0.21, n, logw, d_max)
        eng.quantise_scan_dev(logw, d_max, n, d_sum)
        assert int(host(d_sum)[0]) == ssum and host(d_max)[0] == wm
        base, total, n_total = 7 * ssum // 5, 4 * ssum, 3 * n
        eng.offspring_from_scan_dev(n, dev(np.array([base], np.int64)), dev(np.array([total], np.int64)), seed, frame, n_total, first)
        want = orc.offspring_offsets(cdf, base, total, orc.comb_offset(seed, frame, total), n_total)
    assert np.array_equal(bits(host(logw)), bits(wl))
    assert np.array_equal(host(first), want)


def test_prefix_sum_full_width_values(eng):
    """64-bit carries across lanes, waves and tiles."""
    n = 300_001
    rng = np.random.default_rng(0)
    v = rng.integers(0, 1 << 40, n, dtype=np.uint64)
    cdf = torch.empty(n, dtype=torch.int64, device=DEV)
    eng.prefix_sum_dev(dev(v.view(np.int64)), n, cdf)
    assert np.array_equal(host(cdf).view(np.uint64), np.cumsum(v, dtype=np.uint64))


@pytest.mark.parametrize("n,shards,frame", [(4096, 1, 0), (100_000, 1, 3), (65536, 4, 1), (1_000_003, 1, 2), (30_000, 3, 5)])
def test_resample_indices_bit_exact(eng, orc, n, shards, frame):
    """Offspring offsets + ancestors against the oracle, unsharded and with the population cut into
    shards that only know their local CDF, the base offset and the grand total."""
    rng = np.random.default_rng(n + shards)
    w = (rng.random(n) ** 6 * 2**32).astype(np.uint64)
    w[rng.integers(0, n, n // 10)] = 0
    seed = 0x1234567887654321
    cdf_ref = orc.prefix_sum(w)
    total = int(cdf_ref[-1])
    want_first = orc.offspring_offsets(cdf_ref, 0, total, orc.comb_offset(seed, frame, total), n)
    want_anc = orc.ancestors(want_first, 0, n)
    cuts = [0] + sorted(rng.choice(np.arange(1, n), shards - 1, replace=False).tolist()) + [n] if shards > 1 else [0, n]
    first_all = torch.empty(n, dtype=torch.int32, device=DEV)
    d_total = dev(np.array([total], np.int64))
    for a, b in zip(cuts[:-1], cuts[1:]):
        loc = dev(w[a:b].view(np.int64))
        cdf = torch.empty(b - a, dtype=torch.int64, device=DEV)
        eng.prefix_sum_dev(loc, b - a, cdf)
        d_base = dev(np.array([int(cdf_ref[a - 1]) if a else 0], np.int64))
        eng.offspring_offsets_dev(cdf, b - a, d_base if a else None, d_total, seed, frame, n, first_all[a:b])
    assert np.array_equal(host(first_all), want_first)
    anc = torch.empty(n, dtype=torch.int32, device=DEV)
    for a, b in zip(cuts[:-1], cuts[1:]):
        eng.ancestors_dev(first_all, n, a, b - a, anc[a:b])
    got = host(anc)
    assert np.array_equal(got, want_anc)
    assert (np.diff(got) >= 0).all() and (w[got] > 0).all()


def test_resample_8m_equal_weights_128bit_path(eng):
    """N * CDF overflows 64 bits at 8M particles x 2^32: the 128-by-64 division path."""
    n, n_total = 1 << 20, 8 << 20
    wq = torch.full((n,), 1 << 32, dtype=torch.int64, device=DEV)
    cdf = torch.empty(n, dtype=torch.int64, device=DEV)
    eng.prefix_sum_dev(wq, n, cdf)
    first = torch.empty(n, dtype=torch.int32, device=DEV)
    base = dev(np.array([(1 << 32) * 3 * n], np.int64))        # this shard is the 4th of 8
    total = dev(np.array([(1 << 32) * n_total], np.int64))
    eng.offspring_offsets_dev(cdf, n, base, total, 99, 7, n_total, first)
    f = host(first)
    assert np.array_equal(f[1:], np.arange(3 * n + 1, 4 * n, dtype=np.int32)) and f[0] == 3 * n


def test_fused_entries_equal_staged_ones(eng, orc):
    """slam_motion_score_dev == motion then score; EKF with the log-likelihood kept in the engine +
    slam_logweight_ekf_dev == EKF with an explicit loglik + slam_logweight_dev; device-resident
    observation lists == uploaded ones."""
    import _shard_worker as W
    pkg = load_package()
    L = 70
    meta, edt, bx, by, lm = W.make_world(L=L)
    keep = dev(edt)
    eng.grid_set_dev(2, keep, pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
    eng.scan_upload(bx, by)
    for n in (1000, 200_000):   # both scorer mappings (4 lanes per pose / 1 lane per pose)
        rng = np.random.default_rng(n)
        m = n + 50
        src = [dev(rng.normal(0, 0.3, m).astype(np.float32)) for _ in range(3)]
        anc = dev(np.sort(rng.integers(0, m, n)).astype(np.int32))
        dp, sig = [0.01, -0.02, 0.003], [0.02, 0.02, 0.004]
        a = [torch.empty(n, device=DEV) for _ in range(3)]
        b = [torch.empty(n, device=DEV) for _ in range(3)]
        sa, ca = torch.empty(n, device=DEV), torch.empty(n, dtype=torch.int32, device=DEV)
        sb, cb = torch.empty(n, device=DEV), torch.empty(n, dtype=torch.int32, device=DEV)
        eng.motion_sample_dev(src, anc, a, n, 7, dp, sig, 99, 4)
        eng.score_poses_dev(2, a[0], a[1], a[2], n, sa, ca)
        eng.motion_score_dev(2, src, anc, b, n, 7, dp, sig, 99, 4, sb, cb)
        for u, v in zip(a + [sa, ca], b + [sb, cb]):
            assert torch.equal(u, v)
        # EKF: 70 observations (one batch of two groups), rows padded to 96 floats
        Lp = 96
        mp = _rand_map(rng, L, m, Lp)
        ids = rng.permutation(L).astype(np.int32)
        zx, zy = rng.normal(0, 2, L).astype(np.float32), rng.normal(0, 2, L).astype(np.float32)
        d_in = dev(mp)
        o1, o2 = torch.zeros((m, 5, Lp), device=DEV), torch.zeros((m, 5, Lp), device=DEV)
        ll = torch.empty(n, device=DEV)
        lw1, lw2 = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
        m1, m2 = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
        def table(ids_, zx_, zy_):
            t = np.full((2, L), np.nan, np.float32)
            t[0, ids_], t[1, ids_] = zx_, zy_
            return dev(t)

        eng.obs_upload(ids, zx, zy, L)
        eng.ekf_update_dev(d_in, o1, 5 * Lp, Lp, L, a[0], a[1], a[2], anc, n, 0.02, ll)
        eng.logweight_dev(sa, ll, 0.3, n, lw1, m1)
        tab = table(ids, zx, zy)
        eng.obs_set_dev(tab[0], tab[1], L)
        eng.ekf_update_dev(d_in, o2, 5 * Lp, Lp, L, a[0], a[1], a[2], anc, n, 0.02, None)
        eng.logweight_ekf_dev(sa, 0.3, n, lw2, m2)
        assert torch.equal(o1, o2) and torch.equal(lw1, lw2) and torch.equal(m1, m2)
        # a partial observation list, as a list from the host and as a table on the device
        few = ids[:9].copy()
        eng.obs_upload(few, zx[:9], zy[:9], L)
        eng.ekf_update_dev(d_in, o1, 5 * Lp, Lp, L, a[0], a[1], a[2], anc, n, 0.02, ll)
        eng.logweight_dev(sa, ll, 0.3, n, lw1, m1)
        tab = table(few, zx[:9], zy[:9])
        eng.obs_set_dev(tab[0], tab[1], L)
        eng.ekf_update_dev(d_in, o2, 5 * Lp, Lp, L, a[0], a[1], a[2], anc, n, 0.02, None)
        eng.logweight_ekf_dev(sa, 0.3, n, lw2, m2)
        assert torch.equal(o1, o2) and torch.equal(lw1, lw2) and torch.equal(m1, m2)
    with pytest.raises(pkg.SlamError) as ei:
        eng.logweight_ekf_dev(sa, 0.3, 17, lw2, m2)   # no EKF call for 17 particles on this engine
    assert ei.value.status == -4


def test_gathers(eng):
    rng = np.random.default_rng(3)
    n, m, L = 5000, 6000, 9
    src = rng.standard_normal(m).astype(np.float32)
    idx = rng.integers(0, m, n).astype(np.int32)
    dst = torch.empty(n, device=DEV)
    eng.gather_f32_dev(dev(src), dev(idx), n, dst)
    assert np.array_equal(host(dst), src[idx])
    mp = rng.standard_normal((m, 5, L + 2)).astype(np.float32)    # source rows: plane stride L + 2
    out = torch.zeros((n + 3, 5, 32), device=DEV)                 # destination rows: plane stride 32
    eng.gather_map_dev(dev(mp), out, 5 * (L + 2), 5 * 32, L + 2, 32, L, dev(idx), n)
    got = host(out)
    assert np.array_equal(got[:n, :, :L], mp[idx][:, :, :L]) and not got[n:].any() and not got[:, :, L:].any()


@pytest.mark.parametrize("n,world,skew", [(1000, 2, 3.0), (5000, 4, 2.0), (2048, 3, 6.0), (4099, 5, 0.5), (600_000, 2, 4.0),
                                          (777, 16, 3.0)])
def test_exchange_plan_index_and_pack_vs_numpy(eng, orc, n, world, skew):
    """The duplicate-free exchange, entry point by entry point, against a numpy restatement (np.unique over the
    ancestors of a rank's slots): gather index, plan (anything / send / recv counts / send base) and the packed
    rows, for every rank of the world; offspring counts from heavily skewed to nearly uniform, tile counts on both
    sides of the scan's 2048-element tiles and 256-tile chunks."""
    from _oracle_ops import OracleOps

    load_package()
    from _pf_rehearsal import HipOps

    rng = np.random.default_rng(n + world)
    n_total, L, Lp = n * world, 3, 4
    w = np.exp(skew * rng.standard_normal(n_total))
    counts = rng.multinomial(n_total, w / w.sum())
    first = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)       # what the resample kernels produce
    cpu, gpu = OracleOps(None, None, None, None), HipOps(eng)
    cap = 2 * n
    pose = rng.standard_normal((3, cap)).astype(np.float32)
    mp = rng.standard_normal((cap, 5, Lp)).astype(np.float32)
    moved = 0
    for rank in range(world):
        src_c, plan_c = torch.zeros(n, dtype=torch.int32), torch.zeros(1 + 3 * world, dtype=torch.int32)
        cpu.ancestors_sharded(torch.from_numpy(first), n_total, n, rank, world, src_c, plan_c)
        src_g = torch.zeros(n, dtype=torch.int32, device=DEV)
        plan_g = torch.zeros(1 + 3 * world, dtype=torch.int32, device=DEV)
        gpu.ancestors_sharded(dev(first), n_total, n, rank, world, src_g, plan_g)
        assert np.array_equal(host(plan_g), plan_c.numpy())
        assert gpu.read_plan(plan_g, world) == plan_c.tolist()     # the same plan through mapped host memory
        assert np.array_equal(host(src_g), src_c.numpy())
        plan = plan_c.tolist()
        stot = sum(plan[1:1 + world])
        moved += stot
        rec = 3 + 5 * L
        out_c = torch.zeros(rec * stot)
        cpu.migrate_pack(n, rank, world, plan, torch.from_numpy(pose), cap, torch.from_numpy(mp), 5 * Lp, Lp, L, out_c)
        out_g = torch.zeros(rec * stot, device=DEV)
        gpu.migrate_pack(n, rank, world, plan, dev(pose), cap, dev(mp), 5 * Lp, Lp, L, out_g)
        assert np.array_equal(bits(host(out_g)), bits(out_c.numpy()))
        # and back: unpacking the rows somebody sent fills the staging tail in order
        rtot = min(stot, cap - n)
        if rtot:
            pose_g, mp_g = dev(pose), dev(mp)
            gpu.migrate_unpack(out_g, 1, [rtot], n, pose_g, cap, mp_g, 5 * Lp, Lp, L)
            rows = out_c.numpy()[: rec * rtot].reshape(rtot, rec)
            assert np.array_equal(host(pose_g)[:, n:n + rtot], rows[:, :3].T)
            assert np.array_equal(host(mp_g)[n:n + rtot, :, :L], rows[:, 3:].reshape(rtot, 5, L))
            assert np.array_equal(host(mp_g)[:n], mp[:n])
    assert moved > 0


def test_full_filter_matches_oracle_over_frames(eng, orc):
    """Six frames of the whole loop (motion -> score -> EKF -> weights -> resample, gathers fused)
    on the GPU vs the same loop on the CPU specification: identical particles, maps and weights."""
    import _shard_worker as W
    from _oracle_ops import OracleOps

    load_package()
    from _pf_rehearsal import HipOps, ParticleFilter

    pkg = load_package()
    L, n, frames = 6, 4096, 6
    meta, edt, bx, by, lm = W.make_world(L=L)
    x, y, th, mp = W.init_state(n, L, lm)
    eng.grid_set_dev(3, dev(edt), pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
    keep = dev(edt)
    eng.grid_set_dev(3, keep, pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
    eng.scan_upload(bx, by)
    kw = dict(seed=77, sigma=(0.02, 0.02, 0.004), meas_var=0.02, score_gain=0.05)
    gpu = ParticleFilter(HipOps(eng), n, L, device=DEV, grid_slot=3, **kw)
    cpu = ParticleFilter(OracleOps(meta, edt, bx, by), n, L, device="cpu", **kw)
    for f in (gpu, cpu):
        f.set_poses(x, y, th)
        f.set_map(mp)
    for fr in range(frames):
        obs = W.observations(lm, fr)
        gpu.step([0.01, -0.005, 0.002], obs)
        cpu.step([0.01, -0.005, 0.002], obs)
        assert np.array_equal(host(gpu.src_idx), cpu.src_idx.numpy()), fr          # resample indices: exact
        assert np.array_equal(bits(host(gpu.logw)), bits(cpu.logw.numpy())), fr
    assert np.array_equal(bits(host(gpu.poses())), bits(cpu.poses().numpy()))
    assert np.array_equal(bits(host(gpu.maps())), bits(cpu.maps().numpy()))
    assert gpu.best_particle() == cpu.best_particle()


@pytest.mark.parametrize("L,world", [(6, 2), (0, 2), (6, 4)])
def test_ranks_on_one_card_equal_unsharded_oracle(orc, tmp_path, L, world):
    """The multi-GPU path with the real HIP stages: 2 or 4 processes share this GPU, exchange through gloo
    (host-staged), and must reproduce the unsharded CPU specification bit for bit — poses, landmark maps,
    log-weights — with particles really migrating between the ranks."""
    import socket

    import torch.multiprocessing as mp

    import _shard_worker as W

    n_total, frames = 4096, 6
    ref = W.run_filter(0, 1, n_total, L, frames)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(W.worker_gpu, args=(world, port, n_total, L, frames, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in parts], axis=1)), bits(ref["pose"]))
    assert np.array_equal(bits(np.concatenate([p["logw"] for p in parts])), bits(ref["logw"]))
    if L:
        assert np.array_equal(bits(np.concatenate([p["map"] for p in parts], axis=0)), bits(ref["map"]))
    assert parts[-1]["migrated"].max() > 10   # rows = distinct ancestors (each travels once per destination)
    for p in parts:
        assert tuple(p["best"]) == tuple(np.array(ref["best"]))


def test_c_session_equals_python_frame_loop_and_oracle(eng, orc):
    """slam_pf_* (the C-level session a plain C host uses) runs the same stages as pf.py: identical particles,
    maps and best particle as the CPU specification over several frames, with and without landmarks."""
    import _shard_worker as W
    from _oracle_ops import OracleOps

    pkg = load_package()
    from _pf_rehearsal import ParticleFilter

    for L in (6, 0):
        n, frames = 3000, 5
        meta, edt, bx, by, lm = W.make_world(L=max(L, 1))
        lm = lm[:L]
        keep = dev(edt)
        eng.grid_set_dev(1, keep, pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
        eng.scan_upload(bx, by)
        x, y, th, mp = W.init_state(n, L, lm)
        kw = dict(seed=77, sigma=(0.02, 0.02, 0.004), meas_var=0.02, score_gain=0.05 if L else 1.0)
        ses = pkg.PfSession(eng, n, L, **kw)
        cpu = ParticleFilter(OracleOps(meta, edt, bx, by), n, L, device="cpu", **kw)
        ses.set_poses(x, y, th)
        cpu.set_poses(x, y, th)
        if L:
            ses.set_map(mp)
            cpu.set_map(mp)
        for fr in range(frames):
            use = L > 0 and fr != 2                       # one frame without observations: maps just follow
            if use:
                obs = W.observations(lm, fr)
                eng.obs_upload(obs[0], obs[1], obs[2], L)
            ses.step(1, [0.01, -0.005, 0.002], use)
            cpu.step([0.01, -0.005, 0.002], obs if use else None)
            pose, lw, idx = ses.best()
            blw, bidx = cpu.best_particle()
            assert idx == bidx and float(lw) == blw
            assert np.array_equal(bits(pose), bits(cpu.pose[cpu.cur][:, bidx].numpy()))
        assert np.array_equal(bits(ses.poses()), bits(cpu.poses().numpy()))
        if L:
            assert np.array_equal(bits(ses.maps()), bits(cpu.maps().numpy()))
        ses.close()


def test_ranks_on_one_card_equal_one_rank_at_scale(tmp_path):
    """Sharded == unsharded at a scale the CPU specification cannot check: 4 ranks x 8192 particles x 40 landmarks
    sharing one MI355X against the same filter as ONE rank of 32 768 particles on that MI355X, 15 frames, bit for
    bit (poses, maps, weights, heaviest particle).  (4 ranks + this process = 5 processes on the card.)"""
    import socket

    import _shard_worker as W
    import torch.multiprocessing as mp

    world, n_total, L, frames = 4, 32768, 40, 15
    ctx = mp.get_context("spawn")
    outs = {}
    for tag, g in (("one", 1), ("many", world)):
        d = tmp_path / tag
        d.mkdir()
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(W.worker_gpu, args=(g, port, n_total, L, frames, str(d)), nprocs=g, join=True)
        outs[tag] = [np.load(d / f"rank{r}.npz") for r in range(g)]
    ref, parts = outs["one"][0], outs["many"]
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in parts], axis=1)), bits(ref["pose"]))
    assert np.array_equal(bits(np.concatenate([p["logw"] for p in parts])), bits(ref["logw"]))
    assert np.array_equal(bits(np.concatenate([p["map"] for p in parts], axis=0)), bits(ref["map"]))
    for p in parts:
        assert tuple(p["best"]) == tuple(ref["best"])
    assert max(int(p["migrated"].max()) for p in parts) > 20      # the exchange really carried rows


@pytest.mark.parametrize("form", [0, 1, 2])
def test_full_size_ekf_config2_vs_oracle(eng, orc, form):
    """BASELINE config 2 at full size — 65 536 particles x 500 landmarks, every landmark observed, resample
    gather fused in: the whole 2 x 655 MB update against the CPU specification, bit for bit (both kernel forms)."""
    import ctypes as C
    eng.ekf_form_set(form)
    n, L, Lp = 65536, 500, 512
    rng = np.random.default_rng(65536)
    mp = _rand_map(rng, L, n, Lp)
    x, y, th = (rng.normal(0, 1, n).astype(np.float32) for _ in range(3))
    u = rng.random(n) ** 3
    anc = np.sort(rng.choice(n, n, p=u / u.sum())).astype(np.int32)    # resample-like: sorted, with repeats and gaps
    ids = rng.permutation(L).astype(np.int32)
    zx, zy = rng.normal(0, 2, L).astype(np.float32), rng.normal(0, 2, L).astype(np.float32)
    d_out = torch.full((n, 5, Lp), -555.0, device=DEV)
    ll = torch.empty(n, device=DEV)
    eng.obs_upload(ids, zx, zy, L)
    eng.ekf_update_dev(dev(mp), d_out, 5 * Lp, Lp, L, dev(x), dev(y), dev(th), dev(anc), n, 0.0016, ll)
    want = np.full((n, 5, Lp), -555.0, np.float32)
    wl = np.empty(n, np.float32)
    orc.lib().orc_ekf_update(mp, want, 5 * Lp, Lp, L, x, y, th, anc.ctypes.data_as(C.c_void_p), n, ids, zx, zy, L, 0.0016, wl)
    got = host(d_out)
    assert np.array_equal(bits(got), bits(want))
    assert np.array_equal(bits(host(ll)), bits(wl))
    # size-independent properties: covariances stay symmetric positive definite and never grow
    got, prior = got[:, :, :L], mp[anc][:, :, :L]
    seen = prior[:, 2] >= 0
    det = got[:, 2] * got[:, 4] - got[:, 3] * got[:, 3]
    assert (got[:, 2][seen] > 0).all() and (det[seen] > 0).all() and (got[:, 2][seen] <= prior[:, 2][seen] * (1 + 1e-5)).all()
    eng.ekf_form_set(-1)


def test_full_size_resample_8m_vs_oracle(eng, orc):
    """BASELINE config 4 scale: 8M particles (here in one shard).  Ancestors equal the oracle's; plus the
    properties that hold at any size: sorted, offspring counts within 1 of N*w/S, zero weight => no offspring."""
    n = 8 * 1024 * 1024
    rng = np.random.default_rng(8)
    w = (rng.random(n) ** 4 * 2**32).astype(np.uint64)
    w[rng.integers(0, n, n // 20)] = 0
    seed, frame = 2024, 9
    cdf = torch.empty(n, dtype=torch.int64, device=DEV)
    eng.prefix_sum_dev(dev(w.view(np.int64)), n, cdf)
    cdf_ref = np.cumsum(w, dtype=np.uint64)
    assert np.array_equal(host(cdf).view(np.uint64), cdf_ref)
    total = int(cdf_ref[-1])
    first = torch.empty(n, dtype=torch.int32, device=DEV)
    eng.offspring_offsets_dev(cdf, n, None, dev(np.array([total], np.int64)), seed, frame, n, first)
    anc = torch.empty(n, dtype=torch.int32, device=DEV)
    eng.ancestors_dev(first, n, 0, n, anc)
    got = host(anc)
    want_first = orc.offspring_offsets(cdf_ref, 0, total, orc.comb_offset(seed, frame, total), n)
    assert np.array_equal(host(first), want_first)
    assert np.array_equal(got, orc.ancestors(want_first, 0, n))
    assert (np.diff(got) >= 0).all()
    cnt = np.bincount(got, minlength=n)
    assert cnt.sum() == n and (cnt[w == 0] == 0).all()
    assert (np.abs(cnt - n * w.astype(np.float64) / total) < 1.0 + 1e-6).all()


@pytest.mark.parametrize("n", [1, 2, 7, 2048, 2049, 100_000, 1_048_576 + 3, 8 * 1024 * 1024, 8 * 1024 * 1024 + 2049])
def test_ancestors_from_scan_equals_the_two_launch_form(eng, orc, n):
    """slam_ancestors_from_scan_dev (one launch, 96-bit comparisons against the scan) == slam_offspring_from_scan_dev
    + slam_ancestors_dev, on weights with zeros, ties and a dominant particle; tile counts up to the 4096 the
    one-launch form keeps in LDS and beyond (its fallback); small sizes also against the CPU specification."""
    rng = np.random.default_rng(n)
    logw = (rng.normal(0, 3, n) - 5).astype(np.float32)
    logw[rng.integers(0, n, max(n // 10, 1))] = -200.0     # weight exactly 0 after quantisation
    logw[rng.integers(0, n)] = 0.0                          # the maximum
    if n > 4:
        logw[n // 2] = logw[n // 2 - 1]
    lw, d_max = dev(logw), torch.empty(1, device=DEV)
    tmp = torch.empty(n, device=DEV)
    eng.logweight_dev(None, lw, 0.0, n, tmp, d_max)         # leaves the block maxima the scan needs
    seed, frame = 99, 3
    eng.quantise_scan_dev(tmp, None, n, None)
    first = torch.empty(n, dtype=torch.int32, device=DEV)
    eng.offspring_from_scan_dev(n, None, None, seed, frame, n, first)
    two = torch.empty(n, dtype=torch.int32, device=DEV)
    eng.ancestors_dev(first, n, 0, n, two)
    one = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    eng.ancestors_from_scan_dev(n, seed, frame, one)
    assert torch.equal(one, two)
    if n <= 200_000:
        wq, total = orc.quantise_weights(logw, np.float32(logw.max()))
        want = orc.resample(wq, seed, frame)
        assert np.array_equal(host(one), want)
    # all weights zero cannot happen with finite log-weights, but -inf everywhere must stay memory-safe
    if n == 7:
        eng.logweight_dev(None, dev(np.full(n, -np.inf, np.float32)), 0.0, n, tmp, d_max)
        eng.quantise_scan_dev(tmp, None, n, None)
        eng.ancestors_from_scan_dev(n, seed, frame, one)
        eng.offspring_from_scan_dev(n, None, None, seed, frame, n, first)
        eng.ancestors_dev(first, n, 0, n, two)
        assert torch.equal(one, two)


def test_rccl_collectives_single_rank(eng, orc, tmp_path):
    """The multi-GPU code path over the real RCCL backend, as far as one GPU allows: a world_size-1 `nccl` group,
    every collective of the frame loop issued for real (all-reduce MAX on float32, all-gather of int64 totals and of
    int32 offsets, all-to-all with empty splits), the sharded kernels used instead of the single-GPU ones — and the
    result must equal the plain single-GPU path bit for bit."""
    import socket
    import subprocess
    import sys
    import textwrap

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {str(__import__('pathlib').Path(__file__).resolve().parents[1])!r}); sys.path.insert(0, {str(__import__('pathlib').Path(__file__).resolve().parent)!r})
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}")
        import numpy as np, torch, torch.distributed as dist
        import _shard_worker as W
        from __graft_entry__ import load_package
        pkg = load_package()
        from _pf_rehearsal import HipOps, ParticleFilter
        dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        L, n, frames = 6, 8192, 5
        meta, edt, bx, by, lm = W.make_world(L=L)
        eng = pkg.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
        d_edt = torch.from_numpy(edt).to(dev)
        eng.grid_set_dev(0, d_edt, pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y))
        eng.scan_upload(bx, by)
        x, y, th, mp = W.init_state(n, L, lm)
        out = []
        for force in (False, True):
            pf = ParticleFilter(HipOps(eng), n, L, device=dev, seed=77, sigma=(0.02, 0.02, 0.004), meas_var=0.02,
                                score_gain=0.05, force_collectives=force)
            pf.set_poses(x, y, th); pf.set_map(mp)
            for f in range(frames):
                pf.step([0.01, -0.005, 0.002], W.observations(lm, f))
            torch.cuda.synchronize()
            out.append((pf.poses().cpu().numpy(), pf.maps().cpu().numpy(), pf.logw.cpu().numpy(), pf.best_particle()))
        dist.destroy_process_group()
        a, b = out
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), "poses differ"
        assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), "maps differ"
        assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)) and a[3] == b[3]
        print("RCCL single-rank path == single-GPU path")
    """))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL single-rank path == single-GPU path" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_fast_reciprocal_equals_ieee_division_for_every_float_in_range():
    """csrc/ekf_math.h: 1 / det without the scale / fix-up steps of a general division when det's exponent is in [-60, 60].
    Every float of that range, both signs, scalar and packed form, against the compiler's correctly rounded division."""
    pkg = load_package()
    eng = pkg.Engine(0)
    bad, seen = eng.selftest_reciprocal()
    eng.close()
    assert seen == 2 * 121 * (1 << 23), seen
    assert bad == 0, bad
