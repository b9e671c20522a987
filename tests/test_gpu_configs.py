"""BASELINE.json configurations at FULL size on one MI355X, called through the C ABI.

  configs[4] per-GPU share   524 288 particles x 5 000 landmarks  (Lp = 5 024: 100 480-byte rows, 52.7 GB per buffer)
  north-star target          1 048 576 particles x 1 000 landmarks (21.5 GB per buffer)
  configs[3]                 the 8-rank particle shard, rehearsed as 8 ranks sharing this card

At these sizes the CPU specification (oracle/slam_oracle_pf.c — PARITY UNPINNED, the reference has no such
stages) cannot check everything in seconds, so each test compares a SAMPLE of >= 4 096 particles
bit for bit and adds size-independent properties over the whole output: covariances stay symmetric positive
definite and never grow, log-likelihoods are finite, a second run gives the same bits.
"""
import ctypes as C

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


def _device_map(n, L, Lp, seed, chunk=32768):
    """Random maps [n][5][Lp] made on the device (the host could not hold them): means N(0, 3), covariances
    A A^T + 0.02 I, every 10th landmark column 'not seen yet' (P_xx = -1), padding columns -555."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    m = torch.full((n, 5, Lp), -555.0, device=DEV)
    unseen = torch.arange(3, L, 10, device=DEV)
    for i0 in range(0, n, chunk):
        i1 = min(i0 + chunk, n)
        k = i1 - i0
        m[i0:i1, 0:2, :L] = 3.0 * torch.randn((k, 2, L), device=DEV, generator=g)
        a = 0.3 * torch.randn((k, 4, L), device=DEV, generator=g)
        m[i0:i1, 2, :L] = a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1] + 0.02
        m[i0:i1, 3, :L] = a[:, 0] * a[:, 2] + a[:, 1] * a[:, 3]
        m[i0:i1, 4, :L] = a[:, 2] * a[:, 2] + a[:, 3] * a[:, 3] + 0.02
        m[i0:i1, 2, unseen] = -1.0
        del a
    return m


def _resample_like(rng, n):
    """Sorted indices with repeats and gaps, as a resample produces them."""
    u = rng.random(n) ** 3
    return np.sort(rng.choice(n, n, p=u / u.sum())).astype(np.int32)


def _full_size_ekf(eng, orc, n, L, Lp, nobs, seed, nsample=4096, form=-1):
    eng.ekf_form_set(form)
    rng = np.random.default_rng(seed)
    m_in = _device_map(n, L, Lp, seed)
    m_out = torch.full((n, 5, Lp), -777.0, device=DEV)
    x, y, th = (rng.normal(0, 1, n).astype(np.float32) for _ in range(3))
    anc = _resample_like(rng, n)
    ids = rng.permutation(L)[:nobs].astype(np.int32)
    zx, zy = rng.normal(0, 2, nobs).astype(np.float32), rng.normal(0, 2, nobs).astype(np.float32)
    d_x, d_y, d_th, d_anc = (torch.from_numpy(a).to(DEV) for a in (x, y, th, anc))
    ll = torch.empty(n, device=DEV)
    eng.obs_upload(ids, zx, zy, L)
    eng.ekf_update_dev(m_in, m_out, 5 * Lp, Lp, L, d_x, d_y, d_th, d_anc, n, 0.0016, ll)
    torch.cuda.synchronize()

    # ---- sample vs the CPU specification, bit for bit: random slots + both ends + a run of neighbours (shared ancestors)
    s = np.unique(np.concatenate([rng.integers(0, n, nsample), [0, 1, n - 2, n - 1], np.arange(n // 2, n // 2 + 64)]))
    d_s = torch.from_numpy(s).to(DEV)
    rows_in = m_in[d_anc[d_s].long()].cpu().numpy()                      # the ancestors' rows, gathered on the device
    want = np.full((len(s), 5, Lp), -777.0, np.float32)
    wl = np.empty(len(s), np.float32)
    orc.lib().orc_ekf_update(rows_in, want, 5 * Lp, Lp, L, x[s].copy(), y[s].copy(), th[s].copy(), None, len(s),
                             ids, zx, zy, nobs, 0.0016, wl)
    got = m_out[d_s].cpu().numpy()
    assert np.array_equal(bits(got[:, :, :L]), bits(want[:, :, :L]))
    assert np.array_equal(bits(ll[d_s].cpu().numpy()), bits(wl))

    # ---- properties over the WHOLE output, in chunks (prior = the ancestor's row)
    observed = torch.zeros(L, dtype=torch.bool, device=DEV)
    observed[torch.from_numpy(ids).to(DEV).long()] = True
    assert bool(torch.isfinite(ll).all())
    for i0 in range(0, n, 16384):
        i1 = min(i0 + 16384, n)
        o = m_out[i0:i1, :, :L]
        p = m_in[d_anc[i0:i1].long()][:, :, :L]
        seen = p[:, 2] >= 0
        det = o[:, 2] * o[:, 4] - o[:, 3] * o[:, 3]
        assert bool((o[:, 2][seen] > 0).all()) and bool((det[seen] > 0).all())
        assert bool((o[:, 2][seen] <= p[:, 2][seen] * (1 + 1e-5)).all())          # an observation never inflates P
        assert bool(torch.equal(o[:, :, ~observed], p[:, :, ~observed]))         # unobserved landmarks: copied through
        first = (~seen) & observed
        assert bool((o[:, 2][first] == np.float32(0.0016)).all())                 # first sighting: P = R
        del o, p, det
    # ---- a second run gives the same bits (no atomics, no order dependence)
    again = torch.empty_like(m_out)
    ll2 = torch.empty(n, device=DEV)
    eng.ekf_update_dev(m_in, again, 5 * Lp, Lp, L, d_x, d_y, d_th, d_anc, n, 0.0016, ll2)
    torch.cuda.synchronize()
    assert bool(torch.equal(again[:, :, :L], m_out[:, :, :L])) and bool(torch.equal(ll, ll2))
    del m_in, m_out, again
    torch.cuda.empty_cache()
    eng.ekf_form_set(-1)


def test_config4_share_512k_x_5000_landmarks(eng, orc):
    """BASELINE configs[4], one GPU's share: 524 288 particles x 5 000 landmarks, Lp = 5 024 (39 whole 128-landmark
    batches and a tail per row), resample gather fused, every landmark observed."""
    _full_size_ekf(eng, orc, 524288, 5000, 5024, 5000, seed=5000, form=2)


@pytest.mark.parametrize("form", [0, 1])
def test_north_star_1m_x_1000_landmarks(eng, orc, form):
    """The north-star target workload: 1 048 576 particles x 1 000 landmarks on one GPU (Lp = 1 024), with either
    out-of-place kernel (one wavefront per particle / per 4 neighbouring particles)."""
    _full_size_ekf(eng, orc, 1048576, 1000, 1024, 1000, seed=1000, form=form)


def test_north_star_sparse_observations(eng, orc):
    """Same size, only 32 of the 1 000 landmarks observed: the other 968 are copied through."""
    _full_size_ekf(eng, orc, 1048576, 1000, 1024, 32, seed=1032, nsample=4096, form=1)


@pytest.mark.parametrize("L,Lp", [(4999, 4999), (4999, 5024), (5000, 5000), (5000, 5024), (5024, 5024), (8191, 8192),
                                  (8192, 8192)])
def test_ekf_long_rows_small_n(eng, orc, L, Lp):
    """Landmark counts of configs[4] and the engine's maximum (SLAM_MAX_OBS = 8192) with few particles, whole output
    against the CPU specification: in place, out of place, with the fused gather, subsets observed."""
    rng = np.random.default_rng(L * 3 + Lp)
    for n, with_anc, in_place, nobs in [(67, True, False, L), (5, False, True, L // 2), (130, True, False, 37), (64, False, False, 0)]:
        eng.ekf_form_set({67: 1, 5: 2, 130: 2, 64: 0}[n])   # grouped by 4 / by 2 / one wavefront per particle
        rows = n + (9 if with_anc else 0)
        mp = np.full((rows, 5, Lp), -555.0, np.float32)
        mp[:, 0:2, :L] = rng.normal(0, 3, (rows, 2, L))
        A = rng.normal(0, 0.3, (rows, L, 2, 2))
        P = A @ np.swapaxes(A, -1, -2) + 0.02 * np.eye(2)
        mp[:, 2, :L], mp[:, 3, :L], mp[:, 4, :L] = P[..., 0, 0], P[..., 0, 1], P[..., 1, 1]
        mp[:, 2, rng.integers(0, L, L // 10)] = -1.0
        x, y, th = (rng.normal(0, 1, n).astype(np.float32) for _ in range(3))
        anc = np.sort(rng.integers(0, rows, n)).astype(np.int32) if with_anc else None
        ids = rng.permutation(L)[:nobs].astype(np.int32)
        zx, zy = rng.normal(0, 2, nobs).astype(np.float32), rng.normal(0, 2, nobs).astype(np.float32)
        eng.obs_upload(ids, zx, zy, L)
        d_in = torch.from_numpy(mp).to(DEV)
        d_out = d_in if in_place else torch.full((rows, 5, Lp), -777.0, device=DEV)
        ll = torch.empty(n, device=DEV)
        eng.ekf_update_dev(d_in, d_out, 5 * Lp, Lp, L, *(torch.from_numpy(a).to(DEV) for a in (x, y, th)),
                           torch.from_numpy(anc).to(DEV) if with_anc else None, n, 0.015, ll)
        want = mp.copy() if in_place else np.full((rows, 5, Lp), -777.0, np.float32)
        wl = np.empty(n, np.float32)
        orc.lib().orc_ekf_update(want if in_place else mp, want, 5 * Lp, Lp, L, x, y, th,
                                 anc.ctypes.data_as(C.c_void_p) if with_anc else None, n, ids, zx, zy, nobs, 0.015, wl)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        tag = f"L={L} Lp={Lp} n={n} nobs={nobs} anc={with_anc} in_place={in_place}"
        assert np.array_equal(bits(got[:n, :, :L]), bits(want[:n, :, :L])), tag
        assert np.array_equal(bits(ll.cpu().numpy()), bits(wl)), tag
    eng.ekf_form_set(-1)


# ------------------------------------------------------------------ the sharded C session (slam_pf_create_sharded)
def _run_c_session_ranks(world, n_total, L, frames, transport="local", recv_capacity=0, ess=0.0, inplace_form=-1, paged=False,
                         fail_rank=None, fail_frame=None, layout=None, sparse_obs=False, dense_from=None, maps_every_frame=False):
    """`world` ranks of the C-level sharded session in THIS process, one host thread per rank, all on cuda:0
    (in-process transport), or a one-rank RCCL communicator.  Returns the concatenated population."""
    import threading

    import _shard_worker as W

    pkg = load_package()
    meta, edt, bx, by, lm = W.make_world(L=max(L, 1))
    lm = lm[:L]
    x, y, th, mp = W.init_state(n_total, L, lm)
    d_edt = torch.from_numpy(edt).to(DEV)
    gm = pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y)
    n = n_total // world
    group = pkg.LocalGroup(world) if transport == "local" else None
    uid = pkg.comm_unique_id() if transport == "rccl" else None
    out, errors = [None] * world, []

    def rank_main(r):
        try:
            eng = pkg.Engine(0)
            eng.ekf_inplace_form_set(inplace_form)   # frames that keep their population: whole rows / the compact observation list
            eng.grid_set_dev(0, d_edt, gm)
            eng.scan_upload(bx, by)
            comm = None
            if transport == "local":
                comm = pkg.Comm.local(eng, group, r)
            elif transport == "rccl":
                comm = pkg.Comm.rccl(eng, r, world, uid)
            ses = pkg.PfSession(eng, n, L, seed=77, sigma=(0.02, 0.02, 0.004), meas_var=0.02,
                                score_gain=0.05 if L else 1.0, comm=comm, recv_capacity=recv_capacity,
                                resample_ess_frac=ess, map_layout=layout or ("pages" if paged else "rows"))
            sl = slice(r * n, (r + 1) * n)
            ses.set_poses(x[sl], y[sl], th[sl])
            if L:
                ses.set_map(mp[sl])
            rows, bests, layouts, frame_maps = [], [], [], []
            for f in range(frames):
                use = L > 0 and f != 2                      # one frame without observations: the maps just follow
                if r == fail_rank and f == fail_frame:
                    comm.abort()                            # this rank gives up: the others must fail, not wait
                    raise RuntimeError("rank gave up")
                if use and sparse_obs and (dense_from is None or f < dense_from):   # 12 neighbouring landmarks: under a quarter of the map
                    ids = ((np.arange(12) + 17 * f) % L).astype(np.int32)
                    z = lm[ids] + 0.01 * np.float32(f)
                    eng.obs_upload(ids, z[:, 0].copy(), z[:, 1].copy(), L)
                elif use and dense_from is not None:         # every landmark, every frame: AUTO's way back to rows
                    z = lm + 0.01 * np.float32(f)
                    eng.obs_upload(np.arange(L, dtype=np.int32), z[:, 0].copy(), z[:, 1].copy(), L)
                elif use:
                    eng.obs_upload(*W.observations(lm, f), L)
                ses.step(0, [0.01, -0.005, 0.002], use)
                rows.append(ses.rows_received())
                layouts.append(ses.layout())
                if maps_every_frame:                        # collective; completes the exchange of this frame early
                    gsel = np.arange(0, n_total, 97)        # every 97th particle of the POPULATION: this rank's share of them
                    frame_maps.append(ses.map_rows((gsel[(gsel >= r * n) & (gsel < (r + 1) * n)] - r * n).astype(np.int32)))
                if f % 3 == 1:
                    bests.append(ses.best())                 # collective; must not disturb the pending exchange
            res = {"pose": ses.poses(), "best": ses.best(), "rows": rows, "bests": bests, "resampled": ses.frames_resampled(),
                   "mean": ses.mean(0.07), "paged_end": ses.is_paged(), "layouts": layouts, "frame_maps": frame_maps,
                   "fused": eng.frame_fusion_count()}
            if L:
                res["map"] = ses.maps()
            out[r] = res
            ses.close()
            if comm:
                comm.close()
            eng.close()
        except BaseException as exc:   # noqa: BLE001 - re-raised by the caller
            errors.append((r, exc))

    th_ = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th_:
        t.start()
    for t in th_:
        t.join()
    if group:
        group.close()
    if errors:
        raise errors[0][1]
    return out


@pytest.mark.parametrize("world,n_total,L", [(2, 4096, 6), (4, 4096, 0), (8, 32768, 40), (3, 3000, 6)])
def test_c_sharded_session_ranks_on_one_card_equal_one_rank(orc, world, n_total, L):
    """slam_pf_create_sharded with 2 / 3 / 4 / 8 ranks sharing this card (threads of one process, in-process transport:
    the whole C frame loop, plan / pack / unpack kernels and every exchange step issued by the engine) against the
    SAME population as one single-GPU session: poses, maps and the heaviest particle, bit for bit.  The 8 x 4 096 x 40
    case is BASELINE configs[3]'s eight-way particle shard in rehearsal (no 8-GPU box here)."""
    frames = 7
    one = _run_c_session_ranks(1, n_total, L, frames, transport=None)[0]
    many = _run_c_session_ranks(world, n_total, L, frames)
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in many], axis=1)), bits(one["pose"]))
    if L:
        assert np.array_equal(bits(np.concatenate([p["map"] for p in many], axis=0)), bits(one["map"]))
    for p in many:   # every rank names the same heaviest particle of the whole population, with its pose
        assert p["best"][2] == one["best"][2] and p["best"][1] == one["best"][1]
        assert np.array_equal(bits(p["best"][0]), bits(one["best"][0]))
        for a, b in zip(p["bests"], one["bests"]):
            assert a[2] == b[2] and a[1] == b[1] and np.array_equal(bits(a[0]), bits(b[0]))
    assert max(max(p["rows"]) for p in many) > 10      # rows really travelled between the ranks
    # posterior mean (slam_pf_mean): exact integer sums on the device -> the same bits on every rank and for one GPU,
    # and equal to the same fixed-point sums made with numpy from the population itself
    for p in many:
        assert np.array_equal(bits(p["mean"]), bits(one["mean"]))
    x, y, th = one["pose"]
    s, c = orc.det_sincos((th - np.float32(0.07)).astype(np.float32))
    fx = lambda a, sh: int(np.trunc(a.astype(np.float64) * 2.0 ** sh).astype(np.int64).sum())
    n = x.size
    assert one["mean"][0] == np.float32(fx(x, 32) / 2.0 ** 32 / n) and one["mean"][1] == np.float32(fx(y, 32) / 2.0 ** 32 / n)
    assert abs(float(one["mean"][2]) - (0.07 + np.arctan2(float(fx(s, 30)), float(fx(c, 30))))) < 1e-6


def test_c_sharded_session_over_rccl_one_rank(orc):
    """The engine-issued RCCL calls themselves, as far as one GPU allows: a one-rank communicator (ncclCommInitRank,
    all-reduce MAX, the all-gathers, the grouped send/recv with empty splits — all issued on the ENGINE's stream in
    program order, csrc/comm.hip) must reproduce the single-GPU session bit for bit.  RCCL with more than one rank needs more
    than one GPU (it refuses two ranks on one device): unverified on this box."""
    one = _run_c_session_ranks(1, 8192, 6, 6, transport=None)[0]
    via = _run_c_session_ranks(1, 8192, 6, 6, transport="rccl")[0]
    assert np.array_equal(bits(via["pose"]), bits(one["pose"])) and np.array_equal(bits(via["map"]), bits(one["map"]))
    assert via["best"][2] == one["best"][2]


def test_c_sharded_session_small_staging_refuses_collectively(orc):
    """recv_capacity smaller than what a frame may need: every rank gets SLAM_ERR_CAPACITY in the same frame (the
    verdict is derived from the all-gathered offsets), nobody is left waiting inside a collective."""
    pkg = load_package()
    with pytest.raises(pkg.SlamError) as ei:
        _run_c_session_ranks(2, 4096, 6, 6, recv_capacity=8)
    assert ei.value.status == -5


def test_slam_pf_main_several_ranks_identical_output(orc, tmp_path):
    """The C host program: 4 ranks (threads, in-process transport, one card) and a one-rank RCCL run print the same
    pose log and write the same map as the single-GPU run."""
    import json
    import subprocess

    from __graft_entry__ import PKG_DIR
    from conftest import GOLDEN

    info = json.loads((GOLDEN / "datasets.json").read_text())["parity"]
    csv = tmp_path / "parity.csv"
    orc.run_tool("gen_dataset", csv, *info["gen_args"])
    exe = PKG_DIR / "lib" / "slam_pf_main"
    outs = {}
    for tag, extra in (("one", []), ("four", ["--gpus", "4", "--transport", "local", "--same-device"]),
                       ("rccl1", ["--gpus", "1", "--transport", "rccl"])):
        r = subprocess.run([str(exe), str(csv), "300", "1079", str(tmp_path / f"map_{tag}.csv"), "8192", "7", *extra],
                           check=True, capture_output=True, text=True, timeout=600)
        outs[tag] = ([ln for ln in r.stdout.splitlines() if ln.startswith("pose =")], (tmp_path / f"map_{tag}.csv").read_bytes())
        print(tag, r.stderr.strip())
    assert len(outs["one"][0]) == 299
    assert outs["four"] == outs["one"] and outs["rccl1"] == outs["one"]


# ------------------------------------------------------------------ ESS-gated resampling
def _gated_reference(orc, n, L, frames, ess):
    """The gated frame loop written with the CPU specification's functions (oracle/): the executable statement of
    DESIGN.md §7 "resample gate".  Returns final poses / maps with the pending gather applied, and the per-frame verdicts."""
    import _shard_worker as W

    meta, edt, bx, by, lm = W.make_world(L=max(L, 1))
    lm = lm[:L]
    x, y, th, mp = W.init_state(n, L, lm)
    fq = orc.ess_frac_q16(ess)
    anc, carry, prev_resampled, verdicts = None, None, True, []
    seed, sigma, gain, mvar = 77, (0.02, 0.02, 0.004), (0.05 if L else 1.0), 0.02
    for f in range(frames):
        x, y, th = orc.motion_sample(x, y, th, anc, n, 0, [0.01, -0.005, 0.002], sigma, seed, f)
        score, _ = orc.score_poses_det(meta, edt, bx, by, x, y, th)
        ll = None
        use = L > 0 and f != 2
        if use:
            ids, zx, zy = W.observations(lm, f)
            mp, ll = orc.ekf_update(mp, x, y, th, anc, ids, zx, zy, mvar)
        elif L and anc is not None:
            mp = mp[anc]
        logw, m = orc.logweight_carry(score, ll, gain, None if prev_resampled else carry)
        wq, _ = orc.quantise_weights(logw, m)
        s16, q16 = orc.ess_terms(wq)
        prev_resampled = orc.ess_resample(s16, q16, n, fq) if fq else True
        carry = orc.weight_carry(logw, m)
        anc = orc.resample(wq, seed, f) if prev_resampled else np.arange(n, dtype=np.int32)
        verdicts.append(prev_resampled)
    return np.stack([x[anc], y[anc], th[anc]]), (mp[anc] if L else None), verdicts


@pytest.mark.parametrize("inplace_form", [0, 1])
@pytest.mark.parametrize("L,ess", [(6, 0.5), (0, 0.1), (6, 0.2), (6, 0.05), (6, 0.9)])
def test_ess_gated_session_matches_specification(orc, L, ess, inplace_form):
    """slam_pf_config.resample_ess_frac: frames whose effective sample size stays above the threshold keep their
    population (ancestor = itself, EKF in place on the observed landmarks, weights carried into the next frame); the
    verdict is integer arithmetic on the device.  Poses and maps after 9 frames equal the CPU specification bit for bit,
    and the scenario contains frames of both kinds."""
    n, frames = 4096, 9
    want_pose, want_map, verdicts = _gated_reference(orc, n, L, frames, ess)
    got = _run_c_session_ranks(1, n, L, frames, transport=None, ess=ess, inplace_form=inplace_form)[0]
    assert np.array_equal(bits(got["pose"]), bits(want_pose))
    if L:
        assert np.array_equal(bits(got["map"]), bits(want_map))
    assert got["resampled"] == sum(verdicts[:-1])          # the host has looked at every frame but the last
    if ess < 0.9:
        assert any(verdicts) and not all(verdicts), verdicts


@pytest.mark.parametrize("world,L", [(4, 6), (2, 0), (8, 6)])
def test_ess_gated_sharded_session_equals_one_rank(orc, world, L):
    """The gate is decided from integer sums over the WHOLE population (all-gathered shard sums), so every rank reaches
    the same verdict and the sharded run equals the single-GPU run bit for bit; frames that keep their population
    exchange nothing."""
    n_total, frames, ess = 4096, 9, (0.5 if L else 0.1)
    one = _run_c_session_ranks(1, n_total, L, frames, transport=None, ess=ess)[0]
    many = _run_c_session_ranks(world, n_total, L, frames, ess=ess, inplace_form=1)   # ranks: list form; one GPU: the engine's choice
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in many], axis=1)), bits(one["pose"]))
    if L:
        assert np.array_equal(bits(np.concatenate([p["map"] for p in many], axis=0)), bits(one["map"]))
    for p in many:
        assert p["resampled"] == one["resampled"] and 0 < p["resampled"] < frames - 1
        assert p["best"][2] == one["best"][2]


def test_north_star_split_8_ranks_x_131072_x_1000_on_one_card():
    """The north-star problem as the eight-way shard it is meant to run as — 8 ranks x 131 072 particles x 1 000 landmarks —
    rehearsed on ONE card (in-process transport: the whole sharded C session, every exchange step, except RCCL itself) against
    the same 1 048 576 particles on one rank: poses of every slot, maps of 4 096 sampled slots and the heaviest particle, bit
    for bit, after 3 frames.  NOT a scaling figure (the ranks share a GPU), and no scaling curve exists yet: RCCL with more than
    one rank has never run (single-GPU boxes)."""
    import threading

    import _shard_worker as W

    pkg = load_package()
    world, n, L, frames = 8, 131072, 1000, 3
    n_total, Lp = world * n, 1024
    meta, edt, bx, by, lm = W.make_world(L=L)
    d_edt = torch.from_numpy(edt).to(DEV)
    gm = pkg.grid_meta(meta.rows, meta.cols, meta.ld, meta.pixel, meta.min_x, meta.min_y)
    rng = np.random.default_rng(8)
    x, y, th = ((s * rng.standard_normal(n_total)).astype(np.float32) for s in (0.3, 0.3, 0.05))
    g = torch.Generator(device=DEV).manual_seed(8)
    d_lm = torch.from_numpy(lm).to(DEV)
    m0 = torch.zeros((n_total, 5, Lp), device=DEV)
    for i0 in range(0, n_total, 65536):           # means near the landmarks; covariances shared by runs of 512 particles
        m0[i0:i0 + 65536, 0, :L] = d_lm[:, 0] + 0.1 * torch.randn((65536, L), device=DEV, generator=g)
        m0[i0:i0 + 65536, 1, :L] = d_lm[:, 1] + 0.1 * torch.randn((65536, L), device=DEV, generator=g)
        for j0 in range(i0, i0 + 65536, 512):
            a = 0.02 + 0.1 * torch.rand((3, L), device=DEV, generator=g)
            m0[j0:j0 + 512, 2, :L], m0[j0:j0 + 512, 4, :L] = a[0], a[1]
            m0[j0:j0 + 512, 3, :L] = (a[2] - 0.07) * 0.3
    m0[:, 2, 7:L:13] = -1.0                        # some landmarks nobody has seen yet
    torch.cuda.synchronize()
    sel = np.unique(np.concatenate([rng.integers(0, n_total, 4096), [0, n - 1, n, n_total - 1]])).astype(np.int64)
    obs = [(np.arange(L, dtype=np.int32), (lm[:, 0] + 0.01 * f).astype(np.float32), (lm[:, 1] - 0.01 * f).astype(np.float32))
           for f in range(frames)]

    def run(rank, world_, group, out):
        eng = pkg.Engine(0)
        eng.grid_set_dev(0, d_edt, gm)
        eng.scan_upload(bx, by)
        comm = pkg.Comm.local(eng, group, rank) if group else None
        k = n_total // world_
        ses = pkg.PfSession(eng, k, L, seed=77, sigma=(0.02, 0.02, 0.004), meas_var=0.02, score_gain=0.05, comm=comm)
        sl = slice(rank * k, (rank + 1) * k)
        ses.set_poses(x[sl], y[sl], th[sl])
        ses.set_map_dev(m0[sl], 5 * Lp, Lp)
        eng.sync()
        for f in range(frames):
            eng.obs_upload(*obs[f], L)
            ses.step(0, [0.01, -0.005, 0.002], True)
        mine = sel[(sel >= rank * k) & (sel < (rank + 1) * k)] - rank * k
        out[rank] = {"pose": ses.poses(), "maps": ses.map_rows(mine.astype(np.int32)), "best": ses.best(), "layout": ses.layout(),
                     "rows": ses.rows_received()}
        ses.close()
        if comm:
            comm.close()
        eng.close()

    one = [None]
    run(0, 1, None, one)
    torch.cuda.empty_cache()
    group = pkg.LocalGroup(world)
    many, errors = [None] * world, []

    def guarded(r):
        try:
            run(r, world, group, many)
        except BaseException as exc:   # noqa: BLE001
            errors.append(exc)

    ths = [threading.Thread(target=guarded, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    group.close()
    del m0
    torch.cuda.empty_cache()
    if errors:
        raise errors[0]
    assert one[0]["layout"] == "split" and all(p["layout"] == "split" for p in many)
    assert np.array_equal(bits(np.concatenate([p["pose"] for p in many], axis=1)), bits(one[0]["pose"]))
    assert np.array_equal(bits(np.concatenate([p["maps"] for p in many], axis=0)), bits(one[0]["maps"]))
    for p in many:
        assert p["best"][2] == one[0]["best"][2] and np.array_equal(bits(p["best"][0]), bits(one[0]["best"][0]))
