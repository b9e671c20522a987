"""Known-answer tests of the particle-filter SPECIFICATION (oracle/slam_oracle_pf.c, rows A9-A12).

PARITY UNPINNED: the reference has no particle filter, so these stages cannot
be checked against reference outputs.  What is checked instead: published known answers (Philox),
float64 re-derivations (EKF, elementary functions), exact integer identities (resampling) and the
one reference anchor — zero-noise motion == the constant-velocity predict of main.c:875-898.
"""
import numpy as np
import pytest


def test_philox_known_answers(orc):
    # Random123 kat_vectors: philox4x32-10
    assert [hex(v) for v in orc.philox([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(v) for v in orc.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(v) for v in orc.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])] == [
        "0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_det_math_accuracy(orc):
    a = np.linspace(-20, 20, 400001).astype(np.float32)
    s, c = orc.det_sincos(a)
    assert np.abs(s - np.sin(a.astype(np.float64))).max() < 2.5e-7
    assert np.abs(c - np.cos(a.astype(np.float64))).max() < 2.5e-7
    x = np.linspace(-79.9, 0, 200001).astype(np.float32)
    e = orc.det_exp(x)
    ref = np.exp(x.astype(np.float64))
    assert (np.abs(e - ref) / ref).max() < 3e-7
    assert orc.det_exp([0.0])[0] == 1.0 and orc.det_exp([-80.0])[0] == 0.0 and orc.det_exp([-1e30])[0] == 0.0
    assert orc.det_exp([np.nan])[0] == 0.0 and orc.det_exp([3.0])[0] == 1.0   # clamped: weights never exceed 1
    v = np.exp(np.linspace(-40, 40, 200001)).astype(np.float32)
    lg = orc.det_log(v)
    ref = np.log(v.astype(np.float64))
    assert (np.abs(lg - ref) / np.maximum(np.abs(ref), 1.0)).max() < 2e-7
    assert orc.det_log([1.0])[0] == 0.0


def test_motion_zero_noise_is_the_reference_predict(orc):
    """sigma = 0, one particle: guess = pose + (pose - prev), main.c:884-891, bit for bit."""
    pose = np.array([1.2345678, -0.7654321, 0.1234567], np.float32)
    prev = np.array([1.2301234, -0.7612345, 0.1229876], np.float32)
    dp = (pose - prev).astype(np.float32)                      # DiffPose, main.c:812-817
    want = (pose + dp).astype(np.float32)                      # main.c:889-891
    x, y, th = orc.motion_sample(pose[0:1], pose[1:2], pose[2:3], None, 1, 0, dp, [0, 0, 0], seed=99, frame=5)
    assert (x[0], y[0], th[0]) == tuple(want)


def test_motion_noise_statistics_and_shard_independence(orc):
    n = 200000
    z = np.zeros(n, np.float32)
    sig = np.array([0.05, 0.03, 0.01], np.float32)
    x, y, th = orc.motion_sample(z, z, z, None, n, 0, [1.0, 2.0, 3.0], sig, seed=1234, frame=7)
    for arr, mu, s in ((x, 1.0, sig[0]), (y, 2.0, sig[1]), (th, 3.0, sig[2])):
        assert abs(arr.mean() - mu) < 4 * s / np.sqrt(n)
        assert abs(arr.std() / s - 1) < 0.01
        k = ((arr - mu) / s) ** 4
        assert abs(k.mean() - 3.0) < 0.1          # gaussian kurtosis
    assert abs(np.corrcoef(x, y)[0, 1]) < 0.01 and abs(np.corrcoef(x, th)[0, 1]) < 0.01
    # same global ids => same noise, however the population is cut into shards
    h = n // 2
    xa, ya, ta = orc.motion_sample(z[:h], z[:h], z[:h], None, h, 0, [1.0, 2.0, 3.0], sig, 1234, 7)
    xb, yb, tb = orc.motion_sample(z[h:], z[h:], z[h:], None, n - h, h, [1.0, 2.0, 3.0], sig, 1234, 7)
    assert np.array_equal(np.concatenate([xa, xb]), x) and np.array_equal(np.concatenate([ta, tb]), th)
    # different frame or seed => different noise
    x2, _, _ = orc.motion_sample(z, z, z, None, n, 0, [1.0, 2.0, 3.0], sig, 1234, 8)
    assert not np.array_equal(x, x2)
    # fused ancestor gather
    src = np.arange(10, dtype=np.float32)
    anc = np.array([3, 3, 0, 9], np.int32)
    xg, _, _ = orc.motion_sample(src, src, src, anc, 4, 0, [0, 0, 0], [0, 0, 0], 1, 1)
    assert xg.tolist() == [3, 3, 0, 9]


def _ekf_f64(mu, P, pose, z, q):
    ct, st = np.cos(pose[2]), np.sin(pose[2])
    H = np.array([[ct, -st], [st, ct]])
    v = z - H @ (mu - pose[:2])
    S = H @ P @ H.T + q * np.eye(2)
    K = P @ H.T @ np.linalg.inv(S)
    ll = -0.5 * v @ np.linalg.solve(S, v) - 0.5 * np.log(np.linalg.det(S)) - np.log(2 * np.pi)
    return mu + K @ v, (np.eye(2) - K @ H) @ P, ll


def _rows(planes):
    """[5][L][n] (convenient to fill) -> the engine's layout, one row per particle: [n][5][L]"""
    return np.ascontiguousarray(np.asarray(planes, np.float32).transpose(2, 0, 1))


def _planes(rows):
    return np.asarray(rows).transpose(1, 2, 0)


def test_ekf_single_landmark_against_float64(orc):
    rng = np.random.default_rng(3)
    n, L = 64, 7
    pose = np.stack([rng.normal(0, 1, n), rng.normal(0, 1, n), rng.normal(0, 0.5, n)], 1)
    mp = np.zeros((5, L, n), np.float32)
    mp[0] = rng.normal(3, 1, (L, n)); mp[1] = rng.normal(-2, 1, (L, n))
    A = rng.normal(0, 0.3, (L, n, 2, 2))
    Pm = A @ np.swapaxes(A, -1, -2) + 0.05 * np.eye(2)
    mp[2], mp[3], mp[4] = Pm[..., 0, 0], Pm[..., 0, 1], Pm[..., 1, 1]
    obs_id = np.array([4, 1], np.int32)
    z = rng.normal(0, 2, (2, 2)).astype(np.float32)
    q = 0.01
    out, ll = orc.ekf_update(_rows(mp), pose[:, 0], pose[:, 1], pose[:, 2], None, obs_id, z[:, 0], z[:, 1], q)
    out = _planes(out)
    for i in range(n):
        tot = 0.0
        for k, l in enumerate(obs_id):
            mu64, P64, l64 = _ekf_f64(mp[0:2, l, i].astype(np.float64),
                                      np.array([[mp[2, l, i], mp[3, l, i]], [mp[3, l, i], mp[4, l, i]]], np.float64),
                                      pose[i].astype(np.float32).astype(np.float64), z[k].astype(np.float64), q)
            assert np.allclose(out[0:2, l, i], mu64, rtol=2e-4, atol=2e-4)
            assert np.allclose([out[2, l, i], out[3, l, i], out[4, l, i]], [P64[0, 0], P64[0, 1], P64[1, 1]], rtol=2e-3, atol=2e-5)
            tot += l64
        assert abs(ll[i] - tot) < 2e-3 * max(1.0, abs(tot))
    # landmarks that were not observed are untouched; covariance shrinks where observed
    for l in set(range(L)) - set(obs_id.tolist()):
        assert np.array_equal(out[:, l], mp[:, l])
    assert (out[2, 4] < mp[2, 4]).all() and (out[4, 1] < mp[4, 1]).all()


def test_ekf_first_sighting_gather_and_loglik_summation_order(orc):
    n, L = 5, 70
    rng = np.random.default_rng(4)
    mp = np.zeros((5, L, n), np.float32)
    mp[0:2] = rng.normal(0, 1, (2, L, n)); mp[2] = 0.2; mp[4] = 0.3; mp[3] = 0.01
    mp[2, 9] = -1.0                                   # landmark 9 never seen
    x = rng.normal(0, 1, n).astype(np.float32); y = rng.normal(0, 1, n).astype(np.float32)
    th = rng.normal(0, 1, n).astype(np.float32)
    ids = np.arange(L, dtype=np.int32)[::-1].copy()
    zx = rng.normal(0, 1, L).astype(np.float32); zy = rng.normal(0, 1, L).astype(np.float32)
    out, ll = orc.ekf_update(_rows(mp), x, y, th, None, ids, zx, zy, 0.02)
    out = _planes(out)
    # first sighting: world point = R^T-convention inverse of the observation, P = R
    s, c = orc.det_sincos(th)
    k9 = int(np.where(ids == 9)[0][0])
    assert np.allclose(out[0, 9], x + (c * zx[k9] + s * zy[k9]), atol=1e-6)
    assert np.allclose(out[1, 9], y + (c * zy[k9] - s * zx[k9]), atol=1e-6)
    assert (out[2, 9] == np.float32(0.02)).all() and (out[3, 9] == 0).all() and (out[4, 9] == np.float32(0.02)).all()
    # the specified summation order, restated independently: the term of landmark l (what a one-observation
    # call returns; 0 when l has no observation) goes to accumulator l mod 128 in order of l; accumulators j and
    # j+64 are added; then a 6-level xor butterfly over the 64 sums
    def tree(terms, ids_, nl):                         # terms: [nobs][n] float32, one per observation
        lane = np.zeros((128, terms.shape[1]), np.float32)
        by_landmark = {int(l): terms[k] for k, l in enumerate(ids_)}
        for l in range(nl):
            if l in by_landmark:
                lane[l % 128] = lane[l % 128] + by_landmark[l]
        t = lane[:64] + lane[64:]
        for sft in (1, 2, 4, 8, 16, 32):
            t = t + t[np.arange(64) ^ sft]
        return t[0]

    def terms_of(mp_, ids_, zx_, zy_):
        return np.stack([orc.ekf_update(_rows(mp_), x, y, th, None, ids_[k:k + 1], zx_[k:k + 1], zy_[k:k + 1], 0.02)[1]
                         for k in range(len(ids_))])

    assert np.array_equal(tree(terms_of(mp, ids, zx, zy), ids, L), ll)
    _, ll_shuffled = orc.ekf_update(_rows(mp), x, y, th, None, ids[::-1].copy(), zx[::-1].copy(), zy[::-1].copy(), 0.02)
    assert np.array_equal(ll_shuffled, ll)             # the order of the observation list is irrelevant
    # more than 128 landmarks: accumulators are reused in order (l and l+128 share one)
    L2 = 300
    mp2 = np.zeros((5, L2, n), np.float32)
    mp2[0:2] = rng.normal(0, 1, (2, L2, n)); mp2[2] = 0.2; mp2[4] = 0.3; mp2[3] = -0.02
    ids2 = rng.permutation(L2).astype(np.int32)[:290]
    zx2 = rng.normal(0, 1, 290).astype(np.float32); zy2 = rng.normal(0, 1, 290).astype(np.float32)
    _, ll2 = orc.ekf_update(_rows(mp2), x, y, th, None, ids2, zx2, zy2, 0.02)
    assert np.array_equal(tree(terms_of(mp2, ids2, zx2, zy2), ids2, L2), ll2)
    # fused gather: particle i continues from ancestor anc[i]'s map
    anc = np.array([2, 2, 0, 4, 4], np.int32)
    out_g, ll_g = orc.ekf_update(_rows(mp), x, y, th, anc, ids[:10], zx[:10], zy[:10], 0.02)
    out_ref, ll_ref = orc.ekf_update(_rows(mp[:, :, anc]), x, y, th, None, ids[:10], zx[:10], zy[:10], 0.02)
    assert np.array_equal(out_g, out_ref) and np.array_equal(ll_g, ll_ref)


def test_weights_known_answers(orc):
    score = np.array([3.0, 1.0, 1.0, 50.0], np.float32)
    logw, m = orc.logweight(score, None, 2.0)
    assert logw.tolist() == [-6.0, -2.0, -2.0, -100.0] and m == -2.0
    wq, s = orc.quantise_weights(logw, m)
    assert wq[1] == wq[2] == 1 << 32 and wq[3] == 0            # max weight is exactly 2^32, < e^-80 is exactly 0
    assert abs(int(wq[0]) / 2**32 - np.exp(-4.0)) < 1e-7 and s == int(wq.sum())
    ll = np.array([0.5, -0.5, 0.0, 0.0], np.float32)
    logw2, m2 = orc.logweight(score, ll, 2.0)
    assert logw2.tolist() == [-5.5, -2.5, -2.0, -100.0] and m2 == -2.0
    logw3, _ = orc.logweight(None, ll, 0.0)
    assert np.array_equal(logw3, ll)


def test_resample_identities(orc):
    n = 1000
    # uniform weights => every particle survives exactly once, whatever the comb offset
    for frame in range(5):
        anc = orc.resample(np.full(n, 1 << 32, np.uint64), seed=5, frame=frame)
        assert np.array_equal(anc, np.arange(n))
    # one-hot => everybody descends from the hot particle
    w = np.zeros(n, np.uint64); w[617] = 12345
    assert (orc.resample(w, 5, 0) == 617).all()
    # general weights: sorted ancestors, offspring counts within 1 of N*w/S, zero weight => no offspring
    rng = np.random.default_rng(8)
    w = (rng.random(n) ** 8 * 2**32).astype(np.uint64)
    w[::7] = 0
    anc = orc.resample(w, 77, 3)
    assert (np.diff(anc) >= 0).all()
    cnt = np.bincount(anc, minlength=n)
    expect = n * w.astype(np.float64) / w.sum()
    assert (np.abs(cnt - expect) < 1.0 + 1e-9).all() and (cnt[w == 0] == 0).all() and cnt.sum() == n


def test_resample_exact_integer_arithmetic_and_sharding(orc):
    rng = np.random.default_rng(9)
    n = 4096
    w = rng.integers(0, 1 << 32, n, dtype=np.uint64)
    w[100:200] = 0
    cdf = orc.prefix_sum(w)
    total = int(cdf[-1])
    for frame in (0, 1, 2):
        u = orc.comb_offset(42, frame, total)
        first = orc.offspring_offsets(cdf, 0, total, u, n)
        # Python big-int restatement of first[i] = ceil((N*C_excl - u) / total), clamped at 0
        cex = np.concatenate([[0], cdf[:-1]]).astype(object)
        want = [0 if int(c) * n <= u else (int(c) * n - u - 1) // total + 1 for c in cex]
        assert first.tolist() == want
        anc = orc.ancestors(first, 0, n)
        # brute-force definition: tooth j at j*total + u falls in [N*C_excl(i), N*C_incl(i))
        teeth = np.array([j * total + u for j in range(n)], object)
        edges = np.array([int(c) * n for c in cdf], object)
        brute = np.searchsorted(np.array(edges, np.float64), np.array(teeth, np.float64), side="right")
        # float64 searchsorted can be off at exact ties; verify with exact integers instead
        for j in range(0, n, 37):
            i = int(anc[j])
            lo = int(cex[i]) * n
            hi = int(cdf[i]) * n
            assert lo <= int(teeth[j]) < hi
        # sharding: two ranks with their own local CDFs + base offsets produce the same `first`
        h = 1500
        c0, c1 = orc.prefix_sum(w[:h]), orc.prefix_sum(w[h:])
        f0 = orc.offspring_offsets(c0, 0, total, u, n)
        f1 = orc.offspring_offsets(c1, int(c0[-1]), total, u, n)
        assert np.array_equal(np.concatenate([f0, f1]), first)
        a0 = orc.ancestors(first, 0, 2048)
        a1 = orc.ancestors(first, 2048, 2048)
        assert np.array_equal(np.concatenate([a0, a1]), anc)


def test_resample_large_population_no_overflow(orc):
    """N*C exceeds 64 bits at the BASELINE sizes (8M particles x 2^32 weights): 128-bit path."""
    n = 1 << 16
    w = np.full(n, (1 << 32), np.uint64)
    cdf = orc.prefix_sum(w)
    total = int(cdf[-1])
    n_total = 8 * 1024 * 1024   # pretend this shard is the first 64k of 8M equal-weight particles
    big_total = (1 << 32) * n_total
    u = orc.comb_offset(1, 1, big_total)
    first = orc.offspring_offsets(cdf, 0, big_total, u, n_total)
    assert first[0] == 0 and np.array_equal(first[1:], np.arange(1, n))   # equal weights: one tooth each
    assert total == (1 << 32) * n


def test_scorer_rounding_identity_holds_for_every_float(orc):
    """The HIP scorer selects cells with trunc(v + copysign(0.5 - 1 ulp, v)) instead of roundf(v) (main.c:483, 501):
    the two agree on ALL 2^32 float bit patterns (NaN aside: both convert to cell 0 on the device)."""
    assert orc.lib().orc_round_trick_mismatches(0, 0xFFFFFFFF, 1) == 0
