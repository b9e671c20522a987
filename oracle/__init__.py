"""oracle — CPU checker for the HIP engine.  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product package never does, and has no CPU fallback.

``oracle.lib()`` loads ``oracle/_build/liboracle.so`` (the plain-C restatement in
``slam_oracle.c`` / ``slam_oracle_pf.c``, built by ``make -C oracle``) and the thin numpy
wrappers below call it.  Parity status is documented in ``slam_oracle.h``: rows A1-A8 are pinned
against the compiled reference (``tests/test_oracle_vs_reference.py``); the particle-filter
stages have no reference counterpart and are "parity unpinned".
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
BUILD = HERE / "_build"
REF = HERE / "_ref"

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


class GridMeta(C.Structure):
    """Mirror of ``orc_grid_meta`` (slam_oracle.h)."""

    _fields_ = [
        ("rows", C.c_int),
        ("cols", C.c_int),
        ("ld", C.c_int),
        ("pixel", C.c_float),
        ("min_x", C.c_float),
        ("min_y", C.c_float),
    ]


def build(ref: bool = True) -> None:
    """Compile the oracle (and, when /root/reference exists, the reference harness)."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", str(HERE), *targets], check=True)


_LIB = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    so = BUILD / "liboracle.so"
    if not so.exists():
        build(ref=False)
    L = C.CDLL(str(so))
    P = C.POINTER
    L.orc_beam_angles.argtypes = [C.c_float, C.c_float, C.c_int, _f32p]
    L.orc_clean_scan.argtypes = [_f32p, _f32p, C.c_int, C.c_float, C.c_float, _f32p, _f32p]
    L.orc_clean_scan.restype = C.c_int
    L.orc_transform.argtypes = [_f32p, _f32p, C.c_int, _f32p, _f32p, _f32p]
    L.orc_local_map.argtypes = [_f32p, _f32p, C.c_int, _f32p, _f32p, C.c_int, C.c_float, _f32p, _f32p]
    L.orc_local_map.restype = C.c_int
    L.orc_rasterise.argtypes = [_f32p, _f32p, C.c_int, C.c_float, C.c_int, C.c_int, _i32p, P(GridMeta)]
    for name in ("orc_edt_gather", "orc_edt_scatter", "orc_edt_window"):
        getattr(L, name).argtypes = [_i32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float]
    L.orc_score_pose.argtypes = [P(GridMeta), _f32p, _f32p, _f32p, C.c_int, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_void_p, P(C.c_int)]
    L.orc_score_pose.restype = C.c_float
    L.orc_score_poses.argtypes = [P(GridMeta), _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p, C.c_int, _f32p, _i32p]
    L.orc_fastmatch.argtypes = [P(GridMeta), _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, P(C.c_int),
                                P(C.c_float)]
    # ---- particle-filter specification (slam_oracle_pf.c)
    _i64, _u64, _u32 = C.c_int64, C.c_uint64, C.c_uint32
    L.orc_det_sincosf_array.argtypes = [_f32p, C.c_int, _f32p, _f32p]
    L.orc_det_expf_array.argtypes = [_f32p, C.c_int, _f32p]
    L.orc_det_logf_array.argtypes = [_f32p, C.c_int, _f32p]
    L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
    L.orc_score_poses_det.argtypes = [P(GridMeta), _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p, C.c_int, _f32p, _i32p]
    L.orc_motion_sample.argtypes = [_f32p, _f32p, _f32p, C.c_void_p, _f32p, _f32p, _f32p, C.c_int, _i64, _f32p, _f32p,
                                    _u64, _u32]
    L.orc_ekf_update.argtypes = [_f32p, _f32p, _i64, C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_void_p, C.c_int, _i32p,
                                 _f32p, _f32p, C.c_int, C.c_float, _f32p]
    L.orc_logweight.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, _f32p, P(C.c_float)]
    L.orc_quantise_weights.argtypes = [_f32p, C.c_float, C.c_int, _u64p, P(_u64)]
    L.orc_round_trick_mismatches.argtypes = [_u32, _u32, _u32]
    L.orc_round_trick_mismatches.restype = _u64
    L.orc_ess_terms.argtypes = [_u64p, C.c_int, P(_u64), P(_u64)]
    L.orc_ess_resample.argtypes = [_u64, _u64, _i64, _u32]
    L.orc_ess_resample.restype = C.c_int
    L.orc_logweight_carry.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, _f32p, P(C.c_float)]
    L.orc_weight_carry.argtypes = [_f32p, C.c_float, C.c_int, _f32p]
    L.orc_prefix_sum.argtypes = [_u64p, C.c_int, _u64p]
    L.orc_comb_offset.argtypes = [_u64, _u32, _u64]
    L.orc_comb_offset.restype = _u64
    L.orc_offspring_offsets.argtypes = [_u64p, C.c_int, _u64, _u64, _u64, _i64, _i32p]
    L.orc_ancestors.argtypes = [_i32p, _i64, _i64, C.c_int, _i32p]
    _LIB = L
    return L


# ------------------------------------------------------------------ numpy-level helpers


def meta(rows: int, cols: int, ld: int, pixel: float, min_x: float, min_y: float) -> GridMeta:
    return GridMeta(int(rows), int(cols), int(ld), np.float32(pixel), np.float32(min_x), np.float32(min_y))


def beam_angles(angle_min: float, angle_inc: float, n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().orc_beam_angles(angle_min, angle_inc, n, out)
    return out


def clean_scan(ranges, angles, range_min=0.023, usable=24):
    n = len(ranges)
    x = np.empty(n, np.float32)
    y = np.empty(n, np.float32)
    m = lib().orc_clean_scan(np.ascontiguousarray(ranges, np.float32), np.ascontiguousarray(angles, np.float32), n,
                             range_min, usable, x, y)
    return x[:m].copy(), y[:m].copy()


def transform(x, y, pose):
    tx = np.empty(len(x), np.float32)
    ty = np.empty(len(x), np.float32)
    lib().orc_transform(np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32), len(x),
                        np.ascontiguousarray(pose, np.float32), tx, ty)
    return tx, ty


def local_map(map_x, map_y, tx, ty, border=1.0):
    lx = np.empty(max(len(map_x), 1), np.float32)
    ly = np.empty(max(len(map_x), 1), np.float32)
    m = lib().orc_local_map(np.ascontiguousarray(map_x, np.float32), np.ascontiguousarray(map_y, np.float32),
                            len(map_x), np.ascontiguousarray(tx, np.float32), np.ascontiguousarray(ty, np.float32),
                            len(tx), border, lx, ly)
    return lx[:m].copy(), ly[:m].copy()


def rasterise(px, py, pixel, ld):
    grid = np.empty((ld, ld), np.int32)
    m = GridMeta()
    lib().orc_rasterise(np.ascontiguousarray(px, np.float32), np.ascontiguousarray(py, np.float32), len(px), pixel, ld,
                        ld, grid, C.byref(m))
    return grid, m


def edt(occ: np.ndarray, rows: int, cols: int, cap: float = 10.0, variant: str = "window", out=None) -> np.ndarray:
    """EDT of ``occ[:rows, :cols]`` (int32, row-major, leading dimension occ.shape[1])."""
    occ = np.ascontiguousarray(occ, np.int32)
    if out is None:
        out = np.zeros(occ.shape, np.float32)
    getattr(lib(), "orc_edt_" + variant)(occ, out, occ.shape[1], rows, cols, cap)
    return out


def score_pose(m: GridMeta, edt_grid, bx, by, x, y, ct, st):
    """-> (score, count, hits[count]) for one pose with the given heading cos/sin."""
    hits = np.empty(len(bx), np.float32)
    cnt = C.c_int(0)
    s = lib().orc_score_pose(C.byref(m), np.ascontiguousarray(edt_grid, np.float32),
                             np.ascontiguousarray(bx, np.float32), np.ascontiguousarray(by, np.float32), len(bx),
                             x, y, ct, st, hits.ctypes.data_as(C.c_void_p), C.byref(cnt))
    return np.float32(s), cnt.value, hits[: cnt.value].copy()


def score_poses(m: GridMeta, edt_grid, bx, by, x, y, theta):
    n = len(x)
    score = np.empty(n, np.float32)
    count = np.empty(n, np.int32)
    lib().orc_score_poses(C.byref(m), np.ascontiguousarray(edt_grid, np.float32), np.ascontiguousarray(bx, np.float32),
                          np.ascontiguousarray(by, np.float32), len(bx), np.ascontiguousarray(x, np.float32),
                          np.ascontiguousarray(y, np.float32), np.ascontiguousarray(theta, np.float32), n, score, count)
    return score, count


def fastmatch(m: GridMeta, edt_grid, bx, by, pose, res):
    """-> (pose[3], best_hits (full scratch, nbeams long), best_hits_size, best_score)"""
    out = np.empty(3, np.float32)
    hits = np.zeros(max(len(bx), 1), np.float32)
    n = C.c_int(0)
    sc = C.c_float(0)
    lib().orc_fastmatch(C.byref(m), np.ascontiguousarray(edt_grid, np.float32), np.ascontiguousarray(bx, np.float32),
                        np.ascontiguousarray(by, np.float32), len(bx), np.ascontiguousarray(pose, np.float32),
                        np.ascontiguousarray(res, np.float32), out, hits, C.byref(n), C.byref(sc))
    return out, hits, n.value, np.float32(sc.value)


def libm_cos_sin(theta):
    """cosf/sinf of the platform libm — what the reference itself calls for its headings."""
    m = C.CDLL("libm.so.6")
    m.cosf.restype = C.c_float
    m.cosf.argtypes = [C.c_float]
    m.sinf.restype = C.c_float
    m.sinf.argtypes = [C.c_float]
    th = np.atleast_1d(np.asarray(theta, np.float32))
    return (np.array([m.cosf(float(t)) for t in th], np.float32), np.array([m.sinf(float(t)) for t in th], np.float32))


def run_tool(name: str, *args, **kw) -> subprocess.CompletedProcess:
    """Run one of the oracle's own executables (gen_dataset, main_cpu)."""
    exe = BUILD / name
    if not exe.exists():
        build(ref=False)
    return subprocess.run([str(exe), *map(str, args)], check=True, **kw)


# ------------------------------------------------------------------ particle-filter specification


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _opt_i32(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, np.int32)
    return a, a.ctypes.data_as(C.c_void_p)


def det_sincos(a):
    a = _f32(np.atleast_1d(a))
    s = np.empty_like(a)
    c = np.empty_like(a)
    lib().orc_det_sincosf_array(a, len(a), s, c)
    return s, c


def det_exp(x):
    x = _f32(np.atleast_1d(x))
    y = np.empty_like(x)
    lib().orc_det_expf_array(x, len(x), y)
    return y


def det_log(x):
    x = _f32(np.atleast_1d(x))
    y = np.empty_like(x)
    lib().orc_det_logf_array(x, len(x), y)
    return y


def philox(ctr, key):
    out = np.empty(4, np.uint32)
    lib().orc_philox4x32_10(np.ascontiguousarray(ctr, np.uint32), np.ascontiguousarray(key, np.uint32), out)
    return out


def score_poses_det(m: GridMeta, edt_grid, bx, by, x, y, theta):
    n = len(x)
    score = np.empty(n, np.float32)
    count = np.empty(n, np.int32)
    lib().orc_score_poses_det(C.byref(m), _f32(edt_grid), _f32(bx), _f32(by), len(bx), _f32(x), _f32(y), _f32(theta), n,
                              score, count)
    return score, count


def motion_sample(src_x, src_y, src_th, anc, n, first_id, dp, sigma, seed, frame):
    x, y, th = (np.empty(n, np.float32) for _ in range(3))
    keep, ancp = _opt_i32(anc)
    lib().orc_motion_sample(_f32(src_x), _f32(src_y), _f32(src_th), ancp, x, y, th, n, first_id, _f32(dp), _f32(sigma),
                            seed, frame)
    return x, y, th


def ekf_update(map_in, x, y, th, anc, obs_id, obs_zx, obs_zy, meas_var, n=None):
    """map_in: float32 [rows][5][L] (one row per particle).  Returns (map_out, loglik[n]); out of place."""
    map_in = _f32(map_in)
    rows, five, L_ = map_in.shape
    n = len(x) if n is None else n
    if five != 5 or rows < n or (len(obs_id) and (np.min(obs_id) < 0 or np.max(obs_id) >= L_)):
        raise ValueError("ekf_update: map must be [rows >= n][5][L] and landmark ids < L")
    out = map_in.copy()
    ll = np.empty(n, np.float32)
    keep, ancp = _opt_i32(anc)
    lib().orc_ekf_update(map_in, out, 5 * L_, L_, L_, _f32(x), _f32(y), _f32(th), ancp, n,
                         np.ascontiguousarray(obs_id, np.int32), _f32(obs_zx), _f32(obs_zy), len(obs_id), meas_var, ll)
    return out, ll


def logweight(score, loglik, gain):
    n = len(score) if score is not None else len(loglik)
    logw = np.empty(n, np.float32)
    m = C.c_float(0)
    sp = _f32(score).ctypes.data_as(C.c_void_p) if score is not None else None
    keep_s = _f32(score) if score is not None else None
    keep_l = _f32(loglik) if loglik is not None else None
    lib().orc_logweight(keep_s.ctypes.data_as(C.c_void_p) if keep_s is not None else None,
                        keep_l.ctypes.data_as(C.c_void_p) if keep_l is not None else None, gain, n, logw, C.byref(m))
    return logw, np.float32(m.value)


def logweight_carry(score, loglik, gain, carry):
    """logw = carry + (loglik - gain*score) (carry None: no carry term) and its maximum."""
    n = len(score) if score is not None else len(loglik)
    logw = np.empty(n, np.float32)
    m = C.c_float(0)
    keep = [(_f32(a) if a is not None else None) for a in (score, loglik, carry)]
    ptr = [(a.ctypes.data_as(C.c_void_p) if a is not None else None) for a in keep]
    lib().orc_logweight_carry(ptr[0], ptr[1], gain, ptr[2], n, logw, C.byref(m))
    return logw, np.float32(m.value)


def weight_carry(logw, m):
    out = np.empty(len(logw), np.float32)
    lib().orc_weight_carry(_f32(logw), m, len(logw), out)
    return out


def ess_terms(wq):
    s, q = C.c_uint64(0), C.c_uint64(0)
    lib().orc_ess_terms(np.ascontiguousarray(wq, np.uint64), len(wq), C.byref(s), C.byref(q))
    return int(s.value), int(q.value)


def ess_resample(s16, q16, n_total, frac_q16):
    return bool(lib().orc_ess_resample(s16, q16, n_total, frac_q16))


def ess_frac_q16(frac: float) -> int:
    """The gate threshold as the engine quantises it: round(frac * 65536); 0 = resample every frame."""
    return int(np.rint(np.float32(frac) * np.float32(65536.0))) if 0.0 < frac < 1.0 else 0


def quantise_weights(logw, m):
    wq = np.empty(len(logw), np.uint64)
    s = C.c_uint64(0)
    lib().orc_quantise_weights(_f32(logw), m, len(logw), wq, C.byref(s))
    return wq, int(s.value)


def prefix_sum(wq):
    cdf = np.empty(len(wq), np.uint64)
    lib().orc_prefix_sum(np.ascontiguousarray(wq, np.uint64), len(wq), cdf)
    return cdf


def comb_offset(seed, frame, total):
    return int(lib().orc_comb_offset(seed, frame, total))


def offspring_offsets(cdf, base, total, comb_u, n_total):
    first = np.empty(len(cdf), np.int32)
    lib().orc_offspring_offsets(np.ascontiguousarray(cdf, np.uint64), len(cdf), base, total, comb_u, n_total, first)
    return first


def ancestors(first_all, slot0, nslots):
    anc = np.empty(nslots, np.int32)
    lib().orc_ancestors(np.ascontiguousarray(first_all, np.int32), len(first_all), slot0, nslots, anc)
    return anc


def resample(wq, seed, frame):
    """Whole systematic resample of one unsharded population: -> ancestors[n]."""
    cdf = prefix_sum(wq)
    total = int(cdf[-1])
    first = offspring_offsets(cdf, 0, total, comb_offset(seed, frame, total), len(wq))
    return ancestors(first, 0, len(wq))
