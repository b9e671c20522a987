"""MI355X-native scan-matching / particle-filter engine — Python binding of the C ABI.

The product is the shared library ``lib/libslam_hip.so`` (hand-written HIP kernels for gfx950 behind
the plain-C interface of ``include/slam_hip.h``).  This module is only a ctypes view of that ABI so
that tests, ``bench.py`` and Python hosts can call it; the C host program under ``host/`` links the
same library directly.  There is NO CPU fallback: importing works without a GPU (so the build check
can load the library and verify its symbols), but creating an ``Engine`` raises when no gfx950 device
is present, and a missing library raises at import of the symbols.

Reference mapping (paths relative to the reference repository):
  Engine.edt_host / grid_upload  <- euclidean_distance_transform{,2}  Subsystem_1/main_accelerated.c:215-283
  Engine.fastmatch               <- FastMatch / FastMatch2            Subsystem_1/main.c:381-809
  Engine.score_poses*            <- the per-pose body of FastMatch    Subsystem_1/main.c:459-518
  motion / EKF / weights / resample: no reference counterpart
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
import os as _os

# SLAM_HIP_LIB: another build of the same library (kernel-tuning measurements); the product is lib/libslam_hip.so
LIB_PATH = Path(_os.environ["SLAM_HIP_LIB"]) if _os.environ.get("SLAM_HIP_LIB") else PKG_DIR / "lib" / "libslam_hip.so"
HEADER_PATH = PKG_DIR.parent / "include" / "slam_hip.h"

SLAM_OK = 0
SLAM_MAX_BEAMS = 4096
SLAM_MAX_OBS = 8192
EKF_OBS_CHUNK = 32


class SlamError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: {status_string(status)} ({status})" + (f": {detail}" if detail else ""))


class GridMeta(C.Structure):
    """``slam_grid_meta`` — rows, cols, ld, pixel, min_x, min_y (reference: MyGrid, main.c:200-213)."""

    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("ld", C.c_int32), ("pixel", C.c_float),
                ("min_x", C.c_float), ("min_y", C.c_float)]


_vp, _i, _i64, _u64, _u32, _f = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32, C.c_float
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); every entry point include/slam_hip.h declares
SIGNATURES = {
    "slam_abi_version": (_i, []),
    "slam_status_string": (C.c_char_p, [_i]),
    "slam_last_error": (C.c_char_p, [_vp]),
    "slam_engine_create": (_i, [_i, C.POINTER(_vp)]),
    "slam_engine_destroy": (_i, [_vp]),
    "slam_engine_set_stream": (_i, [_vp, _vp]),
    "slam_engine_sync": (_i, [_vp]),
    "slam_profile_enable": (_i, [_vp, _i]),
    "slam_profile_read": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "slam_profile_bracket_overhead": (_i, [_vp, C.POINTER(C.c_double)]),
    "slam_profile_copy_ceiling": (_i, [_vp, _vp, _vp, _i64, _i, _i, C.POINTER(C.c_double)]),
    "slam_edt_dev": (_i, [_vp, _vp, _i, _i, _i, _f, _vp]),
    "slam_edt_host": (_i, [_vp, _vp, _i, _i, _i, _f, _vp]),
    "slam_grid_upload_host": (_i, [_vp, _i, _vp, C.POINTER(GridMeta), _f, _vp]),
    "slam_grid_set_dev": (_i, [_vp, _i, _vp, C.POINTER(GridMeta)]),
    "slam_grid_set_meta": (_i, [_vp, _i, C.POINTER(GridMeta)]),
    "slam_scan_upload_host": (_i, [_vp, _vp, _vp, _i]),
    "slam_scan_set_dev": (_i, [_vp, _vp, _vp, _i]),
    "slam_score_poses_cs_dev": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "slam_score_poses_dev": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "slam_score_poses_cs_host": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "slam_score_poses_host": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "slam_pose_hits_host": (_i, [_vp, _i, _f, _f, _f, _f, _vp, C.POINTER(C.c_int32)]),
    "slam_fastmatch_host": (_i, [_vp, _i, _fp, _fp, _fp, _vp, C.POINTER(C.c_int32), _fp]),
    "slam_motion_sample_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _fp, _fp, _u64, _u32]),
    "slam_motion_score_dev": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _fp, _fp, _u64, _u32, _vp, _vp]),
    "slam_obs_upload_host": (_i, [_vp, _vp, _vp, _vp, _i, _i]),
    "slam_obs_set_dev": (_i, [_vp, _vp, _vp, _i]),
    "slam_logweight_ekf_dev": (_i, [_vp, _vp, _f, _i, _vp, _vp]),
    "slam_ekf_update_dev": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _vp]),
    "slam_logweight_dev": (_i, [_vp, _vp, _vp, _f, _i, _vp, _vp]),
    "slam_quantise_weights_dev": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "slam_quantise_scan_dev": (_i, [_vp, _vp, _vp, _i, _vp]),
    "slam_offspring_from_scan_dev": (_i, [_vp, _i, _vp, _vp, _u64, _u32, _i64, _vp]),
    "slam_offspring_from_scan_sharded_dev": (_i, [_vp, _i, _vp, _i, _i, _u64, _u32, _i64, _vp]),
    "slam_prefix_sum_dev": (_i, [_vp, _vp, _i, _vp]),
    "slam_offspring_offsets_dev": (_i, [_vp, _vp, _i, _vp, _vp, _u64, _u32, _i64, _vp]),
    "slam_ancestors_dev": (_i, [_vp, _vp, _i64, _i64, _i, _vp]),
    "slam_comb_offset": (_u64, [_u64, _u32, _u64]),
    "slam_ancestors_from_scan_dev": (_i, [_vp, _i, _u64, C.c_uint32, _vp]),
    "slam_ancestors_sharded_dev": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "slam_exchange_plan_host": (_i, [_vp, _i, _vp]),
    "slam_migrate_pack_dev": (_i, [_vp, _i, _i, _i, _vp, _vp, _i64, _vp, _i64, _i, _i, _vp]),
    "slam_migrate_unpack_dev": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i64, _vp, _i64, _i, _i]),
    "slam_argmax_dev": (_i, [_vp, _vp, _i, _vp, _vp]),
    "slam_gather_f32_dev": (_i, [_vp, _vp, _vp, _i, _vp]),
    "slam_gather_map_dev": (_i, [_vp, _vp, _vp, _i64, _i64, _i, _i, _i, _vp, _i]),
    "slam_mapper_create": (_i, [_vp, _i, _f, _f, C.POINTER(_vp)]),
    "slam_mapper_create_ex": (_i, [_vp, _i, _f, _f, _vp, C.POINTER(_vp)]),
    "slam_mapper_params_default": (None, [_vp]),
    "slam_mapper_destroy": (_i, [_vp]),
    "slam_mapper_first_frame": (_i, [_vp, _vp]),
    "slam_mapper_next_frame": (_i, [_vp, _vp, _fp]),
    "slam_mapper_get_map_host": (_i, [_vp, _vp, _vp, C.c_int32, C.POINTER(C.c_int32)]),
    "slam_exchange_set_capacity": (_i, [_vp, _i]),
    "slam_ekf_form_set": (_i, [_vp, _i]),
    "slam_ekf_form_counts": (_i, [_vp, _vp]),
    "slam_selftest_reciprocal": (_i, [_vp, _vp, _vp]),
    "slam_frame_fusion_set": (_i, [_vp, _i]),
    "slam_frame_fusion_count": (_i, [_vp, _vp]),
    "slam_frame_front_last": (_i, [_vp, _vp]),
    "slam_ekf_inplace_form_set": (_i, [_vp, _i]),
    "slam_pf_paged_set": (_i, [_vp, _i]),
    "slam_pf_is_paged": (_i, [_vp]),
    "slam_pf_set_map_dev": (_i, [_vp, _vp, _i64, _i]),
    "slam_ekf_inplace_form_counts": (_i, [_vp, _vp]),
    "slam_resample_gate_set": (_i, [_vp, _f]),
    "slam_resample_happened_host": (_i, [_vp, C.POINTER(C.c_int)]),
    "slam_comm_unique_id": (_i, [_vp]),
    "slam_comm_create_rccl": (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "slam_local_group_create": (_i, [_i, C.POINTER(_vp)]),
    "slam_local_group_destroy": (_i, [_vp]),
    "slam_comm_create_local": (_i, [_vp, _vp, _i, C.POINTER(_vp)]),
    "slam_comm_abort": (_i, [_vp]),
    "slam_comm_rank": (_i, [_vp]),
    "slam_comm_world": (_i, [_vp]),
    "slam_comm_destroy": (_i, [_vp]),
    "slam_pf_create_sharded": (_i, [_vp, _vp, _vp, _i, C.POINTER(_vp)]),
    "slam_pf_mean": (_i, [_vp, _f, _fp]),
    "slam_pf_rows_received": (_i, [_vp]),
    "slam_pf_device_view": (_i, [_vp, _vp]),
    "slam_pf_frames_resampled": (_i64, [_vp]),
    "slam_pf_layout_changes": (_i64, [_vp]),
    "slam_pf_create": (_i, [_vp, _vp, C.POINTER(_vp)]),
    "slam_pf_destroy": (_i, [_vp]),
    "slam_pf_reset": (_i, [_vp, _fp]),
    "slam_pf_set_poses_host": (_i, [_vp, _vp, _vp, _vp]),
    "slam_pf_set_map_host": (_i, [_vp, _vp]),
    "slam_pf_step": (_i, [_vp, _i, _fp, _i]),
    "slam_pf_best": (_i, [_vp, _fp, _fp, C.POINTER(C.c_int32)]),
    "slam_pf_get_poses_host": (_i, [_vp, _vp, _vp, _vp]),
    "slam_pf_get_map_host": (_i, [_vp, _vp]),
    "slam_pf_get_map_rows_host": (_i, [_vp, _vp, _i, _vp]),
    "slam_pf_paged_device_view": (_i, [_vp, _vp]),
    "slam_pf_layout": (_i, [_vp]),
    "slam_pf_split_device_view": (_i, [_vp, _vp]),
}

_LIB = None


def load_library() -> C.CDLL:
    """Load libslam_hip.so and bind every C-ABI symbol.  Raises if the library or a symbol is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP engine has not been built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C hardware-acceleration-of-lidar-slam_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def status_string(status: int) -> str:
    return load_library().slam_status_string(status).decode()


def _ptr(a):
    """Device pointer of a torch tensor / raw int, host pointer of a numpy array, or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(a.data_ptr())


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _f3(v):
    return (C.c_float * 3)(*[float(np.float32(x)) for x in v])


class Engine:
    """One engine per GPU / host thread (``slam_engine``)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.slam_engine_create(device, C.byref(h))
        if rc != SLAM_OK:
            raise SlamError(rc, "slam_engine_create", self.lib.slam_last_error(None).decode())
        self.h = h
        self.nbeams = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.slam_engine_destroy(self.h)
            self.h = None

    __del__ = close

    def _ck(self, rc, where):
        if rc != SLAM_OK:
            raise SlamError(rc, where, self.lib.slam_last_error(self.h).decode())

    def set_stream(self, stream_ptr: int | None):
        """stream_ptr: a hipStream_t as an integer (0 = HIP's default stream); None = the engine's own stream."""
        p = C.c_void_p(-1 & (2 ** (8 * C.sizeof(C.c_void_p)) - 1)) if stream_ptr is None else C.c_void_p(stream_ptr)
        self._ck(self.lib.slam_engine_set_stream(self.h, p), "set_stream")

    def sync(self):
        self._ck(self.lib.slam_engine_sync(self.h), "sync")

    PROF_SCORE, PROF_EDT, PROF_EKF, PROF_WEIGHTS, PROF_SCAN, PROF_ANCESTORS, PROF_PLAN, PROF_PACK, PROF_UNPACK = range(9)
    PROF_COLLECTIVES, PROF_PAGES, PROF_EKF_TAIL, PROF_COUNT = 9, 10, 11, 12
    PROF_NAMES = ("score", "edt", "ekf", "weights", "scan", "ancestors", "plan", "pack", "unpack", "collectives", "pages", "ekf_tail")

    def profile_enable(self, *kernels):
        """profile_enable(PROF_EKF, ...) times those kernels; profile_enable() switches timing off."""
        mask = 0
        for k in kernels:
            mask |= 1 << k
        self._ck(self.lib.slam_profile_enable(self.h, mask), "profile_enable")

    def profile_read(self, kernel: int):
        """-> (total milliseconds, launches) of that kernel since the last read (HIP events on the engine stream)."""
        ms, n = C.c_double(0), C.c_int64(0)
        self._ck(self.lib.slam_profile_read(self.h, kernel, C.byref(ms), C.byref(n)), "profile_read")
        return ms.value, n.value

    def profile_bracket_overhead(self) -> float:
        """Milliseconds an empty event bracket measures on the engine's stream."""
        ms = C.c_double(0)
        self._ck(self.lib.slam_profile_bracket_overhead(self.h, C.byref(ms)), "profile_bracket_overhead")
        return ms.value

    def profile_copy_ceiling(self, d_src, d_dst, rows: int, plane_stride: int, reps: int = 12) -> float:
        """``slam_profile_copy_ceiling``: milliseconds of one pure copy of `rows` rows of 5 x plane_stride floats with the
        landmark update's access shape."""
        ms = C.c_double(0)
        self._ck(self.lib.slam_profile_copy_ceiling(self.h, _ptr(d_src), _ptr(d_dst), rows, plane_stride, reps, C.byref(ms)),
                 "profile_copy_ceiling")
        return ms.value

    # ---------------------------------------------------------------- host-buffer level (drop-in)
    def edt_host(self, occ: np.ndarray, rows: int, cols: int, cap: float = 10.0, out: np.ndarray | None = None):
        occ = _np(occ, np.int32)
        if out is None:
            out = np.zeros(occ.shape, np.float32)
        assert out.flags.c_contiguous and out.dtype == np.float32 and out.shape == occ.shape
        self._ck(self.lib.slam_edt_host(self.h, _ptr(occ), occ.shape[1], rows, cols, cap, _ptr(out)), "edt_host")
        return out

    def grid_upload(self, slot: int, occ: np.ndarray, meta: GridMeta, cap: float = 10.0, want_edt: bool = False):
        occ = _np(occ, np.int32)
        assert occ.ndim == 2 and occ.shape[1] == meta.ld and occ.shape[0] >= meta.rows
        out = np.zeros(occ.shape, np.float32) if want_edt else None
        self._ck(self.lib.slam_grid_upload_host(self.h, slot, _ptr(occ), C.byref(meta), cap, _ptr(out)), "grid_upload")
        return out

    def grid_set_dev(self, slot: int, d_edt, meta: GridMeta):
        self._ck(self.lib.slam_grid_set_dev(self.h, slot, _ptr(d_edt), C.byref(meta)), "grid_set_dev")
        # the engine adopts the buffer without copying it: keep the caller's array alive as long as the slot names it (a
        # temporary tensor would go back to its allocator and be handed out again under the engine's feet)
        self.__dict__.setdefault("_adopted", {})[("grid", slot)] = d_edt

    def grid_set_meta(self, slot: int, meta: GridMeta):
        self._ck(self.lib.slam_grid_set_meta(self.h, slot, C.byref(meta)), "grid_set_meta")

    def scan_upload(self, bx, by):
        bx, by = _np(bx, np.float32), _np(by, np.float32)
        assert bx.shape == by.shape and bx.ndim == 1
        self._ck(self.lib.slam_scan_upload_host(self.h, _ptr(bx), _ptr(by), len(bx)), "scan_upload")
        self.nbeams = len(bx)

    def scan_set_dev(self, d_bx, d_by, nbeams: int):
        self._ck(self.lib.slam_scan_set_dev(self.h, _ptr(d_bx), _ptr(d_by), nbeams), "scan_set_dev")
        self.nbeams = nbeams
        self.__dict__.setdefault("_adopted", {})["scan"] = (d_bx, d_by)

    def score_poses_cs_host(self, slot, x, y, ct, st):
        x, y, ct, st = (_np(a, np.float32) for a in (x, y, ct, st))
        score = np.empty(len(x), np.float32)
        count = np.empty(len(x), np.int32)
        self._ck(self.lib.slam_score_poses_cs_host(self.h, slot, _ptr(x), _ptr(y), _ptr(ct), _ptr(st), len(x),
                                                   _ptr(score), _ptr(count)), "score_poses_cs_host")
        return score, count

    def score_poses_host(self, slot, x, y, theta):
        x, y, theta = (_np(a, np.float32) for a in (x, y, theta))
        score = np.empty(len(x), np.float32)
        count = np.empty(len(x), np.int32)
        self._ck(self.lib.slam_score_poses_host(self.h, slot, _ptr(x), _ptr(y), _ptr(theta), len(x), _ptr(score),
                                                _ptr(count)), "score_poses_host")
        return score, count

    def pose_hits(self, slot, x, y, ct, st):
        hits = np.zeros(max(self.nbeams, 1), np.float32)
        n = C.c_int32(0)
        self._ck(self.lib.slam_pose_hits_host(self.h, slot, x, y, ct, st, _ptr(hits), C.byref(n)), "pose_hits")
        return hits[: n.value].copy(), n.value

    def fastmatch(self, slot, pose, res, hits: np.ndarray | None = None):
        """-> (pose[3], hit buffer, best_hits_size, best_score).  `hits` (float32, >= nbeams) plays the role of
        the reference's persistent FastMatchParameters.bestHits; a zeroed one is made when omitted."""
        out = (C.c_float * 3)()
        if hits is None:
            hits = np.zeros(max(self.nbeams, 1), np.float32)
        n = C.c_int32(-1)
        sc = C.c_float(0)
        self._ck(self.lib.slam_fastmatch_host(self.h, slot, _f3(pose), _f3(res), out, _ptr(hits), C.byref(n),
                                              C.byref(sc)), "fastmatch")
        return np.array(list(out), np.float32), hits, n.value, np.float32(sc.value)

    # ---------------------------------------------------------------- device level (async on the engine stream)
    def edt_dev(self, d_occ, ld, rows, cols, cap, d_out):
        self._ck(self.lib.slam_edt_dev(self.h, _ptr(d_occ), ld, rows, cols, cap, _ptr(d_out)), "edt_dev")

    def score_poses_dev(self, slot, d_x, d_y, d_th, n, d_score, d_count):
        self._ck(self.lib.slam_score_poses_dev(self.h, slot, _ptr(d_x), _ptr(d_y), _ptr(d_th), n, _ptr(d_score),
                                               _ptr(d_count)), "score_poses_dev")

    def score_poses_cs_dev(self, slot, d_x, d_y, d_ct, d_st, n, d_score, d_count):
        self._ck(self.lib.slam_score_poses_cs_dev(self.h, slot, _ptr(d_x), _ptr(d_y), _ptr(d_ct), _ptr(d_st), n,
                                                  _ptr(d_score), _ptr(d_count)), "score_poses_cs_dev")

    def motion_sample_dev(self, src, anc, dst, n, first_id, dp, sigma, seed, frame):
        """src / dst: triples (x, y, theta) of device arrays."""
        self._ck(self.lib.slam_motion_sample_dev(self.h, _ptr(src[0]), _ptr(src[1]), _ptr(src[2]), _ptr(anc),
                                                 _ptr(dst[0]), _ptr(dst[1]), _ptr(dst[2]), n, first_id, _f3(dp),
                                                 _f3(sigma), seed, frame), "motion_sample_dev")

    def motion_score_dev(self, slot, src, anc, dst, n, first_id, dp, sigma, seed, frame, d_score, d_count):
        self._ck(self.lib.slam_motion_score_dev(self.h, slot, _ptr(src[0]), _ptr(src[1]), _ptr(src[2]), _ptr(anc),
                                                _ptr(dst[0]), _ptr(dst[1]), _ptr(dst[2]), n, first_id, _f3(dp),
                                                _f3(sigma), seed, frame, _ptr(d_score), _ptr(d_count)), "motion_score_dev")

    def obs_set_dev(self, d_zx_by_landmark, d_zy_by_landmark, nlandmarks):
        """Observation table on the device: entry l = observation of landmark l, NaN in zx = not observed."""
        self._ck(self.lib.slam_obs_set_dev(self.h, _ptr(d_zx_by_landmark), _ptr(d_zy_by_landmark), nlandmarks),
                 "obs_set_dev")
        self.__dict__.setdefault("_adopted", {})["obs"] = (d_zx_by_landmark, d_zy_by_landmark)

    def logweight_ekf_dev(self, d_score, gain, n, d_logw, d_max):
        self._ck(self.lib.slam_logweight_ekf_dev(self.h, _ptr(d_score), gain, n, _ptr(d_logw), _ptr(d_max)),
                 "logweight_ekf_dev")

    def obs_upload(self, landmark_id, zx, zy, nlandmarks):
        ids, zx, zy = _np(landmark_id, np.int32), _np(zx, np.float32), _np(zy, np.float32)
        self._ck(self.lib.slam_obs_upload_host(self.h, _ptr(ids), _ptr(zx), _ptr(zy), len(ids), nlandmarks),
                 "obs_upload")

    def ekf_update_dev(self, d_map_in, d_map_out, row_stride, plane_stride, nlandmarks, d_x, d_y, d_th, d_anc, n, meas_var,
                       d_loglik):
        """Maps are one row per particle: [particle][5 planes][plane_stride] floats (include/slam_hip.h)."""
        self._ck(self.lib.slam_ekf_update_dev(self.h, _ptr(d_map_in), _ptr(d_map_out), row_stride, plane_stride, nlandmarks,
                                              _ptr(d_x), _ptr(d_y), _ptr(d_th), _ptr(d_anc), n, meas_var,
                                              _ptr(d_loglik)), "ekf_update_dev")

    def logweight_dev(self, d_score, d_loglik, gain, n, d_logw, d_max):
        self._ck(self.lib.slam_logweight_dev(self.h, _ptr(d_score), _ptr(d_loglik), gain, n, _ptr(d_logw), _ptr(d_max)),
                 "logweight_dev")

    def quantise_weights_dev(self, d_logw, d_max, n, d_wq, d_sum):
        self._ck(self.lib.slam_quantise_weights_dev(self.h, _ptr(d_logw), _ptr(d_max), n, _ptr(d_wq), _ptr(d_sum)),
                 "quantise_weights_dev")

    def quantise_scan_dev(self, d_logw, d_max, n, d_sum):
        self._ck(self.lib.slam_quantise_scan_dev(self.h, _ptr(d_logw), _ptr(d_max), n, _ptr(d_sum)), "quantise_scan_dev")

    def offspring_from_scan_dev(self, n, d_base, d_total, seed, frame, n_total, d_first):
        self._ck(self.lib.slam_offspring_from_scan_dev(self.h, n, _ptr(d_base), _ptr(d_total), seed, frame, n_total,
                                                       _ptr(d_first)), "offspring_from_scan_dev")

    def offspring_from_scan_sharded_dev(self, n, d_shard_totals, rank, world, seed, frame, n_total, d_first):
        self._ck(self.lib.slam_offspring_from_scan_sharded_dev(self.h, n, _ptr(d_shard_totals), rank, world, seed, frame,
                                                               n_total, _ptr(d_first)), "offspring_from_scan_sharded_dev")

    def prefix_sum_dev(self, d_wq, n, d_cdf):
        self._ck(self.lib.slam_prefix_sum_dev(self.h, _ptr(d_wq), n, _ptr(d_cdf)), "prefix_sum_dev")

    def offspring_offsets_dev(self, d_cdf, n, d_base, d_total, seed, frame, n_total, d_first):
        self._ck(self.lib.slam_offspring_offsets_dev(self.h, _ptr(d_cdf), n, _ptr(d_base), _ptr(d_total), seed, frame,
                                                     n_total, _ptr(d_first)), "offspring_offsets_dev")

    def ancestors_dev(self, d_first_all, n_total, slot0, nslots, d_anc):
        self._ck(self.lib.slam_ancestors_dev(self.h, _ptr(d_first_all), n_total, slot0, nslots, _ptr(d_anc)),
                 "ancestors_dev")

    def ancestors_from_scan_dev(self, n, seed, frame, d_anc):
        self._ck(self.lib.slam_ancestors_from_scan_dev(self.h, n, seed, frame, _ptr(d_anc)), "ancestors_from_scan_dev")

    def ancestors_sharded_dev(self, d_first_all, n_total, n_local, rank, world, d_src, d_plan, d_pose_idx=None):
        """d_plan: int32[plan_words(world)] on the device: [0] anything moves, send_cnt[world], recv_cnt[world], ...
        d_pose_idx (optional): ancestor's position in an all-gather of the ranks' [x | y | theta] pose blocks."""
        self._ck(self.lib.slam_ancestors_sharded_dev(self.h, _ptr(d_first_all), n_total, n_local, rank, world,
                                                     _ptr(d_src), _ptr(d_plan), _ptr(d_pose_idx)), "ancestors_sharded_dev")

    def ekf_form_set(self, form: int):
        """-1: the engine chooses the out-of-place EKF kernel; 0: one wavefront per particle; 1 / 2: per 4 / 2 particles."""
        self._ck(self.lib.slam_ekf_form_set(self.h, int(form)), "ekf_form_set")

    def ekf_form_counts(self):
        c = (C.c_int64 * 2)()
        self._ck(self.lib.slam_ekf_form_counts(self.h, c), "ekf_form_counts")
        return int(c[0]), int(c[1])

    def selftest_reciprocal(self):
        """(mismatches, values checked) of the landmark update's fast reciprocal against IEEE division, every float in range."""
        bad, seen = C.c_int64(-1), C.c_int64(0)
        self._ck(self.lib.slam_selftest_reciprocal(self.h, C.byref(bad), C.byref(seen)), "selftest_reciprocal")
        return int(bad.value), int(seen.value)

    def frame_fusion_set(self, on: bool):
        """The front of a single-GPU frame on rows (motion + score and the landmark update) as one launch (default) or two."""
        self._ck(self.lib.slam_frame_fusion_set(self.h, 1 if on else 0), "frame_fusion_set")

    def frame_front_last(self):
        """(particles per updating wavefront, lanes per pose) of the last fused front launch; (0, 0) before the first."""
        info = (C.c_int32 * 2)()
        self._ck(self.lib.slam_frame_front_last(self.h, info), "frame_front_last")
        return int(info[0]), int(info[1])

    def frame_fusion_count(self) -> int:
        c = C.c_int64(0)
        self._ck(self.lib.slam_frame_fusion_count(self.h, C.byref(c)), "frame_fusion_count")
        return int(c.value)

    def pf_paged_set(self, on: bool):
        """Sessions made from now on keep their maps as copy-on-write pages (one GPU; same results as rows)."""
        self._ck(self.lib.slam_pf_paged_set(self.h, int(bool(on))), "pf_paged_set")

    def ekf_inplace_form_set(self, form: int):
        """-1: the engine chooses the in-place EKF kernel; 0: whole rows; 1: the observed landmarks only (compact list)."""
        self._ck(self.lib.slam_ekf_inplace_form_set(self.h, int(form)), "ekf_inplace_form_set")

    def ekf_inplace_form_counts(self):
        c = (C.c_int64 * 2)()
        self._ck(self.lib.slam_ekf_inplace_form_counts(self.h, c), "ekf_inplace_form_counts")
        return int(c[0]), int(c[1])

    def resample_gate_set(self, ess_frac: float):
        self._ck(self.lib.slam_resample_gate_set(self.h, float(ess_frac)), "resample_gate_set")

    def resample_happened(self) -> bool:
        r = C.c_int(1)
        self._ck(self.lib.slam_resample_happened_host(self.h, C.byref(r)), "resample_happened")
        return bool(r.value)

    def exchange_set_capacity(self, rows: int):
        self._ck(self.lib.slam_exchange_set_capacity(self.h, int(rows)), "exchange_set_capacity")

    def exchange_plan_host(self, world):
        """The plan of the last ancestors_sharded_dev call, delivered through mapped host memory (no copy, no sync)."""
        plan = np.zeros(plan_words(world), np.int32)
        self._ck(self.lib.slam_exchange_plan_host(self.h, world, _ptr(plan)), "exchange_plan_host")
        return plan.tolist()

    def migrate_pack_dev(self, n_local, rank, world, plan, d_pose, pose_ld, d_map, row_stride, plane_stride, nlandmarks,
                         d_out):
        plan = _np(plan, np.int32)
        self._ck(self.lib.slam_migrate_pack_dev(self.h, n_local, rank, world, _ptr(plan), _ptr(d_pose), pose_ld,
                                                _ptr(d_map), row_stride, plane_stride, nlandmarks, _ptr(d_out)),
                 "migrate_pack_dev")

    def migrate_unpack_dev(self, d_in, world, recv_cnt, n_local, d_pose, pose_ld, d_map, row_stride, plane_stride,
                           nlandmarks):
        cnt = _np(recv_cnt, np.int32)
        self._ck(self.lib.slam_migrate_unpack_dev(self.h, _ptr(d_in), world, _ptr(cnt), n_local, _ptr(d_pose), pose_ld,
                                                  _ptr(d_map), row_stride, plane_stride, nlandmarks), "migrate_unpack_dev")

    def gather_f32_dev(self, d_src, d_idx, n, d_dst):
        self._ck(self.lib.slam_gather_f32_dev(self.h, _ptr(d_src), _ptr(d_idx), n, _ptr(d_dst)), "gather_f32_dev")

    def gather_map_dev(self, d_in, d_out, in_row_stride, out_row_stride, in_plane_stride, out_plane_stride, nlandmarks,
                       d_idx, n):
        self._ck(self.lib.slam_gather_map_dev(self.h, _ptr(d_in), _ptr(d_out), in_row_stride, out_row_stride,
                                              in_plane_stride, out_plane_stride, nlandmarks, _ptr(d_idx), n),
                 "gather_map_dev")


class MapperParams(C.Structure):
    """``slam_mapper_params``: the reference's run-time parameters (main.c:832-839, :50, :846, :224, :943)."""

    _fields_ = [("fast_res", C.c_float * 3), ("fast_res2", C.c_float * 3), ("border", C.c_float), ("pixel", C.c_float),
                ("pixel2", C.c_float), ("key_dt", C.c_float), ("key_dr", C.c_float), ("range_min", C.c_float),
                ("usable_range", C.c_float), ("edt_cap", C.c_float), ("new_point_threshold", C.c_float)]

    @classmethod
    def default(cls):
        p = cls()
        load_library().slam_mapper_params_default(C.byref(p))
        return p

    def as_list(self):
        return list(self.fast_res) + list(self.fast_res2) + [self.border, self.pixel, self.pixel2, self.key_dt, self.key_dr,
                                                             self.range_min, self.usable_range, self.edt_cap, self.new_point_threshold]


class PfConfig(C.Structure):
    """``slam_pf_config``"""

    _fields_ = [("n_particles", C.c_int32), ("n_landmarks", C.c_int32), ("sigma", C.c_float * 3),
                ("meas_var", C.c_float), ("score_gain", C.c_float), ("seed", C.c_uint64), ("resample_ess_frac", C.c_float),
                ("map_layout", C.c_int32)]


MAP_AUTO, MAP_ROWS, MAP_PAGES, MAP_SPLIT, MAP_SPLIT_PAGES = 0, 1, 2, 3, 4   # slam_map_layout
_LAYOUTS = {"auto": MAP_AUTO, "rows": MAP_ROWS, "pages": MAP_PAGES, "split": MAP_SPLIT, "split_pages": MAP_SPLIT_PAGES, None: MAP_AUTO}
_LAYOUT_NAMES = {MAP_ROWS: "rows", MAP_PAGES: "pages", MAP_SPLIT: "split", MAP_SPLIT_PAGES: "split_pages"}


COMM_ID_BYTES = 128


class PfView(C.Structure):
    """``slam_pf_view``"""

    _fields_ = [("pose", C.c_void_p), ("map", C.c_void_p), ("map_spare", C.c_void_p), ("anc", C.c_void_p), ("row_stride", C.c_int64),
                ("plane_stride", C.c_int32), ("map_rows", C.c_int32), ("score", C.c_void_p), ("logw", C.c_void_p),
                ("loglik", C.c_void_p), ("count", C.c_void_p)]


class PfSplitView(C.Structure):
    """``slam_pf_split_view``"""

    _fields_ = [("mean", C.c_void_p), ("cov", C.c_void_p), ("cls", C.c_void_p), ("live", C.c_void_p), ("live_count", C.c_void_p),
                ("plane_stride", C.c_int32), ("rows", C.c_int32)]


class PfPagedView(C.Structure):
    """``slam_pf_paged_view``"""

    _fields_ = [("pool", C.c_void_p), ("table", C.c_void_p), ("freelist", C.c_void_p), ("state", C.c_void_p),
                ("stamp", C.c_void_p), ("stamp_now", C.c_uint32), ("page_landmarks", C.c_int32),
                ("pages_per_particle", C.c_int32), ("table_rows", C.c_int32), ("npages", C.c_int64),
                ("planes", C.c_int32), ("reserved", C.c_int32), ("half_pages", C.c_int64), ("gap_floats", C.c_int64)]


class DeviceArray:
    """A device buffer owned by the engine, exposed through ``__cuda_array_interface__`` so that array libraries
    (``torch.as_tensor(a, device="cuda")``, cupy) can view it without a copy.  Plumbing only."""

    def __init__(self, ptr: int, shape, typestr: str, owner=None):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2, "strides": None}


def comm_unique_id() -> bytes:
    """``slam_comm_unique_id``: the rendezvous token rank 0 makes and hands to the other ranks."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = load_library().slam_comm_unique_id(buf)
    if rc != 0:
        raise SlamError(rc, "comm_unique_id")
    return bytes(buf)


class LocalGroup:
    """``slam_local_group``: the rendezvous of an in-process group of ranks (one host thread per rank)."""

    def __init__(self, world: int):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.slam_local_group_create(world, C.byref(h))
        if rc != 0:
            raise SlamError(rc, "local_group_create")
        self.h, self.world = h, world

    def close(self):
        if getattr(self, "h", None):
            self.lib.slam_local_group_destroy(self.h)
            self.h = None


class Comm:
    """``slam_comm``: how one rank of a sharded filter reaches the others.  ``Comm.rccl`` (one rank per GPU, RCCL over
    xGMI) or ``Comm.local`` (threads of one process).  Creation over RCCL is collective."""

    def __init__(self, engine: Engine, handle, keep=None):
        self.e, self.h, self._keep = engine, handle, keep

    @classmethod
    def rccl(cls, engine: Engine, rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == COMM_ID_BYTES
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        engine._ck(engine.lib.slam_comm_create_rccl(engine.h, rank, world, buf, C.byref(h)), "comm_create_rccl")
        return cls(engine, h)

    @classmethod
    def local(cls, engine: Engine, group: LocalGroup, rank: int):
        h = C.c_void_p()
        engine._ck(engine.lib.slam_comm_create_local(engine.h, group.h, rank, C.byref(h)), "comm_create_local")
        return cls(engine, h, keep=group)

    @property
    def rank(self):
        return self.e.lib.slam_comm_rank(self.h)

    @property
    def world(self):
        return self.e.lib.slam_comm_world(self.h)

    def abort(self):
        """``slam_comm_abort``: give up, so that the other ranks fail with SLAM_ERR_COMM instead of waiting."""
        self.e._ck(self.e.lib.slam_comm_abort(self.h), "comm_abort")

    def close(self):
        if getattr(self, "h", None):
            self.e.lib.slam_comm_destroy(self.h)
            self.h = None


class PfSession:
    """``slam_pf`` — the C-level particle-filter session (what a plain C host uses).  With ``comm`` it is one rank
    of a population sharded over several GPUs (``n_particles`` = this rank's share); every call is then collective."""

    def __init__(self, engine: Engine, n_particles, n_landmarks=0, sigma=(0.01, 0.01, 0.002), meas_var=0.01,
                 score_gain=1.0, seed=1, comm: Comm | None = None, recv_capacity: int = 0, resample_ess_frac: float = 0.0,
                 map_layout: str | int | None = None):
        """map_layout: "auto" (default: the session chooses), "rows" or "pages" (``slam_map_layout``)."""
        self.e, self.n, self.L, self.comm = engine, n_particles, n_landmarks, comm
        cfg = PfConfig(n_particles, n_landmarks, (C.c_float * 3)(*sigma), meas_var, score_gain, seed, resample_ess_frac,
                       _LAYOUTS.get(map_layout, map_layout))
        h = C.c_void_p()
        if comm is None:
            engine._ck(engine.lib.slam_pf_create(engine.h, C.byref(cfg), C.byref(h)), "pf_create")
        else:
            engine._ck(engine.lib.slam_pf_create_sharded(engine.h, C.byref(cfg), comm.h, recv_capacity, C.byref(h)),
                       "pf_create_sharded")
        self.h = h

    def rows_received(self) -> int:
        return self.e.lib.slam_pf_rows_received(self.h)

    def frames_resampled(self) -> int:
        return int(self.e.lib.slam_pf_frames_resampled(self.h))

    def layout_changes(self) -> int:
        return int(self.e.lib.slam_pf_layout_changes(self.h))

    def device_view(self):
        """dict of DeviceArray views of the session's CURRENT buffers (``slam_pf_device_view``): pose [3][n], map and
        map_spare [map_rows][5][plane_stride] (None without landmarks), anc [n] (None when no gather is pending).
        They move with every step."""
        v = PfView()
        self.e._ck(self.e.lib.slam_pf_device_view(self.h, C.byref(v)), "pf_device_view")
        shape = (v.map_rows, 5, v.plane_stride)
        return {"pose": DeviceArray(v.pose, (3, self.n), "<f4", self),
                "map": DeviceArray(v.map, shape, "<f4", self) if v.map else None,
                "map_spare": DeviceArray(v.map_spare, shape, "<f4", self) if v.map_spare else None,
                "anc": DeviceArray(v.anc, (self.n,), "<i4", self) if v.anc else None,
                "score": DeviceArray(v.score, (self.n,), "<f4", self) if v.score else None,
                "logw": DeviceArray(v.logw, (self.n,), "<f4", self) if v.logw else None,
                "loglik": DeviceArray(v.loglik, (self.n,), "<f4", self) if v.loglik else None,
                "count": DeviceArray(v.count, (self.n,), "<i4", self) if v.count else None,
                "plane_stride": v.plane_stride, "row_stride": v.row_stride}

    def layout(self) -> str:
        """"rows", "pages" or "split": how the landmark maps are kept right now (``slam_pf_layout``)."""
        return _LAYOUT_NAMES[self.e.lib.slam_pf_layout(self.h)]

    def split_view(self):
        """``slam_pf_split_device_view``: means, classes, class covariance rows and the list of classes in use."""
        v = PfSplitView()
        self.e._ck(self.e.lib.slam_pf_split_device_view(self.h, C.byref(v)), "pf_split_device_view")
        return {"mean": DeviceArray(v.mean, (v.rows, 2, v.plane_stride), "<f4", self) if v.mean else None,   # None: split pages
                "cov": DeviceArray(v.cov, (v.rows, 3, v.plane_stride), "<f4", self),
                "cls": DeviceArray(v.cls, (v.rows,), "<i4", self), "live": DeviceArray(v.live, (v.rows,), "<i4", self),
                "live_count": DeviceArray(v.live_count, (1,), "<i4", self), "plane_stride": v.plane_stride}

    def paged_view(self):
        """``slam_pf_paged_device_view``: the page pool, tables, stamps and free list of a session that is on pages."""
        v = PfPagedView()
        self.e._ck(self.e.lib.slam_pf_paged_device_view(self.h, C.byref(v)), "pf_paged_device_view")
        P, nb = int(v.npages), v.pages_per_particle
        if v.planes == 2:   # split pages: two buffers of half_pages pages of [2][page_landmarks] floats, gap_floats apart
            H, pf_ = int(v.half_pages), 2 * v.page_landmarks
            pool = [DeviceArray(v.pool + 4 * k * (H * pf_ + int(v.gap_floats)), (H, 2, v.page_landmarks), "<f4", self) for k in (0, 1)]
        else:
            pool = DeviceArray(v.pool, (P, 5, v.page_landmarks), "<f4", self)
        return {"pool": pool, "planes": v.planes, "half_pages": int(v.half_pages),
                "table": DeviceArray(v.table, (v.table_rows, nb), "<i4", self),
                "freelist": DeviceArray(v.freelist, (P,), "<i4", self), "state": DeviceArray(v.state, (4,), "<i4", self),
                "stamp": DeviceArray(v.stamp, (P,), "<i4", self), "stamp_now": int(v.stamp_now),
                "page_landmarks": v.page_landmarks, "pages_per_particle": nb, "table_rows": v.table_rows, "npages": P}

    def close(self):
        if getattr(self, "h", None):
            self.e.lib.slam_pf_destroy(self.h)
            self.h = None

    def reset(self, pose):
        self.e._ck(self.e.lib.slam_pf_reset(self.h, _f3(pose)), "pf_reset")

    def set_poses(self, x, y, th):
        x, y, th = (_np(a, np.float32) for a in (x, y, th))
        self.e._ck(self.e.lib.slam_pf_set_poses_host(self.h, _ptr(x), _ptr(y), _ptr(th)), "pf_set_poses")

    def set_map(self, rows):
        """rows: float32 [n_particles][5][n_landmarks] (mu_x, mu_y, P_xx, P_xy, P_yy)"""
        rows = _np(rows, np.float32)
        assert rows.shape == (self.n, 5, self.L)
        self.e._ck(self.e.lib.slam_pf_set_map_host(self.h, _ptr(rows)), "pf_set_map")

    def set_map_dev(self, d_rows, row_stride: int, plane_stride: int):
        """Maps from device memory: rows [n_particles][5][plane_stride] floats, row_stride floats apart (asynchronous)."""
        self.e._ck(self.e.lib.slam_pf_set_map_dev(self.h, _ptr(d_rows), int(row_stride), int(plane_stride)), "pf_set_map_dev")

    def is_paged(self) -> bool:
        return bool(self.e.lib.slam_pf_is_paged(self.h))

    def step(self, slot, dp, use_observations=False):
        self.e._ck(self.e.lib.slam_pf_step(self.h, slot, _f3(dp), 1 if use_observations else 0), "pf_step")

    def best(self):
        pose = (C.c_float * 3)()
        lw, idx = C.c_float(0), C.c_int32(0)
        self.e._ck(self.e.lib.slam_pf_best(self.h, pose, C.byref(lw), C.byref(idx)), "pf_best")
        return np.array(list(pose), np.float32), np.float32(lw.value), idx.value

    def mean(self, ref_theta: float):
        """``slam_pf_mean``: posterior mean of the current population (heading averaged on the circle around ref_theta)."""
        pose = (C.c_float * 3)()
        self.e._ck(self.e.lib.slam_pf_mean(self.h, float(ref_theta), pose), "pf_mean")
        return np.array(list(pose), np.float32)

    def poses(self):
        x, y, th = (np.empty(self.n, np.float32) for _ in range(3))
        self.e._ck(self.e.lib.slam_pf_get_poses_host(self.h, _ptr(x), _ptr(y), _ptr(th)), "pf_get_poses")
        return np.stack([x, y, th])

    def maps(self):
        m = np.empty((self.n, 5, self.L), np.float32)
        self.e._ck(self.e.lib.slam_pf_get_map_host(self.h, _ptr(m)), "pf_get_map")
        return m

    def map_rows(self, particles):
        """``slam_pf_get_map_rows_host``: the maps of the chosen current particles, [len][5][n_landmarks]."""
        sel = _np(particles, np.int32)
        m = np.empty((len(sel), 5, self.L), np.float32)
        self.e._ck(self.e.lib.slam_pf_get_map_rows_host(self.h, _ptr(sel), len(sel), _ptr(m)), "pf_get_map_rows")
        return m


def plan_words(world: int) -> int:
    """int32 words of the exchange plan slam_ancestors_sharded_dev writes (SLAM_PLAN_WORDS in slam_hip.h)."""
    return 1 + 3 * world


def comb_offset(seed: int, frame: int, total: int) -> int:
    return int(load_library().slam_comb_offset(seed, frame, total))


def grid_meta(rows, cols, ld, pixel, min_x, min_y) -> GridMeta:
    return GridMeta(int(rows), int(cols), int(ld), float(np.float32(pixel)), float(np.float32(min_x)),
                    float(np.float32(min_y)))
