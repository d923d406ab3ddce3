"""ctypes binding of include/cfear_hip.h (no torch types cross this boundary)."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_LIB = None
CFEAR_MAX_OUTER = 64


class CfearError(RuntimeError):
    pass


class Params(C.Structure):
    """cfear_params (include/cfear_hip.h)."""
    _fields_ = [
        ("z_min", C.c_float), ("range_res", C.c_float), ("min_distance", C.c_float),
        ("k_strongest", C.c_int32),
        ("res", C.c_double), ("downsample_factor", C.c_double),
        ("weight_intensity", C.c_int32), ("cost", C.c_int32), ("loss", C.c_int32),
        ("weight_opt", C.c_int32),
        ("loss_limit", C.c_double), ("covar_scale", C.c_double), ("regularization", C.c_double),
        ("submap_scan_size", C.c_int32), ("compensate", C.c_int32), ("radar_ccw", C.c_int32),
        ("use_keyframe", C.c_int32),
        ("min_keyframe_dist", C.c_double), ("min_keyframe_rot_deg", C.c_double),
        ("max_itr_association", C.c_int32), ("min_itr", C.c_int32),
        ("max_solver_iterations", C.c_int32), ("filter_type", C.c_int32),
        ("assoc_radius", C.c_double),
        ("cfar_window_size", C.c_int32), ("cfar_nb_guard_cells", C.c_int32), ("cfar_false_alarm_rate", C.c_float), ("cfar_max_points", C.c_int32),
        ("cfar_max_distance", C.c_double),
    ]


FILTER_KSTRONG, FILTER_CACFAR = 0, 1


class Cell(C.Structure):
    _fields_ = [
        ("mean", C.c_double * 2), ("cov", C.c_double * 3), ("normal", C.c_double * 2),
        ("orth", C.c_double * 2), ("lambda_min", C.c_double), ("lambda_max", C.c_double),
        ("scale", C.c_double), ("sum_intensity", C.c_double), ("avg_intensity", C.c_double),
        ("nsamples", C.c_int32), ("valid", C.c_int32),
    ]


CELL_DTYPE = np.dtype([
    ("mean", "f8", 2), ("cov", "f8", 3), ("normal", "f8", 2), ("orth", "f8", 2),
    ("lambda_min", "f8"), ("lambda_max", "f8"), ("scale", "f8"), ("sum_intensity", "f8"),
    ("avg_intensity", "f8"), ("nsamples", "i4"), ("valid", "i4")])


class RegSummary(C.Structure):
    _fields_ = [
        ("success", C.c_int32), ("usable", C.c_int32), ("outer_iterations", C.c_int32),
        ("num_residuals", C.c_int32), ("num_residual_blocks", C.c_int32), ("assoc_path", C.c_int32),
        ("final_cost", C.c_double), ("score", C.c_double),
        ("inner_iterations", C.c_int32 * CFEAR_MAX_OUTER), ("termination", C.c_int32 * CFEAR_MAX_OUTER),
        ("outer_cost", C.c_double * CFEAR_MAX_OUTER), ("outer_pose", (C.c_double * 3) * CFEAR_MAX_OUTER),
    ]


# cfear_sweep_record (include/cfear_hip.h): what a caller of pointcloudCallback sees after every sweep of a replay
SWEEP_RECORD_DTYPE = np.dtype([("pose", "f8", 3), ("final_cost", "f8"), ("outer_iterations", "i4"), ("num_residuals", "i4"),
                               ("n_keyframes", "i4"), ("n_cells", "i4"), ("inner_iterations", "i4", 8)])
assert SWEEP_RECORD_DTYPE.itemsize == 80

EXPORTS = [
    "cfear_version", "cfear_default_params", "cfear_create", "cfear_destroy", "cfear_last_error",
    "cfear_set_params", "cfear_synchronize", "cfear_tune", "cfear_kstrongest_device", "cfear_kstrongest_host",
    "cfear_rotate_polar", "cfear_rotate_polar_device", "cfear_filter_polar", "cfear_filter_polar_device", "cfear_filter_cfar", "cfear_filter_cfar_device", "cfear_filter_cfar_batch_device", "cfear_cloud_upload", "cfear_cloud_size",
    "cfear_cloud_download", "cfear_clouds_download", "cfear_cloud_release", "cfear_compensate", "cfear_compensate_pair", "cfear_scan_create",
    "cfear_scan_from_cells", "cfear_scan_release", "cfear_scan_size", "cfear_scan_download_cells", "cfear_scan_closest",
    "cfear_register", "cfear_register_soft", "cfear_get_cost", "cfear_cov_by_sampling", "cfear_odometry_create", "cfear_odometry_destroy", "cfear_odometry_reset",
    "cfear_odometry_step_device", "cfear_odometry_step_cloud_device", "cfear_odometry_step_host", "cfear_odometry_poses",
    "cfear_odometry_replay_host", "cfear_odometry_replay_device", "cfear_host_alloc", "cfear_host_free",
    "cfear_odometry_covariances", "cfear_odometry_status", "cfear_odometry_summary", "cfear_odometry_profile", "cfear_odometry_profile_read", "cfear_odometry_profile_read_stages", "cfear_odometry_phase_times", "cfear_time_kstrongest",
]


def lib_path():
    return _build.LIB


def lib():
    """Loads libcfear_hip.so; raises if it is not built (no fallback of any kind)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    try:  # PyTorch bundles its own HIP runtime: when both live in one process, torch's must be loaded first
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise CfearError("libcfear_hip.so is not built: run __graft_entry__.build() "
                         "(python -m cfear_radarodometry_code_public_amd.build)")
    L = C.CDLL(path)
    vp, u8p, u32p, f32p, f64p, i32p = (C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    sig = {
        "cfear_version": (C.c_char_p, []),
        "cfear_default_params": (None, [C.POINTER(Params)]),
        "cfear_create": (C.c_int, [C.POINTER(vp), C.c_int, vp, C.POINTER(Params), C.c_int, C.c_int]),
        "cfear_destroy": (None, [vp]),
        "cfear_last_error": (C.c_char_p, [vp]),
        "cfear_set_params": (C.c_int, [vp, C.POINTER(Params)]),
        "cfear_synchronize": (C.c_int, [vp]),
        "cfear_tune": (C.c_int, [vp, C.c_int, C.c_int]),
        "cfear_kstrongest_device": (C.c_int, [vp, u8p, C.c_int, u32p]),
        "cfear_kstrongest_host": (C.c_int, [vp, u8p, C.c_int, u32p]),
        "cfear_rotate_polar": (C.c_int, [vp, u8p, C.c_int, C.c_int, u8p]),
        "cfear_rotate_polar_device": (C.c_int, [vp, u8p, C.c_int, C.c_int, C.c_int, u8p]),
        "cfear_filter_polar": (C.c_int, [vp, u8p, C.POINTER(vp), C.POINTER(vp)]),
        "cfear_filter_polar_device": (C.c_int, [vp, u8p, C.POINTER(vp), C.POINTER(vp)]),
        "cfear_filter_cfar": (C.c_int, [vp, u8p, C.c_int, C.c_int, C.c_float, C.c_double, C.POINTER(vp)]),
        "cfear_filter_cfar_device": (C.c_int, [vp, u8p, C.c_int, C.c_int, C.c_float, C.c_double, C.POINTER(vp)]),
        "cfear_filter_cfar_batch_device": (C.c_int, [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_double, f32p, C.c_int, i32p]),
        "cfear_cloud_upload": (C.c_int, [vp, f32p, C.c_int, C.POINTER(vp)]),
        "cfear_cloud_size": (C.c_int, [vp, vp, C.POINTER(C.c_int)]),
        "cfear_cloud_download": (C.c_int, [vp, vp, f32p, C.c_int, C.POINTER(C.c_int)]),
        "cfear_clouds_download": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "cfear_cloud_release": (None, [vp, vp]),
        "cfear_compensate": (C.c_int, [vp, vp, f64p, C.c_int]),
        "cfear_compensate_pair": (C.c_int, [vp, vp, vp, f64p, C.c_int]),
        "cfear_scan_create": (C.c_int, [vp, vp, C.POINTER(vp)]),
        "cfear_scan_from_cells": (C.c_int, [vp, vp, C.c_int, C.POINTER(vp)]),
        "cfear_scan_release": (None, [vp, vp]),
        "cfear_scan_size": (C.c_int, [vp, vp, C.POINTER(C.c_int)]),
        "cfear_scan_download_cells": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(C.c_int)]),
        "cfear_scan_closest": (C.c_int, [vp, vp, f64p, C.c_int, C.c_double, i32p]),
        "cfear_register": (C.c_int, [vp, C.POINTER(vp), C.c_int, f64p, f64p, C.POINTER(RegSummary)]),
        "cfear_register_soft": (C.c_int, [vp, C.POINTER(vp), C.c_int, f64p, f64p, f64p, C.POINTER(RegSummary)]),
        "cfear_get_cost": (C.c_int, [vp, C.POINTER(vp), C.c_int, f64p, C.c_int, f64p, f64p, C.c_int, C.POINTER(C.c_int)]),
        "cfear_cov_by_sampling": (C.c_int, [vp, C.POINTER(vp), C.c_int, f64p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double,
                                            C.c_double, C.c_int, f64p, C.POINTER(C.c_int), f64p]),
        "cfear_odometry_create": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
        "cfear_odometry_destroy": (None, [vp, vp]),
        "cfear_odometry_reset": (C.c_int, [vp, vp]),
        "cfear_odometry_step_device": (C.c_int, [vp, vp, u8p]),
        "cfear_odometry_step_host": (C.c_int, [vp, vp, u8p]),
        "cfear_odometry_step_cloud_device": (C.c_int, [vp, vp, f32p, C.c_int, i32p]),
        "cfear_odometry_poses": (C.c_int, [vp, vp, f64p]),
        "cfear_odometry_covariances": (C.c_int, [vp, vp, f64p]),
        "cfear_odometry_status": (C.c_int, [vp, vp, i32p]),
        "cfear_odometry_replay_host": (C.c_int, [vp, vp, u8p, C.c_int, vp]),
        "cfear_odometry_replay_device": (C.c_int, [vp, vp, u8p, C.c_int, vp]),
        "cfear_host_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
        "cfear_host_free": (None, [vp, vp]),
        "cfear_odometry_summary": (C.c_int, [vp, vp, C.c_int, C.POINTER(RegSummary), C.POINTER(C.c_int),
                                             C.POINTER(C.c_int)]),
        "cfear_odometry_profile": (C.c_int, [vp, vp, C.c_int]),
        "cfear_odometry_profile_read": (C.c_int, [vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
        "cfear_odometry_profile_read_stages": (C.c_int, [vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
        "cfear_odometry_phase_times": (C.c_int, [vp, vp, C.c_int, vp]),
        "cfear_time_kstrongest": (C.c_int, [vp, u8p, C.c_int, u32p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the ABI is incomplete (tests/test_abi.py checks every export)
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def default_params(**kw):
    p = Params()
    lib().cfear_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _addr(a):
    """Device pointer (int), torch tensor (data_ptr) or numpy array -> address."""
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data


TUNE_FILTER_OCCUPANCY, TUNE_FILTER_ROWS_PER_WAVE, TUNE_ODOMETRY_OVERLAP, TUNE_REPLAY_PERSISTENT_MAX, TUNE_FILTER_CUS, TUNE_REPEAT_SHORTCUT, TUNE_MAX_CELLS, TUNE_REGISTRATION_ORDER, TUNE_LARGE_SUBMAP_KERNEL, TUNE_NN_TIE_RULE, TUNE_VOXEL_ORDER = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
TUNE_DEFAULTS = {TUNE_ODOMETRY_OVERLAP: 0, TUNE_FILTER_CUS: 0, TUNE_MAX_CELLS: 0, TUNE_REGISTRATION_ORDER: 1, TUNE_LARGE_SUBMAP_KERNEL: 0}  # include/cfear_hip.h


class Context:
    """cfear_ctx: one HIP device + stream. `stream` = raw hipStream_t handle (int) or None."""

    def __init__(self, params, A, R, device=0, stream=None):
        self._L = lib()
        self._h = C.c_void_p()
        self.params = params
        self.A, self.R = int(A), int(R)
        self._tuned = {}
        rc = self._L.cfear_create(C.byref(self._h), int(device), C.c_void_p(stream or 0), C.byref(params),
                                  self.A, self.R)
        if rc != 0:
            self._h = None
            raise CfearError("cfear_create failed rc=%d (is a gfx950 GPU visible?)" % rc)

    def close(self):
        if getattr(self, "_h", None):
            for ptr in list(getattr(self, "_pinned", {}).values()):
                self._L.cfear_host_free(self._h, ptr)
            self._pinned = {}
            self._L.cfear_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise CfearError("%s failed rc=%d: %s" % (what, rc, self._L.cfear_last_error(self._h).decode()))

    @property
    def handle(self):
        return self._h

    def set_params(self, params):
        self._check(self._L.cfear_set_params(self._h, C.byref(params)), "cfear_set_params")
        self.params = params

    def synchronize(self):
        self._check(self._L.cfear_synchronize(self._h), "cfear_synchronize")

    def tune(self, key, value):
        """cfear_tune (include/cfear_hip.h): launch-shape knobs - TUNE_FILTER_OCCUPANCY, TUNE_FILTER_ROWS_PER_WAVE, TUNE_ODOMETRY_OVERLAP,
        TUNE_FILTER_CUS, TUNE_REPLAY_PERSISTENT_MAX, TUNE_REPEAT_SHORTCUT, TUNE_REGISTRATION_ORDER: results do not depend on them - and
        TUNE_MAX_CELLS, the cell capacity batched odometry objects created afterwards are sized for (an overflow is reported, never silent)"""
        self._check(self._L.cfear_tune(self._h, int(key), int(value)), "cfear_tune")
        # what the library holds after its clamps (csrc/cabi.hip cfear_tune), so that a later restore puts back a value the context really had
        k, v = int(key), int(value)
        clamp = {TUNE_FILTER_ROWS_PER_WAVE: lambda x: max(x, 0), TUNE_ODOMETRY_OVERLAP: lambda x: min(max(x, 0), 8), TUNE_FILTER_CUS: lambda x: max(x, 0),
                 TUNE_REGISTRATION_ORDER: lambda x: int(x != 0), TUNE_MAX_CELLS: lambda x: max(x, 0), TUNE_REPEAT_SHORTCUT: lambda x: int(x != 0),
                 TUNE_VOXEL_ORDER: lambda x: int(x == 1), TUNE_NN_TIE_RULE: lambda x: x if 0 <= x <= 2 else 0, TUNE_LARGE_SUBMAP_KERNEL: lambda x: x if 0 <= x <= 2 else 0,
                 TUNE_REPLAY_PERSISTENT_MAX: lambda x: max(x, 0)}
        self._tuned[k] = clamp.get(k, lambda x: x)(v)

    # ---- stage 1 ----
    def kstrongest_host(self, polar):
        polar = np.ascontiguousarray(polar, dtype=np.uint8)
        if polar.ndim == 2:
            polar = polar[None]
        n, A, R = polar.shape
        assert (A, R) == (self.A, self.R)
        out = np.zeros((n, A, self.params.k_strongest), dtype=np.uint32)
        self._check(self._L.cfear_kstrongest_host(self._h, polar.ctypes.data, n, out.ctypes.data), "cfear_kstrongest_host")
        return out

    def kstrongest_device(self, d_polar, n_scans, d_slots):
        self._check(self._L.cfear_kstrongest_device(self._h, _addr(d_polar), int(n_scans), _addr(d_slots)),
                    "cfear_kstrongest_device")

    def time_kstrongest(self, d_polar, n_scans, d_slots, warmup, iters):
        t = C.c_double()
        self._check(self._L.cfear_time_kstrongest(self._h, _addr(d_polar), int(n_scans), _addr(d_slots), int(warmup),
                                                  int(iters), C.byref(t)), "cfear_time_kstrongest")
        return t.value

    def rotate_polar(self, img):
        """radarDriver::Callback (non-Oxford): rows = range -> rows = azimuth (cv::ROTATE_90_COUNTERCLOCKWISE)"""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        out = np.empty((img.shape[1], img.shape[0]), dtype=np.uint8)
        self._check(self._L.cfear_rotate_polar(self._h, img.ctypes.data, img.shape[0], img.shape[1], out.ctypes.data), "cfear_rotate_polar")
        return out

    # ---- stage 1 -> clouds (radarDriver::CallbackOffline) ----
    def filter_polar(self, polar, peaks=True):
        """polar: uint8 [A,R] numpy (host) or a device pointer/torch tensor (device=True path via _addr)."""
        c, cp = C.c_void_p(), C.c_void_p()
        if isinstance(polar, np.ndarray):
            polar = np.ascontiguousarray(polar, dtype=np.uint8)
            assert polar.shape == (self.A, self.R)
            rc = self._L.cfear_filter_polar(self._h, polar.ctypes.data, C.byref(c), C.byref(cp) if peaks else None)
        else:
            rc = self._L.cfear_filter_polar_device(self._h, _addr(polar), C.byref(c), C.byref(cp) if peaks else None)
        self._check(rc, "cfear_filter_polar")
        return Cloud(self, c), (Cloud(self, cp) if peaks else None)

    def filter_cfar(self, polar, window_size=10, nb_guard_cells=20, false_alarm_rate=0.01, max_distance=400.0):
        """radarDriver::Process with filter_type CA-CFAR (defaults of radarDriver::Parameters, radar_driver.h:43-44)."""
        c = C.c_void_p()
        args = (int(window_size), int(nb_guard_cells), C.c_float(false_alarm_rate), C.c_double(max_distance), C.byref(c))
        if isinstance(polar, np.ndarray):
            polar = np.ascontiguousarray(polar, dtype=np.uint8)
            assert polar.shape == (self.A, self.R)
            rc = self._L.cfear_filter_cfar(self._h, polar.ctypes.data, *args)
        else:
            rc = self._L.cfear_filter_cfar_device(self._h, _addr(polar), *args)
        self._check(rc, "cfear_filter_cfar")
        return Cloud(self, c)

    def filter_cfar_batch(self, d_polar, n_scans, d_xyi, capacity, d_counts, window_size=10, nb_guard_cells=20, false_alarm_rate=0.01, max_distance=400.0):
        """CA-CFAR over n_scans device-resident sweeps; d_xyi [n_scans, capacity, 3] float32 and d_counts [n_scans] int32 on the device"""
        self._check(self._L.cfear_filter_cfar_batch_device(self._h, _addr(d_polar), int(n_scans), int(window_size), int(nb_guard_cells),
                                                           C.c_float(false_alarm_rate), C.c_double(max_distance), _addr(d_xyi), int(capacity),
                                                           _addr(d_counts)), "cfear_filter_cfar_batch_device")

    def cloud_upload(self, xyi):
        xyi = np.ascontiguousarray(xyi, dtype=np.float32).reshape(-1, 3)
        c = C.c_void_p()
        self._check(self._L.cfear_cloud_upload(self._h, xyi.ctypes.data, xyi.shape[0], C.byref(c)), "cfear_cloud_upload")
        return Cloud(self, c)

    def compensate(self, cloud, motion_xyt, ccw):
        m = np.asarray(motion_xyt, dtype=np.float64).copy()
        self._check(self._L.cfear_compensate(self._h, cloud._h, m.ctypes.data, int(ccw)), "cfear_compensate")

    def compensate_pair(self, cloud, cloud_peaks, motion_xyt, ccw):
        """cfear_compensate_pair: a sweep's two clouds by the same motion in one launch (odometrykeyframefuser.cpp:148-149)"""
        m = np.asarray(motion_xyt, dtype=np.float64).copy()
        self._check(self._L.cfear_compensate_pair(self._h, cloud._h, cloud_peaks._h, m.ctypes.data, int(ccw)), "cfear_compensate_pair")

    # ---- stage 2 (MapPointNormal) ----
    def scan_create(self, cloud):
        s = C.c_void_p()
        self._check(self._L.cfear_scan_create(self._h, cloud._h, C.byref(s)), "cfear_scan_create")
        return Scan(self, s)

    def scan_from_cells(self, cells):
        """cells: numpy array of CELL_DTYPE (raw = true identity cells, transformed copies) -> Scan"""
        cells = np.ascontiguousarray(cells, dtype=CELL_DTYPE)
        s = C.c_void_p()
        self._check(self._L.cfear_scan_from_cells(self._h, cells.ctypes.data, len(cells), C.byref(s)), "cfear_scan_from_cells")
        return Scan(self, s)

    # ---- stage 3 (n_scan_normal_reg::Register) ----
    def register(self, scans, poses):
        n = len(scans)
        arr = (C.c_void_p * n)(*[s._h for s in scans])
        P = np.ascontiguousarray(poses, dtype=np.float64).reshape(n, 3).copy()
        cov = np.zeros(36)
        S = RegSummary()
        self._check(self._L.cfear_register(self._h, arr, n, P.ctypes.data, cov.ctypes.data, C.byref(S)), "cfear_register")
        return bool(S.success), P, cov.reshape(6, 6), S

    def register_soft(self, scans, poses, prior_cov6):
        """Register(..., soft_constraints=true) with reg_cov.back() = prior_cov6"""
        n = len(scans)
        arr = (C.c_void_p * n)(*[s._h for s in scans])
        P = np.ascontiguousarray(poses, dtype=np.float64).reshape(n, 3).copy()
        pc = np.ascontiguousarray(prior_cov6, dtype=np.float64).reshape(36).copy()
        cov = np.zeros(36)
        S = RegSummary()
        self._check(self._L.cfear_register_soft(self._h, arr, n, P.ctypes.data, pc.ctypes.data, cov.ctypes.data, C.byref(S)), "cfear_register_soft")
        return bool(S.success), P, cov.reshape(6, 6), S

    def get_cost(self, scans, poses, itr=2):
        """n_scan_normal_reg::GetCost -> (score, residuals) or None where the reference returns false."""
        n = len(scans)
        arr = (C.c_void_p * n)(*[s._h for s in scans])
        P = np.ascontiguousarray(poses, dtype=np.float64).reshape(n, 3).copy()
        cap = 2 * (n - 1) * max(scans[-1].size, 1)
        res = np.zeros(cap)
        score, m = C.c_double(), C.c_int()
        rc = self._L.cfear_get_cost(self._h, arr, n, P.ctypes.data, int(itr), C.byref(score), res.ctypes.data, cap, C.byref(m))
        if rc == -4:  # CFEAR_ERR_EMPTY
            return None
        self._check(rc, "cfear_get_cost")
        return score.value, res[:m.value].copy()

    def cov_by_sampling(self, scans, poses, final_cost, num_residuals, itr=2, xy_range=0.4, yaw_range=0.0043625, steps=3, cov_scaler=4.0):
        """approximateCovarianceBySampling -> (success, cov6x6, sampled costs)"""
        n = len(scans)
        arr = (C.c_void_p * n)(*[s._h for s in scans])
        P = np.ascontiguousarray(poses, dtype=np.float64).reshape(n, 3).copy()
        cov, costs, ok = np.zeros(36), np.zeros(steps ** 3), C.c_int()
        self._check(self._L.cfear_cov_by_sampling(self._h, arr, n, P.ctypes.data, int(itr), float(xy_range), float(yaw_range), int(steps),
                                                  float(cov_scaler), float(final_cost), int(num_residuals), cov.ctypes.data, C.byref(ok),
                                                  costs.ctypes.data), "cfear_cov_by_sampling")
        return bool(ok.value), cov.reshape(6, 6), costs

    def pinned(self, shape, dtype=np.uint8):
        """Page-locked host array (cfear_host_alloc): copies from it overlap with kernels. Freed by pinned_free(arr) - arr or any
        view / slice of it - or when the context closes: the array and its views must not be touched after either (the memory is
        gone; NumPy cannot know)."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = C.c_void_p()
        self._check(self._L.cfear_host_alloc(self._h, nbytes, C.byref(ptr)), "cfear_host_alloc")
        buf = (C.c_uint8 * nbytes).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = ptr.value
        return arr

    def pinned_free(self, arr):
        root = arr
        while isinstance(getattr(root, "base", None), np.ndarray):  # a view / slice: the allocation is its root array's
            root = root.base
        ptr = getattr(self, "_pinned", {}).pop(root.ctypes.data, None)
        if ptr is None:
            raise ValueError("pinned_free: not an array of Context.pinned() of this context (or already freed)")
        if self._h:
            self._L.cfear_host_free(self._h, ptr)

    def odometry(self, n_sequences, overlap=None, filter_cus=None, max_cells=None, reg_order=None, large_kernel=None):
        """overlap: None = the context's setting; 0 / False = the three kernels in turn on the context stream; n >= 1 = the filter one
        sweep ahead on a low-priority stream, features / registration of n ranges of the sequences on n high-priority streams.
        The keyword settings apply to THIS object only: the context's own cfear_tune values are put back afterwards (an object
        created later without the keyword does not inherit them)."""
        want = [(TUNE_ODOMETRY_OVERLAP, overlap), (TUNE_FILTER_CUS, filter_cus), (TUNE_REGISTRATION_ORDER, reg_order), (TUNE_MAX_CELLS, max_cells),
                (TUNE_LARGE_SUBMAP_KERNEL, large_kernel)]
        saved = []
        try:
            for key, v in want:
                if v is not None:
                    saved.append((key, self._tuned.get(key, TUNE_DEFAULTS[key])))
                    self.tune(key, int(v))
            return Odometry(self, n_sequences)
        finally:
            for key, v in reversed(saved):
                self.tune(key, v)


class Cloud:
    def __init__(self, ctx, h):
        self._ctx, self._h = ctx, h

    def release(self):
        if self._h and self._ctx._h:
            self._ctx._L.cfear_cloud_release(self._ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @property
    def size(self):
        n = C.c_int()
        self._ctx._check(self._ctx._L.cfear_cloud_size(self._ctx._h, self._h, C.byref(n)), "cfear_cloud_size")
        return n.value

    def download(self):
        n = self.size
        out = np.zeros((max(n, 1), 3), dtype=np.float32)
        m = C.c_int()
        self._ctx._check(self._ctx._L.cfear_cloud_download(self._ctx._h, self._h, out.ctypes.data, n, C.byref(m)),
                         "cfear_cloud_download")
        return out[:n]


class Scan:
    def __init__(self, ctx, h):
        self._ctx, self._h = ctx, h

    def release(self):
        if self._h and self._ctx._h:
            self._ctx._L.cfear_scan_release(self._ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @property
    def size(self):
        n = C.c_int()
        self._ctx._check(self._ctx._L.cfear_scan_size(self._ctx._h, self._h, C.byref(n)), "cfear_scan_size")
        return n.value

    def cells(self):
        n = self.size
        out = np.zeros(max(n, 1), dtype=CELL_DTYPE)
        m = C.c_int()
        self._ctx._check(self._ctx._L.cfear_scan_download_cells(self._ctx._h, self._h, out.ctypes.data, n, C.byref(m)),
                         "cfear_scan_download_cells")
        return out[:n]

    def closest(self, qxy, d):
        q = np.ascontiguousarray(qxy, dtype=np.float64).reshape(-1, 2)
        idx = np.zeros(q.shape[0], dtype=np.int32)
        self._ctx._check(self._ctx._L.cfear_scan_closest(self._ctx._h, self._h, q.ctypes.data, q.shape[0], float(d),
                                                         idx.ctypes.data), "cfear_scan_closest")
        return idx


class Odometry:
    """Batched OdometryKeyframeFuser: B independent sequences, all state on the device."""

    def __init__(self, ctx, n_sequences):
        self._ctx, self.B = ctx, int(n_sequences)
        self._h = C.c_void_p()
        ctx._check(ctx._L.cfear_odometry_create(ctx._h, self.B, C.byref(self._h)), "cfear_odometry_create")

    def release(self):
        if self._h and self._ctx._h:
            self._ctx._L.cfear_odometry_destroy(self._ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def reset(self):
        self._ctx._check(self._ctx._L.cfear_odometry_reset(self._ctx._h, self._h), "cfear_odometry_reset")

    def step_device(self, d_polar):
        self._ctx._check(self._ctx._L.cfear_odometry_step_device(self._ctx._h, self._h, _addr(d_polar)),
                         "cfear_odometry_step_device")

    def step_cloud_device(self, d_xyi, capacity, d_counts):
        """clouds on the device (raw pointers): [B][capacity][3] floats, [B] int32 counts"""
        self._ctx._check(self._ctx._L.cfear_odometry_step_cloud_device(self._ctx._h, self._h, C.c_void_p(int(d_xyi)), int(capacity), C.c_void_p(int(d_counts))),
                         "cfear_odometry_step_cloud_device")

    def step_host(self, polar):
        polar = np.ascontiguousarray(polar, dtype=np.uint8)
        assert polar.shape == (self.B, self._ctx.A, self._ctx.R)
        self._ctx._check(self._ctx._L.cfear_odometry_step_host(self._ctx._h, self._h, polar.ctypes.data),
                         "cfear_odometry_step_host")

    def replay_host(self, frames, records=True):
        """frames: uint8 [n, B, A, R] (or [n, A, R] for one sequence), e.g. a view of Context.pinned(). Runs the n sweeps with
        no host round trip in between; -> structured array [n, B] of SWEEP_RECORD_DTYPE (or None)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.ndim == 3:
            frames = frames[:, None]
        assert frames.shape[1:] == (self.B, self._ctx.A, self._ctx.R), frames.shape
        n = frames.shape[0]
        rec = np.zeros((n, self.B), dtype=SWEEP_RECORD_DTYPE) if records else None
        self._ctx._check(self._ctx._L.cfear_odometry_replay_host(self._ctx._h, self._h, frames.ctypes.data, n,
                                                                 rec.ctypes.data if records else None), "cfear_odometry_replay_host")
        return rec

    def replay_device(self, d_frames, n_sweeps, d_records=None):
        """d_frames: device pointer / torch tensor of n_sweeps x B x A x R bytes; d_records: device buffer of n_sweeps x B x 80 bytes or
        None. Asynchronous on the context stream."""
        self._ctx._check(self._ctx._L.cfear_odometry_replay_device(self._ctx._h, self._h, _addr(d_frames), int(n_sweeps),
                                                                   _addr(d_records) if d_records is not None else None), "cfear_odometry_replay_device")

    def profile(self, enable):
        self._ctx._check(self._ctx._L.cfear_odometry_profile(self._ctx._h, self._h, int(enable)), "cfear_odometry_profile")

    def profile_read(self):
        """-> (filter seconds, filter launches)"""
        tf, nf = C.c_double(), C.c_int()
        self._ctx._check(self._ctx._L.cfear_odometry_profile_read(self._ctx._h, self._h, C.byref(tf), C.byref(nf)),
                         "cfear_odometry_profile_read")
        return tf.value, nf.value

    def profile_read_stages(self):
        """-> (features seconds, registration seconds, launches of each)"""
        tf, tr, n = C.c_double(), C.c_double(), C.c_int()
        self._ctx._check(self._ctx._L.cfear_odometry_profile_read_stages(self._ctx._h, self._h, C.byref(tf), C.byref(tr), C.byref(n)),
                         "cfear_odometry_profile_read_stages")
        return tf.value, tr.value, n.value

    def phase_times(self, mode, light=False, workgroups_only=False, controller=False):
        """None: switch to the timed kernel instantiations (light: phase stamps only, without the per-LM-command clock reads that
        halve the speed of the registration kernel; workgroups_only: production kernels, start / end clock per workgroup; controller: slots 0..7 of a sequence hold the
        breakdown of the registration's command loop instead of the feature kernel's stamps); True: read + clear the [B][32] tick table of the steps since the last read;
        False: back to the production kernels."""
        L, c = self._ctx._L, self._ctx
        if mode is None:
            c._check(L.cfear_odometry_phase_times(c._h, self._h, 3 if workgroups_only else (4 if controller else (2 if light else 1)), None), "cfear_odometry_phase_times")
            return None
        if mode is False:
            c._check(L.cfear_odometry_phase_times(c._h, self._h, 0, None), "cfear_odometry_phase_times")
            return None
        buf = np.zeros((self.B, 32), dtype=np.int64)
        c._check(L.cfear_odometry_phase_times(c._h, self._h, -1, buf.ctypes.data), "cfear_odometry_phase_times")
        return buf

    def poses(self):
        out = np.zeros((self.B, 3))
        self._ctx._check(self._ctx._L.cfear_odometry_poses(self._ctx._h, self._h, out.ctypes.data), "cfear_odometry_poses")
        return out

    def status(self, per_sequence=False):
        """raises CfearError (rc=-6) if a scan of this object was truncated (cfear_odometry_status); per_sequence=True: returns the per-sequence
        flag words instead of raising (bit 0 cells lost, bit 1 points lost)"""
        if per_sequence:
            out = np.zeros(self.B, dtype=np.int32)
            rc = self._ctx._L.cfear_odometry_status(self._ctx._h, self._h, out.ctypes.data)
            if rc not in (0, -6):  # (0 and CFEAR_ERR_CAPACITY are what the flag words describe; anything else - a HIP error, a failed join - is not)
                self._ctx._check(rc, "cfear_odometry_status")
            return out
        self._ctx._check(self._ctx._L.cfear_odometry_status(self._ctx._h, self._h, None), "cfear_odometry_status")

    def covariances(self):
        """cov_current of every sequence after the last sweep: [B, 6, 6]"""
        out = np.zeros((self.B, 36))
        self._ctx._check(self._ctx._L.cfear_odometry_covariances(self._ctx._h, self._h, out.ctypes.data), "cfear_odometry_covariances")
        return out.reshape(self.B, 6, 6)

    def summary(self, sequence):
        S = RegSummary()
        nc, nk = C.c_int(), C.c_int()
        self._ctx._check(self._ctx._L.cfear_odometry_summary(self._ctx._h, self._h, int(sequence), C.byref(S), C.byref(nc),
                                                             C.byref(nk)), "cfear_odometry_summary")
        return S, nc.value, nk.value
