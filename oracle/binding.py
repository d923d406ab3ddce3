"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module. The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CFO_MAX_OUTER = 64


class Params(C.Structure):
    """Mirror of cfo_params / cfear_params (oracle/cfear_oracle.h, include/cfear_hip.h)."""
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
        ("max_solver_iterations", C.c_int32), ("reserved0", C.c_int32),
        ("assoc_radius", C.c_double),
    ]


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
assert CELL_DTYPE.itemsize == C.sizeof(Cell)


class RegSummary(C.Structure):
    _fields_ = [
        ("success", C.c_int32), ("usable", C.c_int32), ("outer_iterations", C.c_int32),
        ("num_residuals", C.c_int32), ("num_residual_blocks", C.c_int32), ("reserved", C.c_int32),
        ("final_cost", C.c_double), ("score", C.c_double),
        ("inner_iterations", C.c_int32 * CFO_MAX_OUTER), ("termination", C.c_int32 * CFO_MAX_OUTER),
        ("outer_cost", C.c_double * CFO_MAX_OUTER), ("outer_pose", (C.c_double * 3) * CFO_MAX_OUTER),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libcfear_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("cfear_oracle.c", "cfear_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcfear_oracle.so"], stdout=subprocess.DEVNULL)
    return so


PERT = {"voxel_reverse": 1, "voxel_random": 2, "voxel_stdsort": 4, "sum_reverse": 8, "sum_pairwise": 16, "wsum_eigen_redux": 32,
        "eig_jacobi": 64, "nn_tie_high": 128, "nn_tie_flann": 256}
_STDSORT = None


def set_perturbation(mask=0, seed=1):
    """[3P] sensitivity modes of the oracle (cfear_oracle.h; process-wide; 0 = the oracle as specified). mask: an int or a list of
    PERT names. 'voxel_stdsort' loads oracle/libcfear_stdsort.so (libstdc++'s std::sort called as PCL <= 1.9 calls it)."""
    global _STDSORT
    L = lib()
    if not isinstance(mask, int):
        mask = sum(PERT[m] for m in mask)
    if (mask & PERT["voxel_stdsort"]) and _STDSORT is None:
        so = os.path.join(_HERE, "libcfear_stdsort.so")
        src = os.path.join(_HERE, "stdsort_perm.cpp")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libcfear_stdsort.so"], stdout=subprocess.DEVNULL)
        _STDSORT = C.CDLL(so)
        L.cfo_set_voxel_sorter(C.cast(_STDSORT.cfo_stdsort_perm, C.c_void_p))
    L.cfo_set_perturbation(C.c_uint(mask), C.c_uint64(seed))
    return mask


def flann_nearest(pts, queries):
    """cfo_flann_nearest: (indices, squared distances) of the restated FLANN 1-NN search"""
    pts = np.ascontiguousarray(pts, dtype=np.float32); q = np.ascontiguousarray(queries, dtype=np.float32)
    idx = np.zeros(len(q), dtype=np.int32); d = np.zeros(len(q), dtype=np.float32)
    L = lib()
    L.cfo_flann_nearest.restype = None
    L.cfo_flann_nearest(pts.ctypes.data_as(C.c_void_p), C.c_int(len(pts)), q.ctypes.data_as(C.c_void_p), C.c_int(len(q)), idx.ctypes.data_as(C.c_void_p),
                        d.ctypes.data_as(C.c_void_p))
    return idx, d


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    u8p, u32p, f32p, f64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                             C.POINTER(C.c_double))
    L.cfo_default_params.argtypes = [C.POINTER(Params)]
    L.cfo_set_perturbation.argtypes = [C.c_uint, C.c_uint64]
    L.cfo_set_voxel_sorter.argtypes = [C.c_void_p]
    L.cfo_get_perturbation.restype = C.c_uint
    L.cfo_filter.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u32p]
    L.cfo_filter_bruteforce.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u32p]
    L.cfo_cloud.argtypes = [u32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, f32p]
    L.cfo_compensate.argtypes = [f32p, C.c_int, f64p, C.c_int]
    L.cfo_loss_eval.argtypes = [C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]
    L.cfo_loss_eval.restype = None
    L.cfo_cfar_scaling.argtypes = [C.c_int, C.c_double]
    L.cfo_cfar_scaling.restype = C.c_double
    L.cfo_cfar.argtypes = [u8p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_double, C.c_int, C.c_int, C.c_float, f32p, C.c_int]
    L.cfo_cfar_prefix.argtypes = L.cfo_cfar.argtypes
    L.cfo_cfar_prefix.restype = C.c_int
    L.cfo_scan_create.argtypes = [f32p, C.c_int, C.POINTER(Params), C.c_int]
    L.cfo_scan_create.restype = C.c_void_p
    L.cfo_scan_free.argtypes = [C.c_void_p]
    L.cfo_scan_size.argtypes = [C.c_void_p]
    L.cfo_scan_cells.argtypes = [C.c_void_p]
    L.cfo_scan_cells.restype = C.POINTER(Cell)
    L.cfo_scan_num_samples.argtypes = [C.c_void_p]
    L.cfo_scan_samples.argtypes = [C.c_void_p]
    L.cfo_scan_samples.restype = f32p
    L.cfo_scan_closest.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int]
    L.cfo_register.argtypes = [C.POINTER(C.c_void_p), C.c_int, f64p, f64p, C.POINTER(Params), C.c_int,
                               C.POINTER(RegSummary)]
    L.cfo_register_soft.argtypes = [C.POINTER(C.c_void_p), C.c_int, f64p, f64p, f64p, C.POINTER(Params), C.c_int, C.POINTER(RegSummary)]
    L.cfo_get_cost.argtypes = [C.POINTER(C.c_void_p), C.c_int, f64p, C.POINTER(Params), C.c_int, C.c_int, f64p, f64p, C.c_int]
    L.cfo_cov_by_sampling.argtypes = [C.POINTER(C.c_void_p), C.c_int, f64p, C.POINTER(Params), C.c_int, C.c_int, C.c_double, C.c_double,
                                      C.c_int, C.c_double, C.c_double, C.c_int, f64p, f64p]
    L.cfo_fuser_create.argtypes = [C.POINTER(Params)]
    L.cfo_fuser_create.restype = C.c_void_p
    L.cfo_fuser_free.argtypes = [C.c_void_p]
    L.cfo_fuser_process_polar.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, f64p]
    L.cfo_fuser_process_cloud.argtypes = [C.c_void_p, f32p, C.c_int, f64p]
    L.cfo_fuser_num_keyframes.argtypes = [C.c_void_p]
    L.cfo_fuser_last_summary.argtypes = [C.c_void_p]
    L.cfo_fuser_last_summary.restype = C.POINTER(RegSummary)
    L.cfo_fuser_last_scan.argtypes = [C.c_void_p]
    L.cfo_fuser_last_scan.restype = C.c_void_p
    L.cfo_fuser_timers.argtypes = [C.c_void_p, f64p]
    L.cfo_fuser_set_cov_sampling.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double]
    L.cfo_fuser_last_cov.argtypes = [C.c_void_p, f64p]
    _LIB = L
    return L


def default_params(**kw):
    p = Params()
    lib().cfo_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def filter_polar(img, z_min, k, brute=False):
    """img uint8 [A,R] -> packed slots uint32 [A,k]."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    A, R = img.shape
    out = np.zeros((A, k), dtype=np.uint32)
    fn = lib().cfo_filter_bruteforce if brute else lib().cfo_filter
    rc = fn(_ptr(img, C.c_uint8), A, R, int(z_min), int(k), _ptr(out, C.c_uint32))
    if rc != 0:
        raise ValueError("cfo_filter rc=%d" % rc)
    return out


def cloud(slots, range_res, min_distance, peaks=False):
    slots = np.ascontiguousarray(slots, dtype=np.uint32)
    A, k = slots.shape
    xyi = np.zeros((A * k, 3), dtype=np.float32)
    n = lib().cfo_cloud(_ptr(slots, C.c_uint32), A, k, np.float32(range_res), np.float32(min_distance),
                        int(peaks), _ptr(xyi, C.c_float))
    return xyi[:n].copy()


def cfar(img, range_res, static_threshold, min_distance, window_size=10, nb_guard_cells=20, false_alarm_rate=0.01, max_distance=400.0, prefix=False):
    """AzimuthCACFAR::getFilteredPointCloud with the defaults of radarDriver::Parameters (radar_driver.h:43-44). prefix=True: the window sums off a
    prefix sum (cfo_cfar_prefix: the same arithmetic, exact sums; for long windows)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    A, R = img.shape
    args = (_ptr(img, C.c_uint8), A, R, np.float32(range_res), np.float32(static_threshold), np.float32(min_distance),
            float(max_distance), int(window_size), int(nb_guard_cells), np.float32(false_alarm_rate))
    fn = lib().cfo_cfar_prefix if prefix else lib().cfo_cfar
    n = fn(*args, None, 0)
    xyi = np.zeros((max(n, 1), 3), dtype=np.float32)
    fn(*args, _ptr(xyi, C.c_float), n)
    return xyi[:n].copy()


def loss_eval(loss, loss_limit, s):
    """(rho, rho', rho'') of the loss the registration builds (registration.cpp:78-97), at squared residual norm s"""
    out = (C.c_double * 3)()
    lib().cfo_loss_eval(int(loss), float(loss_limit), float(s), out)
    return np.array(out[:])


def compensate(xyi, mot, ccw):
    xyi = np.ascontiguousarray(xyi, dtype=np.float32).copy()
    m = np.asarray(mot, dtype=np.float64).copy()
    lib().cfo_compensate(_ptr(xyi, C.c_float), xyi.shape[0], _ptr(m, C.c_double), int(ccw))
    return xyi


class Scan:
    def __init__(self, xyi, params, brute=False):
        xyi = np.ascontiguousarray(xyi, dtype=np.float32)
        self._h = lib().cfo_scan_create(_ptr(xyi, C.c_float), xyi.shape[0], C.byref(params), int(brute))
        if not self._h:
            raise ValueError("empty cloud")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().cfo_scan_free(self._h)
            self._h = None

    @property
    def size(self):
        return lib().cfo_scan_size(self._h)

    def cells(self):
        n = self.size
        if n == 0:
            return np.zeros(0, dtype=CELL_DTYPE)
        p = lib().cfo_scan_cells(self._h)
        buf = C.string_at(p, n * C.sizeof(Cell))
        return np.frombuffer(buf, dtype=CELL_DTYPE).copy()

    def samples(self):
        n = lib().cfo_scan_num_samples(self._h)
        p = lib().cfo_scan_samples(self._h)
        return np.ctypeslib.as_array(p, shape=(n, 3)).copy()

    def closest(self, x, y, d, brute=False):
        return lib().cfo_scan_closest(self._h, float(x), float(y), float(d), int(brute))


def register(scans, poses, params, brute=False):
    """scans: list[Scan]; poses [n,3] -> (ret, poses_out, cov6, RegSummary)."""
    n = len(scans)
    arr = (C.c_void_p * n)(*[s._h for s in scans])
    P = np.ascontiguousarray(poses, dtype=np.float64).copy()
    cov = np.zeros(36, dtype=np.float64)
    S = RegSummary()
    ret = lib().cfo_register(arr, n, _ptr(P, C.c_double), _ptr(cov, C.c_double), C.byref(params), int(brute),
                             C.byref(S))
    return ret, P, cov.reshape(6, 6), S


def register_soft(scans, poses, prior_cov6, params, brute=False):
    """Register(..., soft_constraints=true): prior_cov6 = the 6x6 covariance passed in for the last pose."""
    n = len(scans)
    arr = (C.c_void_p * n)(*[s._h for s in scans])
    P = np.ascontiguousarray(poses, dtype=np.float64).copy()
    pc = np.ascontiguousarray(prior_cov6, dtype=np.float64).reshape(36).copy()
    cov = np.zeros(36, dtype=np.float64)
    S = RegSummary()
    ret = lib().cfo_register_soft(arr, n, _ptr(P, C.c_double), _ptr(pc, C.c_double), _ptr(cov, C.c_double), C.byref(params), int(brute),
                                  C.byref(S))
    return ret, P, cov.reshape(6, 6), S


def get_cost(scans, poses, params, itr=2, brute=False):
    """n_scan_normal_reg::GetCost -> (score, residuals) or None where the reference returns false."""
    n = len(scans)
    arr = (C.c_void_p * n)(*[s._h for s in scans])
    P = np.ascontiguousarray(poses, dtype=np.float64).copy()
    cap = 2 * (n - 1) * max(len(scans[-1].cells()), 1)
    res = np.zeros(cap, dtype=np.float64)
    score = np.zeros(1, dtype=np.float64)
    m = lib().cfo_get_cost(arr, n, _ptr(P, C.c_double), C.byref(params), int(itr), int(brute), _ptr(score, C.c_double),
                           _ptr(res, C.c_double), cap)
    if m < 0:
        return None
    return float(score[0]), res[:m].copy()


def cov_by_sampling(scans, poses, params, final_cost, num_residuals, itr=2, xy_range=0.4, yaw_range=0.0043625, steps=3,
                    cov_scaler=4.0, brute=False):
    """approximateCovarianceBySampling with the defaults of OdometryKeyframeFuser::Parameters (odometrykeyframefuser.h:107-110)
    -> (success, cov6x6, sampled costs)"""
    n = len(scans)
    arr = (C.c_void_p * n)(*[s._h for s in scans])
    P = np.ascontiguousarray(poses, dtype=np.float64).copy()
    cov = np.zeros(36)
    costs = np.zeros(steps ** 3)
    ok = lib().cfo_cov_by_sampling(arr, n, _ptr(P, C.c_double), C.byref(params), int(itr), int(brute), float(xy_range), float(yaw_range),
                                   int(steps), float(cov_scaler), float(final_cost), int(num_residuals), _ptr(cov, C.c_double),
                                   _ptr(costs, C.c_double))
    return bool(ok), cov.reshape(6, 6), costs


class Fuser:
    def __init__(self, params):
        self._h = lib().cfo_fuser_create(C.byref(params))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().cfo_fuser_free(self._h)
            self._h = None

    def process_polar(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        pose = np.zeros(3)
        rc = lib().cfo_fuser_process_polar(self._h, _ptr(img, C.c_uint8), img.shape[0], img.shape[1],
                                           _ptr(pose, C.c_double))
        if rc != 0:
            raise RuntimeError("cfo_fuser_process_polar rc=%d" % rc)
        return pose

    def process_cloud(self, xyi):
        xyi = np.ascontiguousarray(xyi, dtype=np.float32).copy()
        pose = np.zeros(3)
        rc = lib().cfo_fuser_process_cloud(self._h, _ptr(xyi, C.c_float), xyi.shape[0], _ptr(pose, C.c_double))
        if rc != 0:
            raise RuntimeError("cfo_fuser_process_cloud rc=%d" % rc)
        return pose

    def set_cov_sampling(self, enable=True, xy_range=0.4, yaw_range=0.0043625, steps=3, scaler=4.0):
        """Parameters::estimate_cov_by_sampling and its companions (odometrykeyframefuser.h:104-110)"""
        lib().cfo_fuser_set_cov_sampling(self._h, int(enable), float(xy_range), float(yaw_range), int(steps), float(scaler))

    def last_cov(self):
        """cov_current after the last sweep, 6 x 6"""
        c = np.zeros(36)
        lib().cfo_fuser_last_cov(self._h, _ptr(c, C.c_double))
        return c.reshape(6, 6)

    @property
    def num_keyframes(self):
        return lib().cfo_fuser_num_keyframes(self._h)

    def last_summary(self):
        return lib().cfo_fuser_last_summary(self._h).contents

    def last_cells(self):
        h = lib().cfo_fuser_last_scan(self._h)
        n = lib().cfo_scan_size(h)
        if n == 0:
            return np.zeros(0, dtype=CELL_DTYPE)
        buf = C.string_at(lib().cfo_scan_cells(h), n * C.sizeof(Cell))
        return np.frombuffer(buf, dtype=CELL_DTYPE).copy()

    def timers(self):
        t = np.zeros(4)
        lib().cfo_fuser_timers(self._h, _ptr(t, C.c_double))
        return dict(zip(("Filtering", "compensate", "build_normals", "register"), t.tolist()))


def unpack_slots(slots):
    s = np.asarray(slots, dtype=np.uint32)
    return {"range": (s & 0xFFFF).astype(np.int32), "intensity": ((s >> 16) & 0xFF).astype(np.int32),
            "valid": ((s >> 24) & 1).astype(bool), "peak": ((s >> 25) & 1).astype(bool)}
