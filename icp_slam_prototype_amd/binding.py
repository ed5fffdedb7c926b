"""ctypes binding of include/icpk.h (lib/libicpk.so).  No torch types cross the
boundary: device buffers are passed as integer addresses.

There is no CPU fallback: if the library is missing `load()` raises, and
`Context()` raises when no HIP device is usable (ICPK_E_NO_DEVICE).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICPK_LIB_PATH") or os.path.join(HERE, "lib", "libicpk.so")  # (ICPK_LIB_PATH: a diagnostic variant build)

OK = 0
W_TOO_FEW_PAIRS = 1
E_ARG, E_EMPTY_TARGET, E_HIP, E_NOT_SET, E_NO_DEVICE, E_RCCL = -1, -2, -3, -4, -5, -6
COMM_ID_BYTES = 128
SOLVE_REFERENCE, SOLVE_KABSCH, SOLVE_POINT_TO_PLANE = 0, 1, 2
W_DEGENERATE = 2
W_EMPTY_MAP = 3
MAX_NN_KEYPOINT_DISTANCE = 0.1  # icp.hpp:10
NORMALS_CROSS, NORMALS_REFERENCE = 0, 1
SUBSAMPLE_FACTOR = 40  # pointcloud.hpp:11
NP2L = 28
NN_EXACT, NN_FILTERED, NN_PRUNED, NN_GRID = 0, 1, 2, 3
NSUM = 19

# every symbol include/icpk.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "icpk_version", "icpk_create", "icpk_destroy", "icpk_last_error", "icpk_default_params",
    "icpk_set_log_callback", "icpk_stream", "icpk_set_target", "icpk_set_source", "icpk_set_target_device",
    "icpk_set_source_device", "icpk_reset_source", "icpk_commit_source", "icpk_get_source", "icpk_get_target", "icpk_source_size", "icpk_target_size",
    "icpk_nn", "icpk_reduce", "icpk_transform_source", "icpk_transform_target", "icpk_get_trace",
    "icpk_get_associations", "icpk_align",
    "icpk_align_batch", "icpk_align_batch_device", "icpk_backproject", "icpk_backproject_with_normals", "icpk_set_target_normals",
    "icpk_get_target_normals", "icpk_reduce_p2l", "icpk_solve_point_to_plane", "icpk_pair_distance", "icpk_pair_distance3", "icpk_distance3", "icpk_make_rotation_matrix",
    "icpk_matrix_to_quaternion", "icpk_quaternion_to_euler", "icpk_solve_reference", "icpk_solve_kabsch",
    "icpk_associate_keypoints", "icpk_filter_depth_image", "icpk_backproject_filtered", "icpk_backproject_pair", "icpk_set_subsample", "icpk_backproject_keypoints", "icpk_register_host_buffer", "icpk_unregister_host_buffer",
    "icpk_comm_unique_id", "icpk_comm_init_rccl", "icpk_comm_destroy", "icpk_comm_rank", "icpk_comm_world",
    "icpk_comm_partition", "icpk_comm_broadcast_target", "icpk_comm_gather_results", "icpk_comm_allreduce_sums",
    "icpk_comm_barrier", "icpk_align_query_sharded",
]


class Params(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32),
        ("threshold", C.c_float),
        ("max_nn_dist", C.c_float),
        ("min_pairs", C.c_int32),
        ("solve", C.c_int32),
        ("fixed_iterations", C.c_int32),
        ("nn_mode", C.c_int32),
        ("profile", C.c_int32),
        ("last_rotation", C.c_float * 9),
        ("last_translation", C.c_float * 3),
        ("host_loop", C.c_int32),
        ("profile_stride", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("status", C.c_int32),
        ("final_pairs", C.c_int32),
        ("final_mse", C.c_float),
        ("nn_launches", C.c_int32),
        ("nn_timed_launches", C.c_int32),
        ("nn_ms_total", C.c_float),
        ("reduce_ms_total", C.c_float),
        ("transform_ms_total", C.c_float),
        ("total_ms", C.c_float),
    ]


class Pair(C.Structure):
    _fields_ = [
        ("sx", C.POINTER(C.c_float)), ("sy", C.POINTER(C.c_float)), ("sz", C.POINTER(C.c_float)),
        ("ns", C.c_int32),
        ("tx", C.POINTER(C.c_float)), ("ty", C.POINTER(C.c_float)), ("tz", C.POINTER(C.c_float)),
        ("nt", C.c_int32),
        ("idx_out", C.POINTER(C.c_int32)), ("dist_out", C.POINTER(C.c_float)),
    ]


LOG_FN = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_double, C.c_void_p)

_lib = None


class IcpkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"icpk status {code}: {msg}")
        self.code = code


def load():
    """Load libicpk.so.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    lib.icpk_version.restype = C.c_char_p
    lib.icpk_last_error.restype = C.c_char_p
    lib.icpk_last_error.argtypes = [C.c_void_p]
    lib.icpk_stream.restype = C.c_void_p
    lib.icpk_stream.argtypes = [C.c_void_p]
    lib.icpk_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.icpk_destroy.argtypes = [C.c_void_p]
    lib.icpk_destroy.restype = None
    fp = C.POINTER(C.c_float)
    for name in ("icpk_set_target", "icpk_set_source"):
        getattr(lib, name).argtypes = [C.c_void_p, fp, fp, fp, C.c_int32]
    for name in ("icpk_set_target_device", "icpk_set_source_device"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.icpk_reset_source.argtypes = [C.c_void_p]
    lib.icpk_commit_source.argtypes = [C.c_void_p]
    lib.icpk_get_source.argtypes = [C.c_void_p, fp, fp, fp]
    lib.icpk_get_target.argtypes = [C.c_void_p, fp, fp, fp]
    lib.icpk_source_size.argtypes = [C.c_void_p]
    lib.icpk_target_size.argtypes = [C.c_void_p]
    lib.icpk_nn.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), fp]
    lib.icpk_reduce.argtypes = [C.c_void_p, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.icpk_transform_source.argtypes = [C.c_void_p, fp, fp]
    lib.icpk_transform_target.argtypes = [C.c_void_p, fp, fp]
    lib.icpk_get_trace.argtypes = [C.c_void_p, C.POINTER(C.c_int32), fp, fp, C.POINTER(C.c_int32), fp]
    lib.icpk_get_associations.argtypes = [C.c_void_p, C.POINTER(C.c_int32), fp]
    lib.icpk_align.argtypes = [C.c_void_p, C.POINTER(Params), fp, C.POINTER(Stats)]
    lib.icpk_align_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Pair), C.POINTER(Params), fp, C.POINTER(Stats)]
    lib.icpk_align_batch_device.argtypes = lib.icpk_align_batch.argtypes
    lib.icpk_set_subsample.argtypes = [C.c_void_p, C.c_int32, C.c_uint64]
    lib.icpk_register_host_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.icpk_unregister_host_buffer.argtypes = [C.c_void_p, C.c_void_p]
    lib.icpk_backproject.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), C.c_int32, C.c_int32, C.c_float, C.c_float,
                                     fp, C.c_int32]
    lib.icpk_pair_distance.argtypes = [C.c_void_p, fp, fp, fp, C.c_int32]
    lib.icpk_pair_distance3.argtypes = [C.c_void_p, fp, fp, fp, C.c_int32]
    lib.icpk_distance3.argtypes = [fp, fp]
    lib.icpk_distance3.restype = C.c_float
    lib.icpk_backproject_with_normals.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), C.c_int32, C.c_int32, C.c_float,
                                                  C.c_float, fp, C.c_int32]
    lib.icpk_set_target_normals.argtypes = [C.c_void_p, fp, fp, fp, C.c_int32]
    lib.icpk_get_target_normals.argtypes = [C.c_void_p, fp, fp, fp]
    lib.icpk_reduce_p2l.argtypes = [C.c_void_p, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.icpk_solve_point_to_plane.argtypes = [C.POINTER(C.c_double)] * 3
    lib.icpk_default_params.argtypes = [C.POINTER(Params)]
    lib.icpk_default_params.restype = None
    lib.icpk_set_log_callback.argtypes = [C.c_void_p, LOG_FN, C.c_void_p]
    lib.icpk_make_rotation_matrix.argtypes = [C.c_float, C.c_float, C.c_float, fp]
    lib.icpk_backproject_keypoints.argtypes = [C.POINTER(C.c_uint16), C.c_int32, C.c_int32, fp, C.c_int32, C.c_float, C.c_float, fp,
                                               C.POINTER(C.c_int32)]
    lib.icpk_make_rotation_matrix.restype = None
    lib.icpk_matrix_to_quaternion.argtypes = [fp, fp]
    lib.icpk_matrix_to_quaternion.restype = None
    lib.icpk_quaternion_to_euler.argtypes = [fp, fp]
    lib.icpk_quaternion_to_euler.restype = None
    lib.icpk_solve_reference.argtypes = [fp, fp]
    lib.icpk_solve_reference.restype = None
    dp = C.POINTER(C.c_double)
    lib.icpk_solve_kabsch.argtypes = [C.c_int64, dp, dp, dp, dp, dp]
    lib.icpk_solve_kabsch.restype = None
    ip = C.POINTER(C.c_int32)
    lib.icpk_associate_keypoints.argtypes = [C.c_void_p, C.c_int32, C.c_float, ip, ip, fp, ip, ip, C.c_int32, ip]
    u16 = C.POINTER(C.c_uint16)
    lib.icpk_filter_depth_image.argtypes = [C.c_void_p, u16, u16, C.c_int32, C.c_int32] + [C.c_int32] * 5
    lib.icpk_backproject_filtered.argtypes = [C.c_void_p, u16, C.c_int32, C.c_int32, C.c_float, C.c_float, fp,
                                              C.c_int32, C.c_int32] + [C.c_int32] * 5
    lib.icpk_backproject_pair.argtypes = [C.c_void_p, u16, u16, C.c_int32, C.c_int32, C.c_float, C.c_float, fp, fp, fp] + \
        [C.c_int32] * 6 + [C.POINTER(C.c_int32)] * 2
    lib.icpk_comm_unique_id.argtypes = [C.c_void_p]
    lib.icpk_comm_init_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.icpk_comm_destroy.argtypes = [C.c_void_p]
    lib.icpk_comm_rank.argtypes = [C.c_void_p]
    lib.icpk_comm_world.argtypes = [C.c_void_p]
    lib.icpk_comm_partition.argtypes = [C.c_int32, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.icpk_comm_partition.restype = None
    lib.icpk_comm_broadcast_target.argtypes = [C.c_void_p, C.c_int]
    lib.icpk_comm_gather_results.argtypes = [C.c_void_p, fp, C.POINTER(Stats), C.c_int32, C.c_int32, fp, fp]
    lib.icpk_comm_allreduce_sums.argtypes = [C.c_void_p, dp, C.c_int32, C.POINTER(C.c_int64)]
    lib.icpk_comm_barrier.argtypes = [C.c_void_p]
    lib.icpk_align_query_sharded.argtypes = [C.c_void_p, C.POINTER(Params), fp, C.POINTER(Stats)]
    _lib = lib
    return lib


def comm_unique_id():
    """ncclGetUniqueId through the C ABI: 128 opaque bytes rank 0 ships to the other ranks."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().icpk_comm_unique_id(buf)
    if rc != OK:
        raise IcpkError(rc, "icpk_comm_unique_id failed (librccl.so.1 not loadable?)")
    return buf.raw


def comm_partition(n_items, world, rank):
    s, c = C.c_int32(0), C.c_int32(0)
    load().icpk_comm_partition(n_items, world, rank, C.byref(s), C.byref(c))
    return s.value, c.value


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def default_params(**kw):
    p = Params()
    load().icpk_default_params(C.byref(p))
    for k, v in kw.items():
        if k in ("last_rotation", "last_translation"):
            arr = _f(v).reshape(-1)
            getattr(p, k)[:] = [float(x) for x in arr]
        else:
            setattr(p, k, v)
    return p


# ---- host helpers (no device) ------------------------------------------------
def backproject_keypoints(depth, kp_xy, fx=468.60, cx=318.27):
    """pointcloud.cpp:60-98 (host only): (points (3, m) float32, kept (m,) indices into kp_xy)."""
    depth = np.ascontiguousarray(depth, np.uint16)
    kp = np.ascontiguousarray(kp_xy, np.float32).reshape(-1, 2)
    out = np.zeros((max(len(kp), 1), 3), np.float32)
    kept = np.zeros(max(len(kp), 1), np.int32)
    m = load().icpk_backproject_keypoints(depth.ctypes.data_as(C.POINTER(C.c_uint16)), depth.shape[0], depth.shape[1], _fp(kp),
                                          len(kp), fx, cx, _fp(out), kept.ctypes.data_as(C.POINTER(C.c_int32)))
    if m < 0:
        raise IcpkError(m, "icpk_backproject_keypoints")
    return np.ascontiguousarray(out[:m].T), kept[:m].copy()


def make_rotation_matrix(x, y, z):
    out = np.zeros(9, np.float32)
    load().icpk_make_rotation_matrix(x, y, z, _fp(out))
    return out.reshape(3, 3)


def matrix_to_quaternion(m):
    m = _f(m).reshape(9)
    q = np.zeros(4, np.float32)
    load().icpk_matrix_to_quaternion(_fp(m), _fp(q))
    return q


def quaternion_to_euler(q):
    q = _f(q).reshape(4)
    e = np.zeros(3, np.float32)
    load().icpk_quaternion_to_euler(_fp(q), _fp(e))
    return e


def distance3(a, b):
    """icp.cpp:595-602 distance(cv::Point3f, cv::Point3f) on the host."""
    a = _f(a).reshape(3)
    b = _f(b).reshape(3)
    return np.float32(load().icpk_distance3(_fp(a), _fp(b)))


def solve_reference(M):
    M = _f(M).reshape(9)
    R = np.zeros(9, np.float32)
    load().icpk_solve_reference(_fp(M), _fp(R))
    return R.reshape(3, 3)


def solve_point_to_plane(sums):
    dp = C.POINTER(C.c_double)
    sums = np.ascontiguousarray(sums, np.float64)
    R = np.zeros(9)
    t = np.zeros(3)
    rc = load().icpk_solve_point_to_plane(sums.ctypes.data_as(dp), R.ctypes.data_as(dp), t.ctypes.data_as(dp))
    return R.reshape(3, 3), t, rc


def solve_kabsch(n, sa, sb, sab):
    dp = C.POINTER(C.c_double)
    sa = np.ascontiguousarray(sa, np.float64)
    sb = np.ascontiguousarray(sb, np.float64)
    sab = np.ascontiguousarray(sab, np.float64).reshape(9)
    R = np.zeros(9)
    t = np.zeros(3)
    load().icpk_solve_kabsch(int(n), sa.ctypes.data_as(dp), sb.ctypes.data_as(dp), sab.ctypes.data_as(dp),
                             R.ctypes.data_as(dp), t.ctypes.data_as(dp))
    return R.reshape(3, 3), t


class Context:
    """One GPU + one HIP stream (icpk_ctx)."""

    def __init__(self, device=0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.icpk_create(C.byref(h), int(device))
        if rc != OK:
            raise IcpkError(rc, "icpk_create failed (no usable HIP device: there is no CPU fallback)")
        self._h = h
        self._log_ref = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.icpk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise IcpkError(rc, self._lib.icpk_last_error(self._h).decode())
        return rc

    @property
    def stream(self):
        return self._lib.icpk_stream(self._h)

    # -- clouds ---------------------------------------------------------------
    def set_target(self, pts):
        x, y, z = (_f(pts[k]) for k in range(3))
        self._chk(self._lib.icpk_set_target(self._h, _fp(x), _fp(y), _fp(z), x.size))

    def set_source(self, pts):
        x, y, z = (_f(pts[k]) for k in range(3))
        self._chk(self._lib.icpk_set_source(self._h, _fp(x), _fp(y), _fp(z), x.size))

    def set_target_device(self, px, py, pz, n):
        self._chk(self._lib.icpk_set_target_device(self._h, px, py, pz, n))

    def set_source_device(self, px, py, pz, n):
        self._chk(self._lib.icpk_set_source_device(self._h, px, py, pz, n))

    def reset_source(self):
        self._chk(self._lib.icpk_reset_source(self._h))

    def commit_source(self):
        self._chk(self._lib.icpk_commit_source(self._h))

    def get_source(self):
        n = self._lib.icpk_source_size(self._h)
        out = np.empty((3, n), np.float32)
        self._chk(self._lib.icpk_get_source(self._h, _fp(out[0]), _fp(out[1]), _fp(out[2])))
        return out

    def get_target(self):
        n = self._lib.icpk_target_size(self._h)
        out = np.empty((3, n), np.float32)
        self._chk(self._lib.icpk_get_target(self._h, _fp(out[0]), _fp(out[1]), _fp(out[2])))
        return out

    @property
    def source_size(self):
        return self._lib.icpk_source_size(self._h)

    @property
    def target_size(self):
        return self._lib.icpk_target_size(self._h)

    # -- steps ------------------------------------------------------------------
    def nn(self, nn_mode=NN_EXACT, fetch=True):
        n = self.source_size
        if not fetch:
            self._chk(self._lib.icpk_nn(self._h, nn_mode, None, None))
            return None
        idx = np.empty(n, np.int32)
        dist = np.empty(n, np.float32)
        self._chk(self._lib.icpk_nn(self._h, nn_mode, idx.ctypes.data_as(C.POINTER(C.c_int32)), _fp(dist)))
        return idx, dist

    def get_associations(self):
        n = self.source_size
        idx = np.empty(n, np.int32)
        dist = np.empty(n, np.float32)
        self._chk(self._lib.icpk_get_associations(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), _fp(dist)))
        return idx, dist

    def reduce(self, max_dist=0.75):
        sums = np.zeros(NSUM, np.float64)
        cnt = C.c_int64(0)
        self._chk(self._lib.icpk_reduce(self._h, max_dist, sums.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt)))
        return sums, cnt.value

    def transform_source(self, R, t):
        R = _f(R).reshape(9)
        t = _f(t).reshape(3)
        self._chk(self._lib.icpk_transform_source(self._h, _fp(R), _fp(t)))

    def transform_target(self, R, t):
        R = _f(R).reshape(9)
        t = _f(t).reshape(3)
        self._chk(self._lib.icpk_transform_target(self._h, _fp(R), _fp(t)))

    def get_trace(self, max_iterations=64):
        n = C.c_int32(0)
        R = np.zeros((max_iterations, 9), np.float32)
        t = np.zeros((max_iterations, 3), np.float32)
        pairs = np.zeros(max_iterations, np.int32)
        mse = np.zeros(max_iterations, np.float32)
        self._chk(self._lib.icpk_get_trace(self._h, C.byref(n), _fp(R), _fp(t),
                                           pairs.ctypes.data_as(C.POINTER(C.c_int32)), _fp(mse)))
        k = n.value
        return [dict(R=R[i].reshape(3, 3).copy(), t=t[i].copy(), n_pairs=int(pairs[i]), mse=np.float32(mse[i]))
                for i in range(k)]

    def pair_distance(self, a, b, point3=False):
        """point3: the cv::Point3f overload (icp.cpp:595-602) instead of the loop's distance."""
        a = _f(a)
        b = _f(b)
        n = a.shape[1]
        out = np.empty(n, np.float32)
        fn = self._lib.icpk_pair_distance3 if point3 else self._lib.icpk_pair_distance
        self._chk(fn(self._h, _fp(a), _fp(b), _fp(out), n))
        return out

    def register_host_buffer(self, arr):
        """Pins a C-contiguous numpy array for the device (icpk_register_host_buffer): depth images that lie inside it
        are read by icpk_backproject_pair where they are.  Keep the array alive until unregister_host_buffer / close."""
        if not arr.flags["C_CONTIGUOUS"]:
            raise ValueError("a C-contiguous array expected")
        self._chk(self._lib.icpk_register_host_buffer(self._h, arr.ctypes.data, arr.nbytes))

    def unregister_host_buffer(self, arr):
        self._chk(self._lib.icpk_unregister_host_buffer(self._h, arr.ctypes.data))

    def set_subsample(self, factor=40, seed=0):
        """pointcloud.cpp:27-30 with a reproducible choice: every back-projection keeps one valid pixel in `factor`
        (0 / 1: all); see include/icpk.h.  oracle.subsample_keep gives the same mask."""
        self._chk(self._lib.icpk_set_subsample(self._h, int(factor), int(seed)))

    def backproject(self, depth, which=0, fx=468.60, cx=318.27, offset=None):
        depth = np.ascontiguousarray(depth, np.uint16)
        rows, cols = depth.shape
        off = None if offset is None else _f(offset)
        n = self._chk(self._lib.icpk_backproject(self._h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), rows, cols,
                                                 fx, cx, None if off is None else _fp(off), which))
        return n

    def filter_depth_image(self, depth, max_d=25000, min_d=1000, morph=True, anchor=(-1, -1)):
        """SLAM.cpp:553-574 on the device; anchor = (x, y) of dilate/erode, -1 = element centre."""
        depth = np.ascontiguousarray(depth, np.uint16)
        out = np.empty_like(depth)
        rows, cols = depth.shape
        u16 = C.POINTER(C.c_uint16)
        self._chk(self._lib.icpk_filter_depth_image(self._h, depth.ctypes.data_as(u16), out.ctypes.data_as(u16), rows, cols,
                                                    int(max_d), int(min_d), int(bool(morph)), int(anchor[0]), int(anchor[1])))
        return out

    def backproject_filtered(self, depth, which=0, normals_mode=-1, fx=468.60, cx=318.27, offset=None, max_d=25000,
                             min_d=1000, morph=True, anchor=(-1, -1)):
        depth = np.ascontiguousarray(depth, np.uint16)
        rows, cols = depth.shape
        off = None if offset is None else _f(offset)
        return self._chk(self._lib.icpk_backproject_filtered(
            self._h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), rows, cols, fx, cx, None if off is None else _fp(off),
            which, normals_mode, int(max_d), int(min_d), int(bool(morph)), int(anchor[0]), int(anchor[1])))

    def backproject_pair(self, depth_source, depth_target, R=None, t=None, fx=468.60, cx=318.27, offset=None, filter=False,
                         max_d=25000, min_d=1000, morph=True, anchor=(-1, -1)):
        """icp.cpp:38-71 in one call: both frames back-projected (filtered first if filter), posed by (R, t),
        the posed source committed.  Returns (n_source, n_target)."""
        ds = np.ascontiguousarray(depth_source, np.uint16)
        # depth_target None: the frame this context received as depth_source last time, still on the device (SLAM.cpp:305)
        dt = None if depth_target is None else np.ascontiguousarray(depth_target, np.uint16)
        if ds.ndim != 2 or (dt is not None and ds.shape != dt.shape):
            raise ValueError("two depth images of the same rows x cols shape expected")
        rows, cols = ds.shape
        off = None if offset is None else _f(offset)
        Rm = None if R is None else _f(R)
        tv = None if t is None else _f(t)
        ns, nt = C.c_int32(-1), C.c_int32(-1)
        u16 = C.POINTER(C.c_uint16)
        self._chk(self._lib.icpk_backproject_pair(
            self._h, ds.ctypes.data_as(u16), None if dt is None else dt.ctypes.data_as(u16), rows, cols, fx, cx, None if off is None else _fp(off),
            None if Rm is None else _fp(Rm), None if tv is None else _fp(tv), int(bool(filter)), int(max_d), int(min_d),
            int(bool(morph)), int(anchor[0]), int(anchor[1]), C.byref(ns), C.byref(nt)))
        return ns.value, nt.value

    def associate_keypoints(self, max_dist=MAX_NN_KEYPOINT_DISTANCE, nn_mode=NN_GRID, rejected=None, capacity=None):
        """icp.cpp:488-515 on the context's clouds.  rejected: the caller's running list of rejected
        query indices (None = empty).  Returns (status, assoc_q, assoc_t, assoc_d, rejected) --
        with status W_EMPTY_MAP the other entries are None / the unchanged list."""
        n = self.source_size
        prev = np.asarray([] if rejected is None else rejected, np.int32)
        cap = prev.size + max(n, 1) if capacity is None else int(capacity)
        aq = np.full(max(n, 1), -1, np.int32)
        at = np.full(max(n, 1), -1, np.int32)
        ad = np.full(max(n, 1), np.nan, np.float32)
        rj = np.full(max(cap, 1), -1, np.int32)
        rj[:prev.size] = prev
        na, nr = C.c_int32(-1), C.c_int32(prev.size)
        ip = C.POINTER(C.c_int32)
        rc = self._chk(self._lib.icpk_associate_keypoints(self._h, nn_mode, max_dist, aq.ctypes.data_as(ip), at.ctypes.data_as(ip),
                                                          _fp(ad), C.byref(na), rj.ctypes.data_as(ip), cap, C.byref(nr)))
        if rc == W_EMPTY_MAP:
            assert na.value == -1 and nr.value == prev.size  # untouched
            return rc, None, None, None, prev
        return rc, aq[:na.value].copy(), at[:na.value].copy(), ad[:na.value].copy(), rj[:nr.value].copy()

    # -- point-to-plane extension ------------------------------------------------
    def backproject_with_normals(self, depth, normals_mode=NORMALS_CROSS, fx=468.60, cx=318.27, offset=None):
        depth = np.ascontiguousarray(depth, np.uint16)
        rows, cols = depth.shape
        off = None if offset is None else _f(offset)
        return self._chk(self._lib.icpk_backproject_with_normals(
            self._h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), rows, cols, fx, cx,
            None if off is None else _fp(off), normals_mode))

    def set_target_normals(self, nrm):
        x, y, z = (_f(nrm[k]) for k in range(3))
        self._chk(self._lib.icpk_set_target_normals(self._h, _fp(x), _fp(y), _fp(z), x.size))

    def get_target_normals(self):
        out = np.empty((3, self.target_size), np.float32)
        self._chk(self._lib.icpk_get_target_normals(self._h, _fp(out[0]), _fp(out[1]), _fp(out[2])))
        return out

    def reduce_p2l(self, max_dist=0.75):
        sums = np.zeros(NP2L, np.float64)
        cnt = C.c_int64(0)
        self._chk(self._lib.icpk_reduce_p2l(self._h, max_dist, sums.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt)))
        return sums, cnt.value

    # -- loop ---------------------------------------------------------------------
    def align(self, params=None, **kw):
        p = params if params is not None else default_params(**kw)
        T = np.zeros(16, np.float32)
        st = Stats()
        rc = self._chk(self._lib.icpk_align(self._h, C.byref(p), _fp(T), C.byref(st)))
        return T.reshape(4, 4), st, rc

    def align_query_sharded(self, params=None, **kw):
        """One pair, queries sharded over the communicator's ranks (collective call): this context's source is the
        rank's slice, the target is the same everywhere; one in-stream all-reduce of 20 doubles per iteration, no host
        round trip.  Returns (T, stats, rc), identical on every rank."""
        p = params if params is not None else default_params(**kw)
        T = np.zeros(16, np.float32)
        st = Stats()
        rc = self._chk(self._lib.icpk_align_query_sharded(self._h, C.byref(p), _fp(T), C.byref(st)))
        return T.reshape(4, 4), st, rc

    def align_batch(self, pairs, params=None, associations=False, **kw):
        """pairs: list of (source (3,Ns), target (3,Nt)) host arrays.  Returns (T (n,4,4), stats list,
        rc) and, with associations=True, additionally a list of (idx, dist) per pair."""
        p = params if params is not None else default_params(**kw)
        n = len(pairs)
        arr = (Pair * max(n, 1))()
        keep = []
        assoc = []
        for b, (s, t) in enumerate(pairs):
            sx, sy, sz = (_f(s[k]) for k in range(3))
            tx, ty, tz = (_f(t[k]) for k in range(3))
            keep.append((sx, sy, sz, tx, ty, tz))
            arr[b].sx, arr[b].sy, arr[b].sz, arr[b].ns = _fp(sx), _fp(sy), _fp(sz), sx.size
            arr[b].tx, arr[b].ty, arr[b].tz, arr[b].nt = _fp(tx), _fp(ty), _fp(tz), tx.size
            if associations:
                idx = np.full(sx.size, -1, np.int32)
                dist = np.full(sx.size, np.nan, np.float32)
                assoc.append((idx, dist))
                arr[b].idx_out = idx.ctypes.data_as(C.POINTER(C.c_int32))
                arr[b].dist_out = _fp(dist)
        T = np.zeros((max(n, 1), 16), np.float32)
        st = (Stats * max(n, 1))()
        rc = self._lib.icpk_align_batch(self._h, n, arr, C.byref(p), _fp(T), st)
        out = (T[:n].reshape(n, 4, 4), list(st)[:n], rc)
        return out + (assoc,) if associations else out

    def align_batch_device(self, pairs, params=None, **kw):
        """pairs: list of (src_ptr, ns, tgt_ptr, nt): device addresses of (3, N) float32 xyz-SoA
        blocks (plane stride = N floats) resident on this context's device."""
        p = params if params is not None else default_params(**kw)
        n = len(pairs)
        arr = (Pair * max(n, 1))()
        fpt = C.POINTER(C.c_float)
        for b, (sp, ns, tp, nt) in enumerate(pairs):
            arr[b].sx, arr[b].sy, arr[b].sz = (C.cast(C.c_void_p(sp + 4 * ns * k), fpt) for k in range(3))
            arr[b].tx, arr[b].ty, arr[b].tz = (C.cast(C.c_void_p(tp + 4 * nt * k), fpt) for k in range(3))
            arr[b].ns, arr[b].nt = ns, nt
        T = np.zeros((max(n, 1), 16), np.float32)
        st = (Stats * max(n, 1))()
        rc = self._lib.icpk_align_batch_device(self._h, n, arr, C.byref(p), _fp(T), st)
        return T[:n].reshape(n, 4, 4), list(st)[:n], rc

    # -- RCCL collectives behind the C ABI (icpk_comm.cpp) -------------------------------
    def comm_init(self, unique_id, rank, world):
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        self._chk(self._lib.icpk_comm_init_rccl(self._h, buf, rank, world))

    def comm_destroy(self):
        self._chk(self._lib.icpk_comm_destroy(self._h))

    @property
    def comm_rank(self):
        return self._lib.icpk_comm_rank(self._h)

    @property
    def comm_world(self):
        return self._lib.icpk_comm_world(self._h)

    def comm_broadcast_target(self, root=0):
        self._chk(self._lib.icpk_comm_broadcast_target(self._h, root))

    def comm_gather_results(self, T_local, stats_local, n_total):
        """T_local (b, 4, 4) float32 and the list of Stats of this rank's block -> (T (n_total, 4, 4),
        S (n_total, 4) float64 = iterations, status, pairs, mse), identical on every rank.  (The integers cross the
        collective as int32 bit patterns: exact for any cloud size.)"""
        T_local = np.ascontiguousarray(T_local, np.float32).reshape(-1, 16)
        b = T_local.shape[0]
        st = (Stats * max(b, 1))()
        for k in range(b):
            if isinstance(stats_local[k], Stats):
                st[k] = stats_local[k]
            else:  # a row [iterations, status, pairs, mse] (batch.stats_rows)
                it, status, pairs, mse = stats_local[k]
                st[k].iterations, st[k].status, st[k].final_pairs, st[k].final_mse = int(it), int(status), int(pairs), float(mse)
        T = np.zeros((max(n_total, 1), 16), np.float32)
        S = np.zeros((max(n_total, 1), 4), np.float32)
        self._chk(self._lib.icpk_comm_gather_results(self._h, _fp(T_local) if b else None, st, b, n_total, _fp(T), _fp(S)))
        out = np.empty((n_total, 4), np.float64)
        out[:, :3] = S[:n_total, :3].view(np.int32)
        out[:, 3] = S[:n_total, 3]
        return T[:n_total].reshape(n_total, 4, 4), out

    def comm_allreduce_sums(self, sums, count):
        sums = np.ascontiguousarray(sums, np.float64).copy()
        cnt = C.c_int64(int(count))
        self._chk(self._lib.icpk_comm_allreduce_sums(self._h, sums.ctypes.data_as(C.POINTER(C.c_double)), sums.size,
                                                     C.byref(cnt)))
        return sums, cnt.value

    def comm_barrier(self):
        self._chk(self._lib.icpk_comm_barrier(self._h))

    def set_log_callback(self, fn):
        """fn(key, quantity, usec) -- same shape as logDeltaTime (SLAM.hpp:30)."""
        if fn is None:
            self._log_ref = LOG_FN()
        else:
            self._log_ref = LOG_FN(lambda k, q, us, _u: fn(k, q, us))
        self._chk(self._lib.icpk_set_log_callback(self._h, self._log_ref, None))
