"""ctypes wrapper around oracle/_build/libicp_oracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
See oracle/icp_oracle.c for the parity status (NN / reference solve: parity
unpinned; Kabsch: pinned by tests/golden/kabsch_*.npz).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libicp_oracle.so")

NSUM = 19
NP2L = 28
RED_THREADS = 256
RED_MAX_BLOCKS = 256


def build(force=False):
    src = os.path.join(_HERE, "icp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class Params(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32),
        ("threshold", C.c_float),
        ("max_nn_dist", C.c_float),
        ("min_pairs", C.c_int32),
        ("solve", C.c_int32),
        ("sum_order", C.c_int32),
        ("fixed_iterations", C.c_int32),
        ("threads", C.c_int32),
        ("last_rotation", C.c_float * 9),
        ("last_translation", C.c_float * 3),
    ]


class IterTrace(C.Structure):
    _fields_ = [
        ("n_pairs", C.c_int32),
        ("mse", C.c_float),
        ("M", C.c_float * 9),
        ("R", C.c_float * 9),
        ("t", C.c_float * 3),
    ]


class Result(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("status", C.c_int32),
        ("final_pairs", C.c_int32),
        ("final_mse", C.c_float),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        assert _lib.orc_sizeof_params() == C.sizeof(Params)
        assert _lib.orc_sizeof_trace() == C.sizeof(IterTrace)
        assert _lib.orc_sizeof_result() == C.sizeof(Result)
        _lib.orc_distance.restype = C.c_float
        _lib.orc_distance.argtypes = [C.c_float] * 6
        _lib.orc_distance3.restype = C.c_float
        _lib.orc_distance3.argtypes = [C.c_float] * 6
        _lib.orc_mse_seq.restype = C.c_float
        _lib.orc_sums_canonical.restype = C.c_int64
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def max_threads():
    return int(lib().orc_max_threads())


def distance(a, b):
    a = np.float32(a)
    b = np.float32(b)
    return np.float32(lib().orc_distance(*[C.c_float(float(v)) for v in (*a, *b)]))


def distance3(a, b):
    """icp.cpp:595-602 distance(cv::Point3f, cv::Point3f)."""
    a = np.float32(a)
    b = np.float32(b)
    return np.float32(lib().orc_distance3(*[C.c_float(float(v)) for v in (*a, *b)]))


def keypoint_associations(src, tgt, max_dist=0.1, rejected=None):
    """icp.cpp:488-539.  Returns None when the target (map key points) is empty (the
    reference returns before touching its outputs), else (assoc_q, assoc_t, assoc_d,
    rejected) with `rejected` = the caller's list (or []) with this sweep's rejected query
    indices appended."""
    sx, sy, sz = (_f(src[k]) for k in range(3))
    tx, ty, tz = (_f(tgt[k]) for k in range(3))
    nq, nt = sx.size, tx.size
    prev = np.asarray([] if rejected is None else rejected, np.int32)
    aq = np.empty(max(nq, 1), np.int32)
    at = np.empty(max(nq, 1), np.int32)
    ad = np.empty(max(nq, 1), np.float32)
    rj = np.empty(prev.size + max(nq, 1), np.int32)
    rj[:prev.size] = prev
    na = C.c_int32(0)
    nr = C.c_int32(prev.size)
    rc = lib().orc_keypoint_associations(_p(sx), _p(sy), _p(sz), C.c_int(nq), _p(tx), _p(ty), _p(tz), C.c_int(nt),
                                         C.c_float(max_dist), _p(aq, C.c_int32), _p(at, C.c_int32), _p(ad),
                                         C.byref(na), _p(rj, C.c_int32), C.byref(nr))
    if rc == 1:
        return None
    return aq[:na.value].copy(), at[:na.value].copy(), ad[:na.value].copy(), rj[:nr.value].copy()


def nn_bruteforce(src, tgt, threads=1):
    """src, tgt: (3, N) float32 SoA.  Returns idx int32[Nq], dist float32[Nq]."""
    sx, sy, sz = (_f(src[k]) for k in range(3))
    tx, ty, tz = (_f(tgt[k]) for k in range(3))
    nq, nt = sx.size, tx.size
    idx = np.empty(nq, np.int32)
    dist = np.empty(nq, np.float32)
    rc = lib().orc_nn_bruteforce(_p(sx), _p(sy), _p(sz), C.c_int(nq), _p(tx), _p(ty), _p(tz),
                                 C.c_int(nt), _p(idx, C.c_int32), _p(dist), C.c_int(threads))
    if rc != 0:
        raise ValueError("oracle: empty target")
    return idx, dist


def _assoc_args(src, tgt, idx, dist):
    sx, sy, sz = (_f(src[k]) for k in range(3))
    tx, ty, tz = (_f(tgt[k]) for k in range(3))
    idx = np.ascontiguousarray(idx, np.int32)
    dist = _f(dist)
    keep = (sx, sy, sz, tx, ty, tz, idx, dist)
    return keep, [_p(sx), _p(sy), _p(sz), C.c_int(sx.size), _p(tx), _p(ty), _p(tz),
                  _p(idx, C.c_int32), _p(dist)]


def calculate_offset_seq(src, tgt, idx, dist, max_dist):
    keep, a = _assoc_args(src, tgt, idx, dist)
    out = np.zeros(3, np.float32)
    n = lib().orc_calculate_offset_seq(*a, C.c_float(max_dist), _p(out))
    return out, int(n)


def mse_seq(dist, max_dist):
    d = _f(dist)
    return np.float32(lib().orc_mse_seq(_p(d), C.c_int(d.size), C.c_float(max_dist)))


def cross_moment_seq(src, tgt, idx, dist, max_dist):
    keep, a = _assoc_args(src, tgt, idx, dist)
    out = np.zeros(9, np.float32)
    n = lib().orc_cross_moment_seq(*a, C.c_float(max_dist), _p(out))
    return out.reshape(3, 3), int(n)


def sums_canonical(src, tgt, idx, dist, max_dist):
    keep, a = _assoc_args(src, tgt, idx, dist)
    out = np.zeros(NSUM, np.float64)
    n = lib().orc_sums_canonical(*a, C.c_float(max_dist), _p(out, C.c_double))
    return out, int(n)


def svd3(A):
    A = np.ascontiguousarray(A, np.float64)
    U = np.zeros((3, 3))
    S = np.zeros(3)
    V = np.zeros((3, 3))
    lib().orc_svd3(_p(A, C.c_double), _p(U, C.c_double), _p(S, C.c_double), _p(V, C.c_double))
    return U, S, V


def solve_reference(M):
    M = _f(M).reshape(9)
    R = np.zeros(9, np.float32)
    lib().orc_solve_reference(_p(M), _p(R))
    return R.reshape(3, 3)


def inv3(R):
    R = _f(R).reshape(9)
    out = np.zeros(9, np.float32)
    lib().orc_inv3_f(_p(R), _p(out))
    return out.reshape(3, 3)


def rigid_transform_3D(A, B):
    """A, B: (n, 3) float64 row-major, same call shape as the reference script."""
    A = np.ascontiguousarray(A, np.float64)
    B = np.ascontiguousarray(B, np.float64)
    R = np.zeros((3, 3))
    t = np.zeros(3)
    lib().orc_rigid_transform_3D(_p(A, C.c_double), _p(B, C.c_double), C.c_int(A.shape[0]),
                                 _p(R, C.c_double), _p(t, C.c_double))
    return R, t


def solve_kabsch_from_sums(n, sa, sb, sab):
    sa = np.ascontiguousarray(sa, np.float64)
    sb = np.ascontiguousarray(sb, np.float64)
    sab = np.ascontiguousarray(sab, np.float64).reshape(9)
    R = np.zeros((3, 3))
    t = np.zeros(3)
    lib().orc_solve_kabsch_from_sums(C.c_int64(n), _p(sa, C.c_double), _p(sb, C.c_double),
                                     _p(sab, C.c_double), _p(R, C.c_double), _p(t, C.c_double))
    return R, t


def transform_points(pts, R, t):
    """pts (3, N) float32 -> new (3, N) float32:  fl32(fl32(R p) + t)."""
    x, y, z = (_f(pts[k]).copy() for k in range(3))
    R = _f(R).reshape(9)
    t = _f(t).reshape(3)
    lib().orc_transform_points(_p(x), _p(y), _p(z), C.c_int(x.size), _p(R), _p(t))
    return np.stack([x, y, z])


def make_rotation_matrix(x, y, z):
    out = np.zeros(9, np.float32)
    lib().orc_make_rotation_matrix(C.c_float(x), C.c_float(y), C.c_float(z), _p(out))
    return out.reshape(3, 3)


def quaternion_from_matrix(m):
    m = _f(m).reshape(9)
    q = np.zeros(4, np.float32)
    lib().orc_quaternion_from_matrix(_p(m), _p(q))
    return q


def to_euler(q):
    q = _f(q).reshape(4)
    e = np.zeros(3, np.float32)
    lib().orc_to_euler(_p(q), _p(e))
    return e


def backproject_keypoints(depth, kp_xy, fx=468.60, cx=318.27):
    """pointcloud.cpp:60-98: key points (x, y pixel positions as cv::KeyPoint::pt holds them) -> Point2i by cvRound
    (to nearest, ties to even), depth 0 skipped (:67-70), back-projected by :86-88 in float32, in list order.  Pixels
    outside the image (undefined in the reference) are skipped.  Returns (points (3, m), kept indices)."""
    depth = np.ascontiguousarray(depth, np.uint16)
    rows, cols = depth.shape
    kp = np.asarray(kp_xy, np.float32).reshape(-1, 2)
    pts, kept = [], []
    f32 = np.float32
    for i, (xf, yf) in enumerate(kp):
        if not (np.isfinite(xf) and np.isfinite(yf)):
            continue
        x, y = int(np.rint(xf)), int(np.rint(yf))
        if not (0 <= x < cols and 0 <= y < rows) or depth[y, x] == 0:
            continue
        pz = f32(depth[y, x]) / f32(5000.0)
        px = (f32(x) - f32(cx)) * pz / f32(fx)
        py = (f32(y) - f32(cx)) * pz / f32(fx)
        pts.append((px, py, pz))
        kept.append(i)
    out = np.array(pts, np.float32).reshape(-1, 3).T
    return np.ascontiguousarray(out), np.array(kept, np.int32)


def subsample_keep(rows, cols, factor, seed, stream):
    """The seeded stand-in for pointcloud.cpp:27-30 (`rand() % SUBSAMPLE_FACTOR`) as include/icpk.h defines it for
    icpk_set_subsample: uint8 mask over the rows x cols pixels of image number `stream` (0-based count of the images
    back-projected since the subsample was set).  Pass it as `keep` to backproject."""
    if factor <= 1:
        return np.ones((rows, cols), np.uint8)
    with np.errstate(over="ignore"):
        p = np.arange(rows * cols, dtype=np.uint64)
        z = np.uint64(seed) + np.uint64(stream + 1) * np.uint64(0x9E3779B97F4A7C15) + p * np.uint64(0xD1B54A32D192ED03)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (((z >> np.uint64(32)).astype(np.uint32) % np.uint32(factor)) == 0).astype(np.uint8).reshape(rows, cols)


def backproject(depth, keep=None, fx=468.60, cx=318.27):
    depth = np.ascontiguousarray(depth, np.uint16)
    rows, cols = depth.shape
    n = rows * cols
    x = np.empty(n, np.float32)
    y = np.empty(n, np.float32)
    z = np.empty(n, np.float32)
    kp = None
    if keep is not None:
        keep = np.ascontiguousarray(keep, np.uint8)
        kp = _p(keep, C.c_uint8)
    m = lib().orc_backproject(_p(depth, C.c_uint16), C.c_int(rows), C.c_int(cols), kp,
                              C.c_float(fx), C.c_float(cx), _p(x), _p(y), _p(z))
    return np.stack([x[:m], y[:m], z[:m]])


def backproject_normals(depth, mode=0, fx=468.60, cx=318.27):
    """Returns (points (3,N), normals (3,N)) in the row-major order of valid pixels."""
    depth = np.ascontiguousarray(depth, np.uint16)
    rows, cols = depth.shape
    n = rows * cols
    arrs = [np.empty(n, np.float32) for _ in range(6)]
    m = lib().orc_backproject_normals(_p(depth, C.c_uint16), C.c_int(rows), C.c_int(cols), C.c_float(fx),
                                      C.c_float(cx), C.c_int(mode), *[_p(a) for a in arrs])
    return np.stack([a[:m] for a in arrs[:3]]), np.stack([a[:m] for a in arrs[3:]])


def rotate_normals(nrm, R):
    x, y, z = (_f(nrm[k]).copy() for k in range(3))
    R = _f(R).reshape(9)
    lib().orc_rotate_normals(_p(x), _p(y), _p(z), C.c_int(x.size), _p(R))
    return np.stack([x, y, z])


def sums_p2l_canonical(src, tgt, nrm, idx, dist, max_dist):
    keep, a = _assoc_args(src, tgt, idx, dist)
    nx, ny, nz = (_f(nrm[k]) for k in range(3))
    out = np.zeros(NP2L, np.float64)
    args = a[:7] + [_p(nx), _p(ny), _p(nz)] + a[7:]
    n = lib().orc_sums_p2l_canonical(*args, C.c_float(max_dist), _p(out, C.c_double))
    return out, int(n)


def solve_p2l(sums):
    sums = np.ascontiguousarray(sums, np.float64)
    R = np.zeros((3, 3))
    t = np.zeros(3)
    rc = lib().orc_solve_p2l(_p(sums, C.c_double), _p(R, C.c_double), _p(t, C.c_double))
    return R, t, rc


def depth_range_filter(depth, max_d=25000, min_d=1000):
    d = np.ascontiguousarray(depth, np.uint16).copy()
    lib().orc_depth_range_filter(_p(d, C.c_uint16), C.c_int(d.size), C.c_int(max_d), C.c_int(min_d))
    return d


def filter_depth_image(depth, max_d=25000, min_d=1000, anchor=(2, 2)):
    """SLAM.cpp:553-574 as a whole: range clamp, 5x5 dilate, 5x5 erode (anchor = (x, y))."""
    d = np.ascontiguousarray(depth, np.uint16)
    out = np.empty_like(d)
    lib().orc_filter_depth_image(_p(d, C.c_uint16), _p(out, C.c_uint16), C.c_int(d.shape[0]), C.c_int(d.shape[1]),
                                 C.c_int(max_d), C.c_int(min_d), C.c_int(anchor[0]), C.c_int(anchor[1]))
    return out


def align(src, tgt, max_iterations=16, threshold=1e-4, max_nn_dist=0.75, min_pairs=3,
          solve=0, sum_order=0, fixed_iterations=False, threads=1,
          last_rotation=None, last_translation=None, normals=None):
    """Runs the restated loop.  Returns dict(T, src_out, idx, dist, trace, result)."""
    sx, sy, sz = (_f(src[k]).copy() for k in range(3))
    tx, ty, tz = (_f(tgt[k]) for k in range(3))
    p = Params()
    p.max_iterations = max_iterations
    p.threshold = threshold
    p.max_nn_dist = max_nn_dist
    p.min_pairs = min_pairs
    p.solve = solve
    p.sum_order = sum_order
    p.fixed_iterations = int(bool(fixed_iterations))
    p.threads = threads
    lr = np.eye(3, dtype=np.float32) if last_rotation is None else _f(last_rotation)
    lt = np.zeros(3, np.float32) if last_translation is None else _f(last_translation)
    p.last_rotation[:] = [float(v) for v in lr.reshape(9)]
    p.last_translation[:] = [float(v) for v in lt.reshape(3)]
    T = np.zeros(16, np.float32)
    idx = np.zeros(sx.size, np.int32)
    dist = np.zeros(sx.size, np.float32)
    trace = (IterTrace * max(1, max_iterations))()
    res = Result()
    if normals is not None:
        nx, ny, nz = (_f(normals[k]) for k in range(3))
        npt = [_p(nx), _p(ny), _p(nz)]
    else:
        npt = [None, None, None]
    lib().orc_align2(_p(sx), _p(sy), _p(sz), C.c_int(sx.size), _p(tx), _p(ty), _p(tz),
                     C.c_int(tx.size), *npt, C.byref(p), _p(T), _p(idx, C.c_int32), _p(dist),
                     trace, C.byref(res))
    tr = []
    for i in range(res.iterations):
        tr.append(dict(n_pairs=trace[i].n_pairs, mse=np.float32(trace[i].mse),
                       M=np.array(trace[i].M, np.float32).reshape(3, 3),
                       R=np.array(trace[i].R, np.float32).reshape(3, 3),
                       t=np.array(trace[i].t, np.float32)))
    return dict(T=T.reshape(4, 4), src_out=np.stack([sx, sy, sz]), idx=idx, dist=dist, trace=tr,
                iterations=res.iterations, status=res.status, final_pairs=res.final_pairs,
                final_mse=np.float32(res.final_mse))
