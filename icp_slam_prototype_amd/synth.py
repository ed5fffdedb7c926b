"""Synthetic workloads for the BASELINE.json configs.

Pure numpy; used by bench.py and tests/.  No dataset ships with the reference
(CoRBS/TUM sequences are not fetchable), so every workload is generated:
depth images are ray-cast from a small analytic room and back-projected with
the reference's own formula (pointcloud.cpp:37-39, including its use of CX/FX
for the y axis) WITHOUT the rand()%40 subsample (pointcloud.cpp:28; the library's seeded stand-in is icpk_set_subsample).
"""
import numpy as np

# pointcloud.hpp:7-10
FX = np.float32(468.60)
FY = np.float32(468.61)
CX = np.float32(318.27)
CY = np.float32(243.99)
# SLAM.cpp:25-35,135 (Kinect v2 depth intrinsics)
K2_FX = np.float32(363.58)
K2_CX = np.float32(250.32)

CAMERA_START = np.float32(5.0)  # icp.cpp:53 cameraPosition = (5,5,5)


def rot_xyz_deg(x, y, z):
    """float64 restatement of the sign convention of icp.cpp:640-653 (Rx*Ry*Rz)."""
    ax, ay, az = np.deg2rad([x, y, z])
    rx = np.array([[1, 0, 0], [0, np.cos(ax), np.sin(ax)], [0, -np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, -np.sin(ay)], [0, 1, 0], [np.sin(ay), 0, np.cos(ay)]])
    rz = np.array([[np.cos(az), np.sin(az), 0], [-np.sin(az), np.cos(az), 0], [0, 0, 1]])
    return rx @ ry @ rz


def backproject(depth, keep=None, fx=FX, cx=CX):
    """numpy restatement of pointcloud.cpp:19-58 (float32 arithmetic, row-major
    order of the non-zero pixels).  Returns (3, N) float32 SoA."""
    depth = np.asarray(depth, np.uint16)
    m = depth != 0
    if keep is not None:
        m &= np.asarray(keep, bool)
    r, c = np.nonzero(m)  # row-major order
    d = depth[r, c].astype(np.float32)
    pz = d / np.float32(5000.0)
    px = (c.astype(np.float32) - np.float32(cx)) * pz / np.float32(fx)
    py = (r.astype(np.float32) - np.float32(cx)) * pz / np.float32(fx)
    return np.stack([px, py, pz]).astype(np.float32)


def render_room_depth(rows, cols, R_wc, c_w, fx=FX, cx=CX, noise_sigma=0.0, rng=None):
    """Ray-cast a room (floor, back wall, left wall, one sphere) from a camera
    with world-from-camera rotation R_wc and centre c_w.  Ray directions use the
    same pinhole the reference back-projects with, so back-projection of the
    result reproduces the hit points.  Returns uint16 depth = metres * 5000."""
    v, u = np.mgrid[0:rows, 0:cols].astype(np.float64)
    d_cam = np.stack([(u - float(cx)) / float(fx), (v - float(cx)) / float(fx), np.ones_like(u)], -1)
    w = d_cam @ np.asarray(R_wc, np.float64).T
    o = np.asarray(c_w, np.float64)
    t_best = np.full((rows, cols), np.inf)

    def plane(n, h):
        n = np.asarray(n, np.float64)
        den = w @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (h - o @ n) / den
        t[~(t > 1e-6)] = np.inf
        return t

    t_best = np.minimum(t_best, plane([0, 1, 0], 1.3))    # floor (y down)
    t_best = np.minimum(t_best, plane([0, 0, 1], 3.6))    # back wall
    t_best = np.minimum(t_best, plane([-1, 0, 0], 2.1))   # left wall x = -2.1
    # sphere
    sc = np.array([0.35, 0.45, 2.3])
    sr = 0.55
    oc = o - sc
    a = np.einsum("ijk,ijk->ij", w, w)
    b = 2.0 * (w @ oc)
    cc = oc @ oc - sr * sr
    disc = b * b - 4 * a * cc
    with np.errstate(invalid="ignore"):
        ts = (-b - np.sqrt(disc)) / (2 * a)
    ts[~(disc > 0) | ~(ts > 1e-6)] = np.inf
    t_best = np.minimum(t_best, ts)
    if noise_sigma > 0:
        t_best = t_best + rng.normal(0.0, noise_sigma, t_best.shape)
    d = np.rint(t_best * 5000.0)
    d[~np.isfinite(d)] = 0
    d[(d < 1000) | (d > 25000)] = 0  # SLAM.cpp:229 filterDepthImage range (SLAM.hpp:15-16)
    return d.astype(np.uint16)


def kinect_pair(rows=480, cols=640, valid=0.30, seed=2, rot_deg=(0.0, 2.0, 0.0),
                shift=(0.03, 0.0, 0.0), noise_sigma=0.002, fx=FX, cx=CX, world_offset=True):
    """Config 2 / 3 / 4 workload: two depth frames of the same room, the second
    after a small camera motion, independent Bernoulli validity masks.
    Returns dict(source (3,Ns), target (3,Nt), depth_src, depth_tgt, R_true, t_true)
    where source = current frame (`data`), target = previous frame."""
    rng_t = np.random.default_rng(seed)
    rng_s = np.random.default_rng(seed + 1)
    depth_t = render_room_depth(rows, cols, np.eye(3), np.zeros(3), fx, cx, noise_sigma, rng_t)
    Rm = rot_xyz_deg(*rot_deg)
    depth_s = render_room_depth(rows, cols, Rm, np.asarray(shift, np.float64), fx, cx, noise_sigma, rng_s)
    keep_t = rng_t.random((rows, cols)) < valid
    keep_s = rng_s.random((rows, cols)) < valid
    depth_t = np.where(keep_t, depth_t, 0).astype(np.uint16)
    depth_s = np.where(keep_s, depth_s, 0).astype(np.uint16)
    tgt = backproject(depth_t, None, fx, cx)
    src = backproject(depth_s, None, fx, cx)
    if world_offset:
        tgt = tgt + CAMERA_START
        src = src + CAMERA_START
    return dict(source=src.astype(np.float32), target=tgt.astype(np.float32),
                depth_src=depth_s, depth_tgt=depth_t, R_true=Rm,
                t_true=np.asarray(shift, np.float64), fx=float(fx), cx=float(cx))


def frustum_pair(n=10000, seed=1, rot_deg=(0.0, 5.0, 0.0), shift=(0.02, -0.01, 0.03)):
    """Config 1: n points uniform in the Kinect frustum (z in [0.5, 4] m, pixel
    coordinates uniform), source = target rotated by the reference's
    makeRotationMatrix convention about the cloud centroid plus a shift."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(0.5, 4.0, n)
    u = rng.uniform(0, 640, n)
    v = rng.uniform(0, 480, n)
    tgt = np.stack([(u - float(CX)) * z / float(FX), (v - float(CX)) * z / float(FX), z])
    R = rot_xyz_deg(*rot_deg)
    c = tgt.mean(axis=1, keepdims=True)
    src = R @ (tgt - c) + c + np.asarray(shift, np.float64)[:, None]
    return dict(source=src.astype(np.float32), target=tgt.astype(np.float32), R_true=R,
                t_true=np.asarray(shift, np.float64))


def dense_pair(n=1_000_000, seed=5, rot_deg=(0.0, 1.0, 0.0), shift=(0.01, 0.0, 0.0)):
    """Config 5: n points on a noisy wavy surface inside a 4 x 3 x 3 m box."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-2.0, 2.0, n)
    y = rng.uniform(-1.5, 1.5, n)
    z = 2.5 + 0.4 * np.sin(1.7 * x) * np.cos(2.3 * y) + rng.normal(0, 0.003, n)
    tgt = np.stack([x, y, z]) + float(CAMERA_START)
    R = rot_xyz_deg(*rot_deg)
    c = tgt.mean(axis=1, keepdims=True)
    src = R @ (tgt - c) + c + np.asarray(shift, np.float64)[:, None]
    perm = rng.permutation(n)
    return dict(source=src[:, perm].astype(np.float32), target=tgt.astype(np.float32), R_true=R,
                t_true=np.asarray(shift, np.float64))


def lattice_wall(rows=60, cols=80, z=2.0, shift_px=0.5):
    """Tie-heavy case: a flat wall sampled on the
    pixel lattice, source shifted by half a pixel so that many queries have two
    or more targets at exactly the same float distance."""
    v, u = np.mgrid[0:rows, 0:cols].astype(np.float32)
    step = np.float32(0.01)
    tgt = np.stack([(u * step).ravel(), (v * step).ravel(), np.full(rows * cols, z, np.float32)])
    src = tgt.copy()
    src[0] += np.float32(shift_px) * step
    return dict(source=src.astype(np.float32), target=tgt.astype(np.float32))
