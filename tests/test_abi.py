"""CPU-side checks of the C-ABI library: it loads, exports every symbol the
header declares, refuses to run without a GPU (no CPU fallback), and its host
helpers (no device work) agree with the oracle / the pinned Kabsch goldens."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from icp_slam_prototype_amd import binding, build, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def lib():
    build.build()
    return binding.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "icpk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(icpk_[a-z_0-9]+)\s*\(", text)) - {"icpk_log_fn"})


def test_library_exports_every_declared_symbol(lib):
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/icpk.h but not exported"
    assert sorted(binding.SYMBOLS) == syms
    assert lib.icpk_version().decode().startswith("icpk ")


def test_struct_layouts_match_header(lib):
    assert C.sizeof(binding.Params) == 8 * 4 + 12 * 4 + 4 + 4
    assert C.sizeof(binding.Stats) == 6 * 4 + 4 * 4
    p = binding.default_params()
    assert (p.max_iterations, p.min_pairs, p.solve, p.nn_mode) == (16, 3, 0, binding.NN_GRID)
    assert p.threshold == np.float32(1e-4) and p.max_nn_dist == np.float32(0.75)
    assert list(p.last_rotation) == [1, 0, 0, 0, 1, 0, 0, 0, 1]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback_without_gpu(lib):
    with pytest.raises(binding.IcpkError) as e:
        binding.Context(0)
    assert e.value.code == binding.E_NO_DEVICE


def test_null_arguments_are_rejected(lib):
    assert lib.icpk_create(None, 0) == binding.E_ARG
    T = np.zeros(16, np.float32)
    assert lib.icpk_align(None, None, T.ctypes.data_as(C.POINTER(C.c_float)), None) == binding.E_ARG
    assert np.array_equal(T.reshape(4, 4), np.eye(4))  # output initialised even on failure
    assert lib.icpk_reduce(None, 0.75, None, None) == binding.E_ARG
    assert lib.icpk_source_size(None) == 0


def test_host_helpers_match_oracle(lib, oracle):
    for ang in [(0, 0, 0), (0, 5, 0), (10, 20, 30), (-3, 91, 179), (0.5, -0.25, 2)]:
        assert np.array_equal(binding.make_rotation_matrix(*ang), oracle.make_rotation_matrix(*ang))
    rng = np.random.default_rng(0)
    for _ in range(30):
        R = synth.rot_xyz_deg(*rng.uniform(-180, 180, 3)).astype(np.float32)
        q = binding.matrix_to_quaternion(R)
        assert np.array_equal(q, oracle.quaternion_from_matrix(R))
        assert np.array_equal(binding.quaternion_to_euler(q), oracle.to_euler(q))
    for k in range(30):
        M = rng.normal(size=(3, 3)).astype(np.float32)
        if k % 2:
            M = (np.outer([5, 5, 7], [5, 5, 7]) * 100 + rng.normal(size=(3, 3))).astype(np.float32)
        a, b = binding.solve_reference(M), oracle.solve_reference(M)
        assert np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) < 1e-5


def test_keypoint_backprojection_matches_oracle(lib, oracle):
    """pointcloud.cpp:60-98 (host helper, no device): rounding of the key point to a pixel (ties to even), empty pixels
    and positions outside the image skipped, the full-cloud formula -- equal to the oracle's restatement bit for bit and
    to the full back-projection at the same pixels."""
    rng = np.random.default_rng(3)
    rows, cols = 48, 64
    depth = rng.integers(500, 30000, (rows, cols)).astype(np.uint16)
    depth[rng.random(depth.shape) < 0.3] = 0
    kp = np.stack([rng.uniform(-3, cols + 3, 300), rng.uniform(-3, rows + 3, 300)], 1).astype(np.float32)
    kp[:8] = [[0.5, 0.5], [1.5, 2.5], [62.5, 46.5], [63.49, 47.49], [63.5, 47.0], [-0.5, 0.0], [np.nan, 3.0], [5.0, np.inf]]
    got, kept = binding.backproject_keypoints(depth, kp)
    want, wkept = oracle.backproject_keypoints(depth, kp)
    assert np.array_equal(kept, wkept) and got.tobytes() == want.tobytes() and 100 < len(kept) < 300
    # the same numbers as the cloud's own points at those pixels (pointcloud.cpp:37-39 == :86-88)
    full = oracle.backproject(depth)
    rank = np.cumsum(depth.reshape(-1) != 0) - 1
    for k, i in enumerate(kept[:50]):
        x, y = int(np.rint(kp[i, 0])), int(np.rint(kp[i, 1]))
        assert np.array_equal(got[:, k], full[:, rank[y * cols + x]])
    # ties to even: 0.5 -> 0, 1.5 -> 2, 2.5 -> 2
    d = np.arange(1, 13, dtype=np.uint16).reshape(3, 4) * 1000
    p, k = binding.backproject_keypoints(d, np.float32([[0.5, 0.5], [1.5, 2.5], [2.5, 1.5]]))
    assert [float(v) for v in p[2]] == [float(np.float32(d[0, 0]) / np.float32(5000)), float(np.float32(d[2, 2]) / np.float32(5000)),
                                        float(np.float32(d[2, 2]) / np.float32(5000))]
    assert binding.backproject_keypoints(d, np.zeros((0, 2), np.float32))[0].shape == (3, 0)


def test_host_kabsch_matches_reference_python_golden(lib):
    """PINNED by the outputs of the reference's rigid_transform_3D.py."""
    g = np.load(os.path.join(GOLD, "kabsch_golden.npz"))
    for nme in sorted({k[:-2] for k in g.files}):
        A, B, R, t = g[nme + "_A"], g[nme + "_B"], g[nme + "_R"], g[nme + "_t"]
        R2, t2 = binding.solve_kabsch(A.shape[0], A.sum(0), B.sum(0), A.T @ B)
        assert np.linalg.norm(R2 - R) < 1e-8, nme
        assert np.linalg.norm(t2 - t) < 1e-8, nme


def test_polar_solve_matches_jacobi_oracle_on_random_moments(oracle):
    """Round 2 replaced the Jacobi SVD of the loop's solve by Newton's iteration for the orthogonal
    polar factor (csrc/solve_impl.h, polar3), with the SVD as the fall-back for near-singular
    moments.  Host entry points only (no device work): reference flavour from random un-centred
    moments of every conditioning, reflections included, against the oracle's float64 Jacobi SVD;
    Kabsch flavour from random sums.  R must be orthogonal and agree to float accuracy."""
    from icp_slam_prototype_amd import binding

    rng = np.random.default_rng(123)
    worst = 0.0
    for t in range(3000):
        c = rng.normal(5, 1, 3)
        kind = t % 6
        noise = [0.3, 1e-2, 1e-4, 3.0, 1e-7, 0.0][kind]            # 4: almost rank 1, 5: exactly rank 1
        M = (np.outer(c, c) * (0 if kind == 3 else 1) + noise * rng.normal(0, 1, (3, 3))) * 10.0 ** rng.uniform(-3, 6)
        if t % 7 == 0:
            M = -M                                                 # det < 0: the column-2 flip of icp.cpp:220-223
        M = M.astype(np.float32)
        R = binding.solve_reference(M)
        Ro = oracle.solve_reference(M)
        assert np.isfinite(R).all()
        assert np.abs(R.astype(np.float64) @ R.astype(np.float64).T - np.eye(3)).max() < 1e-5
        if kind not in (4, 5):  # a (numerically) rank-deficient moment has no unique polar factor
            worst = max(worst, float(np.abs(R - Ro).max()))
    assert worst < 2e-5, worst
    for t in range(500):
        n = int(rng.integers(3, 5000))
        A = rng.normal(0, 1, (3, n)) * rng.uniform(0.01, 3, (3, 1))
        if t % 5 == 0:
            A[2] = 0.0  # exactly planar: rank-2 covariance, the Newton path must hand over to the SVD
        ang = rng.uniform(-0.5, 0.5, 3)
        Rt = oracle.make_rotation_matrix(*np.degrees(ang)).astype(np.float64)
        B = Rt @ A + rng.normal(0, 1, (3, 1))
        Rd, td = binding.solve_kabsch(n, A.sum(1), B.sum(1), A @ B.T)
        assert np.abs(Rd @ Rd.T - np.eye(3)).max() < 1e-9 and np.linalg.det(Rd) > 0.999
        if t % 5 != 0 and n > 10:
            assert np.abs(Rd - Rt).max() < 1e-6 and np.abs(Rd @ A.mean(1) + td - B.mean(1)).max() < 1e-9
