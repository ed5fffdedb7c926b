"""Generate tests/golden/kabsch_golden.npz by EXECUTING the reference's own
rigid_transform_3D.py (read from /root/reference at generation time; the
reference text is never copied into this repository).

Only runs in the build container (the GPU box has no /root/reference); the
resulting .npz (plain float64 arrays, no pickles) is the committed fixture.

The script targets NumPy < 2 (`mat` was removed in NumPy 2.0), so the name
`mat` is bound to numpy.asmatrix in the namespace it is executed in -- the same
object `numpy.mat` used to be.  Its module-level self-test (lines 42-97) runs as
a side effect and is ignored; its prints are swallowed.

usage: python tests/golden/make_kabsch_golden.py
"""
import contextlib
import io
import os

import numpy as np

REF = "/root/reference/rigid_transform_3D.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kabsch_golden.npz")


def load_reference_function():
    ns = {"mat": np.asmatrix, "__name__": "rigid_transform_3D_ref"}
    with open(REF) as f:
        code = compile(f.read(), REF, "exec")
    np.random.seed(12345)  # the script's self-test draws from the global RNG
    with contextlib.redirect_stdout(io.StringIO()):
        exec(code, ns)
    return ns["rigid_transform_3D"]


def rot(axis, deg):
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def main():
    fn = load_reference_function()
    rng = np.random.default_rng(20261004)
    cases = {}

    def run(name, A, B):
        with contextlib.redirect_stdout(io.StringIO()):
            R, t = fn(np.asmatrix(A), np.asmatrix(B))
        cases[name + "_A"] = np.asarray(A, np.float64)
        cases[name + "_B"] = np.asarray(B, np.float64)
        cases[name + "_R"] = np.asarray(R, np.float64)
        cases[name + "_t"] = np.asarray(t, np.float64).reshape(3)

    # (a) the self-test's shape: n = 10 uniform points, random rigid motion
    A = rng.random((10, 3))
    Rt = rot("z", 33.0) @ rot("x", -12.0) @ rot("y", 71.0)
    run("selftest10", A, A @ Rt.T + np.array([0.3, 0.9, 0.1]))
    # (b) 1000 points in a Kinect-like frustum, 5 degree rotation, small shift
    A = np.column_stack([rng.uniform(-2, 2, 1000), rng.uniform(-1.5, 1.5, 1000), rng.uniform(0.5, 4, 1000)])
    run("rot5deg1000", A, A @ rot("y", 5.0).T + np.array([0.02, -0.01, 0.03]))
    # (c) noisy correspondences (no exact solution)
    B = A @ rot("x", 2.0).T + np.array([0.0, 0.03, 0.0]) + rng.normal(0, 0.01, A.shape)
    run("noisy1000", A, B)
    # (d) minimal n = 3
    A = rng.random((3, 3))
    run("min3", A, A @ rot("z", -20.0).T + np.array([1.0, 2.0, 3.0]))
    # (e) reflection branch: B is a mirrored copy of a nearly planar A, so that
    #     det(Vt^T U^T) < 0 and the script negates Vt[2,:]
    A = np.column_stack([rng.uniform(-1, 1, 200), rng.uniform(-1, 1, 200), rng.normal(0, 1e-3, 200)])
    B = A.copy()
    B[:, 2] *= -1.0
    B = B @ rot("z", 10.0).T
    run("reflect200", A, B)
    # (f) world-offset coordinates as in the reference (camera at (5,5,5), icp.cpp:53)
    A = np.column_stack([rng.uniform(-2, 2, 500), rng.uniform(-1.5, 1.5, 500), rng.uniform(0.5, 4, 500)]) + 5.0
    run("offset555", A, (A - 5.0) @ rot("y", 2.0).T + 5.0 + np.array([0.03, 0.0, 0.0]))

    np.savez(OUT, **cases)
    print("wrote", OUT, "with", len(cases) // 4, "cases")


if __name__ == "__main__":
    main()
