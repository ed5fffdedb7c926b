"""Generate tests/golden/oracle_regression.npz: outputs of THIS repository's oracle
(oracle/icp_oracle.c) on small seeded inputs (SURVEY.md 8c fixtures F1-F3, F5).

These vectors are NOT pinned by the reference (its C++ cannot be built here and it
ships no golden data): they freeze the oracle's behaviour so that a later change to the
checker itself is noticed.  The Kabsch fixture (kabsch_golden.npz) is the one that
comes from running reference code.

usage: python tests/golden/make_oracle_regression.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_slam_prototype_amd import synth  # noqa: E402
from oracle import icp_oracle as o  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.npz")


def main():
    d = {}
    # F1: NN on a tie-heavy lattice and on a rotated frustum cloud
    p = synth.lattice_wall(30, 40)
    d["f1_lattice_idx"], d["f1_lattice_dist"] = o.nn_bruteforce(p["source"], p["target"])
    q = synth.frustum_pair(1200, seed=1)
    src, tgt = q["source"] + np.float32(5), q["target"] + np.float32(5)
    idx, dist = o.nn_bruteforce(src, tgt)
    d["f1_frustum_idx"], d["f1_frustum_dist"] = idx, dist
    # F2: reference-order offset / MSE / moment and the canonical sums
    d["f2_offset"], _ = o.calculate_offset_seq(src, tgt, idx, dist, 0.75)
    d["f2_mse"] = np.array([o.mse_seq(dist, 0.75)], np.float32)
    d["f2_moment"], _ = o.cross_moment_seq(src, tgt, idx, dist, 0.75)
    d["f2_sums"], _ = o.sums_canonical(src, tgt, idx, dist, 0.75)
    # F3: per-iteration trace of the reference-flavour loop, 8 iterations
    r = o.align(src, tgt, max_iterations=8, threshold=0.0, solve=0, sum_order=1)
    d["f3_T"] = r["T"]
    d["f3_R"] = np.stack([t["R"] for t in r["trace"]])
    d["f3_t"] = np.stack([t["t"] for t in r["trace"]])
    d["f3_mse"] = np.array([t["mse"] for t in r["trace"]], np.float32)
    d["f3_pairs"] = np.array([t["n_pairs"] for t in r["trace"]], np.int32)
    r = o.align(src, tgt, max_iterations=8, threshold=0.0, solve=1, sum_order=1)
    d["f3_kabsch_T"] = r["T"]
    # F5: makeRotationMatrix / quaternion / Euler
    d["f5_rot_0_5_0"] = o.make_rotation_matrix(0, 5, 0)
    d["f5_rot_10_20_30"] = o.make_rotation_matrix(10, 20, 30)
    d["f5_quat"] = o.quaternion_from_matrix(d["f5_rot_10_20_30"])
    d["f5_euler"] = o.to_euler(d["f5_quat"])
    np.savez(OUT, **d)
    print("wrote", OUT, len(d), "arrays")


if __name__ == "__main__":
    main()
