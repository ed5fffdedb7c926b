"""Soak test (tools only): many random clouds, grid scan vs exact kernel, bit for bit, over first and
seeded sweeps with the source moving in between.  usage: python tools/soak_grid.py [n_cases] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import _fuzz_cloud

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
use_oracle = len(sys.argv) > 3 and sys.argv[3] == "oracle"  # compare with the CPU oracle instead of the exact kernel
if use_oracle:
    from oracle import icp_oracle as orc
kinds = ["uniform", "clusters", "line", "plane_lattice", "duplicates", "tiny", "huge"]
ctx = binding.Context(0)
t0 = time.time()
bad = 0
for c in range(n_cases):
    rng = np.random.default_rng(seed0 + c)
    kt, ks = kinds[rng.integers(0, 7)], kinds[rng.integers(0, 7)]
    nt, nq = int(rng.integers(1, 60000)), int(rng.integers(1, 40000))
    if use_oracle:
        nt, nq = 1 + nt // 3, 1 + nq // 3
    off = rng.uniform(-10, 10, (3, 1))
    scale = float(10.0 ** rng.uniform(-2, 1))
    tgt = (_fuzz_cloud(rng, nt, kt) * scale + off).astype(np.float32)
    src = (_fuzz_cloud(rng, nq, ks) * scale + off + rng.normal(0, 0.01 * scale, (3, 1))).astype(np.float32)
    if rng.random() < 0.3:
        tgt = tgt[:, rng.permutation(nt)]
    ctx.set_target(tgt)
    ctx.set_source(src)
    cur = src
    for sweep in range(3):
        if use_oracle:
            ie, de = orc.nn_bruteforce(cur, tgt, threads=orc.max_threads())
        else:
            ie, de = ctx.nn(binding.NN_EXACT)
        if sweep == 0:
            ctx.reset_source()  # first grid sweep unseeded (expanding search)
        ig, dg = ctx.nn(binding.NN_GRID)
        ok = np.array_equal(ie, ig) and np.array_equal(de.view(np.uint32), dg.view(np.uint32))
        if not ok:
            bad += 1
            print("MISMATCH case", c, kt, ks, nt, nq, "sweep", sweep, int((ie != ig).sum()), flush=True)
        R = synth.rot_xyz_deg(*rng.uniform(-2, 2, 3)).astype(np.float32)
        tr = (rng.normal(0, 0.02, 3) * scale).astype(np.float32)
        ctx.transform_source(R, tr)
        if use_oracle:
            cur = orc.transform_points(cur, R, tr)
    if c % 20 == 19:
        print(f"{c + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
