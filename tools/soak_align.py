"""Soak test (tools only): whole alignments, grid scan + device loop vs exact kernel + host loop,
bit for bit (transform, pair count, associations, moved source).  usage: soak_align.py [n] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import _fuzz_cloud

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
ctx = binding.Context(0)
t0 = time.time()
bad = 0
for c in range(n_cases):
    rng = np.random.default_rng(seed0 + c)
    if rng.random() < 0.5:
        p = synth.kinect_pair(rows=int(rng.integers(40, 200)), cols=int(rng.integers(60, 260)), valid=float(rng.uniform(0.2, 1.0)),
                              seed=int(rng.integers(0, 1 << 30)), rot_deg=tuple(rng.uniform(-3, 3, 3)),
                              shift=tuple(rng.uniform(-0.05, 0.05, 3)))
        src, tgt = p["source"], p["target"]
    else:
        nt, nq = int(rng.integers(50, 30000)), int(rng.integers(50, 20000))
        tgt = (_fuzz_cloud(rng, nt, "clusters" if rng.random() < 0.5 else "uniform") + 5).astype(np.float32)
        src = (tgt[:, rng.integers(0, nt, nq)] + rng.normal(0, 0.02, (3, nq))).astype(np.float32)
    if min(src.shape[1], tgt.shape[1]) < 10:
        continue
    solve = int(rng.integers(0, 2))
    kw = dict(solve=solve, max_iterations=int(rng.integers(1, 12)), max_nn_dist=float(rng.choice([0.75, 0.1, 0.03])))
    if rng.random() < 0.5:
        kw["fixed_iterations"] = 1
    else:
        kw["threshold"] = float(10 ** rng.uniform(-6, -3))
    res = []
    for mode, host_loop in ((binding.NN_GRID, 0), (binding.NN_EXACT, 1)):
        ctx.set_target(tgt)
        ctx.set_source(src)
        T, st, rc = ctx.align(nn_mode=mode, host_loop=host_loop, **kw)
        idx, dist = ctx.get_associations()
        res.append((T.copy(), rc, st.iterations, st.final_pairs, st.final_mse, idx, dist, ctx.get_source()))
    a, b = res
    ok = (np.array_equal(a[0], b[0]) and a[1:5] == b[1:5] and np.array_equal(a[5], b[5]) and
          np.array_equal(a[6].view(np.uint32), b[6].view(np.uint32)) and np.array_equal(a[7].view(np.uint32), b[7].view(np.uint32)))
    if not ok:
        bad += 1
        print("MISMATCH case", c, kw, src.shape, tgt.shape, a[1:5], b[1:5], flush=True)
    if c % 50 == 49:
        print(f"{c + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
