"""Diagnostic (tools only): per-iteration movement of the source points against their NN distance
(decides whether bound-based skipping of NN searches can pay).  usage: python tools/probe_motion.py [solve]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth

solve = int(sys.argv[1]) if len(sys.argv) > 1 else 0
p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
ctx = binding.Context(0)
ctx.set_target(p["target"]); ctx.set_source(p["source"])
T, st, rc = ctx.align(max_iterations=20, fixed_iterations=1, solve=solve)
tr = ctx.get_trace(32)
idx, dist = ctx.get_associations()
print("final NN distance: p10 %.4f p50 %.4f p90 %.4f" % tuple(np.quantile(dist, [.1, .5, .9])))
pts = p["source"].astype(np.float64)
prev_idx = None
for i, t in enumerate(tr):
    R = t["R"].astype(np.float64)
    if solve == 0:
        Rinv = np.linalg.inv(R)
        new = Rinv @ pts - t["t"].astype(np.float64)[:, None]
    else:
        new = R @ pts + t["t"].astype(np.float64)[:, None]
    d = np.linalg.norm(new - pts, axis=0)
    pts = new
    print(f"iter {i:2d}: pairs {t['n_pairs']} mse {t['mse']:.3e} movement p50 {np.median(d)*1e3:8.4f} mm p90 {np.quantile(d,.9)*1e3:8.4f} mm max {d.max()*1e3:8.4f} mm")
