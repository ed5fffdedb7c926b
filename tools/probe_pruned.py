"""Probe: pruned NN kernel time on clouds with / without outliers (tools only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth

def run(name, src, tgt, mode=binding.NN_PRUNED, iters=60, solve=binding.SOLVE_KABSCH):
    ctx = binding.Context(0)
    ctx.set_target(tgt); ctx.set_source(src)
    p = binding.default_params(max_iterations=iters, fixed_iterations=1, profile=1, nn_mode=mode, solve=solve)
    ctx.align(p)
    T, st, rc = ctx.align(p)
    print(f"{name:28s} nq={src.shape[1]} nt={tgt.shape[1]} nn_ms/sweep={st.nn_ms_total/st.nn_launches:.4f} total_ms={st.total_ms:.3f} pairs={st.final_pairs}")
    ctx.close()

p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
run("kinect pair (outliers)", p["source"], p["target"])
run("kinect pair ref-solve", p["source"], p["target"], solve=binding.SOLVE_REFERENCE)
t = p["target"]
run("self + 0.5mm", (t + np.float32(0.0005)).astype(np.float32), t)
run("self + 2cm", (t + np.float32(0.02)).astype(np.float32), t)
# drop source points whose NN after alignment is far (emulate no outliers)
ctx = binding.Context(0); ctx.set_target(t); ctx.set_source(p["source"])
ctx.align(max_iterations=20, fixed_iterations=1, solve=binding.SOLVE_KABSCH)
idx, dist = ctx.get_associations(); ctx.close()
keep = dist < 0.03
print("inlier fraction", keep.mean(), "dist quantiles", np.quantile(dist, [0.5, 0.9, 0.99, 0.999]))
run("kinect inliers only", np.ascontiguousarray(p["source"][:, keep]), t)
