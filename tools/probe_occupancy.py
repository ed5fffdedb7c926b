"""tools only: pruned NN sweep time vs number of queries (is the 5742-wave launch paying for
a second residency round at 5 waves/SIMD = 5120 slots?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth

p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
for nq in (60000, 70000, 78000, 81000, 81920, 83000, 86000, 91870):
    ctx = binding.Context(0)
    ctx.set_target(p["target"]); ctx.set_source(np.ascontiguousarray(p["source"][:, :nq]))
    prm = binding.default_params(max_iterations=40, fixed_iterations=1, profile=1, solve=binding.SOLVE_KABSCH)
    ctx.align(prm)
    T, st, rc = ctx.align(prm)
    print(f"nq={nq:6d} waves={nq//16:5d} nn_us/sweep={1e3*st.nn_ms_total/st.nn_launches:.1f}")
    ctx.close()
