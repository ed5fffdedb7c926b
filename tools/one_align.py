"""Diagnostic (tools only): one alignment of config 2 (for counter collection)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icp_slam_prototype_amd import binding, synth
p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
ctx = binding.Context(0)
ctx.set_target(p["target"]); ctx.set_source(p["source"])
mode = int(os.environ.get("ICPK_AB_MODE", binding.NN_GRID))
T, st, rc = ctx.align(max_iterations=10, fixed_iterations=1, nn_mode=mode)
print("done", st.iterations, flush=True)
