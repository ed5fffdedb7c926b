"""Diagnostic: wall time of each C-ABI call of the drop-in frame path (tools/bench_tracker.py's loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth
frames = []
rng = np.random.default_rng(0)
for k in range(6):
    d = synth.render_room_depth(480, 640, synth.rot_xyz_deg(0, 0.5 * k, 0), np.array([0.01 * k, 0, 0]), noise_sigma=0.002, rng=rng)
    d[rng.random(d.shape) > 0.3] = 0
    frames.append(d.astype(np.uint16))
ctx = binding.Context(0)
camR = np.eye(3, dtype=np.float32); camP = np.full(3, 5, np.float32)
acc = {}
def tm(name, fn):
    t0 = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0) + (time.perf_counter() - t0); return r
n = 0
for rep in range(12):
    if rep == 2:
        acc.clear(); n = 0
    for i in range(1, len(frames)):
        tm("bp_pair", lambda: ctx.backproject_pair(frames[i], frames[i - 1], R=camR, t=camP))
        st = tm("align", lambda: ctx.align(max_iterations=16, threshold=1e-4))[1]
        tm("trace", lambda: ctx.get_trace(16))
        n += 1
print({k: round(v / n * 1e6) for k, v in acc.items()}, "us per call; iterations", st.iterations)
acc.clear()
for rep in range(50):
    tm("align, same clouds (grid cached)", lambda: (ctx.reset_source(), ctx.align(max_iterations=16, threshold=1e-4)))
    tm("align fixed 5", lambda: (ctx.reset_source(), ctx.align(max_iterations=5, fixed_iterations=1)))
    tm("align fixed 1", lambda: (ctx.reset_source(), ctx.align(max_iterations=1, fixed_iterations=1)))
print({k: round(v / 50 * 1e6) for k, v in acc.items()}, "us per call")
