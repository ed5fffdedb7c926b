import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from icp_slam_prototype_amd import binding, synth
frames = []
rng = np.random.default_rng(0)
for k in range(4):
    d = synth.render_room_depth(480, 640, synth.rot_xyz_deg(0, 0.5 * k, 0), np.array([0.01 * k, 0, 0]), noise_sigma=0.002, rng=rng)
    d[rng.random(d.shape) > 0.3] = 0
    frames.append(d.astype(np.uint16))
ctx = binding.Context(0)
camR = np.eye(3, dtype=np.float32); camP = np.full(3, 5, np.float32)
acc = {}
def tm(name, fn):
    t0 = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0) + (time.perf_counter() - t0); return r
for rep in range(20):
    for i in range(1, 4):
        tm("bp_tgt", lambda: ctx.backproject(frames[i-1], which=1))
        tm("tr_tgt", lambda: ctx.transform_target(camR, camP))
        tm("bp_src", lambda: ctx.backproject(frames[i], which=0))
        tm("tr_src", lambda: ctx.transform_source(camR, camP))
        tm("commit", lambda: ctx.commit_source())
        st = tm("align", lambda: ctx.align(max_iterations=16, threshold=1e-4))[1]
        tm("trace", lambda: ctx.get_trace(16))
n = 60
print({k: round(v / n * 1e6) for k, v in acc.items()}, "us per call; iterations", st.iterations)
acc.clear()
for rep in range(30):
    tm("bp_src_first", lambda: ctx.backproject(frames[1], which=0))
    tm("bp_tgt_second", lambda: ctx.backproject(frames[0], which=1))
    tm("bp_tgt_again", lambda: ctx.backproject(frames[0], which=1))
    tm("bp_src_again", lambda: ctx.backproject(frames[1], which=0))
print({k: round(v / 30 * 1e6) for k, v in acc.items()}, "us per call")
