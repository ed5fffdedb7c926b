"""Diagnostic: the drop-in path as SLAM.cpp would drive it (icp::Tracker::getTransformation's call
sequence through the C ABI): per frame pair two uint16 depth images cross PCIe, are filtered
(optional) and back-projected on the device, posed, aligned (16 iterations max, threshold 1e-4,
SLAM.cpp:277) and the trace is read back.  Prints frame pairs per second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth

filt = "--filter" in sys.argv
frames = []
rng = np.random.default_rng(0)
for k in range(6):
    d = synth.render_room_depth(480, 640, synth.rot_xyz_deg(0, 0.5 * k, 0), np.array([0.01 * k, 0, 0]), noise_sigma=0.002, rng=rng)
    d[rng.random(d.shape) > 0.3] = 0
    frames.append(d.astype(np.uint16))
ctx = binding.Context(0)
camR = np.eye(3, dtype=np.float32)
camP = np.full(3, 5, np.float32)
separate = "--separate" in sys.argv  # the five set-up calls of round 2's first half instead of icpk_backproject_pair
both = "--both" in sys.argv  # upload the previous frame too (round 2); default: it stayed on the device (SLAM.cpp:305)
first = [True]
def pair(prev, cur):
    if separate:
        bp = (lambda d, w: ctx.backproject_filtered(d, which=w)) if filt else (lambda d, w: ctx.backproject(d, which=w))
        bp(prev, 1); ctx.transform_target(camR, camP)
        bp(cur, 0); ctx.transform_source(camR, camP); ctx.commit_source()
    else:
        ctx.backproject_pair(cur, prev if (both or first[0]) else None, R=camR, t=camP, filter=filt)
        first[0] = False
    T, st, rc = ctx.align(max_iterations=16, threshold=1e-4)
    ctx.get_trace(16)
    return st
for i in range(1, len(frames)):
    pair(frames[i - 1], frames[i])
t0 = time.perf_counter()
n = 0
for rep in range(4):
    for i in range(1, len(frames)):
        first[0] = first[0] or i == 1  # (the sequence wraps around)
        st = pair(frames[i - 1], frames[i]); n += 1
dt = time.perf_counter() - t0
print(f"tracker path{' + filterDepthImage' if filt else ''}{' (separate set-up calls)' if separate else ''}{' (both frames uploaded)' if both else ''}: {n / dt:.0f} frame pairs/s ({dt / n * 1e3:.2f} ms per pair; last pair {st.iterations} iterations, {ctx.source_size} x {ctx.target_size} points)")
