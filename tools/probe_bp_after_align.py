"""Diagnostic: why the target back-projection that follows an alignment takes 2.5 x its isolated time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
def tm(acc, name, fn):
    t0 = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0) + (time.perf_counter() - t0); return r
def run(label, seq, n=40):
    acc = {}
    for rep in range(n + 5):
        if rep == 5: acc.clear()
        for name, fn in seq: tm(acc, name, fn)
    print(label, {k: round(v / n * 1e6) for k, v in acc.items()}, flush=True)
bp_t = ("bp_tgt", lambda: ctx.backproject(frames[0], which=1))
bp_s = ("bp_src", lambda: ctx.backproject(frames[1], which=0))
al = ("align", lambda: ctx.align(max_iterations=16, threshold=1e-4))
al_fixed = ("align", lambda: ctx.align(max_iterations=5, fixed_iterations=1))
nn1 = ("nn", lambda: ctx.nn(binding.NN_GRID))
tr = ("trace", lambda: ctx.get_trace(16))
slp = ("sleep", lambda: time.sleep(0.0005))
trt = ("tr_tgt", lambda: ctx.transform_target(camR, camP))
trs = ("tr_src", lambda: ctx.transform_source(camR, camP))
com = ("commit", lambda: ctx.commit_source())
run("bp + align           ", [bp_t, bp_s, al])
run("bp + align + trace   ", [bp_t, bp_s, al, tr])
run("bp,tr_tgt + align    ", [bp_t, trt, bp_s, al])
run("bp,tr_src,com + align", [bp_t, bp_s, trs, com, al])
run("all                  ", [bp_t, trt, bp_s, trs, com, al, tr])
k = [0]
def nxt(which):
    def f():
        k[0] += 1
        return ctx.backproject(frames[(k[0] // 2) % 4], which=which)
    return f
run("all, frames vary     ", [("bp_tgt", nxt(1)), trt, ("bp_src", nxt(0)), trs, com, al, tr])
