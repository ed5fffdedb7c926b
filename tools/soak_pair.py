"""Soak test (tools only): icpk_backproject_pair against the step-by-step set-up calls on random frame
sizes, validity masks, poses, offsets and filters: clouds, the alignment that follows (threshold mode: the
throttled loop) and the aligned source, bit for bit.  usage: soak_pair.py [n] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 31000
a, b = binding.Context(0), binding.Context(0)
bad = 0
t0 = time.time()
for c in range(n_cases):
    rng = np.random.default_rng(seed0 + c)
    rows, cols = int(rng.integers(8, 300)), int(rng.integers(8, 400))
    fx = float(rng.uniform(100, 600)); cx = float(rng.uniform(0, cols))
    p = synth.kinect_pair(rows, cols, valid=float(rng.uniform(0.05, 1.0)), seed=int(rng.integers(0, 1 << 30)), fx=fx, cx=cx,
                          rot_deg=tuple(rng.uniform(-2, 2, 3)), shift=tuple(rng.uniform(-0.03, 0.03, 3)), noise_sigma=0.001)
    ds, dt = p["depth_src"].copy(), p["depth_tgt"].copy()
    filt = bool(rng.random() < 0.4)
    if filt:
        ds[rng.random(ds.shape) < 0.02] = 40000
        dt[rng.random(dt.shape) < 0.02] = 200
    posed = rng.random() < 0.8
    R = binding.make_rotation_matrix(*rng.uniform(-20, 20, 3)) if posed else None
    t = rng.uniform(-6, 6, 3).astype(np.float32) if posed else None
    off = None if rng.random() < 0.5 else rng.uniform(-2, 2, 3).astype(np.float32)
    morph = bool(rng.random() < 0.7)
    kw = dict(fx=fx, cx=cx, offset=off)
    if filt:
        nt = a.backproject_filtered(dt, which=1, morph=morph, **kw)
    else:
        nt = a.backproject(dt, which=1, **kw)
    if posed: a.transform_target(R, t)
    ns = a.backproject_filtered(ds, which=0, morph=morph, **kw) if filt else a.backproject(ds, which=0, **kw)
    if posed: a.transform_source(R, t)
    a.commit_source()
    got = b.backproject_pair(ds, dt, R=R, t=t, filter=filt, morph=morph, **kw)
    ok = got == (ns, nt)
    if ok and nt > 0 and ns > 0:
        ok = a.get_source().tobytes() == b.get_source().tobytes() and a.get_target().tobytes() == b.get_target().tobytes()
        par = dict(max_iterations=int(rng.integers(1, 14)), threshold=float(10 ** rng.uniform(-7, -3)), solve=int(rng.integers(0, 2)))
        ra, rb = a.align(**par), b.align(**par)
        ok = ok and ra[0].tobytes() == rb[0].tobytes() and ra[2] == rb[2] and ra[1].iterations == rb[1].iterations and \
            ra[1].final_pairs == rb[1].final_pairs and a.get_source().tobytes() == b.get_source().tobytes()
    if not ok:
        bad += 1
        print("MISMATCH case", c, rows, cols, filt, posed, got, (ns, nt), flush=True)
    if c % 50 == 49:
        print(f"{c + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
