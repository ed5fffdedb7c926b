"""Diagnostic (tools only): a few alignments of one workload, for rocprofv3 (kernel trace or
--pmc counter passes).  No torch: libicpk.so alone initialises the GPU.
usage: python3 tools/one_align.py [--workload W] [--nn-mode grid|filtered|pruned|exact] [--iters I] [--reps R]
       [--batch P]   (P > 0: P config-2 pairs through icpk_align_batch instead)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icp_slam_prototype_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="kinect640x480_30pct")
ap.add_argument("--nn-mode", default=None)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--batch", type=int, default=0)
a = ap.parse_args()
modes = {"exact": binding.NN_EXACT, "filtered": binding.NN_FILTERED, "pruned": binding.NN_PRUNED, "grid": binding.NN_GRID}
mode = modes[a.nn_mode] if a.nn_mode else int(os.environ.get("ICPK_AB_MODE", binding.NN_GRID))
ctx = binding.Context(0)
if a.batch > 0:
    pairs = []
    for k in range(a.batch):
        p = synth.kinect_pair(480, 640, valid=0.30, seed=100 + k)
        pairs.append((p["source"], p["target"]))
    for _ in range(a.reps):
        T, st, rc = ctx.align_batch(pairs, max_iterations=a.iters, fixed_iterations=1)
    print("done batch", rc, st[0].iterations, flush=True)
else:
    if a.workload == "kinect640x480_30pct":
        p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
    elif a.workload == "kinect640x480_dense":
        p = synth.kinect_pair(480, 640, valid=1.0, seed=2)
    elif a.workload == "kinect_v2_512x424":
        p = synth.kinect_pair(424, 512, valid=1.0, seed=2, fx=synth.K2_FX, cx=synth.K2_CX)
    elif a.workload == "dense1m":
        p = synth.dense_pair(1_000_000, seed=5)
    else:
        p = synth.frustum_pair(10000, seed=2)
    ctx.set_target(p["target"])
    ctx.set_source(p["source"])
    for _ in range(a.reps):
        T, st, rc = ctx.align(max_iterations=a.iters, fixed_iterations=1, nn_mode=mode)
    print("done", st.iterations, flush=True)
ctx.close()
