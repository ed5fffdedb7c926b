"""Diagnostic: icpk_align_batch with HOST buffers (64 config-2 pairs, every cloud crossing PCIe inside the call)
for ICPK_BATCH_SETUP = 0 / 1 / 2 (2: the batched set-up launches for host-pointer batches too)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pairs = []
for k in range(n):
    p = synth.kinect_pair(480, 640, valid=0.30, seed=100 + k)
    pairs.append((p["source"], p["target"]))
ref = None
for mode in ("0", "1", "2", "1", "2"):
    os.environ["ICPK_BATCH_SETUP"] = mode
    with binding.Context(0) as c:
        for _ in range(3):
            T, st, rc = c.align_batch(pairs, max_iterations=20, fixed_iterations=1)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            T, st, rc = c.align_batch(pairs, max_iterations=20, fixed_iterations=1)
            ts.append(time.perf_counter() - t0)
    if ref is None:
        ref = T.copy()
    dt = sorted(ts)[2]
    print(f"ICPK_BATCH_SETUP={mode}: {n} pairs from host buffers {dt * 1e3:.2f} ms  {n * 20 / dt:.0f} iter/s  same bits {np.array_equal(ref, T)}", flush=True)
