"""Diagnostic (tools only): does the HIP runtime stall once after some thousand launches of a process?
Times chunks of small launches (icpk_transform_source = one kernel + one stream sync each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth
p = synth.frustum_pair(2000, seed=1)
ctx = binding.Context(0)
ctx.set_target(p["target"]); ctx.set_source(p["source"])
R = np.eye(3, dtype=np.float32); t = np.zeros(3, np.float32)
chunk = 250
out = []
for c in range(int(os.environ.get("CHUNKS", 60))):
    t0 = time.perf_counter()
    for _ in range(chunk):
        ctx.transform_source(R, t)
    out.append((time.perf_counter() - t0) * 1e3)
print("ms per chunk of", chunk, "launches:", " ".join(f"{v:.1f}" for v in out))
