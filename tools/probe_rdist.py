"""Diagnostic (tools only): distribution of NN distances over the iterations of config 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth

for name, p in (("config2", synth.kinect_pair(480, 640, valid=0.30, seed=2)),
                ("config5_1M", synth.dense_pair(1_000_000))):
    ctx = binding.Context(0)
    ctx.set_target(p["target"]); ctx.set_source(p["source"])
    for k in (0, 1, 2, 5, 19):
        ctx.reset_source()
        ctx.align(max_iterations=k, fixed_iterations=1)
        idx, d = ctx.get_associations()
        qs = np.quantile(d, [0.5, 0.9, 0.95, 0.99, 0.999, 1.0])
        print(name, "iter", k, "quantiles 50/90/95/99/99.9/100 (m):", " ".join(f"{q:.4f}" for q in qs),
              "frac>0.04:", float((d > 0.04).mean()), "frac>0.08:", float((d > 0.08).mean()), flush=True)
    ctx.close()
