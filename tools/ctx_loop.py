"""Diagnostic (tools only): device memory after many context create/destroy cycles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icp_slam_prototype_amd import binding, synth
p = synth.kinect_pair(rows=120, cols=160, seed=5)
def loop(n, work):
    f0 = torch.cuda.mem_get_info()[0]
    for k in range(n):
        c = binding.Context(0)
        if work:
            c.set_target(p["target"]); c.set_source(p["source"])
            c.align(max_iterations=3, fixed_iterations=1)
        c.close()
    f1 = torch.cuda.mem_get_info()[0]
    print(f"{n} contexts, work={work}: delta {(f0 - f1) / 2**20:.1f} MiB")
def loop_batch(n):
    pairs = [(p["source"], p["target"])] * 5
    f0 = torch.cuda.mem_get_info()[0]
    for k in range(n):
        os.environ["ICPK_BATCH_GROUP"] = "2"
        c = binding.Context(0)
        c.align_batch(pairs, max_iterations=3, fixed_iterations=1)  # slots, pools, set-up streams, events
        c.align(max_iterations=8, threshold=1e-9) if False else None
        c.close()
    f1 = torch.cuda.mem_get_info()[0]
    print(f"{n} contexts with a 5-pair batch each (groups of 2): delta {(f0 - f1) / 2**20:.1f} MiB")
loop(50, False); loop(50, False); loop(50, True); loop(50, True); loop(200, True)
loop_batch(10); loop_batch(10); loop_batch(30)
