import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
import numpy as np
from icp_slam_prototype_amd import binding, synth
dev = []
for k in range(32):
    p = synth.kinect_pair(480, 640, valid=0.30, seed=100 + k)
    dev.append((torch.from_numpy(np.ascontiguousarray(p["source"])).cuda(), torch.from_numpy(np.ascontiguousarray(p["target"])).cuda()))
torch.cuda.synchronize()
args = [(s.data_ptr(), s.shape[1], t.data_ptr(), t.shape[1]) for s, t in dev]
par = binding.default_params(max_iterations=20, fixed_iterations=1)
c = binding.Context(0)
if os.environ.get("NOGC"):
    import gc
    gc.collect(); gc.disable()
for i in range(int(os.environ.get("CALLS", 6))):
    t0 = time.perf_counter()
    c.align_batch_device(args, par)
    print(f"call {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", file=sys.stderr, flush=True)
