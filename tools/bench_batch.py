"""Diagnostic: frame-batch throughput on one GPU (config-2 pairs resident in HBM) against the
single-pair rate, for a few group sizes.  usage: python tools/bench_batch.py [n_pairs] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # first: its HIP runtime must be the one libicpk.so binds to
torch.cuda.init()
import numpy as np
from icp_slam_prototype_amd import binding, synth

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows, cols = (int(os.environ.get("ROWS", 480)), int(os.environ.get("COLS", 640)))
dev = []
for k in range(n_pairs):
    p = synth.kinect_pair(rows, cols, valid=0.30, seed=100 + k)
    s = torch.from_numpy(np.ascontiguousarray(p["source"])).cuda()
    t = torch.from_numpy(np.ascontiguousarray(p["target"])).cuda()
    dev.append((s, t))
torch.cuda.synchronize()
args = [(s.data_ptr(), s.shape[1], t.data_ptr(), t.shape[1]) for s, t in dev]
par = binding.default_params(max_iterations=iters, fixed_iterations=1)

ctx = binding.Context(0)
s, t = dev[0]
ctx.set_target_device(t.data_ptr(), t.data_ptr() + 4 * t.shape[1], t.data_ptr() + 8 * t.shape[1], t.shape[1])
ctx.set_source_device(s.data_ptr(), s.data_ptr() + 4 * s.shape[1], s.data_ptr() + 8 * s.shape[1], s.shape[1])
for _ in range(3):
    ctx.align(par)
t0 = time.perf_counter()
for _ in range(10):
    ctx.align(par)
dt = (time.perf_counter() - t0) / 10
print(f"single pair resident: {dt*1e3:.3f} ms/alignment  {iters/dt:.0f} iter/s", flush=True)
ctx.close()
for g in [int(v) for v in os.environ.get("BGROUPS", "1,2,4,8,16").split(",")]:
    os.environ["ICPK_BATCH_GROUP"] = str(g)
    c = binding.Context(0)
    for _ in range(4):  # the runtime's pools settle within the first three calls of a process
        c.align_batch_device(args, par)
    reps = int(os.environ.get("REPS", 3))
    each = []
    t0 = time.perf_counter()
    for _ in range(reps):
        t1 = time.perf_counter()
        T, st, rc = c.align_batch_device(args, par)
        each.append(round((time.perf_counter() - t1) * 1e3, 2))
    dt = sorted(each)[len(each) // 2] / 1e3  # median
    if os.environ.get("EACH"):
        print("   per rep ms:", each, flush=True)
    assert rc == 0 and all(x.iterations == iters for x in st)
    print(f"group {g:2d}: {n_pairs} pairs in {dt*1e3:.2f} ms  = {dt/n_pairs*1e3:.3f} ms/pair  {n_pairs*iters/dt:.0f} iter/s", flush=True)
    c.close()
