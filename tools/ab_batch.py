"""Diagnostic (tools only): build libicpk with extra -D flags into /tmp and time a 16-pair
lock-step batch of config-2 pairs (device resident).  usage: python tools/ab_batch.py [-DFLAG ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--build-only" not in sys.argv:
    import torch
    torch.cuda.init()
import numpy as np
from icp_slam_prototype_amd import build, binding, synth

flags = [a for a in sys.argv[1:] if a.startswith("-D")]
tag = "_".join(f[2:].replace("=", "") for f in flags) or "base"
# variants are built where the snapshot travels from (build them in the container, run on the GPU box)
vdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "icp_slam_prototype_amd", "lib_variants", tag)
os.makedirs(vdir, exist_ok=True)
vlib = os.path.join(vdir, "libicpk.so")
if flags and not os.path.exists(vlib):
    build.build(force=True, extra=flags, out=vlib)
if flags:
    binding.LIB_PATH = vlib
if "--build-only" in sys.argv:
    sys.exit(0)
n = int(os.environ.get("PAIRS", 16))
dev = []
for k in range(n):
    p = synth.kinect_pair(480, 640, valid=0.30, seed=100 + k)
    dev.append((torch.from_numpy(np.ascontiguousarray(p["source"])).cuda(), torch.from_numpy(np.ascontiguousarray(p["target"])).cuda()))
torch.cuda.synchronize()
args = [(s.data_ptr(), s.shape[1], t.data_ptr(), t.shape[1]) for s, t in dev]
par = binding.default_params(max_iterations=20, fixed_iterations=1)
c = binding.Context(0)
for _ in range(4):
    c.align_batch_device(args, par)
each = []
for _ in range(7):
    t0 = time.perf_counter()
    T, st, rc = c.align_batch_device(args, par)
    each.append(time.perf_counter() - t0)
dt = sorted(each)[len(each) // 2]
# single pair too
cs = binding.Context(0)
s0, t0_ = dev[0]
cs.set_target_device(t0_.data_ptr(), t0_.data_ptr() + 4 * t0_.shape[1], t0_.data_ptr() + 8 * t0_.shape[1], t0_.shape[1])
cs.set_source_device(s0.data_ptr(), s0.data_ptr() + 4 * s0.shape[1], s0.data_ptr() + 8 * s0.shape[1], s0.shape[1])
for _ in range(3):
    cs.align(par)
ts = []
for _ in range(15):
    t1 = time.perf_counter()
    cs.align(par)
    ts.append(time.perf_counter() - t1)
single = 20 / sorted(ts)[len(ts) // 2]
print(f"{tag} env={ {k: v for k, v in os.environ.items() if k.startswith('ICPK_')} }: single {single:8.0f} iter/s;  {n} pairs {dt * 1e3:.3f} ms  {n * 20 / dt:9.0f} iter/s  T[0][0,3]={T[0][0,3]:.6f}", flush=True)
