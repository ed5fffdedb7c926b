"""Diagnostic (tools only): build libicpk with extra -D flags into /tmp and time config 2.
usage: python tools/ab_variant.py [-DFLAG ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icp_slam_prototype_amd import build, binding, synth

flags = [a for a in sys.argv[1:] if a.startswith("-D")]
tag = "_".join(f[2:] for f in flags) or "base"
os.makedirs(f"/tmp/icpk_{tag}", exist_ok=True)
binding.LIB_PATH = build.build(force=True, extra=flags, out=f"/tmp/icpk_{tag}/libicpk.so")
p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
ctx = binding.Context(0)
ctx.set_target(p["target"]); ctx.set_source(p["source"])
mode = int(os.environ.get("ICPK_AB_MODE", binding.NN_PRUNED))
par = binding.default_params(max_iterations=20, fixed_iterations=1, profile=0, nn_mode=mode)
for _ in range(3):
    ctx.align(par)
best = 1e9
for _ in range(5):
    t = time.perf_counter()
    for _ in range(10):
        T, st, rc = ctx.align(par)
    best = min(best, (time.perf_counter() - t) / 10)
par.profile = 2
_, st, _ = ctx.align(par)
print(f"{tag} mode={mode} ppc={os.environ.get('ICPK_GRID_PPC')} gs={os.environ.get('ICPK_GRID_SLICES')}: {20 / best:9.1f} iter/s  ({best * 1e3:.3f} ms/align; nn {st.nn_ms_total:.3f} reduce {st.reduce_ms_total:.3f} ms)  T[0,3]={T[0,3]:.6f}")
