"""Diagnostic (tools only): where a wave of nn_pruned_kernel spends its cycles.
Builds libicpk with -DICPK_NP_STAMPS into /tmp, runs a few sweeps of config 2, prints
per-phase s_memtime statistics (100 MHz constant clock ticks: 10 ns each)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import build, binding, synth

os.makedirs("/tmp/icpk_dbg", exist_ok=True)
lib = build.build(force=True, extra=["-DICPK_NP_STAMPS"], out="/tmp/icpk_dbg/libicpk.so")
binding.LIB_PATH = lib
p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
ctx = binding.Context(0)
ctx.set_target(p["target"]); ctx.set_source(p["source"])
L = binding.load()
ctx.align(max_iterations=6, fixed_iterations=1, solve=binding.SOLVE_KABSCH, host_loop=1)
L.icpk_debug_clear_stamps()
ctx.nn(binding.NN_PRUNED, fetch=False)   # one steady-state sweep (seeded by the previous matches)
buf = np.zeros(8 * 4096, np.uint64)
L.icpk_debug_read_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
b = buf.reshape(4096, 8).astype(np.int64)
t0 = b[:, 0].min()
ph = {"init": b[:, 1] - b[:, 0], "passes": b[:, 2] - b[:, 1], "eager": b[:, 3] - b[:, 2], "scan": b[:, 4] - b[:, 3],
      "total": b[:, 4] - b[:, 0], "start": b[:, 0] - t0, "end": b[:, 4] - t0}
for k, v in ph.items():
    v = v[(b[:, 2] > 0) & (b[:, 3] > 0)] if k in ("passes", "eager", "scan") else v
    print(f"{k:7s} ticks(10ns): mean {v.mean():9.1f} p50 {np.median(v):9.1f} p90 {np.quantile(v,0.9):9.1f} max {v.max():9.1f}")
for k, name in ((5, "coarse candidates"), (6, "survivors"), (7, "scanned")):
    v = b[:, k]
    print(f"{name:18s}: mean {v.mean():7.1f} p50 {np.median(v):6.0f} p90 {np.quantile(v,0.9):6.0f} max {v.max():6d}")
