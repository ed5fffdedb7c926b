"""Diagnostic (tools only): phases of assoc_reduce_kernel (-DICPK_RED_STAMPS build in /tmp)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import build, binding, synth

os.makedirs("/tmp/icpk_red", exist_ok=True)
binding.LIB_PATH = build.build(force=True, extra=["-DICPK_RED_STAMPS"], out="/tmp/icpk_red/libicpk.so")
p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
ctx = binding.Context(0)
ctx.set_target(p["target"]); ctx.set_source(p["source"])
L = binding.load()
for _ in range(3):
    ctx.align(max_iterations=20, fixed_iterations=1)
buf = np.zeros(8 * 256, np.uint64)
L.icpk_debug_read_red_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
b = buf.reshape(256, 8).astype(np.int64)
t0 = b[:, 0].min()
print("span us", (b[:, 3].max() - t0) / 100.0)
for name, a, c in (("loads+acc", 0, 1), ("butterfly", 1, 2), ("lds+store", 2, 3), ("total", 0, 3)):
    v = (b[:, c] - b[:, a]) * 10.0
    print(f"{name:10s} ns: mean {v.mean():7.0f} p50 {np.median(v):7.0f} max {v.max():7.0f}")
print("block start ns: p50", np.median(b[:, 0] - t0) * 10, "max", (b[:, 0] - t0).max() * 10)
