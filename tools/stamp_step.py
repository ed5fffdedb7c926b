"""Diagnostic (tools only): phases of loop_step_kernel (-DICPK_STEP_STAMPS build in /tmp)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import build, binding, synth

os.makedirs("/tmp/icpk_step", exist_ok=True)
binding.LIB_PATH = build.build(force=True, extra=["-DICPK_STEP_STAMPS"], out="/tmp/icpk_step/libicpk.so")
p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
ctx = binding.Context(0)
ctx.set_target(p["target"]); ctx.set_source(p["source"])
L = binding.load()
p2l = "--p2l" in sys.argv
if p2l:  # BASELINE configs[2]: Kinect v2 pair, target normals from the depth image
    p = synth.kinect_pair(424, 512, valid=1.0, seed=2, fx=synth.K2_FX, cx=synth.K2_CX)
    ctx.set_target(p["target"]); ctx.set_source(p["source"])
    ctx.backproject_with_normals(p["depth_tgt"], binding.NORMALS_CROSS, offset=[5, 5, 5], fx=float(synth.K2_FX), cx=float(synth.K2_CX))
for solve in ((binding.SOLVE_POINT_TO_PLANE,) if p2l else (binding.SOLVE_REFERENCE,)):
    for _ in range(3):
        ctx.align(max_iterations=20, fixed_iterations=1, solve=solve, max_nn_dist=0.3 if p2l else 0.75)
    buf = np.zeros(8 * 64, np.uint64)
    L.icpk_debug_read_step_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
    b = buf.reshape(64, 8).astype(np.int64)[:20]
    for name, a, c in (("stage2", 0, 1), ("control", 1, 2), ("solve", 2, 3), ("  polar", 2, 5), ("  rest", 5, 3), ("store", 3, 4), ("total", 0, 4)):
        v = (b[:, c] - b[:, a]) * 10.0
        print(f"solve={solve} {name:8s} ns: mean {v.mean():8.0f} min {v.min():8.0f} max {v.max():8.0f}")
    print("step-to-step period ns:", np.diff(b[:, 0]).mean() * 10.0)
