"""Diagnostic (tools only): one workload, single-pair alignments with the library named by ICPK_LIB_PATH
(a variant under icp_slam_prototype_amd/lib_variants/) or the product build: iterations/s (median of 15),
per-kernel HIP-event times and a hash of T so that variants can be checked for equal results."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
p = {"config2": lambda: synth.kinect_pair(480, 640, valid=0.30, seed=2),
     "dense": lambda: synth.kinect_pair(480, 640, valid=1.0, seed=2),
     "config5": lambda: synth.dense_pair(1_000_000, seed=5),
     "config3": lambda: synth.kinect_pair(424, 512, valid=1.0, seed=2, fx=synth.K2_FX, cx=synth.K2_CX)}[wl]()
c = binding.Context(0)
c.set_target(p["target"]); c.set_source(p["source"])
kw = {}
if wl == "config3":  # point-to-plane: target and normals from the depth image
    c.backproject_with_normals(p["depth_tgt"], binding.NORMALS_CROSS, offset=[5, 5, 5], fx=float(synth.K2_FX), cx=float(synth.K2_CX))
    kw = dict(solve=binding.SOLVE_POINT_TO_PLANE, max_nn_dist=0.3)
for _ in range(3):
    c.reset_source(); T, st, rc = c.align(max_iterations=20, fixed_iterations=1, **kw)
ts = []
for _ in range(15):
    c.reset_source()
    t0 = time.perf_counter(); T, st, rc = c.align(max_iterations=20, fixed_iterations=1, **kw); ts.append(time.perf_counter() - t0)
c.reset_source()
T2, st2, rc = c.align(max_iterations=20, fixed_iterations=1, profile=2, **kw)
print(f"{os.environ.get('ICPK_LIB_PATH', 'product')}: {wl} {20 / sorted(ts)[7]:8.0f} iter/s  nn {st2.nn_ms_total / max(st2.nn_timed_launches, 1) * 1e3:6.2f} us  "
      f"reduce {st2.reduce_ms_total / 21 * 1e3:6.2f} us  total {st2.total_ms * 1e3 / 20:6.2f} us/iter  T crc {zlib.crc32(T.tobytes()):08x} pairs {st.final_pairs}", flush=True)
