"""Diagnostic (tools only): per-wave work counters of the steady grid sweep
(-DICPK_GRID_STAMPS -DICPK_GRID_COUNTS): chunks, candidate batches, exact-path executions.
usage: ICPK_GRID_SLICES=4 python tools/count_grid.py [sweeps_before]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import build, binding, synth

os.makedirs("/tmp/icpk_gc", exist_ok=True)
binding.LIB_PATH = build.build(force=True, extra=["-DICPK_GRID_STAMPS", "-DICPK_GRID_COUNTS"], out="/tmp/icpk_gc/libicpk.so")
p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
ctx = binding.Context(0)
ctx.set_target(p["target"]); ctx.set_source(p["source"])
L = binding.load()
NW = 16384
for iters in [int(a) for a in sys.argv[1:]] or [1, 3, 8, 15]:
    ctx.reset_source()
    ctx.align(max_iterations=iters, fixed_iterations=1, nn_mode=binding.NN_GRID, host_loop=1)
    L.icpk_debug_clear_grid_stamps()
    ctx.nn(binding.NN_GRID, fetch=False)
    buf = np.zeros(8 * NW, np.uint64)
    L.icpk_debug_read_grid_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
    b = buf.reshape(NW, 8).astype(np.int64)
    b = b[b[:, 1] > 0]  # waves that ran at least one chunk
    print(f"sweep after {iters} iterations: waves {len(b)}  chunks/wave {b[:,1].mean():.2f}  batches/wave {b[:,2].mean():.2f}  "
          f"exact executions/wave {b[:,3].mean():.2f}  lanes per execution {b[:,7].sum() / max(b[:,3].sum(), 1):.2f}  "
          f"rows/query {b[:,5].sum() / 91870:.2f}  candidates/query {b[:,6].sum() / 91870:.1f}")
