"""Diagnostic (tools only): is the tail of the grid sweep made of waves with much work?
Two diagnostic builds: stamps only (times), stamps + counters (work); same clouds, same sweep."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import build, binding, synth

def run(flags, tag):
    os.makedirs(f"/tmp/icpk_{tag}", exist_ok=True)
    binding.LIB_PATH = build.build(force=True, extra=flags, out=f"/tmp/icpk_{tag}/libicpk.so")
    binding._lib = None
    p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
    ctx = binding.Context(0)
    ctx.set_target(p["target"]); ctx.set_source(p["source"])
    L = binding.load()
    ctx.align(max_iterations=8, fixed_iterations=1, nn_mode=binding.NN_GRID, host_loop=1)
    L.icpk_debug_clear_grid_stamps()
    ctx.nn(binding.NN_GRID, fetch=False)
    buf = np.zeros(8 * 16384, np.uint64)
    L.icpk_debug_read_grid_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
    ctx.close()
    return buf.reshape(16384, 8).astype(np.int64)

t = run(["-DICPK_GRID_STAMPS"], "gt")
c = run(["-DICPK_GRID_STAMPS", "-DICPK_GRID_COUNTS"], "gc")
ok = t[:, 0] > 0
tot = (t[ok, 4] - t[ok, 0]) * 10.0
end = (t[ok, 4] - t[ok, 0].min()) * 10.0
rows, cand = c[ok, 5], c[ok, 6]
print("waves", ok.sum(), "corr(total time, candidates) =", np.corrcoef(tot, cand)[0, 1], " corr(total, rows) =", np.corrcoef(tot, rows)[0, 1])
order = np.argsort(-tot)
for frac in (0.001, 0.01, 0.1, 1.0):
    k = max(1, int(frac * len(order)))
    sel = order[:k]
    print(f"slowest {frac*100:5.1f}%: time mean {tot[sel].mean():7.0f} ns  rows/wave {rows[sel].mean():7.1f}  candidates/wave {cand[sel].mean():8.1f}  end mean {end[sel].mean():7.0f}")
late = np.argsort(-end)[: max(1, len(end) // 100)]
print("last 1% to finish: time mean", tot[late].mean(), "rows", rows[late].mean(), "cand", cand[late].mean(), "start mean", (end[late] - tot[late]).mean())
