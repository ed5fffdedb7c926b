"""Diagnostic (tools only): phases and work counters of nn_grid_kernel (-DICPK_GRID_STAMPS)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import build, binding, synth

os.makedirs("/tmp/icpk_gs", exist_ok=True)
binding.LIB_PATH = build.build(force=True, extra=["-DICPK_GRID_STAMPS"], out="/tmp/icpk_gs/libicpk.so")
p = synth.kinect_pair(480, 640, valid=0.30, seed=2)
ctx = binding.Context(0)
ctx.set_target(p["target"]); ctx.set_source(p["source"])
L = binding.load()
NW = 16384
for first in (True, False):
    ctx.reset_source()
    if not first:
        ctx.align(max_iterations=8, fixed_iterations=1, nn_mode=binding.NN_GRID, host_loop=1)
    L.icpk_debug_clear_grid_stamps()
    ctx.nn(binding.NN_GRID, fetch=False)
    buf = np.zeros(8 * NW, np.uint64)
    L.icpk_debug_read_grid_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
    b = buf.reshape(NW, 8).astype(np.int64)
    b = b[b[:, 0] > 0]
    t0 = b[:, 0].min()
    print("first sweep" if first else "steady sweep", "waves", len(b), "kernel span us", (b[:, 4].max() - t0) / 100.0)
    for name, a, c in (("init", 0, 1), ("cube", 1, 2), ("ranges", 2, 5), ("batch1", 5, 6), ("rest", 6, 3),
                       ("scan", 2, 3), ("write", 3, 4), ("total", 0, 4)):
        ok = (b[:, a] > 0) & (b[:, c] > 0)
        v = (b[ok, c] - b[ok, a]) * 10.0
        print(f"  {name:7s} ns: mean {v.mean():8.0f} p50 {np.median(v):8.0f} p90 {np.quantile(v, 0.9):8.0f} max {v.max():8.0f}")
    en = (b[:, 4] - t0) * 10.0
    print("  wave end ns: " + " ".join(f"p{int(q*100)} {np.quantile(en, q):.0f}" for q in (0.5, 0.9, 0.99, 0.999, 1.0)))
    tot = (b[:, 4] - b[:, 0]) * 10.0
    print(f"  busy fraction of {7 * 1024} slots x span: {tot.sum() / (7 * 1024 * en.max()):.2f}")
    dec = np.array_split(np.arange(len(b)), 10)
    print("  wave duration by decile of block index (us):", " ".join(f"{tot[d].mean() / 1e3:.1f}/{tot[d].max() / 1e3:.1f}" for d in dec))
    # list scheduling of the measured wave durations over the wave slots: what a heaviest-first start
    # order could buy (durations taken as fixed, which they are not quite: they include queueing)
    import heapq
    def span(order, slots=7 * 1024):
        free = [0.0] * slots
        heapq.heapify(free)
        end = 0.0
        for w in order:
            t = heapq.heappop(free) + tot[w]
            end = max(end, t)
            heapq.heappush(free, t)
        return end / 1e3
    print(f"  simulated span (us): natural order {span(range(len(tot))):.1f}, heaviest first {span(np.argsort(-tot)):.1f}, "
          f"lightest first {span(np.argsort(tot)):.1f}, lower bound {tot.sum() / (7 * 1024) / 1e3:.1f}")
    st = (b[:, 0] - t0) * 10.0
    print(f"  wave start ns: p50 {np.median(st):.0f} p90 {np.quantile(st, 0.9):.0f} max {st.max():.0f}")

