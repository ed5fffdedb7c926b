"""Diagnostic: cost of the throttled enqueue (LoopState::progress) when the loop never exits early --
threshold 0, so every one of max_iterations runs -- for ICPK_LOOP_AHEAD = 0 (everything up front) .. 6,
on a small, the config-2 and a dense cloud.  us per iteration; the throttle must not starve the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_slam_prototype_amd import binding, synth

cases = {
    "1k": synth.frustum_pair(1000, seed=1, rot_deg=(0, 0.5, 0), shift=(0.002, 0, 0)),
    "config2": synth.kinect_pair(seed=2),
    "dense307k": synth.kinect_pair(valid=1.0, seed=3),
}
for ahead in (0, 1, 2, 3, 4, 6):
    os.environ["ICPK_LOOP_AHEAD"] = str(ahead)
    row = []
    with binding.Context(0) as c:
        for name, p in cases.items():
            c.set_target(p["target"])
            c.set_source(p["source"])
            best = 1e9
            for rep in range(12):
                c.reset_source()
                t0 = time.perf_counter()
                T, st, rc = c.align(max_iterations=30, threshold=0.0)
                dt = time.perf_counter() - t0
                assert st.iterations == 30
                if rep >= 2:
                    best = min(best, dt)
            row.append(f"{name} {best / 30 * 1e6:7.1f}")
    print(f"ahead {ahead}: " + "   ".join(row) + "   us/iteration (threshold mode, no exit)", flush=True)
