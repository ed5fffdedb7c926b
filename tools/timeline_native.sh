#!/bin/bash
# Diagnostic: ordered device timeline (kernels + memory copies) of tests/cpp/tracker_bench.cpp on the frames of tools/bench_tracker.py
cd /tmp && export TMPDIR=/tmp
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from icp_slam_prototype_amd import synth
rng = np.random.default_rng(0)
with open("/tmp/frames.u16", "wb") as f:
    for k in range(6):
        d = synth.render_room_depth(480, 640, synth.rot_xyz_deg(0, 0.5 * k, 0), np.array([0.01 * k, 0, 0]), noise_sigma=0.002, rng=rng)
        d[rng.random(d.shape) > 0.3] = 0
        f.write(d.astype(np.uint16).tobytes())
PY
EXE=$GRAFT_REPO_ROOT/icp_slam_prototype_amd/lib/tracker_bench
$EXE /tmp/frames.u16 480 640 6 8 0 1
rm -rf /tmp/tl_nat
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_nat -o t -- $EXE /tmp/frames.u16 480 640 6 2 0 1
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/tl_nat/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void icpk::", "")[:48]))
for f in glob.glob("/tmp/tl_nat/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
ev.sort()
tail = ev[-70:]
t0 = tail[0][0]
prev_end = t0
for s, e, n in tail:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:7.2f}  {n}")
    prev_end = max(prev_end, e)
PY
