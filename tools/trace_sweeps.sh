#!/bin/bash
# Diagnostic: the durations of the NN sweeps of the last alignment of tools/one_align.py, in order
# usage: bash tools/trace_sweeps.sh [one_align arguments]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts
rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o p -- python3 $GRAFT_REPO_ROOT/tools/one_align.py "$@" > /tmp/ts.log 2>&1
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/ts/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void icpk::", "")))
ev.sort()
nn = [(e - s) / 1e3 for s, e, n in ev if "nn_grid" in n]
k = [n for s, e, n in ev if "nn_grid" in n]
per = 21
last = nn[-per:]
print("sweeps of the last alignment (us):", " ".join(f"{v:.1f}" for v in last))
print("kernels:", k[-per][:40], "...", k[-1][:40])
PY
