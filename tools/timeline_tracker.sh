#!/bin/bash
# Diagnostic: ordered device timeline (kernels + memory copies) of the last frame pairs of tools/bench_tracker.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_trk
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_trk -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_tracker.py "$@" > /tmp/tl_trk.out 2>/tmp/tl_trk.err
cat /tmp/tl_trk.out
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/tl_trk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void icpk::", "")[:48]))
for f in glob.glob("/tmp/tl_trk/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
ev.sort()
# last pair: find the last two 'bp_count' style kernels; simply print the last 80 events with gaps
tail = ev[-90:]
t0 = tail[0][0]
prev_end = t0
for s, e, n in tail:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:7.2f}  {n}")
    prev_end = max(prev_end, e)
PY
