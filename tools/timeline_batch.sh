#!/bin/bash
# Diagnostic: ordered device timeline of the last icpk_align_batch_device call of tools/bench_batch.py <pairs>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_b
BGROUPS=${2:-8} REPS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_b -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_batch.py ${1:-8} > /tmp/tl_b.out 2>/tmp/tl_b.err
cat /tmp/tl_b.out
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/tl_b/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void icpk::", "")[:52]))
for f in glob.glob("/tmp/tl_b/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# the last call: walk back from the end to the last 'ingest' kernel run
idx = max(i for i, e in enumerate(ev) if "ingest" in e[2])
while idx > 0 and "ingest" in ev[idx - 1][2]:
    idx -= 1
tail = ev[idx:]
t0 = tail[0][0]
prev_end = t0
shown = 0
for s, e, n in tail:
    if shown < 40 or "step" in n and shown < 60:
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:7.2f}  {n}")
        shown += 1
    prev_end = max(prev_end, e)
print(f"... total {(tail[-1][1] - t0) / 1e3:.1f} us, {len(tail)} events")
PY
