#!/bin/bash
# Diagnostic: idle time between the dependent kernels of one steady iteration (rocprofv3 kernel trace of
# tools/one_align.py): start/end stamps of consecutive dispatches on the alignment's stream.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg
rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python3 $GRAFT_REPO_ROOT/tools/one_align.py --reps 3 > /tmp/tg.log 2>&1
f=$(find /tmp/tg -name '*kernel_trace.csv' | head -n 1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("nn_grid_kernel", "assoc_reduce", "loop_step", "copyBuffer", "grid_qslot", "grid_scan_sums", "grid_scan_apply", "grid_qscatter", "fill"):
        if k in n: return k
    return n[:24]
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 20000:
        gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(g)
    durs[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:12]:
    v.sort(); print(f"{k[0]:18s} -> {k[1]:18s} n={len(v):4d}  gap p50 {v[len(v)//2]/1e3:6.2f} us  p10 {v[len(v)//10]/1e3:6.2f}  p90 {v[len(v)*9//10]/1e3:6.2f}")
for k, v in durs.items():
    v.sort(); print(f"dur {k:18s} n={len(v):4d} p50 {v[len(v)//2]/1e3:7.2f} us")
PY
