#!/bin/bash
# Diagnostic: idle time between the dependent kernels of one steady iteration (rocprofv3 kernel trace of
# tools/one_align.py): start/end stamps of consecutive dispatches on the alignment's stream.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg
rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python3 $GRAFT_REPO_ROOT/tools/one_align.py --reps 3 $ONE_ALIGN_ARGS > /tmp/tg.log 2>&1
f=$(find /tmp/tg -name '*kernel_trace.csv' | head -n 1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("nn_grid_batch", "assoc_reduce_batch", "loop_step_batch", "nn_grid_kernel", "assoc_reduce", "loop_step", "copyBuffer", "grid_qslot", "grid_scan_sums", "grid_scan_apply", "grid_qscatter", "fill"):
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
# one whole alignment (the last one): offsets from its first dispatch
last = [k for k, r in enumerate(rows) if "copyBuffer" in r["Kernel_Name"] or "Memcpy" in r["Kernel_Name"]]
start = None
for k in range(len(rows) - 1, -1, -1):
    if "grid_qslot" in rows[k]["Kernel_Name"]:
        start = k - 1
        break
if start is not None:
    t0 = int(rows[start]["Start_Timestamp"])
    print("timeline of the last alignment (us from its first dispatch):")
    for r in rows[start:start + 12] + rows[-6:]:
        print(f'  {(int(r["Start_Timestamp"]) - t0) / 1e3:8.2f} +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:6.2f}  {short(r["Kernel_Name"])}')
    print(f'  total {(int(rows[-1]["End_Timestamp"]) - t0) / 1e3:.1f} us, {len(rows) - start} dispatches')
for k, v in durs.items():
    v.sort(); print(f"dur {k:18s} n={len(v):4d} p50 {v[len(v)//2]/1e3:7.2f} us")
PY
