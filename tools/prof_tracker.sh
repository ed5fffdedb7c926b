#!/bin/bash
# rocprofv3 kernel trace of the drop-in frame path (tools/bench_tracker.py): which kernels a frame pair costs
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_trk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trk -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_tracker.py > /tmp/prof_trk.out 2>/tmp/prof_trk.err
cat /tmp/prof_trk.out
f=$(find /tmp/prof_trk -name '*kernel_stats.csv' | head -n 1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"GPU kernel time per frame pair (25 pairs): {tot / 25 / 1e3:.1f} us")
for r in rows[:24]:
    print(f'{r["Name"][:70]:70s} calls/pair {int(r["Calls"]) / 25:6.1f}  avg {float(r["AverageNs"]) / 1e3:7.2f} us  per pair {float(r["TotalDurationNs"]) / 25 / 1e3:7.1f} us')
PY
