#!/bin/bash
# Diagnostic: kernel-trace statistics of the grid sweep on config 2
cd /tmp && export TMPDIR=/tmp
export ICPK_AB_MODE=3 ICPK_GRID_SLICES=${1:-8} ICPK_GRID_PPC=${2:-6}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_grid -o g -- python3 $GRAFT_REPO_ROOT/tools/ab_variant.py > /tmp/prof_grid.log 2>&1
tail -n 2 /tmp/prof_grid.log
find /tmp/prof_grid -type f | head -n 20
f=$(find /tmp/prof_grid -name '*kernel_stats*' | head -n 1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5:
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
