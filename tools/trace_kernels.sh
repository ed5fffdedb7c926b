#!/bin/bash
# Diagnostic: rocprofv3 kernel-trace summary (per-kernel calls / average / min / max duration) of tools/one_align.py
# usage: bash tools/trace_kernels.sh <tag> [one_align arguments]
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
rm -rf /tmp/tk_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/one_align.py "$@" > /tmp/tk_$tag.log 2>&1
f=$(find /tmp/tk_$tag -name '*kernel_stats.csv' | head -n 1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/profiles_new
cp $f $GRAFT_REPO_ROOT/gpurun_out/profiles_new/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void icpk::", "")
    print(f"{n:48s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.2f} us min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f} total% {float(r['Percentage']):5.1f}")
PY
