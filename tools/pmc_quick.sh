#!/bin/bash
# Diagnostic: instruction counters of the steady grid sweep for the current build and for variant builds
# (icp_slam_prototype_amd/lib_variants/<tag>/libicpk.so, made by tools/ab_batch.py --build-only), single pair and a
# lock-step group of 8.  usage (GPU box): bash tools/pmc_quick.sh [variant-tag ...]
cd /tmp && export TMPDIR=/tmp
libs=("")
for t in "$@"; do libs+=("$GRAFT_REPO_ROOT/icp_slam_prototype_amd/lib_variants/$t/libicpk.so"); done
for lib in "${libs[@]}"; do
  for arg in "" "--batch 8"; do
    rm -rf /tmp/pq; ICPK_LIB_PATH=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pq -o p -- python3 $GRAFT_REPO_ROOT/tools/one_align.py $arg > /tmp/pq.log 2>&1
    python3 - "$lib" "$arg" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nn_grid" in r["Kernel_Name"] and "false" in r["Kernel_Name"].split("(")[0]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("lib", sys.argv[1][-40:] or "current", "|", sys.argv[2] or "single", "|", {k: round(v / n) for k, (n, v) in acc.items()})
PY
  done
done
