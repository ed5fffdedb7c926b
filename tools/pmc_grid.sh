#!/bin/bash
# Diagnostic: instruction-mix counters of the grid sweep on config 2 (one --pmc pass per group)
cd /tmp && export TMPDIR=/tmp
export ICPK_AB_MODE=3 ICPK_GRID_SLICES=${1:-8}
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_grid_$i -o p -- python3 $GRAFT_REPO_ROOT/tools/one_align.py > /tmp/pmc_grid_$i.log 2>&1
  echo "pass $i done"
  f=$(find /tmp/pmc_grid_$i -name '*counter_collection.csv' | head -n 1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "nn_grid_kernel" in r["Kernel_Name"] and "false" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in acc.items():
    print(f"{k}: dispatches {n} avg/dispatch {v / n:.1f}")
PY
done
