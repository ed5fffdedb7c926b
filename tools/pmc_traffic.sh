#!/bin/bash
# HBM traffic of the NN kernels (separate --pmc passes, one counter each), summary under gpurun_out/
cd /tmp && export TMPDIR=/tmp
export ICPK_AB_MODE=${1:-3}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic_mode$ICPK_AB_MODE.csv
echo "kernel,counter,dispatches,avg_value_KB_per_dispatch" > $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_tr_$c -o p -- python3 $GRAFT_REPO_ROOT/tools/one_align.py > /tmp/pmc_tr_$c.log 2>&1
  echo "pass $c rc=$?"
  f=$(find /tmp/pmc_tr_$c -name '*counter_collection.csv' | head -n 1)
  python3 - "$f" "$out" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0]
    a = acc[(name, r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open(sys.argv[2], "a") as f:
    for (k, c), (n, v) in sorted(acc.items()):
        f.write(f"\"{k}\",{c},{n},{v / n:.3f}\n")
        if "nn_" in k: print(k, c, n, round(v / n, 1))
PY
done
