#!/bin/bash
# Diagnostic: tools/microbench_traffic.hip under rocprofv3 --pmc, one pass per counter group: what the memory-side counters
# report for a KNOWN number of bytes in the sweep's access shapes.  Output: per kernel and counter, the second dispatch's value.
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/microbench_traffic $GRAFT_REPO_ROOT/tools/microbench_traffic.hip || exit 1
/tmp/microbench_traffic
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_BUBBLE_sum" "TCC_REQ_sum TCC_READ_REQ_sum TCC_WRITE_REQ_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_t_$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_t_$i -o p -- /tmp/microbench_traffic > /tmp/pmc_t_$i.log 2>&1
  echo "pass $i ($grp) rc $?"
  f=$(find /tmp/pmc_t_$i -name '*counter_collection.csv' | head -n 1)
  python3 - "$f" <<'PY'
import csv, sys, collections
last = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    last[(k, r["Counter_Name"])] = float(r["Counter_Value"])  # later dispatches overwrite: the warm one remains
for (k, c), v in last.items():
    print(f"  {k:28s} {c:28s} {v:16.1f}")
PY
done
