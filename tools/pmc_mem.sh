#!/bin/bash
# Diagnostic: memory-side counters of the steady grid sweep, ONE counter group per rocprofv3 --pmc
# pass, the TA_* counters one per pass (round 1 ran TA_BUSY_avr + TA_ADDR_STALLED_BY_TC_CYCLES_sum +
# TA_FLAT_READ_WAVEFRONTS_sum in one pass, which ended in `timeout 200` once with nothing kept to
# tell why; each of the three collects in ~1 s on its own, tools/pmc_ta_probe.sh).  Every pass's
# log and exit status are kept under gpurun_out/pmc_mem/.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_mem
mkdir -p $out
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "MemUnitStalled TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_mem_$i -o p -- python3 $GRAFT_REPO_ROOT/tools/one_align.py --iters 10 --reps 1 > $out/pass_$i.log 2>&1
  echo "pass $i ($grp) rc=$?" | tee -a $out/summary.txt
  f=$(find /tmp/pmc_mem_$i -name '*counter_collection.csv' | head -n 1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $out/summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "nn_grid_kernel" in r["Kernel_Name"] and "false" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in acc.items():
    print(f"{k}: dispatches {n} avg/dispatch {v / n:.1f}")
PY
done
