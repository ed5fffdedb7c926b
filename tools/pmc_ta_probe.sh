#!/bin/bash
# One-off diagnostic: which counter of the group
#   TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
# makes rocprofv3 --pmc never finish on this pool?  One counter per pass, the program directly
# after `--`, every pass's log and exit status kept under gpurun_out/pmc_ta/.  Run ONCE.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_ta
mkdir -p $out
for c in TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr; do
  start=$(date +%s)
  timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_ta_$c -o p -- python3 $GRAFT_REPO_ROOT/tools/one_align.py --iters 4 --reps 1 > $out/$c.log 2>&1
  rc=$?
  echo "$c rc=$rc seconds=$(( $(date +%s) - start )) csv=$(find /tmp/pmc_ta_$c -name '*counter_collection.csv' 2>/dev/null | wc -l)" | tee -a $out/summary.txt
  tail -n 3 $out/$c.log | sed 's/^/    /' >> $out/summary.txt
done
