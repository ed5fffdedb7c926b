#!/bin/bash
# tools only: steady-state duration of the pruned NN kernel vs. target-chunk count
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for tb in 400 1500 4096 16384 65536; do
  ICPK_NN_Q=1 ICPK_NN_TARGET_BLOCKS=$tb rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pc_$tb -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
done
