#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command (judged copy goes to profiles/)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_under_rocprof.json 2> /tmp/prof_bench.err
cp /tmp/prof_bench/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/bench_kernel_stats.csv
head -n 6 $GRAFT_REPO_ROOT/gpurun_out/bench_kernel_stats.csv | cut -c1-200
