#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command, headline part only (--no-extras: the
# frame_batch / dense / config-5 blocks would mix their kernels into the same names); the judged
# copies go to profiles/<tag>_bench_kernel_stats.csv and <tag>_bench_under_rocprof.json.
# A second pass profiles the N = 1 frame_batch block alone through tools/one_align.py --batch.
cd /tmp && export TMPDIR=/tmp
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new
mkdir -p $out
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $out/${tag}_bench_under_rocprof.json 2> /tmp/prof_bench.err
f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -n 1)
cp $f $out/${tag}_bench_kernel_stats.csv
head -n 6 $out/${tag}_bench_kernel_stats.csv | cut -c1-160
