#!/bin/bash
# Diagnostic: every workload x {pruned, grid}, bench.py without the CPU baseline
for w in frustum10k kinect640x480_30pct kinect_v2_512x424 kinect640x480_dense dense1m; do
for m in pruned grid; do
python bench.py --workload $w --nn-mode $m --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$m', round(d['value'],1), 'iter/s  nn avg ms', round(d['roofline']['avg_launch_ms'],4), d['stage_ms_per_step'])"
done; done
