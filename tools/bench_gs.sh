#!/bin/bash
# Diagnostic: lanes per query of the grid scan on the larger workloads
for w in frustum10k kinect_v2_512x424 kinect640x480_dense dense1m; do
for gs in 1 2 4 8; do
ICPK_GRID_SLICES=$gs python bench.py --workload $w --no-cpu-baseline --no-extras --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', 'gs=$gs', round(d['value'],1), 'iter/s  nn avg ms', round(d['roofline']['avg_launch_ms'],4))"
done; done
