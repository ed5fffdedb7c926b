#!/bin/bash
# Diagnostic: every workload x {pruned, grid}, bench.py without the CPU baseline
for w in frustum10k kinect640x480_30pct kinect_v2_512x424 kinect640x480_dense dense1m; do
for m in pruned grid; do
python bench.py --workload $w --nn-mode $m --no-cpu-baseline --no-extras --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$m', round(d['value'],1), 'iter/s  nn avg ms', round(d['roofline']['avg_launch_ms'],4), 'Mpts/s', round(d['nn_mpoints_per_s'],1), 'pcie', round(d['pcie_inclusive_iter_s'],1))"
done; done
python bench.py --workload kinect_v2_512x424 --solve p2l --no-cpu-baseline --no-extras --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kinect_v2_512x424 p2l grid', round(d['value'],1), 'iter/s')"
python bench.py --workload dense1m --iters 50 --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense1m 50 iters grid', round(d['value'],1), 'iter/s')"
for m in filtered exact; do python bench.py --nn-mode $m --no-cpu-baseline --no-extras --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 $m', round(d['value'],1), 'iter/s')"; done
