#!/bin/bash
# Long soak of the final build (seeded; every driver stops at its own limit)
echo "== soak_grid 12000"; timeout -k 10 1000 python3 tools/soak_grid.py 12000 301000 | tail -1
echo "== soak_grid vs oracle 600"; timeout -k 10 1000 python3 tools/soak_grid.py 600 302000 oracle | tail -1
echo "== soak_align 8000"; timeout -k 10 1000 python3 tools/soak_align.py 8000 303000 | tail -1
echo "== soak_batch 500"; timeout -k 10 1000 python3 tools/soak_batch.py 500 304000 2>/dev/null | tail -1
echo "== soak_batch device 400"; timeout -k 10 1000 python3 tools/soak_batch.py --device 400 305000 2>/dev/null | tail -1
echo "== soak_pair 2500"; timeout -k 10 1000 python3 tools/soak_pair.py 2500 306000 2>&1 | tail -1
