#!/bin/bash
set -e
export ICPK_AB_MODE=3 ICPK_GRID_PPC=6
for gs in 4 8; do
export ICPK_GRID_SLICES=$gs
python tools/ab_variant.py | tail -n 1
python tools/ab_variant.py -DICPK_GRID_BLOCK=256 | tail -n 1
python tools/ab_variant.py -DICPK_GRID_NOSCAN | tail -n 1
python tools/ab_variant.py -DICPK_GRID_NOSCAN -DICPK_GRID_BLOCK=256 | tail -n 1
done
