#!/bin/bash
set -e
export ICPK_AB_MODE=3
for gs in 4 8; do for ppc in 2 3 4 6 9; do
export ICPK_GRID_SLICES=$gs ICPK_GRID_PPC=$ppc
python tools/ab_variant.py | tail -n 1
done; done
