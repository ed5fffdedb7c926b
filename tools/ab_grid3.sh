#!/bin/bash
set -e
export ICPK_AB_MODE=3
for gs in 4 8; do for ppc in 3 6; do
export ICPK_GRID_SLICES=$gs ICPK_GRID_PPC=$ppc
python tools/ab_variant.py | tail -n 1
done; done
