#!/bin/bash
set -e
export ICPK_AB_MODE=3
for ppc in 3 4 6 8 12; do
export ICPK_GRID_PPC=$ppc
python tools/ab_variant.py | tail -n 1
done
