#!/bin/bash
# Diagnostic: grid scan parameter sweep on config 2 (uses the in-tree library)
set -e
python tools/ab_variant.py | tail -n 1
for gs in 1 2 4 8; do for ppc in 4 8 16; do
ICPK_AB_MODE=3 ICPK_GRID_SLICES=$gs ICPK_GRID_PPC=$ppc python tools/ab_variant.py | tail -n 1
done; done
