#!/bin/bash
set -e
export ICPK_AB_MODE=3 ICPK_GRID_SLICES=8
for gb in 4 6 8; do python tools/ab_variant.py -DICPK_GRID_GB=$gb | tail -n 1; done
export ICPK_GRID_SLICES=4
for gb in 4 8 12; do python tools/ab_variant.py -DICPK_GRID_GB=$gb | tail -n 1; done
