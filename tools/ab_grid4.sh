#!/bin/bash
set -e
export ICPK_AB_MODE=3 ICPK_GRID_SLICES=8
python tools/ab_variant.py | tail -n 1
python tools/ab_variant.py -DICPK_GRID_NOEVAL | tail -n 1
python tools/ab_variant.py -DICPK_GRID_NOLOAD | tail -n 1
python tools/ab_variant.py -DICPK_GRID_NOSCAN | tail -n 1
