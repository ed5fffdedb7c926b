#!/bin/bash
set -e
export ICPK_AB_MODE=3
python tools/ab_variant.py | tail -n 1
python tools/ab_variant.py -DICPK_GRID_WPE_STEADY=8 | tail -n 1
python tools/ab_variant.py -DICPK_GRID_WPE_EXPAND=6 | tail -n 1
python tools/ab_variant.py -DICPK_GRID_WPE_EXPAND=7 | tail -n 1
