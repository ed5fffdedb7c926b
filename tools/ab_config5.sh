for ppc in 6 8 10 12; do ICPK_GRID_PPC=$ppc python tools/ab_single.py config5 | sed "s/^/PPC=$ppc /"; done
for xd in 2 4 6 8; do ICPK_GRID_XDIV=$xd python tools/ab_single.py config5 | sed "s/^/XDIV=$xd /"; done
for sl in 2 4 8; do ICPK_GRID_SLICES=$sl python tools/ab_single.py config5 | sed "s/^/SLICES=$sl /"; done
