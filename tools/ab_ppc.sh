for ppc in 5 6.5 8 10 13; do ICPK_GRID_PPC=$ppc python tools/ab_single.py config2 | sed "s/^/PPC=$ppc /"; done
for xd in 3 4 6 8; do ICPK_GRID_XDIV=$xd python tools/ab_single.py config2 | sed "s/^/XDIV=$xd /"; done
for ppc in 6 8 10; do ICPK_GRID_PPC=$ppc BGROUPS=16 python tools/bench_batch.py 32 2>&1 | grep group | sed "s/^/PPC=$ppc /"; done
for xd in 4 6; do ICPK_GRID_XDIV=$xd BGROUPS=16 python tools/bench_batch.py 32 2>&1 | grep group | sed "s/^/XDIV=$xd /"; done
