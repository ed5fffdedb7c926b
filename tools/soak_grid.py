"""Soak test (tools only): many random clouds, grid scan vs exact kernel, bit for bit, over first and
seeded sweeps with the source moving in between (cases: tests/soak_cases.py).
usage: python tools/soak_grid.py [n_cases] [seed0] [oracle]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from icp_slam_prototype_amd import binding
import soak_cases

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
orc = None
if len(sys.argv) > 3 and sys.argv[3] == "oracle":  # compare with the CPU oracle instead of the exact kernel
    from oracle import icp_oracle as orc
done, bad = soak_cases.soak_grid(binding.Context(0), n_cases, seed0, oracle=orc, log=lambda m: print(m, flush=True))
print("done:", done, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
