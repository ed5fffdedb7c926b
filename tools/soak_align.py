"""Soak test (tools only): whole alignments, grid scan + device loop vs exact kernel + host loop,
bit for bit (transform, pair count, associations, moved source; cases: tests/soak_cases.py).
usage: soak_align.py [n] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from icp_slam_prototype_amd import binding
import soak_cases

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
done, bad = soak_cases.soak_align(binding.Context(0), n_cases, seed0, log=lambda m: print(m, flush=True))
print("done:", done, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
