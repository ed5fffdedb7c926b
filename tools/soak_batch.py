"""Soak test (tools only): random batches through icpk_align_batch against the same pairs one by one
through icpk_align, bit for bit (cases: tests/soak_cases.py).
usage: python tools/soak_batch.py [n_batches] [seed0] [--device]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEVICE = "--device" in sys.argv  # pairs resident in HBM through icpk_align_batch_device (no associations read back)
if DEVICE:
    sys.argv.remove("--device")
    import torch  # first: its HIP runtime must be the one libicpk.so binds to
    torch.cuda.init()
import soak_cases

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
done, bad = soak_cases.soak_batch(n_batches, seed0, device=DEVICE, log=lambda m: print(m, flush=True))
print("done:", n_batches, "batches,", done, "pairs,", bad, "mismatches")
sys.exit(1 if bad else 0)
