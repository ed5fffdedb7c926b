"""Soak test (tools only): random batches through icpk_align_batch against the same pairs one by one
through icpk_align, bit for bit (transform, statistics, associations).  Random group sizes, ragged
pair sizes (1 ... 30k points), both lock-step flavours, threshold exits, far-apart pairs (fallback),
empty sources.  usage: python tools/soak_batch.py [n_batches] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEVICE = "--device" in sys.argv  # pairs resident in HBM through icpk_align_batch_device (no associations read back)
if DEVICE:
    sys.argv.remove("--device")
    import torch  # first: its HIP runtime must be the one libicpk.so binds to
    torch.cuda.init()
import numpy as np
from icp_slam_prototype_amd import binding, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import _fuzz_cloud

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
kinds = ["uniform", "clusters", "line", "plane_lattice", "duplicates", "tiny", "huge"]
single = binding.Context(0)
t0 = time.time()
bad = pairs_done = 0
ctxs = {}
for c in range(n_batches):
    rng = np.random.default_rng(seed0 + c)
    g = int(rng.choice([1, 2, 3, 5, 8, 16]))
    if g not in ctxs:
        os.environ["ICPK_BATCH_GROUP"] = str(g)
        ctxs[g] = binding.Context(0)
    ctx = ctxs[g]
    n = int(rng.integers(1, 3 * g + 2))
    pairs = []
    for k in range(n):
        u = rng.random()
        if u < 0.45:
            p = synth.kinect_pair(rows=int(rng.integers(20, 160)), cols=int(rng.integers(30, 200)), valid=float(rng.uniform(0.2, 1.0)),
                                  seed=int(rng.integers(0, 1 << 30)), rot_deg=tuple(rng.uniform(-3, 3, 3)),
                                  shift=tuple(rng.uniform(-0.05, 0.05, 3)))
            s, t = p["source"], p["target"]
        elif u < 0.9:
            nt, nq = int(rng.integers(1, 20000)), int(rng.integers(1, 15000))
            scale = float(10.0 ** rng.uniform(-2, 1))
            off = rng.uniform(-10, 10, (3, 1))
            t = (_fuzz_cloud(rng, nt, kinds[rng.integers(0, 7)]) * scale + off).astype(np.float32)
            s = (_fuzz_cloud(rng, nq, kinds[rng.integers(0, 7)]) * scale + off + rng.normal(0, 0.01 * scale, (3, 1))).astype(np.float32)
        elif u < 0.95:  # far apart: < 3 pairs within reach -> fallback
            q = synth.frustum_pair(int(rng.integers(3, 500)), seed=int(rng.integers(0, 1 << 30)))
            s, t = q["source"] + np.float32(100), q["target"]
        else:
            q = synth.frustum_pair(int(rng.integers(3, 500)), seed=int(rng.integers(0, 1 << 30)))
            s, t = np.zeros((3, 0), np.float32), q["target"]
        pairs.append((np.ascontiguousarray(s, np.float32), np.ascontiguousarray(t, np.float32)))
    kw = dict(solve=int(rng.integers(0, 2)), max_iterations=int(rng.integers(0, 12)), fixed_iterations=int(rng.random() < 0.5),
              max_nn_dist=float(rng.choice([0.75, 0.1, 0.02])), last_translation=rng.normal(0, 0.1, 3).astype(np.float32))
    if DEVICE:
        keep = [(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()) for s, t in pairs]
        torch.cuda.synchronize()
        args = [(a.data_ptr(), a.shape[1], b_.data_ptr(), b_.shape[1]) for a, b_ in keep]
        T, st, rc = ctx.align_batch_device(args, binding.default_params(**kw))
        assoc = None
    else:
        T, st, rc, assoc = ctx.align_batch(pairs, associations=True, **kw)
    for b, (s, t) in enumerate(pairs):
        single.set_target(t)
        single.set_source(s)
        Ts, sts, rcs = single.align(**kw)
        ok = np.array_equal(T[b].view(np.uint32), Ts.view(np.uint32)) and (st[b].iterations, st[b].status, st[b].final_pairs) == (
            sts.iterations, sts.status, sts.final_pairs)
        if s.shape[1] > 0 and assoc is not None:
            i1, d1 = single.get_associations()
            ok = ok and np.array_equal(assoc[b][0], i1) and np.array_equal(assoc[b][1].view(np.uint32), d1.view(np.uint32))
        if not ok:
            bad += 1
            print("MISMATCH batch", c, "pair", b, "group", g, s.shape, t.shape, kw, flush=True)
        pairs_done += 1
    if c % 10 == 9:
        print(f"{c + 1} batches, {pairs_done} pairs, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("done:", n_batches, "batches,", pairs_done, "pairs,", bad, "mismatches")
sys.exit(1 if bad else 0)
