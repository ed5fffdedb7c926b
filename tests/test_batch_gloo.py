"""world_size-2 gloo test (CPU) of the frame-batch mode's sharding and
collectives.  The GPU alignment is replaced by the oracle AS A TEST STAND-IN
(tests may call the oracle); on a GPU box bench.py injects the C-ABI context."""
import os
import socket
import sys

import numpy as np
import pytest

from icp_slam_prototype_amd import batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_everything():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = batch.partition(n, world, r)
                seen += list(range(s, s + c))
            assert seen == list(range(n))
            counts = [batch.partition(n, world, r)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1
    assert batch.partition(64, 8, 3) == (24, 8)  # BASELINE config 4: 64 pairs, 8 per GPU
    with pytest.raises(ValueError):
        batch.partition(4, 2, 2)


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from icp_slam_prototype_amd import batch, synth
    from oracle import icp_oracle as o

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cpu")
    base = synth.frustum_pair(600, seed=1)
    target = base["target"] if rank == 0 else None

    def make_source(i):
        return synth.frustum_pair(600, seed=1, rot_deg=(0, 0.5 + 0.1 * i, 0), shift=(0.002 * i, 0, 0))["source"]

    def align_fn(src, tgt_tensor):
        r = o.align(src, tgt_tensor.numpy(), max_iterations=3, solve=1, sum_order=1, fixed_iterations=True)
        return r["T"], r["iterations"], r["status"], r["final_pairs"], r["final_mse"]

    T, S = batch.align_frame_batch(make_source, n_frames, target, align_fn, dev, dist)
    q.put((rank, T, S))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 4])
def test_frame_batch_two_ranks_matches_serial(n_frames, oracle):
    import torch.multiprocessing as mp

    from icp_slam_prototype_amd import synth

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    target = synth.frustum_pair(600, seed=1)["target"]
    for rank, T, S in res:
        assert T.shape == (n_frames, 4, 4)
        for i in range(n_frames):
            src = synth.frustum_pair(600, seed=1, rot_deg=(0, 0.5 + 0.1 * i, 0), shift=(0.002 * i, 0, 0))["source"]
            r = oracle.align(src, target, max_iterations=3, solve=1, sum_order=1, fixed_iterations=True)
            assert np.array_equal(T[i], r["T"]), (rank, i)
            assert S[i, 0] == 3 and S[i, 1] == 0 and S[i, 2] == r["final_pairs"]
    assert np.array_equal(res[0][1], res[1][1])  # every rank holds the same gathered result


class _OracleSteps:
    """CPU stand-in for batch.ContextSteps (test only): the oracle does the three steps."""

    def __init__(self, oracle, src, tgt):
        self.o, self.src, self.tgt = oracle, src.copy(), tgt
        self.idx = self.dist = None

    def nn(self):
        self.idx, self.dist = self.o.nn_bruteforce(self.src, self.tgt)

    def reduce(self, max_dist):
        return self.o.sums_canonical(self.src, self.tgt, self.idx, self.dist, max_dist)

    def transform(self, R, t):
        self.src = self.o.transform_points(self.src, R, t)


def _sharded_worker(rank, world, port, solve, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from icp_slam_prototype_amd import batch, build, synth
    from oracle import icp_oracle as o

    build.build()  # the host solve comes from libicpk.so (no device work)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    p = synth.frustum_pair(1500, seed=6, rot_deg=(0, 2, 0), shift=(0.01, 0, 0))
    tgt = batch.broadcast_cloud(p["target"] + np.float32(5) if rank == 0 else None, 0, torch.device("cpu"), dist).numpy()
    s, c = batch.partition(1500, world, rank)
    steps = _OracleSteps(o, np.ascontiguousarray((p["source"] + np.float32(5))[:, s:s + c]), tgt)
    T, it, n, mse = batch.align_query_sharded(steps, dist, torch.device("cpu"), max_iterations=6, solve=solve,
                                              fixed_iterations=True)
    q.put((rank, T, it, n, float(mse)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("solve", [0, 1])
def test_query_sharded_alignment_two_ranks(solve, oracle):
    """One pair, queries split over 2 ranks, one all-reduce of the 19 sums per iteration:
    same transform as the single-process loop up to the summation order across ranks."""
    import torch.multiprocessing as mp

    from icp_slam_prototype_amd import synth

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, solve, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pr = synth.frustum_pair(1500, seed=6, rot_deg=(0, 2, 0), shift=(0.01, 0, 0))
    ref = oracle.align(pr["source"] + np.float32(5), pr["target"] + np.float32(5), max_iterations=6, solve=solve,
                       sum_order=1, fixed_iterations=True)
    assert np.array_equal(res[0][1], res[1][1])  # replicated solve: every rank holds the same T
    assert res[0][2] == ref["iterations"] == 6 and res[0][3] == ref["final_pairs"]
    assert np.linalg.norm(res[0][1].astype(np.float64) - ref["T"].astype(np.float64)) < 1e-5
