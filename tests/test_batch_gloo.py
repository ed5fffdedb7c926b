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


def _pair_worker(rank, world, port, n_pairs, q):
    """BASELINE config 4's host logic: distinct pairs, block partition, gathered results."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from icp_slam_prototype_amd import batch, synth
    from oracle import icp_oracle as o

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = batch.TorchComm(dist, torch.device("cpu"))
    made = []

    def make_pair(i):
        made.append(i)
        p = synth.frustum_pair(400 + 10 * i, seed=100 + i, rot_deg=(0, 1, 0), shift=(0.01, 0, 0))
        return p["source"], p["target"]

    def align_batch_fn(pairs):  # CPU stand-in for Context.align_batch (test only)
        T = np.zeros((len(pairs), 4, 4), np.float32)
        S = np.zeros((len(pairs), 4), np.float32)
        for k, (s, t) in enumerate(pairs):
            r = o.align(s, t, max_iterations=3, solve=0, sum_order=1, fixed_iterations=True)
            T[k] = r["T"]
            S[k] = (r["iterations"], r["status"], r["final_pairs"], r["final_mse"])
        return T, S

    T, S = batch.align_pair_batch(n_pairs, make_pair, align_batch_fn, comm)
    q.put((rank, T, S, made))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [6, 5, 1])
def test_pair_batch_two_ranks_matches_serial(n_pairs, oracle):
    import torch.multiprocessing as mp

    from icp_slam_prototype_amd import synth

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pair_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every pair was built and aligned by exactly one rank, in block order
    assert res[0][3] + res[1][3] == list(range(n_pairs))
    assert res[0][3] == list(range(*[(a, a + c) for a, c in [batch.partition(n_pairs, 2, 0)]][0]))
    for i in range(n_pairs):
        p = synth.frustum_pair(400 + 10 * i, seed=100 + i, rot_deg=(0, 1, 0), shift=(0.01, 0, 0))
        r = oracle.align(p["source"], p["target"], max_iterations=3, solve=0, sum_order=1, fixed_iterations=True)
        for rank, T, S, _ in res:
            assert np.array_equal(T[i], r["T"]) and S[i, 0] == 3 and S[i, 2] == r["final_pairs"]


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
    T, it, n, mse, status = batch.align_query_sharded(steps, dist, torch.device("cpu"), max_iterations=6, solve=solve,
                                                      fixed_iterations=True)
    assert status == 0
    q.put((rank, T, it, n, float(mse)))
    dist.barrier()
    dist.destroy_process_group()


def test_query_sharded_too_few_pairs_fallback(oracle):
    """icp.cpp:163-182 in the sharded loop (single rank, no process group needed): fewer than 3
    associations -> the caller's last motion is applied, offset = -last_translation, status 1,
    exactly like icpk_align / the oracle's loop."""
    from icp_slam_prototype_amd import build, synth

    build.build()

    class _Self:  # world-size-1 transport: the sums are already global
        rank, world = 0, 1

        def allreduce_sums(self, sums, count):
            return np.asarray(sums, np.float64), int(count)

    p = synth.frustum_pair(300, seed=5)
    far = p["source"] + np.float32(100)
    far[:, :2] = p["target"][:, :2] + np.float32(0.05)
    lt = np.array([1, 2, 3], np.float32)
    lr = oracle.make_rotation_matrix(0, 3, 0)
    steps = _OracleSteps(oracle, far, p["target"])
    T, it, n, mse, status = batch.align_query_sharded(steps, _Self(), solve=0, last_rotation=lr, last_translation=lt)
    o = oracle.align(far, p["target"], solve=0, sum_order=1, last_rotation=lr, last_translation=lt)
    assert status == 1 == o["status"] and it == 0 == o["iterations"] and n == 2 == o["final_pairs"]
    assert np.array_equal(T, o["T"]) and np.array_equal(T[:3, 3], -lt)
    assert np.array_equal(steps.src, o["src_out"])


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
