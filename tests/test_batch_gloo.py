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
