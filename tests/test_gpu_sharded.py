"""Query-sharded alignment of one pair through the C ABI: two processes share cuda:0 (a
rehearsal of the 2-GPU layout; the collective runs over gloo here, RCCL on a real node),
each holds half of the queries and the whole target."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, solve, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from icp_slam_prototype_amd import batch, binding, synth

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    p = synth.kinect_pair(valid=0.05, seed=4)
    tgt = batch.broadcast_cloud(p["target"] if rank == 0 else None, 0, torch.device("cpu"), dist).numpy()
    s, c = batch.partition(p["source"].shape[1], world, rank)
    ctx = binding.Context(0)
    ctx.set_target(tgt)
    ctx.set_source(np.ascontiguousarray(p["source"][:, s:s + c]))
    T, it, n, mse, status = batch.align_query_sharded(batch.ContextSteps(ctx), dist, torch.device("cpu"),
                                                      max_iterations=8, solve=solve, fixed_iterations=True)
    q.put((rank, T, it, n, float(mse)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("solve", [0, 1])
def test_query_sharded_matches_single_context(solve):
    import torch.multiprocessing as mp

    from icp_slam_prototype_amd import binding, synth

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, solve, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pr = synth.kinect_pair(valid=0.05, seed=4)
    ctx = binding.Context(0)
    ctx.set_target(pr["target"])
    ctx.set_source(pr["source"])
    T, st, _ = ctx.align(max_iterations=8, solve=solve, fixed_iterations=True)
    assert np.array_equal(res[0][1], res[1][1])
    assert res[0][2] == st.iterations == 8 and res[0][3] == st.final_pairs
    # same pairs, same sums up to the order the two halves are added in
    assert np.linalg.norm(res[0][1].astype(np.float64) - T.astype(np.float64)) < 1e-5
