"""The world > 1 paths of the RCCL layer behind the C ABI (csrc/icpk_comm.cpp) with TWO ranks on the
one GPU of the test box.  Real RCCL refuses two ranks on one device, so the collectives come from
tests/cpp/fake_rccl.cpp (shared memory; test infrastructure, loaded through the ICPK_RCCL_LIB hook):
what is under test is everything of OURS around them -- icpk_comm_init_rccl, the block partition,
staging and row re-ordering of icpk_comm_gather_results, the non-root side of
icpk_comm_broadcast_target (allocation, padding, invalidation of everything derived from the old
target), icpk_comm_allreduce_sums inside the query-sharded loop, and batch.align_pair_batch
(BASELINE config 4's host logic) end to end through the C ABI.  The real library is exercised with
world = 1 in tests/test_gpu_comm.py and with 8 ranks by bench.py on the driver's node."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pairs():
    from icp_slam_prototype_amd import synth

    out = []
    for k in range(5):
        p = synth.kinect_pair(60 + 10 * k, 80 + 10 * k, valid=0.6, seed=900 + k)
        out.append((p["source"], p["target"]))
    return out


def _worker(rank, world, fake, q_id, q_out):
    sys.path.insert(0, ROOT)
    os.environ["ICPK_TEST_HOOKS"] = "1"  # the switch without which icpk_comm_* ignores ICPK_RCCL_LIB
    os.environ["ICPK_RCCL_LIB"] = fake
    from icp_slam_prototype_amd import batch, binding, synth

    def exchange(uid):
        if rank == 0:
            for _ in range(world - 1):
                q_id.put(uid)
            return uid
        return q_id.get(timeout=120)

    ctx = binding.Context(0)
    comm = batch.RcclComm(ctx, rank, world, exchange)
    res = {"rank": ctx.comm_rank, "world": ctx.comm_world}
    # (1) key-frame broadcast: every rank starts with a DIFFERENT target (rank 1's is larger, so its
    # buffers, grid and seeds are all stale afterwards)
    key = synth.kinect_pair(90, 120, valid=0.7, seed=41)
    other = synth.kinect_pair(120, 160, valid=0.9, seed=42 + rank)
    ctx.set_target(key["target"] if rank == 0 else other["target"])
    ctx.set_source(key["source"])
    ctx.nn(binding.NN_GRID, fetch=False)  # builds the grid of the OLD target on the non-root rank
    comm.broadcast_target(0)
    res["bcast_target"] = ctx.get_target()
    res["bcast_nn"] = ctx.nn(binding.NN_GRID)
    # (2) gathered results of a block-partitioned batch: rows tagged with their global pair index
    n_total = 5
    start, count = batch.partition(n_total, world, rank)
    T_local = np.zeros((count, 4, 4), np.float32)
    S_local = np.zeros((count, 4), np.float64)
    for k in range(count):
        T_local[k] = (start + k) * 100 + np.arange(16, dtype=np.float32).reshape(4, 4)
        S_local[k] = (start + k, 0, 100_000_001 + start + k, 0.5 * (start + k))  # (a pair count a float32 cannot hold)
    res["gather"] = comm.gather_results(T_local, S_local, n_total)
    # (3) all-reduce of sums + count
    res["allreduce"] = comm.allreduce_sums(np.arange(19, dtype=np.float64) * (rank + 1), 10 ** 10 * (rank + 1))
    # (4) BASELINE config 4's host logic through the C ABI
    pairs = _pairs()
    res["pair_batch"] = batch.align_pair_batch(
        len(pairs), lambda i: pairs[i],
        lambda ps: (lambda r: (r[0], batch.stats_rows(r[1])))(ctx.align_batch(ps, max_iterations=4, fixed_iterations=1)), comm)
    # (5) query-sharded single pair: one all-reduce of 160 bytes per iteration
    big = synth.kinect_pair(120, 160, valid=0.8, seed=77)
    ctx.set_target(big["target"])  # (same target on every rank; the broadcast was checked above)
    s0, c0 = batch.partition(big["source"].shape[1], world, rank)
    ctx.set_source(np.ascontiguousarray(big["source"][:, s0:s0 + c0]))
    res["sharded"] = batch.align_query_sharded(batch.ContextSteps(ctx), comm, max_iterations=5, solve=1, fixed_iterations=True)
    # (6) the same through the DEVICE-side loop (icpk_align_query_sharded: in-stream all-reduce, no host round trip),
    # both flavours, fixed iterations and threshold exit
    res["sharded_dev"] = []
    for kw in (dict(max_iterations=5, solve=1, fixed_iterations=1), dict(max_iterations=6, solve=0, fixed_iterations=1),
               dict(max_iterations=12, solve=1, threshold=3e-4)):
        res["sharded_dev"].append(comm.align_query_sharded(**kw))
    ctx.reset_source()
    ctx.transform_source(np.eye(3, dtype=np.float32), np.float32([50, 0, 0]))  # out of reach: the < 3 pairs fallback on every rank
    ctx.commit_source()
    res["sharded_dev_far"] = comm.align_query_sharded(max_iterations=4, solve=0, fixed_iterations=1, last_translation=np.float32([0.1, 0, 0]))
    comm.barrier()
    comm.close()
    ctx.close()
    q_out.put((rank, res))


def test_two_ranks_on_one_gpu_through_the_c_abi(oracle):
    import multiprocessing as mp

    from icp_slam_prototype_amd import binding, build, synth

    build.build()
    fake = build.build_fake_rccl()
    mpc = mp.get_context("spawn")
    q_id, q_out = mpc.Queue(), mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, 2, fake, q_id, q_out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q_out.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [got[r]["rank"] for r in (0, 1)] == [0, 1] and got[0]["world"] == got[1]["world"] == 2
    # (1) both ranks hold the root's target and search it correctly
    key = synth.kinect_pair(90, 120, valid=0.7, seed=41)
    oi, od = oracle.nn_bruteforce(key["source"], key["target"], threads=4)
    for r in (0, 1):
        assert np.array_equal(got[r]["bcast_target"], key["target"]), r
        assert np.array_equal(got[r]["bcast_nn"][0], oi) and np.array_equal(got[r]["bcast_nn"][1].view(np.uint32), od.view(np.uint32))
    # (2) every rank holds all rows in global pair order
    for r in (0, 1):
        T, S = got[r]["gather"]
        assert T.shape == (5, 4, 4) and S.shape == (5, 4)
        for k in range(5):
            assert np.array_equal(T[k], k * 100 + np.arange(16, dtype=np.float32).reshape(4, 4))
            assert list(S[k]) == [k, 0, 100_000_001 + k, 0.5 * k]
    # (3)
    for r in (0, 1):
        sums, cnt = got[r]["allreduce"]
        assert np.array_equal(sums, np.arange(19) * 3.0) and cnt == 3 * 10 ** 10
    # (4) the sharded batch equals the batch on one context, bit for bit, on both ranks
    ctx = binding.Context(0)
    T1, st1, rc = ctx.align_batch(_pairs(), max_iterations=4, fixed_iterations=1)
    for r in (0, 1):
        Tg, Sg = got[r]["pair_batch"]
        assert np.array_equal(Tg, T1)
        assert [int(v) for v in Sg[:, 2]] == [s.final_pairs for s in st1] and (Sg[:, 0] == 4).all()
    # (5) the query-sharded loop equals the single-context loop up to the order the halves are added in
    big = synth.kinect_pair(120, 160, valid=0.8, seed=77)
    ctx.set_target(big["target"])
    ctx.set_source(big["source"])
    T2, st2, _ = ctx.align(max_iterations=5, solve=binding.SOLVE_KABSCH, fixed_iterations=1)
    ctx.close()
    assert np.array_equal(got[0]["sharded"][0], got[1]["sharded"][0])  # replicated solve: identical on both ranks
    Ts, it, n, mse, status = got[0]["sharded"]
    assert it == 5 and status == 0 and n == st2.final_pairs
    assert np.linalg.norm(Ts.astype(np.float64) - T2.astype(np.float64)) < 1e-5
    # (6) the device-side sharded loop: identical on both ranks, and equal to the single-context loop up to the order
    # the halves' sums are added in -- same iterations, same exit, same pair count
    ctx = binding.Context(0)
    ctx.set_target(big["target"])
    for k, kw in enumerate((dict(max_iterations=5, solve=1, fixed_iterations=1), dict(max_iterations=6, solve=0, fixed_iterations=1),
                            dict(max_iterations=12, solve=1, threshold=3e-4))):
        ctx.set_source(big["source"])
        T3, st3, rc3 = ctx.align(**kw)
        a, b = got[0]["sharded_dev"][k], got[1]["sharded_dev"][k]
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
        assert (a[1], a[2], a[4]) == (st3.iterations, st3.final_pairs, rc3), (k, a[1:], st3.iterations, st3.final_pairs)
        assert np.linalg.norm(a[0].astype(np.float64) - T3.astype(np.float64)) < 1e-5
    ctx.set_source(big["source"])
    ctx.transform_source(np.eye(3, dtype=np.float32), np.float32([50, 0, 0]))
    ctx.commit_source()
    T4, st4, rc4 = ctx.align(max_iterations=4, solve=0, fixed_iterations=1, last_translation=np.float32([0.1, 0, 0]))
    ctx.close()
    a, b = got[0]["sharded_dev_far"], got[1]["sharded_dev_far"]
    assert np.array_equal(a[0], b[0]) and a[4] == b[4] == rc4 == binding.W_TOO_FEW_PAIRS
    assert np.array_equal(a[0], T4) and a[1] == st4.iterations
