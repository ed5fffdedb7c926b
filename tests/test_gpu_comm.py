"""RCCL behind the C ABI (icpk_comm_*) on the one GPU this box has: a
communicator of world size 1 exercises the whole call path -- dlopen of librccl, unique id,
ncclCommInitRank, broadcast / all-gather / all-reduce on the context's stream, staging and
re-ordering -- with results that are known in closed form.  The partition and ordering logic
for world > 1 is covered by the gloo tests (tests/test_batch_gloo.py); RCCL refuses two ranks
on one device, so real multi-rank runs happen only in bench.py on the 8-GPU node."""
import numpy as np
import pytest

from icp_slam_prototype_amd import batch, binding, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from icp_slam_prototype_amd import build

    build.build()
    c = binding.Context(0)
    yield c
    c.close()


def test_partition_rule_is_the_python_one():
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert binding.comm_partition(n, world, r) == batch.partition(n, world, r)


def test_calls_without_a_communicator_fail_cleanly(ctx):
    assert ctx.comm_rank == -1 and ctx.comm_world == 0
    with pytest.raises(binding.IcpkError) as e:
        ctx.comm_broadcast_target(0)
    assert e.value.code == binding.E_NOT_SET


def test_world_of_one_round_trip(ctx):
    comm = batch.RcclComm(ctx, 0, 1, lambda uid: uid)
    try:
        assert ctx.comm_rank == 0 and ctx.comm_world == 1
        # key-frame broadcast: the root keeps its cloud and everything derived from it stays valid
        p = synth.kinect_pair(120, 160, seed=3)
        ctx.set_target(p["target"])
        ctx.set_source(p["source"])
        before = ctx.nn(binding.NN_GRID)
        comm.broadcast_target(0)
        assert np.array_equal(ctx.get_target(), p["target"])
        after = ctx.nn(binding.NN_GRID)
        assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
        # results of a batch: gathered rows == local rows, in order
        pairs = [(synth.frustum_pair(500 + 50 * k, seed=40 + k)["source"], synth.frustum_pair(500 + 50 * k, seed=40 + k)["target"])
                 for k in range(5)]
        T, st, rc = ctx.align_batch(pairs, max_iterations=3, fixed_iterations=1)
        Tg, Sg = comm.gather_results(T, st, 5)
        assert np.array_equal(Tg, T) and np.array_equal(Sg, batch.stats_rows(st))
        with pytest.raises(binding.IcpkError):  # a block that is not this rank's share
            comm.gather_results(T[:4], st[:4], 5)
        T0, S0 = comm.gather_results(np.zeros((0, 4, 4), np.float32), [], 0)
        assert T0.shape == (0, 4, 4)
        # the whole config-4 host logic through the RCCL transport
        Tb, Sb = batch.align_pair_batch(5, lambda i: pairs[i],
                                        lambda ps: (lambda r: (r[0], batch.stats_rows(r[1])))(ctx.align_batch(ps, max_iterations=3, fixed_iterations=1)),
                                        comm)
        assert np.array_equal(Tb, T) and np.array_equal(Sb, Sg)
        # query-sharded mode's exchange step: the sum over one rank is the identity
        sums = np.arange(19, dtype=np.float64) * 1.25 - 3
        out, cnt = comm.allreduce_sums(sums, 123456789012)
        assert np.array_equal(out, sums) and cnt == 123456789012
        comm.barrier()
        T1, it, n, mse, status = batch.align_query_sharded(batch.ContextSteps(ctx), comm, max_iterations=4, solve=1,
                                                           fixed_iterations=True)
        ctx.reset_source()
        T2, st2, _ = ctx.align(max_iterations=4, solve=binding.SOLVE_KABSCH, fixed_iterations=1, host_loop=1)
        assert status == 0 and it == 4 and n == st2.final_pairs and np.linalg.norm(T1.astype(np.float64) - T2) < 1e-6
        # the device-side sharded loop with ONE rank (real RCCL, in-stream all-reduce): the reduced sums ARE the rank's
        # canonical sums, so it must reproduce icpk_align bit for bit -- transform, statistics, associations, moved source
        for kw in (dict(max_iterations=6, solve=binding.SOLVE_REFERENCE, fixed_iterations=1),
                   dict(max_iterations=6, solve=binding.SOLVE_KABSCH, fixed_iterations=1),
                   dict(max_iterations=16, solve=binding.SOLVE_KABSCH, threshold=1e-4)):
            ctx.reset_source()
            Ta, sta, rca = ctx.align(**kw)
            ia, da = ctx.get_associations()
            sa = ctx.get_source()
            ctx.reset_source()
            Tb2, stb, rcb = ctx.align_query_sharded(**kw)
            ib, db = ctx.get_associations()
            assert np.array_equal(Ta, Tb2) and (rca, sta.iterations, sta.final_pairs, sta.final_mse) == (rcb, stb.iterations, stb.final_pairs, stb.final_mse)
            assert np.array_equal(ia, ib) and np.array_equal(da.view(np.uint32), db.view(np.uint32))
            assert np.array_equal(sa.view(np.uint32), ctx.get_source().view(np.uint32))
    finally:
        comm.close()
    assert ctx.comm_world == 0
