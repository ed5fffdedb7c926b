"""bench.py's N > 1 path (BASELINE configs[3]: pairs block-partitioned over one process per GPU,
results all-gathered every step) rehearsed on the one GPU of this box: two ranks share cuda:0 and
the collectives run over gloo (ICPK_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device).  Keeps
the launch contract -- `python -m torch.distributed.run ... bench.py --gpus N`, one JSON line from
rank 0 -- from rotting between the driver's 8-GPU runs."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, extra_args=()):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, ICPK_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--batch-pairs", "6"] + list(extra_args)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["unit"] == "iter/s" and d["value"] > 0
    assert d["scaling"] == "strong" and d["higher_is_better"] is True and d["vs_baseline"] is None
    if "--shard" not in extra_args:
        assert d["config"]["frame_pairs_per_step"] == 6 and d["config"]["pairs_per_gpu"] == 3
        assert d["results_consistent_on_all_ranks"] is True
        assert len(d["per_rank_ms_per_step"]["align_batch_device"]) == 2 and d["single_gpu_reference"]["value"] > 0
    return d


def test_two_rank_frame_batch_line():
    d = _run({})
    assert d["collectives"].startswith("torch.distributed(gloo")


def test_two_rank_frame_batch_line_through_the_c_abi_communicator():
    """the path the driver's multi-GPU run takes -- RcclComm: id exchange over the launcher's process group,
    icpk_comm_init_rccl, icpk_comm_gather_results every step, the key-frame broadcast -- with the
    shared-memory stand-in for the collectives (tests/cpp/fake_rccl.cpp), two ranks on one GPU"""
    from icp_slam_prototype_amd import build

    d = _run({"ICPK_TEST_HOOKS": "1", "ICPK_RCCL_LIB": build.build_fake_rccl()})
    assert d["collectives"].startswith("icpk_comm") and d["collectives_fallback_reason"] is None
    assert d["keyframe_broadcast_ms"] is not None and d["keyframe_broadcast_error"] is None
    assert d["keyframe_broadcast_bytes"] > 0


def test_two_rank_query_sharded_line_through_the_device_side_loop():
    """bench.py --gpus 2 --shard queries: the target broadcast and the in-stream all-reduce of icpk_align_query_sharded
    through the C-ABI communicator (shared-memory stand-in for the collectives, two ranks on one GPU)"""
    from icp_slam_prototype_amd import build

    d = _run({"ICPK_TEST_HOOKS": "1", "ICPK_RCCL_LIB": build.build_fake_rccl()}, ("--shard", "queries", "--iters", "5"))
    assert d["loop"].startswith("device-side") and d["collectives"].startswith("icpk_comm")
