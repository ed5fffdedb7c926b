"""Host-side pieces of bench.py that need no GPU: the --gpus / WORLD_SIZE guard (a
--gpus N run with one process must not silently measure one GPU) and the CPU baseline child
(quota-aware thread count, pinned OpenMP threads, the JSON the bench line embeds)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_gpus_must_match_world_size():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_cpu_baseline_child_reports_cores_and_rates(oracle):
    env = dict(os.environ, ICPK_CPU_SAMPLE_S="0.05", OMP_PROC_BIND="close", OMP_PLACES="cores")
    r = subprocess.run([sys.executable, BENCH, "--cpu-baseline-child", "--workload", "frustum10k"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-400:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["kind"] == "port" and d["unit"] == "iter/s" and d["value"] > 0
    assert 1 <= d["cores"] <= (os.cpu_count() or 1)
    if d["host"]["cgroup_cpu_quota"]:
        assert d["cores"] <= d["host"]["cgroup_cpu_quota"]
    if d["host"]["cores_per_socket"]:
        assert d["cores"] <= d["host"]["cores_per_socket"]
    assert d["one_thread_ns_per_pair"] > 0 and d["gpairs_per_s"] > 0
    assert "OMP_PROC_BIND=close" in d["sample"]
