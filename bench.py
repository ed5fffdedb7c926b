#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ICP inner loop.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N
          --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

N = 1 (BASELINE.json configs[1]): one synthetic 640x480 Kinect frame pair, 30 % valid pixels
(~92k points per cloud), 20 fixed ICP iterations.  One "step" = one icpk_align call = 21 NN
sweeps + 20 reduce / solve / transform iterations, clouds resident in HBM.  value = ICP
iterations/s of that single pair; NN Mpoints/s and Gpairs/s are reported alongside.  The same
line carries
  roofline            the NN kernel that was timed against what can physically bind it: HBM bytes
                      per launch from the PMC counters (profiles/hbm_traffic.json, written by
                      tools/collect_counters.py from separate rocprofv3 --pmc passes) over the
                      launch duration measured live with HIP events on the kernel's stream,
                      vs 8 TB/s; plus the VALU-issue fraction from SQ_INSTS_VALU;
  roofline_algorithmic the brute-force scan's operand bytes (Nq*Nt*12 + Nq*20)
                      over the same duration -- a YARDSTICK, not traffic (the grid kernel
                      returns the brute-force result without touching those bytes);
  roofline_bruteforce the kernel that does evaluate every pair (K1b), both ways;
  cpu_baseline        the oracle's CPU restatement, pinned OpenMP threads, median of 5;
  frame_batch         BASELINE configs[3] on ONE GPU: 64 distinct config-2 pairs (seeds
                      100..163) through icpk_align_batch_device -- the number the N > 1 lines
                      scale against;
  extra               the dense 307 200-point pair (the metric string's "307k-pt") and config 5
                      (10^6 x 10^6 points, 50 iterations).

N > 1 (BASELINE.json configs[3], frame-batch mode): the SAME 64 pairs,
block-partitioned 64/N per rank, one process per GPU, every rank aligns its block in lock-step
groups on its GPU, no per-iteration collective; one RCCL all-gather of the results per step
(icpk_comm_gather_results, RCCL behind the C ABI).  One step = the whole 64-pair batch; value =
aggregate iterations/s; total work is fixed, so "scaling" is "strong".
"""
import argparse
import gc
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
N_SIMD = 256 * 4           # 256 CUs x 4 SIMDs
VALU_ISSUE_PER_SIMD = 1.2e9  # wave64 fp32 VALU: one wave-instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)
METRIC = "ICP iterations/sec + NN Mpoints/sec at 307k-pt Kinect cloud, 1/2/4/8 GPU"
DTYPE = "f32 filter + f64 exact pair arithmetic (reference float semantics), f64 reductions and solve"
BATCH_PAIRS = 64           # BASELINE configs[3]
BATCH_SEED0 = 100          # seeds 100..163
KERNEL_NAMES = {"exact": "nn_exact_kernel", "filtered": "nn_filtered_kernel<2>", "pruned": "nn_pruned_kernel<4>",
                "grid": "nn_grid_kernel<8,false|true>"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=5, help="N = 1: timed regions of --steps steps each; the median one is reported")
    ap.add_argument("--iters", type=int, default=20, help="fixed ICP iterations per alignment (config 2: 20)")
    ap.add_argument("--workload", default="kinect640x480_30pct",
                    choices=["kinect640x480_30pct", "kinect640x480_dense", "kinect_v2_512x424", "dense1m", "frustum10k"])
    ap.add_argument("--solve", default="reference", choices=["reference", "kabsch", "p2l"])
    ap.add_argument("--nn-mode", default="grid", choices=["exact", "filtered", "pruned", "grid"],
                    help="all four give bit-identical results; grid is the product default")
    ap.add_argument("--shard", default="frames", choices=["frames", "queries"],
                    help="N>1: 'frames' = BASELINE config 4, 64 pairs block-partitioned over the ranks (default); "
                         "'queries' = ONE pair, queries split over ranks, one 160-byte all-reduce per iteration "
                         "(SURVEY.md 8e alternative)")
    ap.add_argument("--comm", default="auto", choices=["auto", "icpk", "torch"],
                    help="N>1 collectives: RCCL behind the C ABI (icpk_comm_*) or torch.distributed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the frame_batch / dense / config-5 blocks at N = 1")
    ap.add_argument("--batch-pairs", type=int, default=BATCH_PAIRS)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def make_workload(name, seed):
    from icp_slam_prototype_amd import synth

    if name == "kinect640x480_30pct":
        return synth.kinect_pair(480, 640, valid=0.30, seed=seed)
    if name == "kinect640x480_dense":
        return synth.kinect_pair(480, 640, valid=1.0, seed=seed)
    if name == "kinect_v2_512x424":
        return synth.kinect_pair(424, 512, valid=1.0, seed=seed, fx=synth.K2_FX, cx=synth.K2_CX)
    if name == "dense1m":
        return synth.dense_pair(1_000_000, seed=seed)
    return synth.frustum_pair(10000, seed=seed)


# ------------------------------------------------------------------------- CPU baseline --
def cpu_topology():
    """Cores this process may really use: the cgroup CPU quota (a box with one GPU gets a
    share of the host), the affinity mask and the socket size from lscpu."""
    info = {"logical_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "quota_cpus": None,
            "model": None, "sockets": None, "cores_per_socket": None, "threads_per_core": None}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            info["quota_cpus"] = float(q) / float(per)
    except Exception:
        pass
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            k, _, v = line.partition(":")
            v = v.strip()
            if k == "Model name":
                info["model"] = v
            elif k == "Socket(s)":
                info["sockets"] = int(v)
            elif k == "Core(s) per socket":
                info["cores_per_socket"] = int(v)
            elif k == "Thread(s) per core":
                info["threads_per_core"] = int(v)
    except Exception:
        pass
    return info


def cpu_baseline(workload):
    """The CPU baseline runs in a CHILD process (it never touches the GPU): libgomp binds the
    calling thread when OMP_PROC_BIND is set, and the parent's host thread -- and the HIP
    runtime's helper threads that inherit its mask -- must not end up pinned to one core."""
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
    env.pop("OMP_NUM_THREADS", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--workload", workload],
                       env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-400:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


def cpu_baseline_child(workload):
    """Oracle (CPU restatement, kind 'port'; gcc -O2 + OpenMP over queries) on the host cores
    this box grants: threads = min(cgroup quota, cores of ONE socket), pinned (OMP_PROC_BIND=close,
    OMP_PLACES=cores), warm-up, then the median of 5 samples of >= ~2 s each of the NN sweep over
    a contiguous query sample against the full target, scaled linearly to Nq, plus the reduce /
    solve / transform of one iteration on the full clouds; and the 1-thread cost per pair."""
    topo = cpu_topology()  # before libgomp is loaded: it narrows this thread's affinity mask
    w = make_workload(workload, 2)
    src, tgt = np.ascontiguousarray(w["source"]), np.ascontiguousarray(w["target"])
    from oracle import icp_oracle as o

    limit = topo["affinity"]
    if topo["quota_cpus"]:
        limit = min(limit, int(topo["quota_cpus"]))
    if topo["cores_per_socket"]:
        limit = min(limit, topo["cores_per_socket"])
    threads = max(1, limit)  # explicit num_threads(): omp_get_max_threads() is whatever an earlier library set
    nq, nt = src.shape[1], tgt.shape[1]

    def sweep(m, th):
        sub = np.ascontiguousarray(src[:, :m])
        t0 = time.perf_counter()
        idx, dist = o.nn_bruteforce(sub, tgt, threads=th)
        return time.perf_counter() - t0, idx, dist

    # one sample = whole sweeps of m queries until at least sample_s seconds have passed (timed, not
    # calibrated: a box that is busy while calibrating must not end up with short samples)
    sweep(min(256, nq), threads)  # warm-up (thread team, caches)
    t_cal, _, _ = sweep(min(2048, nq), threads)
    sample_s = float(os.environ.get("ICPK_CPU_SAMPLE_S", "2.5"))  # seconds per sample (tests shorten it)
    m = int(min(nq, max(2048, 2048 * sample_s / max(t_cal, 1e-4))))
    samples = []
    rounds_each = []
    for _ in range(5):
        tt, rounds = 0.0, 0
        while tt < sample_s or rounds == 0:
            t, idx, dist = sweep(m, threads)
            tt += t
            rounds += 1
        samples.append(tt / rounds * nq / m)
        rounds_each.append(rounds)
    rounds = statistics.median(rounds_each)
    t_nn = statistics.median(samples)
    # one thread: cost per pair of the scalar scan (BASELINE.md B1 stand-in)
    t1_cal, _, _ = sweep(min(128, nq), 1)
    m1 = int(min(nq, max(128, 128 * 0.8 * sample_s / max(t1_cal, 1e-4))))
    t1, _, _ = sweep(m1, 1)
    ns_per_pair_1t = t1 / (float(m1) * nt) * 1e9
    idx_full = np.resize(idx, nq)
    dist_full = np.resize(dist, nq)
    rest = []
    for _ in range(3):
        t0 = time.perf_counter()
        sums, cnt = o.sums_canonical(src, tgt, idx_full, dist_full, 0.75)
        R = o.solve_reference(sums[:9].astype(np.float32))
        o.transform_points(src, o.inv3(R), -(sums[9:12] / max(cnt, 1)).astype(np.float32))
        rest.append(time.perf_counter() - t0)
    t_rest = statistics.median(rest)
    it_s = 1.0 / (t_nn + t_rest)
    spread = (max(samples) - min(samples)) / t_nn
    out = {
        "value": it_s, "unit": "iter/s", "cores": threads, "kind": "port",
        "sample": f"median of 5 NN sweeps of {m} of {nq} queries x {nt} targets on {threads} pinned OpenMP threads "
                  f"(OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}, OMP_PLACES={os.environ.get('OMP_PLACES')}; "
                  f"each sample = {rounds} such sweep(s), {t_nn * m / nq * rounds:.1f} s, scaled linearly to Nq; spread "
                  f"{spread * 100:.1f} %) + median of 3 "
                  f"full reduce/solve/transform; oracle/icp_oracle.c, gcc -O2",
        "nn_s_per_sweep": t_nn, "gpairs_per_s": nq * float(nt) / t_nn / 1e9, "sample_spread": spread,
        "one_thread_ns_per_pair": ns_per_pair_1t,
        # (a noisy neighbour on the box shows up as spread: the fastest sample is what the cores can do)
        "fastest_sample_iter_s": 1.0 / (min(samples) + t_rest),
        "host": {"model": topo["model"], "sockets": topo["sockets"], "cores_per_socket": topo["cores_per_socket"],
                 "threads_per_core": topo["threads_per_core"], "logical_cpus": topo["logical_cpus"],
                 "cgroup_cpu_quota": topo["quota_cpus"]},
    }
    if topo["cores_per_socket"] and threads < topo["cores_per_socket"]:
        # the box's CPU share is smaller than a socket: say what a whole socket would do if the
        # scan scaled linearly (it is embarrassingly parallel over queries) -- an extrapolation
        f = topo["cores_per_socket"] / threads
        out["single_socket_extrapolated_iter_s"] = it_s * f
        out["note"] = (f"this box grants {threads} CPUs (cgroup quota) of a {topo['sockets']} x {topo['cores_per_socket']}-core "
                       f"host: the measured value is for {threads} cores; single_socket_extrapolated_iter_s = value x {f:.1f} "
                       "assumes linear scaling to one whole socket and is NOT measured")
    return out


# --------------------------------------------------------------------------- PMC figures --
def pmc_entry(workload, nn_mode):
    """Per-launch counter figures of the NN kernel for this workload, written by
    tools/collect_counters.py from separate rocprofv3 --pmc passes (None if never collected)."""
    f = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(f):
        return None
    e = json.load(open(f)).get(f"{workload}:{nn_mode}")
    return e if isinstance(e, dict) else None


def roofline_blocks(workload, nn_mode, nq, nt, avg_nn_s, timing, kernel):
    alg_bytes = float(nq) * nt * 12 + nq * 12 + nq * 8
    e = pmc_entry(workload, nn_mode)
    phys = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
            "kernel": kernel, "avg_launch_ms": avg_nn_s * 1e3, "timing": timing}
    if e:
        traffic = float(e["hbm_bytes_per_launch"])
        phys.update({
            "achieved": traffic / avg_nn_s / 1e9, "frac": traffic / avg_nn_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "profiles/hbm_traffic.json <- tools/collect_counters.py: memory-side read requests by size "
                              "(128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B) + WRITE_SIZE per NN launch, separate "
                              "rocprofv3 --pmc passes on the builder's box (constants here, not re-measured by this run); the "
                              "counters are calibrated on known byte counts in this kernel's access shapes: "
                              "profiles/r03_traffic_calibration.txt",
            "traffic_read_bytes": e.get("read_bytes_per_launch"), "traffic_write_bytes": e.get("write_bytes_per_launch"),
            "compulsory_bytes": float(nq + nt) * 12 + nq * 8,
        })
        # the same traffic over the kernel's rocprofv3 --kernel-trace duration (no event / dispatch latency in it): with
        # the event-bracket figure above, `frac` is a range, not a point
        if e.get("rocprof_avg_launch_ns"):
            t_k = float(e["rocprof_avg_launch_ns"]) * 1e-9
            phys["frac_range"] = {"hip_event_bracket": phys["frac"], "rocprof_kernel_duration": traffic / t_k / 1e9 / HBM_PEAK_GBS,
                                  "rocprof_avg_launch_ms": t_k * 1e3,
                                  "note": "event bracket measured live by this run (includes ~3-4 us of event / dispatch latency and "
                                          "the first, unseeded sweep in its share); rocprof duration from profiles/ (builder's box)"}
        if e.get("valu_insts_per_launch"):
            v = float(e["valu_insts_per_launch"])
            phys["valu_insts_per_launch"] = v
            phys["valu_issue_frac"] = v / (N_SIMD * VALU_ISSUE_PER_SIMD * avg_nn_s)
            phys["valu_issue_note"] = ("SQ_INSTS_VALU per launch / (1024 SIMDs x 1.2 G wave-instructions/s x launch time): "
                                       "the fp32 issue bound; f64 and sqrt instructions issue 2-4x slower, so the true "
                                       "issue occupancy is higher")
        vf = phys.get("valu_issue_frac") or 0.0
        if max(vf, phys["frac"]) < 0.5:
            phys["binds"] = ("neither bound binds alone: dependent memory phases per wave + instruction issue "
                             f"(HBM {phys['frac'] * 100:.0f} %, VALU issue >= {vf * 100:.0f} %)")
        elif vf >= phys["frac"]:
            phys["binds"] = f"VALU instruction issue (>= {vf * 100:.0f} % of the fp32 issue bound; HBM {phys['frac'] * 100:.1f} %)"
        else:
            phys["binds"] = f"HBM ({phys['frac'] * 100:.0f} %)"
    yard = {"bound": "hbm", "achieved": alg_bytes / avg_nn_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg_bytes / avg_nn_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes,
            "gpairs_per_s_kernel": nq * float(nt) / avg_nn_s / 1e9,
            "note": "YARDSTICK, not traffic: operand bytes of the brute-force scan this kernel replaces "
                    "(Nq*Nt*12 + Nq*20, SURVEY.md 8d) over the measured launch time; the grid / pruned kernels return "
                    "the same result while only looking at targets within reach"}
    return phys, yard


def batch_roofline(stats, group, pairs_per_launch_hint=None):
    """Physical roofline of the lock-step group's NN kernel (nn_grid_batch_kernel) from the launches
    icpk_align_batch bracketed with HIP events (params.profile = 1: one batched launch per group, booked on
    the group's first pair, covering all its pairs) and the per-pair counter figures of
    profiles/hbm_traffic.json["frame_batch8:grid"] (a group of 8 config-2 pairs under rocprofv3 --pmc)."""
    e16, e8 = pmc_entry("frame_batch16", "grid"), pmc_entry("frame_batch8", "grid")
    e, epairs = (e16, 16.0) if (e16 and group >= 16) else (e8, 8.0)  # counters of the launch shape this run times
    t = 0.0
    pair_launches = 0  # pairs covered by the timed launches
    launches = 0
    for call in stats:  # one list of per-pair stats per profiled call
        n = len(call)
        for g0 in range(0, n, group):
            s = call[g0]
            if s.nn_timed_launches >= 1:
                t += s.nn_ms_total / 1e3
                pair_launches += min(group, n - g0)
                launches += 1
    if launches == 0 or t <= 0:
        return None
    out = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
           "kernel": "nn_grid_batch_kernel<4,false|true>", "avg_launch_ms": t / launches * 1e3,
           "pairs_per_launch": pair_launches / launches,
           "timing": f"two HIP events around ONE batched NN launch per lock-step group, its position among the group's 21 "
                     f"sweeps rotating ({launches} launches over {len(stats)} extra calls after the timed ones; the next group's "
                     "set-up runs beside it on another stream, and the event pair costs ~4 us)"}
    if e:
        per_pair = float(e["hbm_bytes_per_launch"]) / epairs
        v_pair = float(e.get("valu_insts_per_launch") or 0.0) / epairs
        traffic = per_pair * pair_launches / launches
        out.update({"traffic": traffic, "achieved": per_pair * pair_launches / t / 1e9,
                    "frac": per_pair * pair_launches / t / 1e9 / HBM_PEAK_GBS,
                    "traffic_source": f"profiles/hbm_traffic.json[frame_batch{int(epairs)}:grid] / {int(epairs)} pairs x pairs per launch "
                                      "(tools/collect_counters.py: read requests by size + WRITE_SIZE, separate --pmc passes)"})
        if e.get("rocprof_avg_launch_ns") and abs(pair_launches / launches - epairs) < 0.5:
            t_k = float(e["rocprof_avg_launch_ns"]) * 1e-9
            out["frac_range"] = {"hip_event_bracket": out["frac"], "rocprof_kernel_duration": per_pair * epairs / t_k / 1e9 / HBM_PEAK_GBS,
                                 "rocprof_avg_launch_ms": t_k * 1e3}
        if v_pair:
            out["valu_issue_frac"] = v_pair * pair_launches / (N_SIMD * VALU_ISSUE_PER_SIMD * t)
    return out


# ------------------------------------------------------------------------------- helpers --
def upload(torch, dev, a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


def set_clouds_device(ctx, src_d, tgt_d):
    es = src_d.element_size()
    nq, nt = src_d.shape[1], tgt_d.shape[1]
    ctx.set_target_device(tgt_d.data_ptr(), tgt_d.data_ptr() + nt * es, tgt_d.data_ptr() + 2 * nt * es, nt)
    ctx.set_source_device(src_d.data_ptr(), src_d.data_ptr() + nq * es, src_d.data_ptr() + 2 * nq * es, nq)


def timed_alignments(torch, ctx, params, steps, warmup, sync_all, blocks=1):
    """W untimed, then `blocks` timed regions of exactly K steps each, every region bracketed by sync_all(); a step =
    icpk_reset_source (the pair back at its initial pose: three stream-ordered device copies, inside the timed region) +
    icpk_align.  One NN launch per alignment (the position rotating over its sweeps) is bracketed by two HIP events on
    the context's stream.  Returns the MEDIAN block's (elapsed, iterations, launches) -- one scheduler hiccup cannot
    move the number -- plus the event totals over all blocks and every block's elapsed time."""
    params.profile = 0
    for _ in range(warmup):
        ctx.reset_source()
        ctx.align(params)
    params.profile = 1
    params.profile_stride = params.max_iterations + 1
    gc.collect()
    gc.disable()  # a generation-2 collection of the interpreter (tens of ms) must not land in a timed region
    nn_ms = 0.0
    nn_timed = 0
    per_block = []
    for _ in range(blocks):
        sync_all()
        t0 = time.perf_counter()
        nn_launches = iters_done = 0
        for _ in range(steps):
            ctx.reset_source()
            T, st, rc = ctx.align(params)
            nn_ms += st.nn_ms_total
            nn_launches += st.nn_launches
            nn_timed += st.nn_timed_launches
            iters_done += st.iterations
        sync_all()
        per_block.append((time.perf_counter() - t0, iters_done, nn_launches))
    gc.enable()
    elapsed, iters_done, nn_launches = sorted(per_block)[len(per_block) // 2]
    timed_alignments.last_blocks = [round(b[0] * 1e3, 4) for b in per_block]
    return elapsed, iters_done, nn_launches, nn_timed, nn_ms


def batch_pairs_on_device(torch, dev, indices):
    from icp_slam_prototype_amd import synth

    keep, args = [], []
    for i in indices:
        p = synth.kinect_pair(480, 640, valid=0.30, seed=BATCH_SEED0 + i)
        s, t = upload(torch, dev, p["source"]), upload(torch, dev, p["target"])
        keep.append((s, t))
        args.append((s.data_ptr(), s.shape[1], t.data_ptr(), t.shape[1]))
    torch.cuda.synchronize()
    return keep, args


# --------------------------------------------------------------------------------- N = 1 --
def bench_single(args, torch, dev, gpu_index):
    from icp_slam_prototype_amd import binding

    w = make_workload(args.workload, 2)  # seed 2
    src_h, tgt_h = np.ascontiguousarray(w["source"]), np.ascontiguousarray(w["target"])
    src_d, tgt_d = upload(torch, dev, src_h), upload(torch, dev, tgt_h)
    torch.cuda.synchronize()
    nq, nt = src_d.shape[1], tgt_d.shape[1]
    ctx = binding.Context(gpu_index)
    set_clouds_device(ctx, src_d, tgt_d)
    if args.solve == "p2l":
        # point-to-plane (config 3): the target and its normals come from the depth image
        ctx.backproject_with_normals(w["depth_tgt"], binding.NORMALS_CROSS, offset=[5, 5, 5],
                                     fx=float(w.get("fx", 468.60)), cx=float(w.get("cx", 318.27)))
        assert ctx.target_size == nt
    solve = {"reference": binding.SOLVE_REFERENCE, "kabsch": binding.SOLVE_KABSCH, "p2l": binding.SOLVE_POINT_TO_PLANE}
    modes = {"exact": binding.NN_EXACT, "filtered": binding.NN_FILTERED, "pruned": binding.NN_PRUNED, "grid": binding.NN_GRID}
    params = binding.default_params(max_iterations=args.iters, fixed_iterations=1,
                                    max_nn_dist=0.3 if args.solve == "p2l" else 0.75,
                                    solve=solve[args.solve], nn_mode=modes[args.nn_mode])

    def sync_all():
        torch.cuda.synchronize()

    elapsed, iters_done, nn_launches, nn_timed, nn_ms = timed_alignments(torch, ctx, params, args.steps, args.warmup, sync_all,
                                                                         blocks=args.blocks)
    block_ms = list(timed_alignments.last_blocks)
    avg_nn_s = nn_ms / max(nn_timed, 1) / 1e3
    timing = (f"two HIP events on the kernel's own stream around one K1 launch per alignment of the timed regions, the "
              f"position rotating over the {args.iters + 1} sweeps ({nn_timed} of {nn_launches * args.blocks} launches; includes ~4 us of "
              "event/dispatch latency; the rocprofv3 --kernel-trace average of the same command is in profiles/)")
    phys, yard = roofline_blocks(args.workload, args.nn_mode, nq, nt, avg_nn_s, timing, KERNEL_NAMES[args.nn_mode])
    # per-stage device times from one extra, untimed alignment (events around every stage)
    params.profile = 2
    _, st2, _ = ctx.align(params)
    stage_ms = {"nn": st2.nn_ms_total, "reduce": st2.reduce_ms_total, "transform": st2.transform_ms_total, "total": st2.total_ms}
    params.profile = 0
    out = {
        "metric": METRIC, "value": iters_done / elapsed, "unit": "iter/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "timed_blocks": {"blocks": args.blocks, "steps_per_block": args.steps, "ms_each": block_ms,
                         "reported": "the median block: value = its iterations / its wall time, ms_per_step = its wall time / K"},
        "config": {"workload": f"{args.workload}: ONE frame pair, {nq} source x {nt} target points, {args.iters} fixed ICP "
                               f"iterations per step from the pair's initial pose, solve={args.solve}, nn={args.nn_mode} (BASELINE configs[1]); the "
                               f"N > 1 lines run configs[3] (64 such pairs) and scale against frame_batch below",
                   "frame_pairs_per_step": 1, "parallelism": "1 GPU"},
        "nn_mpoints_per_s": nn_launches * nq / elapsed / 1e6,
        "nn_gpairs_per_s_wall": float(nn_launches) * nq * nt / elapsed / 1e9,
        "roofline": phys, "roofline_algorithmic": yard, "stage_ms_per_step": stage_ms,
    }
    if args.nn_mode in ("pruned", "grid"):
        # the same sweep by the brute-force (un-pruned) filtered kernel north_star names: every one
        # of the Nq*Nt pairs is evaluated; HIP events on the context's stream around each launch
        ctx.reset_source()
        ctx.nn(binding.NN_FILTERED, fetch=False)  # seeds
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        s = torch.cuda.ExternalStream(ctx.stream)
        for a, b in ev:
            a.record(s)
            ctx.nn(binding.NN_FILTERED, fetch=False)
            b.record(s)
        torch.cuda.synchronize()
        t_bf = statistics.median(a.elapsed_time(b) for a, b in ev) / 1e3
        p_bf, y_bf = roofline_blocks(args.workload, "filtered", nq, nt, t_bf,
                                     "HIP events on the context's stream around icpk_nn(FILTERED) (fill + kernel + "
                                     "sync), median of 5", KERNEL_NAMES["filtered"])
        p_bf["algorithmic"] = y_bf
        out["roofline_bruteforce"] = p_bf
        ctx.reset_source()
    # PCIe-inclusive rate (never `value`): the boundary handed host buffers, so every step
    # re-uploads both clouds (pageable memory) and rebuilds the grid before aligning
    reps = 5
    t1 = time.perf_counter()
    for _ in range(reps):
        if args.solve == "p2l":  # the depth image crosses PCIe; cloud and normals are built on the device
            ctx.backproject_with_normals(w["depth_tgt"], binding.NORMALS_CROSS, offset=[5, 5, 5],
                                         fx=float(w["fx"]), cx=float(w["cx"]))
        else:
            ctx.set_target(tgt_h)
        ctx.set_source(src_h)
        ctx.align(params)
    out["pcie_inclusive_iter_s"] = reps * args.iters / (time.perf_counter() - t1)
    ctx.close()

    if not args.no_extras and args.workload == "kinect640x480_30pct" and args.solve == "reference" and args.nn_mode == "grid":
        out["frame_batch"] = frame_batch_one_gpu(args, torch, dev, gpu_index, out["value"])
        out["extra"] = {}
        for name, iters, steps in (("kinect640x480_dense", 20, 10), ("dense1m", 50, 4)):
            out["extra"][f"{name}_{iters}iters"] = extra_workload(args, torch, dev, gpu_index, name, iters, steps)
        out["extra"]["kinect_v2_512x424_point_to_plane_20iters"] = extra_point_to_plane(args, torch, dev, gpu_index)
        out["extra"]["tracker_path"] = tracker_path(gpu_index)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload)
        if "value" in out["cpu_baseline"]:
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        if "single_socket_extrapolated_iter_s" in out["cpu_baseline"]:
            out["speedup_vs_single_socket_extrapolated"] = out["value"] / out["cpu_baseline"]["single_socket_extrapolated_iter_s"]
    print(json.dumps(out))


def frame_batch_one_gpu(args, torch, dev, gpu_index, single_value):
    """BASELINE configs[3] on one GPU: all 64 pairs through icpk_align_batch_device."""
    from icp_slam_prototype_amd import binding

    keep, pargs = batch_pairs_on_device(torch, dev, range(args.batch_pairs))
    ctx = binding.Context(gpu_index)
    params = binding.default_params(max_iterations=args.iters, fixed_iterations=1)
    for _ in range(2):  # warm-up: the first call allocates the slots
        T, st, rc = ctx.align_batch_device(pargs, params)
    assert rc == 0 and all(s.iterations == args.iters for s in st)
    reps = 5
    each = []
    gc.collect()
    gc.disable()
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T, st, rc = ctx.align_batch_device(pargs, params)
        torch.cuda.synchronize()
        each.append(time.perf_counter() - t0)
    gc.enable()
    dt = statistics.median(each)
    total_iters = sum(s.iterations for s in st)
    params.profile = 1  # a few more calls, outside the timed ones: a bracketed NN launch per group
    prof = [ctx.align_batch_device(pargs, params)[1] for _ in range(6)]
    params.profile = 0
    roof = batch_roofline(prof, int(os.environ.get("ICPK_BATCH_GROUP", "16")))
    # PCIe-inclusive (never a headline): the same call with HOST buffers, as the reference would hand them
    # over (icpk_align_batch: every cloud crosses PCIe from pageable memory inside the call); 16 pairs
    nh = min(16, args.batch_pairs)
    host_pairs = [(s.cpu().numpy(), t.cpu().numpy()) for s, t in keep[:nh]]
    ctx.align_batch(host_pairs, params)
    th = []
    for _ in range(3):
        t0 = time.perf_counter()
        Th, sth, rch = ctx.align_batch(host_pairs, params)
        th.append(time.perf_counter() - t0)
    pcie_rate = sum(s.iterations for s in sth) / statistics.median(th)
    ctx.close()
    return {"workload": f"{args.batch_pairs} distinct config-2 pairs (seeds {BATCH_SEED0}..{BATCH_SEED0 + args.batch_pairs - 1}), "
                        f"{args.iters} fixed iterations each, resident in HBM, icpk_align_batch_device on ONE GPU "
                        f"(lock-step groups of {os.environ.get('ICPK_BATCH_GROUP', '16')})",
            "value": total_iters / dt, "unit": "iter/s", "ms_per_batch": dt * 1e3, "ms_per_pair": dt * 1e3 / args.batch_pairs,
            "vs_single_pair": total_iters / dt / single_value, "pcie_inclusive_iter_s": pcie_rate, "reps": reps, "timing": "median of 5 calls, host wall clock, "
            "device idle before and after each call", "ms_each": [round(t * 1e3, 3) for t in each], "roofline": roof}


def extra_workload(args, torch, dev, gpu_index, name, iters, steps):
    """the 'also run' variants of SURVEY.md 8d, driver-visible: same timing method as the headline"""
    from icp_slam_prototype_amd import binding

    w = make_workload(name, 2 if name != "dense1m" else 5)
    src_d, tgt_d = upload(torch, dev, w["source"]), upload(torch, dev, w["target"])
    torch.cuda.synchronize()
    nq, nt = src_d.shape[1], tgt_d.shape[1]
    ctx = binding.Context(gpu_index)
    set_clouds_device(ctx, src_d, tgt_d)
    params = binding.default_params(max_iterations=iters, fixed_iterations=1)
    elapsed, iters_done, nn_launches, nn_timed, nn_ms = timed_alignments(torch, ctx, params, steps, 2, torch.cuda.synchronize)
    ctx.close()
    avg_nn_s = nn_ms / max(nn_timed, 1) / 1e3
    phys, yard = roofline_blocks(name, "grid", nq, nt, avg_nn_s, f"as the headline ({nn_timed} of {nn_launches} launches)",
                                 "nn_grid_kernel<8|4,false|true>")
    return {"workload": f"{name}: {nq} x {nt} points, {iters} fixed iterations per step, {steps} steps",
            "value": iters_done / elapsed, "unit": "iter/s", "ms_per_step": elapsed / steps * 1e3,
            "nn_mpoints_per_s": nn_launches * nq / elapsed / 1e6, "roofline": phys, "roofline_algorithmic": yard}


def extra_point_to_plane(args, torch, dev, gpu_index, iters=20, steps=10):
    """BASELINE configs[2], driver-visible: Kinect v2 512x424 pair, surface normals computed on the device
    from the target's depth image, point-to-plane error accumulation (K5) and solve, grid NN."""
    from icp_slam_prototype_amd import binding, synth

    w = make_workload("kinect_v2_512x424", 2)
    src_d, tgt_d = upload(torch, dev, w["source"]), upload(torch, dev, w["target"])
    torch.cuda.synchronize()
    nq, nt = src_d.shape[1], tgt_d.shape[1]
    ctx = binding.Context(gpu_index)
    set_clouds_device(ctx, src_d, tgt_d)
    ctx.backproject_with_normals(w["depth_tgt"], binding.NORMALS_CROSS, offset=[5, 5, 5], fx=float(synth.K2_FX),
                                 cx=float(synth.K2_CX))
    assert ctx.target_size == nt
    params = binding.default_params(max_iterations=iters, fixed_iterations=1, max_nn_dist=0.3,
                                    solve=binding.SOLVE_POINT_TO_PLANE, nn_mode=binding.NN_GRID)
    elapsed, iters_done, nn_launches, nn_timed, nn_ms = timed_alignments(torch, ctx, params, steps, 2, torch.cuda.synchronize)
    ctx.close()
    return {"workload": f"kinect_v2_512x424 (BASELINE configs[2]): {nq} x {nt} points with target normals, point-to-plane, "
                        f"{iters} fixed iterations per step, {steps} steps",
            "value": iters_done / elapsed, "unit": "iter/s", "ms_per_step": elapsed / steps * 1e3,
            "nn_mpoints_per_s": nn_launches * nq / elapsed / 1e6, "nn_kernel_avg_launch_ms": nn_ms / max(nn_timed, 1)}


def tracker_frames():
    from icp_slam_prototype_amd import synth

    rng = np.random.default_rng(0)
    frames = []
    for k in range(6):
        d = synth.render_room_depth(480, 640, synth.rot_xyz_deg(0, 0.5 * k, 0), np.array([0.01 * k, 0, 0]),
                                    noise_sigma=0.002, rng=rng)
        d[rng.random(d.shape) > 0.3] = 0
        frames.append(d.astype(np.uint16))
    return frames


_NATIVE_TRACKER = None  # tracker_path_native's result when main() ran it before this process touched the GPU


def tracker_path(gpu_index):
    """The callers either side of the loop as SLAM.cpp drives them --
    icp::Tracker::getTransformation's call sequence through the C ABI (icpk_backproject_pair,
    icpk_align, icpk_get_trace): per frame pair two 640x480 uint16 depth images cross PCIe, are
    back-projected on the device (with and without filterDepthImage), posed, aligned with the reference's own settings (16 iterations at most,
    threshold 1e-4, SLAM.cpp:277) and the per-iteration trace is read back.  PCIe-inclusive by
    nature; frame pairs per second."""
    from icp_slam_prototype_amd import binding, synth

    frames = tracker_frames()
    ctx = binding.Context(gpu_index)
    camR, camP = np.eye(3, dtype=np.float32), np.full(3, 5, np.float32)
    res = {}
    # resident: the previous frame is the one the context received as `data` last time (SLAM.cpp:305 hands it back):
    # it stays on the device, one 614 KB upload per call instead of two -- icp::Tracker's getTransformation(data, nullptr, ...)
    for key, filt, resident in (("plain", False, True), ("with_filterDepthImage", True, True), ("both_frames_uploaded", False, False)):
        stage = {"backproject_pair": 0.0, "align": 0.0, "get_trace": 0.0}

        def pair(prev, cur, first):
            t0 = time.perf_counter()
            ctx.backproject_pair(cur, prev if (first or not resident) else None, R=camR, t=camP, filter=filt)  # icp.cpp:38-71 in one call
            t1 = time.perf_counter()
            T, st, rc = ctx.align(max_iterations=16, threshold=1e-4)
            t2 = time.perf_counter()
            ctx.get_trace(16)
            t3 = time.perf_counter()
            stage["backproject_pair"] += t1 - t0
            stage["align"] += t2 - t1
            stage["get_trace"] += t3 - t2
            return st

        for i in range(1, len(frames)):
            pair(frames[i - 1], frames[i], i == 1)
        for k in stage:
            stage[k] = 0.0
        t0 = time.perf_counter()
        n = its = 0
        for _ in range(4):
            for i in range(1, len(frames)):
                # (the sequence wraps around: frame 1 follows frame 5 only through an explicit `previous`)
                its += pair(frames[i - 1], frames[i], i == 1).iterations
                n += 1
        dt = time.perf_counter() - t0
        res[key] = {"frame_pairs_per_s": n / dt, "ms_per_pair": dt / n * 1e3, "mean_iterations": its / n,
                    "points": [ctx.source_size, ctx.target_size],
                    "host_ms_per_call": {k: round(v / n * 1e3, 4) for k, v in stage.items()}}
    ctx.close()
    # the same call sequence from a C++ caller (what a drop-in user of icp.cpp is): tests/cpp/tracker_bench.cpp as a
    # child process on the same frames -- no interpreter between the calls
    # (run by main() BEFORE this process initialised the GPU: with a second process holding queues on the card the
    # child's 614 KB uploads take ~260 us instead of ~70 -- an artefact of two processes on one GPU, not of either path)
    if _NATIVE_TRACKER is not None:
        res["native_cpp"] = _NATIVE_TRACKER
    else:
        try:
            res["native_cpp"] = dict(tracker_path_native(frames, gpu_index, rounds=8), beside_the_bench_process=True)
        except Exception as e:  # (the Python-driven figures above stand on their own)
            res["native_cpp"] = {"error": repr(e)[:300]}
    res["workload"] = ("6 synthetic 640x480 frames (30 % valid, camera drifting 0.5 degree / 1 cm per frame), consecutive pairs, "
                       "threshold exit; host depth images in, 4x4 + trace out; plain / with_filterDepthImage: the previous frame "
                       "stays on the device (one upload per call); both_frames_uploaded: as round 2 measured it; these three "
                       "through Python / ctypes (host_ms_per_call includes ~15 us of interpreter per call), native_cpp: the "
                       "same C-ABI calls from tests/cpp/tracker_bench.cpp (native_cpp.registered_frame_buffers: the frames in memory the caller "
                       "pinned once with icpk_register_host_buffer)")
    return res


def tracker_path_native(frames, gpu_index, rounds=8, filt=False, resident=True, registered=False):
    """tests/cpp/tracker_bench.cpp (prebuilt by __graft_entry__.build(); rebuilt here with g++ if missing) on `frames`."""
    import subprocess
    import tempfile

    from icp_slam_prototype_amd import build as b

    exe = b.TRACKER_BENCH if os.path.exists(b.TRACKER_BENCH) else b.build_tracker_bench()
    rows, cols = frames[0].shape
    with tempfile.NamedTemporaryFile(suffix=".u16") as f:
        for d in frames:
            f.write(np.ascontiguousarray(d, np.uint16).tobytes())
        f.flush()
        out = subprocess.run([exe, f.name, str(rows), str(cols), str(len(frames)), str(rounds), str(int(filt)),
                              str(int(resident)), str(gpu_index), str(int(registered))], capture_output=True, text=True, timeout=120)
    if out.returncode != 0:
        raise RuntimeError(f"tracker_bench rc {out.returncode}: {out.stderr[-300:]}")
    return json.loads(out.stdout.strip().splitlines()[-1])


# --------------------------------------------------------------------------------- N > 1 --
def bench_frame_batch(args, torch, dist, rank, world, dev, cdev, gpu_index, rehearsal):
    """BASELINE configs[3]: 64 pairs block-partitioned over the ranks."""
    from icp_slam_prototype_amd import batch, binding

    n_pairs = args.batch_pairs
    start, count = batch.partition(n_pairs, world, rank)
    keep, pargs = batch_pairs_on_device(torch, dev, range(start, start + count))
    ctx = binding.Context(gpu_index)
    comm = None
    comm_err = None
    # (rehearsal on one GPU: the C-ABI communicator only with the shared-memory stand-in of the tests)
    if args.comm in ("auto", "icpk") and (not rehearsal or (os.environ.get("ICPK_TEST_HOOKS") == "1" and os.environ.get("ICPK_RCCL_LIB"))):
        def exchange(uid):  # rank 0's 128-byte id to everybody, over the launcher's process group
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        try:
            comm = batch.RcclComm(ctx, rank, world, exchange)
        except Exception as e:  # noqa: BLE001 -- fall back to torch.distributed, and say so in the line
            if args.comm == "icpk":
                raise
            comm_err = repr(e)
    ok = torch.tensor([1 if comm is not None else 0], device=cdev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:  # not every rank got an RCCL communicator through the C ABI
        if comm is not None:
            comm.close()
        comm = batch.TorchComm(dist, cdev)
    params = binding.default_params(max_iterations=args.iters, fixed_iterations=1)

    def sync_all():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    t_align, t_gather = [], []  # this rank's host wall time per step: its block of pairs / the all-gather (incl. waiting for the slowest rank)

    def step():
        ta = time.perf_counter()
        T, st, rc = ctx.align_batch_device(pargs, params)
        tb = time.perf_counter()
        if isinstance(comm, batch.RcclComm):
            Tg, Sg = comm.gather_results(T, st, n_pairs)
        else:
            Tg, Sg = comm.gather_results(T, batch.stats_rows(st), n_pairs)
        t_align.append(tb - ta)
        t_gather.append(time.perf_counter() - tb)
        return rc, Tg, Sg

    step()  # set-up, before the W warm-up steps: the first call allocates the slots
    for _ in range(args.warmup):
        step()
    gc.collect()
    gc.disable()  # (the interpreter's generation-2 collections take tens of ms: several steps of this workload)
    sync_all()
    del t_align[:], t_gather[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rc, Tg, Sg = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    gc.enable()
    tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    # per-rank step times (mean over the timed steps), so that the spread between ranks is readable from the one line
    mine = torch.tensor([statistics.mean(t_align) * 1e3, statistics.mean(t_gather) * 1e3], dtype=torch.float64, device=cdev)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(per_rank, mine)
    per_rank = [[round(float(v), 4) for v in t.cpu()] for t in per_rank]
    # every rank holds the same gathered result, every pair ran its 20 iterations
    chk = torch.tensor(np.concatenate([Tg.reshape(-1), Sg.reshape(-1)]).astype(np.float64), device=cdev)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    consistent = bool(torch.equal(lo, hi)) and rc == 0 and bool((Sg[:, 0] == args.iters).all()) and bool((Sg[:, 1] == 0).all())
    roof = None
    try:  # (auxiliary: never let it cost the line) one more call with a bracketed NN launch per group
        params.profile = 1
        prof = [ctx.align_batch_device(pargs, params)[1] for _ in range(6)]
        params.profile = 0
        roof = batch_roofline(prof, int(os.environ.get("ICPK_BATCH_GROUP", "16")))
    except Exception as e:  # noqa: BLE001
        roof = {"error": repr(e)}
    # key-frame broadcast (north_star: RCCL broadcast of the target cloud over xGMI), untimed
    # part of the run: rank 0's first target becomes every rank's target; median of 5
    bcast_ms = bcast_err = None
    nt0 = 0
    if isinstance(comm, batch.RcclComm):
        try:  # auxiliary: never let it cost the line
            t_h = keep[0][1].cpu().numpy() if count else np.zeros((3, 1), np.float32)
            if rank == 0:
                ctx.set_target(t_h)
            ts = []
            for _ in range(5):
                sync_all()
                tb = time.perf_counter()
                comm.broadcast_target(0)
                ts.append((time.perf_counter() - tb) * 1e3)
            bcast_ms = statistics.median(ts)
            nt0 = ctx.target_size
        except Exception as e:  # noqa: BLE001
            bcast_err = repr(e)
    # the SAME batch on ONE GPU (rank 0's), outside the timed region: the reference the N-GPU value scales against
    single = None
    try:
        if rank == 0:
            keep1, pargs1 = batch_pairs_on_device(torch, dev, range(n_pairs))
            for _ in range(2):
                ctx.align_batch_device(pargs1, params)
            each = []
            for _ in range(5):
                torch.cuda.synchronize()
                tb = time.perf_counter()
                _, st1, _ = ctx.align_batch_device(pargs1, params)
                torch.cuda.synchronize()
                each.append(time.perf_counter() - tb)
            d1 = statistics.median(each)
            single = {"ms_per_batch": d1 * 1e3, "value": sum(x.iterations for x in st1) / d1, "unit": "iter/s",
                      "timing": "median of 5 calls of icpk_align_batch_device on rank 0's GPU, the other ranks idle"}
            del keep1
    except Exception as e:  # noqa: BLE001 -- auxiliary: never let it cost the line
        single = {"error": repr(e)}
    dist.barrier()
    if rank == 0:
        total_iters = float(Sg[:, 0].sum()) * args.steps
        al = [r[0] for r in per_rank]
        out = {
            "metric": METRIC, "value": total_iters / elapsed, "unit": "iter/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]: {n_pairs} distinct config-2 frame pairs (640x480, 30 % valid, seeds "
                                   f"{BATCH_SEED0}..{BATCH_SEED0 + n_pairs - 1}, ~92k x 92k points each), {args.iters} fixed ICP "
                                   f"iterations each per step, {n_pairs}/{world} pairs per GPU, clouds resident in HBM",
                       "frame_pairs_per_step": n_pairs, "pairs_per_gpu": count,
                       "parallelism": f"frame-batch x{world}: block partition, no per-iteration collective, one all-gather "
                                      f"of the results per step"},
            "roofline": roof,
            "per_rank_ms_per_step": {"align_batch_device": al, "gather_results_incl_wait": [r[1] for r in per_rank],
                                     "align_min": min(al), "align_max": max(al)},
            "single_gpu_reference": single,
            "speedup_vs_single_gpu": (total_iters / elapsed / single["value"]) if single and "value" in single else None,
            "collectives": comm.kind, "collectives_fallback_reason": comm_err,
            "results_consistent_on_all_ranks": consistent,
            "keyframe_broadcast_ms": bcast_ms, "keyframe_broadcast_error": bcast_err,
            "keyframe_broadcast_bytes": (3 * 4 * nt0) if bcast_ms is not None else None,
            "note": "N = 1 of this bench reports the single-pair configs[1] rate as `value` and this same 64-pair batch on "
                    "one GPU as `frame_batch.value`: scale the N > 1 lines against the latter",
        }
        print(json.dumps(out))
    if isinstance(comm, batch.RcclComm):
        comm.close()
    ctx.close()


def bench_query_sharded(args, torch, dist, rank, world, dev, cdev, gpu_index, rehearsal=False):
    """One frame pair, queries split over the ranks.  Default: the C ABI's device-side loop (icpk_align_query_sharded:
    per iteration the sweep and the reduction on the rank's slice, ONE in-stream ncclAllReduce of 20 doubles, the
    replicated loop step on the reduced sums; no host round trip), the target broadcast through
    icpk_comm_broadcast_target.  Falls back -- saying so in the line -- to the host-driven loop over torch.distributed
    (one stream sync + one collective per iteration) if RCCL cannot be initialised through the C ABI."""
    from icp_slam_prototype_amd import batch, binding

    w = make_workload(args.workload, 2)
    nq_total = w["source"].shape[1]
    s0, cnt = batch.partition(nq_total, world, rank)
    src = np.ascontiguousarray(w["source"][:, s0:s0 + cnt])
    ctx = binding.Context(gpu_index)
    solve = {"reference": 0, "kabsch": 1}.get(args.solve)
    if solve is None:
        raise SystemExit("--shard queries supports --solve reference|kabsch")
    comm = comm_err = None
    if args.comm in ("auto", "icpk") and (not rehearsal or (os.environ.get("ICPK_TEST_HOOKS") == "1" and os.environ.get("ICPK_RCCL_LIB"))):
        def exchange(uid):
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        try:
            comm = batch.RcclComm(ctx, rank, world, exchange)
        except Exception as e:  # noqa: BLE001
            if args.comm == "icpk":
                raise
            comm_err = repr(e)
    ok = torch.tensor([1 if comm is not None else 0], device=cdev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    device_loop = int(ok.item()) == 1 and args.nn_mode == "grid"
    if device_loop:
        if rank == 0:
            ctx.set_target(w["target"])
        comm.broadcast_target(0)  # RCCL broadcast of the target cloud over xGMI (north_star)
        ctx.set_source(src)
        nt = ctx.target_size
        params = binding.default_params(max_iterations=args.iters, solve=solve, fixed_iterations=1)

        def one_step():
            return comm.align_query_sharded(params=params)
    else:
        if comm is not None:
            comm.close()
        comm = batch.TorchComm(dist, cdev)
        tgt = comm.broadcast_cloud(w["target"] if rank == 0 else None, 0).cpu().numpy()
        nt = tgt.shape[1]
        ctx.set_target(tgt)
        ctx.set_source(src)
        modes = {"exact": binding.NN_EXACT, "filtered": binding.NN_FILTERED, "pruned": binding.NN_PRUNED, "grid": binding.NN_GRID}
        steps = batch.ContextSteps(ctx, modes[args.nn_mode])

        def one_step():
            ctx.reset_source()
            return batch.align_query_sharded(steps, comm, max_iterations=args.iters, solve=solve, fixed_iterations=True)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 0
    for _ in range(args.steps):
        iters += one_step()[1]
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    if rank == 0:
        print(json.dumps({
            "metric": METRIC, "value": iters / elapsed, "unit": "iter/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {nq_total} source x {nt} target points, {args.iters} "
                                   f"fixed ICP iterations per step, solve={args.solve}, nn={args.nn_mode}",
                       "parallelism": f"query-sharded x{world}, 1 all-reduce(160 B)/iteration"},
            "loop": ("device-side (icpk_align_query_sharded): in-stream ncclAllReduce between K2 and the replicated loop step, "
                     "no host round trip per iteration") if device_loop else
                    "host-driven over torch.distributed: one stream sync + one collective per iteration",
            "collectives": comm.kind, "collectives_fallback_reason": comm_err,
            "nn_mpoints_per_s": iters * nq_total / elapsed / 1e6}))
    if isinstance(comm, batch.RcclComm):
        comm.close()
    ctx.close()


def main():
    args = parse()
    if args.cpu_baseline_child:
        print(json.dumps(cpu_baseline_child(args.workload)))
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: N > 1 must be launched with one process per GPU "
                         f"(python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus})")

    if world == 1 and not args.no_extras and args.workload == "kinect640x480_30pct" and args.solve == "reference" and args.nn_mode == "grid":
        # the frame path from a C++ caller, measured while nothing else holds the GPU (see tracker_path)
        global _NATIVE_TRACKER
        try:
            fr = tracker_frames()

            def median_run(**kw):  # (a process now and then runs 30-40 % slow from start to end on these boxes: three processes, the median one)
                runs = sorted((tracker_path_native(fr, local_rank, rounds=8, **kw) for _ in range(3)), key=lambda r: r["frame_pairs_per_s"])
                return dict(runs[1], runs_frame_pairs_per_s=[round(r["frame_pairs_per_s"], 1) for r in runs])

            _NATIVE_TRACKER = dict(median_run(), beside_the_bench_process=False)
            # ... and for a caller whose frame buffers are long-lived and pinned once (icpk_register_host_buffer): no staging copy
            _NATIVE_TRACKER["registered_frame_buffers"] = median_run(registered=True)
        except Exception as e:
            _NATIVE_TRACKER = {"error": repr(e)[:300]}

    import torch  # before the binding: libicpk.so must bind to the HIP runtime torch ships

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # Rehearsal on a 1-GPU box (never used by the driver): ICPK_BENCH_REHEARSAL=1 maps every
    # rank to cuda:0 and runs the collectives over gloo, because RCCL refuses two ranks on
    # one device.  The real multi-GPU run uses RCCL (behind the C ABI; backend "nccl" for the
    # launcher's own process group).
    rehearsal = os.environ.get("ICPK_BENCH_REHEARSAL") == "1"
    gpu_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    if world == 1:
        return bench_single(args, torch, dev, gpu_index)

    import torch.distributed as dist

    cdev = torch.device("cpu") if rehearsal else dev  # device of the tensors handed to torch collectives
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if rehearsal:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        if args.shard == "queries":
            bench_query_sharded(args, torch, dist, rank, world, dev, cdev, gpu_index, rehearsal)
        else:
            bench_frame_batch(args, torch, dist, rank, world, dev, cdev, gpu_index, rehearsal)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
