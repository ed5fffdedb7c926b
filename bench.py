#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ICP inner loop.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N
          --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): one synthetic 640x480 Kinect frame pair,
30% valid pixels (~92k points per cloud), 20 fixed ICP iterations.  One "step"
= one icpk_align call = 21 brute-force NN sweeps + 20 reduce/solve/transform
iterations, clouds already resident in HBM.  metric = ICP iterations/s (whole
job); NN Mpoints/s and Gpairs/s are reported alongside.

N > 1 is the frame-batch mode: one process per GPU, every rank
aligns its own source frame against a key frame (target cloud) that rank 0
broadcasts once over RCCL/xGMI; no per-iteration collective; weak scaling.

The JSON line also carries
  roofline      -- the NN kernel against the 8 TB/s HBM roofline on ALGORITHMIC
                   bytes (Nq*Nt*12 + Nq*12 + Nq*8 per launch),
                   duration measured with HIP events on the kernel's own stream;
  cpu_baseline  -- the oracle's CPU restatement (OpenMP, all host cores) timed
                   on a bounded query sample of the same workload (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20, help="fixed ICP iterations per step (config 2: 20)")
    ap.add_argument("--workload", default="kinect640x480_30pct",
                    choices=["kinect640x480_30pct", "kinect640x480_dense", "kinect_v2_512x424", "dense1m", "frustum10k"])
    ap.add_argument("--solve", default="reference", choices=["reference", "kabsch", "p2l"])
    ap.add_argument("--nn-mode", default="grid", choices=["exact", "filtered", "pruned", "grid"],
                    help="all four give bit-identical results; grid is the product default")
    ap.add_argument("--shard", default="frames", choices=["frames", "queries"],
                    help="N>1: 'frames' = one frame pair per rank, no per-iteration collective (default, weak "
                         "scaling); 'queries' = ONE pair, queries split over ranks, one 160-byte all-reduce per "
                         "iteration (strong scaling; SURVEY.md 8e alternative)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=32768, help="queries in the CPU baseline sample")
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


def cpu_baseline(src, tgt, sample, solve):
    """Oracle (CPU restatement, kind 'port') on the host cores: NN sweep over a
    contiguous query sample against the full target, scaled linearly to Nq, plus
    the reduce/solve/transform of one iteration measured on the full clouds."""
    from oracle import icp_oracle as o

    threads = o.max_threads()
    nq = src.shape[1]
    m = min(sample, nq)
    sub = np.ascontiguousarray(src[:, :m])
    o.nn_bruteforce(sub[:, :256], tgt, threads=threads)  # warm
    t0 = time.perf_counter()
    idx, dist = o.nn_bruteforce(sub, tgt, threads=threads)
    t_nn = (time.perf_counter() - t0) * nq / m
    idx_full = np.resize(idx, nq)
    dist_full = np.resize(dist, nq)
    t0 = time.perf_counter()
    sums, cnt = o.sums_canonical(src, tgt, idx_full, dist_full, 0.75)
    M = sums[:9].astype(np.float32)
    R = o.solve_reference(M)
    o.transform_points(src, o.inv3(R), -(sums[9:12] / max(cnt, 1)).astype(np.float32))
    t_rest = time.perf_counter() - t0
    it_s = 1.0 / (t_nn + t_rest)
    return {
        "value": it_s, "unit": "iter/s", "cores": threads, "kind": "port",
        "sample": f"NN sweep of {m} of {nq} queries x {tgt.shape[1]} targets on {threads} OpenMP threads "
                  f"(scaled linearly to Nq) + full reduce/solve/transform; oracle/icp_oracle.c, gcc -O2",
        "nn_s_per_sweep": t_nn, "gpairs_per_s": nq * tgt.shape[1] / t_nn / 1e9,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch

    from icp_slam_prototype_amd import binding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # Rehearsal on a 1-GPU box (never used by the driver): ICPK_BENCH_REHEARSAL=1 maps every
    # rank to cuda:0 and runs the collectives over gloo, because RCCL refuses two ranks on
    # one device.  The real multi-GPU run uses backend "nccl" (= RCCL over xGMI).
    rehearsal = os.environ.get("ICPK_BENCH_REHEARSAL") == "1"
    gpu_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    cdev = torch.device("cpu") if rehearsal else dev  # device of the tensors handed to collectives
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.shard == "queries" and world > 1:
        return bench_query_sharded(args, rank, world, dev, cdev, gpu_index, dist)

    # ---- workload: own source frame per rank; key frame (target) from rank 0 ----
    base_seed = 2
    w = make_workload(args.workload, base_seed + 100 * rank if world > 1 else base_seed)
    src_h = np.ascontiguousarray(w["source"])
    src_d = torch.from_numpy(src_h).to(dev)
    if world > 1:
        from icp_slam_prototype_amd import batch

        # RCCL broadcast of the key frame's xyz-SoA over xGMI (one 3*Nt*4-byte message)
        tgt_d = batch.broadcast_cloud(w["target"] if rank == 0 else None, 0, cdev, dist).to(dev)
        tgt_h = tgt_d.cpu().numpy()
    else:
        tgt_h = np.ascontiguousarray(w["target"])
        tgt_d = torch.from_numpy(tgt_h).to(dev)
    torch.cuda.synchronize()
    nq, nt = src_d.shape[1], tgt_d.shape[1]

    ctx = binding.Context(gpu_index)
    es = src_d.element_size()
    ctx.set_target_device(tgt_d.data_ptr(), tgt_d.data_ptr() + nt * es, tgt_d.data_ptr() + 2 * nt * es, nt)
    ctx.set_source_device(src_d.data_ptr(), src_d.data_ptr() + nq * es, src_d.data_ptr() + 2 * nq * es, nq)

    if args.solve == "p2l":
        # point-to-plane (config 3): the target and its normals come from the depth image
        kw = dict(fx=float(w.get("fx", 468.60)), cx=float(w.get("cx", 318.27)))
        ctx.backproject_with_normals(w["depth_tgt"], binding.NORMALS_CROSS, offset=[5, 5, 5], **kw)
        assert ctx.target_size == nt
    params = binding.default_params(
        max_iterations=args.iters, fixed_iterations=1, profile=1,
        max_nn_dist=0.3 if args.solve == "p2l" else 0.75,
        solve={"reference": binding.SOLVE_REFERENCE, "kabsch": binding.SOLVE_KABSCH,
               "p2l": binding.SOLVE_POINT_TO_PLANE}[args.solve],
        nn_mode={"exact": binding.NN_EXACT, "filtered": binding.NN_FILTERED, "pruned": binding.NN_PRUNED, "grid": binding.NN_GRID}[args.nn_mode])

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.align(params)
    # per-stage device times from one extra, untimed alignment (events around every stage)
    params.profile = 2
    _, st2, _ = ctx.align(params)
    stage_ms = {"nn": st2.nn_ms_total, "reduce": st2.reduce_ms_total, "transform": st2.transform_ms_total,
                "total": st2.total_ms}
    # timed region: every 7th K1 launch is bracketed by two HIP events, the offset rotating
    # from alignment to alignment so that all 21 sweep positions are sampled evenly (an event
    # pair costs ~4 us of queue time: ~10 % of an iteration if every launch carried one)
    params.profile = 1
    params.profile_stride = 7
    sync_all()
    t0 = time.perf_counter()
    nn_ms = 0.0
    nn_launches = 0
    nn_timed = 0
    red_ms = tr_ms = 0.0
    iters_done = 0
    for _ in range(args.steps):
        T, st, rc = ctx.align(params)
        nn_ms += st.nn_ms_total
        red_ms += st.reduce_ms_total
        tr_ms += st.transform_ms_total
        nn_launches += st.nn_launches
        nn_timed += st.nn_timed_launches
        iters_done += st.iterations
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        agg = torch.tensor([iters_done, nn_launches * nq, nn_launches * nq * nt], dtype=torch.float64, device=cdev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        total_iters, total_queries, total_pairs = (float(v) for v in agg.tolist())
    else:
        total_iters, total_queries, total_pairs = float(iters_done), float(nn_launches * nq), float(nn_launches) * nq * nt

    if rank == 0:
        avg_nn_s = nn_ms / max(nn_timed, 1) / 1e3
        alg_bytes = float(nq) * nt * 12 + nq * 12 + nq * 8
        achieved = alg_bytes / avg_nn_s / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            traffic = json.load(open(tfile)).get(f"{args.workload}:{args.nn_mode}")
        out = {
            "metric": "ICP iterations/sec + NN Mpoints/sec at 307k-pt Kinect cloud, 1/2/4/8 GPU",
            "value": total_iters / elapsed,
            "unit": "iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 filter + f64 exact pair arithmetic (reference float semantics), f64 reductions",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {nq} source x {nt} target points, {args.iters} fixed ICP "
                                   f"iterations per step, solve={args.solve}, nn={args.nn_mode}",
                       "frame_pairs_per_step": world, "parallelism": f"frame-batch x{world}" if world > 1 else "1 GPU"},
            "nn_mpoints_per_s": total_queries / elapsed / 1e6,
            "nn_gpairs_per_s_wall": total_pairs / elapsed / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": {"exact": "nn_exact_kernel", "filtered": "nn_filtered_kernel<2>",
                                    "pruned": "nn_pruned_kernel<4>", "grid": "nn_grid_kernel<8,false|true>"}[args.nn_mode],
                         "avg_launch_ms": avg_nn_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes,
                         "timing": f"two HIP events recorded on the kernel's own stream immediately around every 7th "
                                   f"K1 launch of the timed region, the offset advancing with every alignment ({nn_timed} of "
                                   f"{nn_launches} launches, first sweeps in proportion; includes "
                                   "~5 us of dispatch latency per launch; the rocprofv3 --kernel-trace average of "
                                   "the same command is in profiles/)",
                         "gpairs_per_s_kernel": nq * nt / avg_nn_s / 1e9,
                         "note": "algorithmic operand bytes of the brute-force scan this kernel replaces "
                                 "(Nq*Nt*12 + Nq*20); the grid / pruned kernels return the same result while "
                                 "only looking at targets within reach, so this is not physical traffic; see "
                                 "roofline_bruteforce for the kernel that evaluates every pair"},
            "stage_ms_per_step": stage_ms,
        }
        if world == 1 and args.nn_mode in ("pruned", "grid"):
            # the same sweep by the brute-force (un-pruned) filtered kernel, for the roofline
            # of the kernel north_star names: every one of the Nq*Nt pairs is evaluated
            ctx.reset_source()
            ctx.nn(binding.NN_FILTERED, fetch=False)  # seeds
            reps = 5
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(reps):
                ctx.nn(binding.NN_FILTERED, fetch=False)
            t_bf = (time.perf_counter() - tb) / reps
            out["roofline_bruteforce"] = {
                "bound": "hbm", "kernel": "nn_filtered_kernel<2>", "achieved": alg_bytes / t_bf / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / t_bf / 1e9 / HBM_PEAK_GBS,
                "avg_launch_ms": t_bf * 1e3, "gpairs_per_s_kernel": nq * nt / t_bf / 1e9,
                "timing": "host wall clock around icpk_nn (includes one launch + sync, ~2%)",
                "traffic": (json.load(open(tfile)).get(f"{args.workload}:filtered") if os.path.exists(tfile) else None)}
            ctx.reset_source()
        if world == 1:
            # PCIe-inclusive rate (never `value`): the boundary handed host buffers, so
            # every step re-uploads both clouds (pageable memory) before aligning
            reps = max(2, min(args.steps, 5))
            params.profile = 0
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
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(src_h, tgt_h, args.cpu_sample, args.solve)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_query_sharded(args, rank, world, dev, cdev, gpu_index, dist):
    """One frame pair, queries split over the ranks (batch.align_query_sharded): per iteration
    every rank runs K1/K2 on its slice, then ONE all-reduce of 19 sums + count (RCCL, 160 bytes)
    and a replicated 3x3 solve.  The loop is host-driven (the collective sits between K2 and
    the solve), so this mode pays one stream sync + one collective latency per iteration."""
    import torch

    from icp_slam_prototype_amd import batch, binding

    w = make_workload(args.workload, 2)
    tgt = batch.broadcast_cloud(w["target"] if rank == 0 else None, 0, cdev, dist).cpu().numpy()
    nq_total = w["source"].shape[1]
    s0, cnt = batch.partition(nq_total, world, rank)
    src = np.ascontiguousarray(w["source"][:, s0:s0 + cnt])
    ctx = binding.Context(gpu_index)
    ctx.set_target(tgt)
    ctx.set_source(src)
    steps = batch.ContextSteps(ctx, {"exact": binding.NN_EXACT, "filtered": binding.NN_FILTERED,
                                     "pruned": binding.NN_PRUNED, "grid": binding.NN_GRID}[args.nn_mode])
    solve = {"reference": 0, "kabsch": 1}.get(args.solve)
    if solve is None:
        raise SystemExit("--shard queries supports --solve reference|kabsch")

    def one_step():
        ctx.reset_source()
        return batch.align_query_sharded(steps, dist, cdev, max_iterations=args.iters, solve=solve,
                                         fixed_iterations=True)

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
            "metric": "ICP iterations/sec + NN Mpoints/sec at 307k-pt Kinect cloud, 1/2/4/8 GPU",
            "value": iters / elapsed, "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32 filter + f64 exact pair arithmetic, f64 reductions", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {nq_total} source x {tgt.shape[1]} target points, {args.iters} "
                                   f"fixed ICP iterations per step, solve={args.solve}, nn={args.nn_mode}",
                       "parallelism": f"query-sharded x{world}, 1 all-reduce(160 B)/iteration"},
            "nn_mpoints_per_s": iters * nq_total / elapsed / 1e6}))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
