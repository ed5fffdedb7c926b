"""Frame-batch mode: independent frame pairs sharded
block-wise over one process per GPU; no data-path collective per iteration.

Collectives (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo"
in the CPU tests):
  * one broadcast of the shared target cloud (key frame) from rank 0: a 3 x Nt
    float32 SoA, 1.1 MB at Nt = 92k -- a single latency-bound message;
  * one all_gather of the per-rank results (B_r x 16 floats + 4 stats).
The alignment itself is injected (`align_fn`): bench.py passes the C-ABI context;
the gloo tests pass a CPU stand-in so the sharding/collective logic is covered
without a GPU.
"""
import numpy as np


def partition(n_items, world, rank):
    """Block-wise shard: returns (start, count) of rank's contiguous slice."""
    if world <= 0 or not (0 <= rank < world) or n_items < 0:
        raise ValueError("bad partition arguments")
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def broadcast_cloud(cloud, src, device, dist):
    """Broadcast a (3, N) float32 cloud from rank `src`; other ranks pass None.
    Returns a torch tensor on `device` holding the cloud on every rank."""
    import torch

    rank = dist.get_rank()
    n = torch.tensor([0 if cloud is None else int(cloud.shape[1])], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    if rank == src:
        t = torch.as_tensor(np.ascontiguousarray(cloud, np.float32)).to(device)
    else:
        t = torch.empty((3, int(n.item())), dtype=torch.float32, device=device)
    dist.broadcast(t, src=src)
    return t


def gather_results(T_local, stats_local, n_total, device, dist):
    """All-gather per-rank results into global frame order.
    T_local: (b, 4, 4) float32, stats_local: (b, 4) float32 [iterations, status,
    pairs, mse].  Returns (n_total,4,4) and (n_total,4) numpy arrays on every rank."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    bmax = max(partition(n_total, world, r)[1] for r in range(world))
    buf = torch.zeros((bmax, 20), dtype=torch.float32, device=device)
    b = T_local.shape[0]
    if b:
        buf[:b, :16] = torch.as_tensor(np.ascontiguousarray(T_local, np.float32).reshape(b, 16)).to(device)
        buf[:b, 16:] = torch.as_tensor(np.ascontiguousarray(stats_local, np.float32)).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    T = np.zeros((n_total, 4, 4), np.float32)
    S = np.zeros((n_total, 4), np.float32)
    for r in range(world):
        s, c = partition(n_total, world, r)
        a = out[r][:c].cpu().numpy()
        T[s:s + c] = a[:, :16].reshape(c, 4, 4)
        S[s:s + c] = a[:, 16:]
    return T, S


def align_frame_batch(make_source, n_frames, target_on_rank0, align_fn, device, dist):
    """Aligns n_frames source frames against one shared target.
    make_source(i) -> (3, N) float32 source cloud of global frame i (only called for
    this rank's frames); target_on_rank0: (3, Nt) cloud on rank 0, None elsewhere;
    align_fn(source_np, target_tensor) -> (T (4,4), iterations, status, pairs, mse).
    Returns the gathered (T, stats) for all frames, identical on every rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    tgt = broadcast_cloud(target_on_rank0, 0, device, dist)
    start, count = partition(n_frames, world, rank)
    T_local = np.zeros((count, 4, 4), np.float32)
    S_local = np.zeros((count, 4), np.float32)
    for k in range(count):
        T, it, status, pairs, mse = align_fn(make_source(start + k), tgt)
        T_local[k] = T
        S_local[k] = (it, status, pairs, mse)
    return gather_results(T_local, S_local, n_frames, device, dist)


# ---------------------------------------------------------------------------------------
# Query-sharded alignment of ONE large pair ("single huge pair"): the target
# is broadcast once, every rank keeps Nq/G queries, and the only per-iteration exchange is
# one all-reduce of 19 doubles + the pair count (152 + 8 bytes: pure latency); the 3x3
# solve is replicated on every rank, so all ranks apply the same transform.
# ---------------------------------------------------------------------------------------
class ContextSteps:
    """The three steps of an iteration on a C-ABI context (icpk_nn / icpk_reduce /
    icpk_transform_source)."""

    def __init__(self, ctx, nn_mode=None):
        from . import binding

        self.ctx = ctx
        self.nn_mode = binding.NN_GRID if nn_mode is None else nn_mode

    def nn(self):
        self.ctx.nn(self.nn_mode, fetch=False)

    def reduce(self, max_dist):
        return self.ctx.reduce(max_dist)

    def transform(self, R, t):
        self.ctx.transform_source(R, t)


def _mul3f(A, B):
    A = np.asarray(A, np.float64)
    B = np.asarray(B, np.float64)
    return ((A[:, 0:1] * B[0:1, :] + A[:, 1:2] * B[1:2, :]) + A[:, 2:3] * B[2:3, :]).astype(np.float32)


def _inv3f(R):
    a, b, c, d, e, f, g, h, i = np.asarray(R, np.float64).reshape(9)
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    s = 1.0 / det
    return np.array([(e * i - f * h) * s, (c * h - b * i) * s, (b * f - c * e) * s,
                     (f * g - d * i) * s, (a * i - c * g) * s, (c * d - a * f) * s,
                     (d * h - e * g) * s, (b * g - a * h) * s, (a * e - b * d) * s]).astype(np.float32).reshape(3, 3)


def align_query_sharded(steps, dist, device, max_iterations=16, threshold=1e-4, max_nn_dist=0.75, min_pairs=3,
                        solve=1, fixed_iterations=False):
    """Runs the ICP loop on this rank's slice of the queries (already uploaded as the
    source of `steps`, target already set on every rank).  solve: 0 reference flavour
    (icp.cpp:199-246), 1 Kabsch (rigid_transform_3D.py).  Returns (T (4,4) float32,
    iterations, total pairs, mse) -- identical on every rank."""
    import torch

    from . import binding

    def global_sums():
        sums, cnt = steps.reduce(max_nn_dist)
        buf = torch.zeros(20, dtype=torch.float64, device=device)
        buf[:19] = torch.as_tensor(sums)
        buf[19] = float(cnt)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)  # RCCL all-reduce: 160 bytes per iteration
        out = buf.cpu().numpy()
        return out[:19], int(round(out[19]))

    def mse_of(sums, n):
        if n <= 0:
            return np.float32(0)
        m = np.float32(sums[12] / float(n))
        return np.float32(np.float64(m) * np.float64(m))

    steps.nn()
    sums, n = global_sums()
    mse = mse_of(sums, n)
    Trot = np.eye(3, dtype=np.float32)
    offset = np.zeros(3, np.float32)
    Tk = np.eye(4)[:3].copy()
    i = 0
    while (fixed_iterations or mse > np.float32(threshold)) and i < max_iterations:
        if n < min_pairs:
            break
        if solve == 0:
            R = binding.solve_reference(sums[:9].astype(np.float32).reshape(3, 3))
            Trot = R.copy() if i == 0 else _mul3f(R, Trot)
            offset = (sums[9:12] / float(n)).astype(np.float32)
            steps.transform(_inv3f(R), -offset)
        else:
            Rd, td = binding.solve_kabsch(n, sums[13:16], sums[16:19], sums[:9].reshape(3, 3).T)
            Rf, tf = Rd.astype(np.float32), td.astype(np.float32)
            steps.transform(Rf, tf)
            Tn = Rf.astype(np.float64) @ Tk
            Tn[:, 3] += tf.astype(np.float64)
            Tk = Tn
        steps.nn()
        sums, n = global_sums()
        mse = mse_of(sums, n)
        i += 1
    T = np.eye(4, dtype=np.float32)
    if solve == 0:
        T[:3, :3] = Trot
        T[:3, 3] = offset
    else:
        T[:3, :] = Tk.astype(np.float32)
    return T, i, n, mse
