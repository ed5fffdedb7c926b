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
