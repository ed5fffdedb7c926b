"""Multi-GPU modes, host side above the C ABI.

Frame-batch mode -- the path shards over independent frame pairs (the frame-pair
formulation of icp.cpp:541-563): pairs are block-partitioned over one process per GPU, every
rank aligns its block with icpk_align_batch(_device) (lock-step groups on its GPU), and there is
NO data-path collective per iteration.  The collectives are
  * one broadcast of a shared target cloud (key frame), 3 x Nt float32 -- a single
    latency-bound message over xGMI -- when all pairs share one target;
  * one all-gather of the results (20 floats per pair).

Two transports implement them:
  * RcclComm  -- the C ABI's own RCCL communicator (icpk_comm_* in include/icpk.h,
                 csrc/icpk_comm.cpp): what a C++ host uses, and what bench.py uses on GPUs;
  * TorchComm -- torch.distributed ("gloo" in the CPU tests and the 1-GPU rehearsal, where
                 RCCL refuses two ranks on one device; "nccl" is RCCL too).
The alignment itself is injected where a test needs a CPU stand-in.
"""
import numpy as np


def partition(n_items, world, rank):
    """Block-wise shard: returns (start, count) of rank's contiguous slice (the same rule as
    icpk_comm_partition)."""
    if world <= 0 or not (0 <= rank < world) or n_items < 0:
        raise ValueError("bad partition arguments")
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def stats_rows(stats):
    """list of binding.Stats -> (b, 4) float64 rows [iterations, status, pairs, mse]"""
    return np.array([[s.iterations, s.status, s.final_pairs, s.final_mse] for s in stats], np.float64).reshape(-1, 4)


class TorchComm:
    """The collectives over torch.distributed (host or device tensors on `device`)."""

    kind = "torch.distributed"

    def __init__(self, dist, device):
        self.dist, self.device = dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.kind = f"torch.distributed({dist.get_backend()})"

    def broadcast_cloud(self, cloud, src=0):
        """(3, N) float32 cloud from rank `src` (others pass None) -> torch tensor on every rank."""
        import torch

        n = torch.tensor([0 if cloud is None else int(cloud.shape[1])], dtype=torch.int64, device=self.device)
        self.dist.broadcast(n, src=src)
        if self.rank == src:
            t = torch.as_tensor(np.ascontiguousarray(cloud, np.float32)).to(self.device)
        else:
            t = torch.empty((3, int(n.item())), dtype=torch.float32, device=self.device)
        self.dist.broadcast(t, src=src)
        return t

    def gather_results(self, T_local, S_local, n_total):
        """T_local (b, 4, 4), S_local (b, 4) of this rank's block -> (n_total, 4, 4), (n_total, 4) float64
        in global pair order, identical on every rank.  (iterations, status and pairs cross the collective as int32
        bit patterns in their float slots, like icpk_comm_gather_results: exact for any cloud size.)"""
        import torch

        bmax = max(partition(n_total, self.world, r)[1] for r in range(self.world))
        buf = torch.zeros((bmax, 20), dtype=torch.float32, device=self.device)
        b = T_local.shape[0]
        if b:
            buf[:b, :16] = torch.as_tensor(np.ascontiguousarray(T_local, np.float32).reshape(b, 16)).to(self.device)
            S_local = np.asarray(S_local, np.float64).reshape(b, 4)
            row = np.empty((b, 4), np.float32)
            row[:, :3] = np.rint(S_local[:, :3]).astype(np.int32).view(np.float32)
            row[:, 3] = S_local[:, 3]
            buf[:b, 16:] = torch.as_tensor(row).to(self.device)
        out = [torch.empty_like(buf) for _ in range(self.world)]
        self.dist.all_gather(out, buf)
        T = np.zeros((n_total, 4, 4), np.float32)
        S = np.zeros((n_total, 4), np.float64)
        for r in range(self.world):
            s, c = partition(n_total, self.world, r)
            a = np.ascontiguousarray(out[r][:c].cpu().numpy())
            T[s:s + c] = a[:, :16].reshape(c, 4, 4)
            S[s:s + c, :3] = np.ascontiguousarray(a[:, 16:19]).view(np.int32)
            S[s:s + c, 3] = a[:, 19]
        return T, S

    def allreduce_sums(self, sums, count):
        import torch

        buf = torch.zeros(len(sums) + 1, dtype=torch.float64, device=self.device)
        buf[:-1] = torch.as_tensor(np.asarray(sums, np.float64))
        buf[-1] = float(count)
        self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
        out = buf.cpu().numpy()
        return out[:-1].copy(), int(round(out[-1]))

    def barrier(self):
        self.dist.barrier()


class RcclComm:
    """The C ABI's RCCL communicator on a binding.Context.  `exchange(id_or_None)` must return
    rank 0's 128-byte id on every rank (any side channel: torch store, MPI, a file)."""

    kind = "icpk_comm (RCCL behind the C ABI)"

    def __init__(self, ctx, rank, world, exchange):
        from . import binding

        self.ctx, self.rank, self.world = ctx, rank, world
        uid = exchange(binding.comm_unique_id() if rank == 0 else None)
        ctx.comm_init(uid, rank, world)

    def broadcast_target(self, root=0):
        """the root context's target cloud becomes the target of every rank's context"""
        self.ctx.comm_broadcast_target(root)

    def gather_results(self, T_local, stats_local, n_total):
        return self.ctx.comm_gather_results(T_local, stats_local, n_total)

    def allreduce_sums(self, sums, count):
        return self.ctx.comm_allreduce_sums(sums, count)

    def align_query_sharded(self, params=None, **kw):
        """the device-side query-sharded loop (icpk_align_query_sharded): the context's source is this rank's slice of
        the queries, the target the same on every rank; one in-stream all-reduce per iteration.
        Returns (T, iterations, total pairs, mse, status) like align_query_sharded below."""
        T, st, rc = self.ctx.align_query_sharded(params, **kw)
        return T, st.iterations, st.final_pairs, st.final_mse, rc

    def barrier(self):
        self.ctx.comm_barrier()

    def close(self):
        self.ctx.comm_destroy()


# ---- frame-batch mode -------------------------------------------------------------------
def align_pair_batch(n_pairs, make_pair, align_batch_fn, comm):
    """BASELINE config 4: n_pairs independent frame pairs, block-partitioned over the ranks.
    make_pair(i) -> (source (3, Ns), target (3, Nt)) of global pair i (called for this rank's
    pairs only); align_batch_fn(list_of_pairs) -> (T (b, 4, 4), S (b, 4) rows
    [iterations, status, pairs, mse]) -- on a GPU: Context.align_batch + stats_rows.
    Returns the gathered (T, S) of all pairs, identical on every rank."""
    start, count = partition(n_pairs, comm.world, comm.rank)
    pairs = [make_pair(start + k) for k in range(count)]
    if count:
        T_local, S_local = align_batch_fn(pairs)
    else:
        T_local, S_local = np.zeros((0, 4, 4), np.float32), np.zeros((0, 4), np.float32)
    return comm.gather_results(np.asarray(T_local, np.float32), np.asarray(S_local, np.float64), n_pairs)


# torch.distributed spellings kept for callers that hold a process group
def broadcast_cloud(cloud, src, device, dist):
    return TorchComm(dist, device).broadcast_cloud(cloud, src)


def gather_results(T_local, stats_local, n_total, device, dist):
    return TorchComm(dist, device).gather_results(T_local, stats_local, n_total)


def align_frame_batch(make_source, n_frames, target_on_rank0, align_fn, device, dist):
    """Scan-to-key-frame variant: n_frames source frames against ONE target broadcast from rank 0.
    make_source(i) -> (3, N) source of global frame i (this rank's frames only);
    align_fn(source_np, target_tensor) -> (T (4,4), iterations, status, pairs, mse).
    Returns the gathered (T, stats) for all frames, identical on every rank."""
    comm = TorchComm(dist, device)
    tgt = comm.broadcast_cloud(target_on_rank0, 0)
    start, count = partition(n_frames, comm.world, comm.rank)
    T_local = np.zeros((count, 4, 4), np.float32)
    S_local = np.zeros((count, 4), np.float64)
    for k in range(count):
        T, it, status, pairs, mse = align_fn(make_source(start + k), tgt)
        T_local[k] = T
        S_local[k] = (it, status, pairs, mse)
    return comm.gather_results(T_local, S_local, n_frames)


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


def align_query_sharded(steps, comm, device=None, max_iterations=16, threshold=1e-4, max_nn_dist=0.75, min_pairs=3,
                        solve=1, fixed_iterations=False, last_rotation=None, last_translation=None):
    """Runs the ICP loop on this rank's slice of the queries (already uploaded as the
    source of `steps`, target already set on every rank).  solve: 0 reference flavour
    (icp.cpp:199-246), 1 Kabsch (rigid_transform_3D.py).  comm: RcclComm / TorchComm (or a
    torch.distributed module together with `device`).
    Returns (T (4,4) float32, iterations, total pairs, mse, status) -- identical on every rank;
    status as icpk_align: 0, or W_TOO_FEW_PAIRS (1) after the fallback of icp.cpp:163-182."""
    from . import binding

    if not hasattr(comm, "allreduce_sums"):  # a torch.distributed module
        comm = TorchComm(comm, device)

    def global_sums():
        sums, cnt = steps.reduce(max_nn_dist)
        return comm.allreduce_sums(sums, cnt)  # 160 bytes per iteration

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
    status = binding.OK
    i = 0
    while (fixed_iterations or mse > np.float32(threshold)) and i < max_iterations:
        if n < min_pairs:
            # icp.cpp:163-182: fewer than 3 associations -- every rank re-applies the caller's
            # last motion to its queries and the loop ends (same as icpk_align's fallback)
            lr = np.eye(3, dtype=np.float32) if last_rotation is None else np.asarray(last_rotation, np.float32).reshape(3, 3)
            lt = np.zeros(3, np.float32) if last_translation is None else np.asarray(last_translation, np.float32).reshape(3)
            steps.transform(lr, lt)
            offset = (-lt).astype(np.float32)
            status = binding.W_TOO_FEW_PAIRS
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
    return T, i, n, mse, status
