"""Frame-batch mode (BASELINE config 4) on the GPU, through the C ABI:
icpk_align_batch / icpk_align_batch_device advance up to ICPK_BATCH_GROUP pairs in lock step
(one launch per stage for the whole group) and must return, for every pair, exactly what
icpk_align returns for that pair alone -- transform, statistics, associations, bit for bit --
and hence the oracle's result wherever icpk_align matches it.

/root/reference does not exist on the GPU box: nothing here reads it.
"""
import numpy as np
import pytest

from icp_slam_prototype_amd import binding, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from icp_slam_prototype_amd import build

    build.build()
    c = binding.Context(0)
    yield c
    c.close()


def _single(ctx, pairs, **kw):
    """every pair on its own through icpk_align"""
    out = []
    for s, t in pairs:
        ctx.set_target(t)
        ctx.set_source(s)
        T, st, rc = ctx.align(**kw)
        idx, dist = ctx.get_associations()
        out.append((T.copy(), st, rc, idx, dist, ctx.get_source()))
    return out


def _same_stats(a, b):
    return (a.iterations, a.status, a.final_pairs, a.nn_launches) == (b.iterations, b.status, b.final_pairs, b.nn_launches) \
        and np.float32(a.final_mse).view(np.uint32) == np.float32(b.final_mse).view(np.uint32)


def _check_batch_equals_single(ctx, pairs, **kw):
    T, st, rc, assoc = ctx.align_batch(pairs, associations=True, **kw)
    ref = _single(ctx, pairs, **kw)
    assert T.shape == (len(pairs), 4, 4)
    for b, (Ts, sts, rcs, idx, dist, _) in enumerate(ref):
        assert np.array_equal(T[b].view(np.uint32), Ts.view(np.uint32)), f"pair {b}: transform"
        assert _same_stats(st[b], sts), f"pair {b}: stats"
        assert np.array_equal(assoc[b][0], idx), f"pair {b}: association indices"
        assert np.array_equal(assoc[b][1].view(np.uint32), dist.view(np.uint32)), f"pair {b}: association distances"
    assert rc == max([r[2] for r in ref] + [0])
    return T, st, ref


def test_one_ranks_share_of_config4_equals_single_pairs(ctx):
    """BASELINE config 4, the share of one GPU at 8 GPUs: 8 distinct config-2 pairs (640x480,
    30 % valid, seeds 100..107, ~92k x 92k points each, ragged sizes), 20 fixed iterations."""
    pairs = []
    for seed in range(100, 108):
        p = synth.kinect_pair(480, 640, valid=0.30, seed=seed)
        pairs.append((p["source"], p["target"]))
    assert len({s.shape[1] for s, _ in pairs}) > 1  # the sizes differ from pair to pair
    T, st, ref = _check_batch_equals_single(ctx, pairs, max_iterations=20, fixed_iterations=1)
    assert all(s.iterations == 20 and s.nn_launches == 21 for s in st)
    # the alignments recover the 2 degree / 3 cm motion of the synthetic camera (reference flavour
    # accumulates only the rotation; icp.cpp:227-233)
    for b in range(8):
        assert abs(np.degrees(np.arccos(np.clip((np.trace(T[b][:3, :3]) - 1) / 2, -1, 1))) - 2.0) < 0.5


@pytest.mark.parametrize("solve", [binding.SOLVE_REFERENCE, binding.SOLVE_KABSCH])
def test_quarter_size_pairs_against_the_oracle(ctx, oracle, solve):
    pairs = []
    for k, (rows, cols) in enumerate([(240, 320), (120, 160), (200, 300), (240, 320), (60, 80)]):
        p = synth.kinect_pair(rows, cols, valid=0.5, seed=300 + k)
        pairs.append((p["source"], p["target"]))
    T, st, rc, assoc = ctx.align_batch(pairs, associations=True, solve=solve, max_iterations=5, fixed_iterations=1)
    assert rc == 0
    for b, (s, t) in enumerate(pairs):
        o = oracle.align(s, t, max_iterations=5, solve=solve, sum_order=1, fixed_iterations=True,
                         threads=oracle.max_threads())
        # north_star: bit-exact correspondence indices, transform within 1e-5 Frobenius
        assert np.array_equal(assoc[b][0], o["idx"]), f"pair {b}"
        assert np.array_equal(assoc[b][1].view(np.uint32), o["dist"].view(np.uint32)), f"pair {b}"
        assert np.linalg.norm(T[b].astype(np.float64) - o["T"].astype(np.float64)) < 1e-5
        assert st[b].iterations == o["iterations"] == 5 and st[b].final_pairs == o["final_pairs"]


def test_more_groups_than_slot_sets_and_ragged_tail(monkeypatch):
    """19 pairs with 4 in lock step: 5 groups over the two alternating slot sets (every slot is
    re-used at least once, the last group holds 3 pairs), sizes from 1 to 6000 points."""
    monkeypatch.setenv("ICPK_BATCH_GROUP", "4")
    c = binding.Context(0)
    try:
        rng = np.random.default_rng(42)
        pairs = []
        for k in range(19):
            n = [1, 2, 3, 64, 65, 700, 6000, 257, 1024, 1025][k % 10] + k
            p = synth.frustum_pair(n, seed=500 + k, rot_deg=(0, 1.0 + 0.1 * k, 0), shift=(0.01, 0, 0.002 * k))
            m = max(1, int(n * rng.uniform(0.5, 1.5)))
            q = synth.frustum_pair(m, seed=900 + k)
            pairs.append((p["source"], q["target"] if k % 3 == 0 else p["target"]))
        for kw in (dict(max_iterations=6, fixed_iterations=1),
                   dict(max_iterations=16, solve=binding.SOLVE_KABSCH)):  # threshold exit: pairs stop at different iterations
            _check_batch_equals_single(c, pairs, **kw)
    finally:
        c.close()


def test_threshold_exit_fallback_and_bad_pairs(ctx, oracle):
    """Per-pair loop control inside a lock-step group: one pair converges early (icp.cpp:155),
    one falls back to the last motion (< 3 pairs, icp.cpp:163-182), one has an empty source, one an
    empty target (ICPK_E_EMPTY_TARGET for that pair only); the others are unaffected."""
    a = synth.frustum_pair(800, seed=5, rot_deg=(0, 0.5, 0), shift=(0.002, 0, 0))
    b = synth.frustum_pair(3000, seed=6, rot_deg=(0, 3, 0), shift=(0.02, 0.01, 0))
    far = a["source"] + np.float32(100)
    far[:, :2] = a["target"][:, :2] + np.float32(0.05)
    empty = np.zeros((3, 0), np.float32)
    pairs = [(a["source"], a["target"]), (far, a["target"]), (empty, b["target"]), (b["source"], empty),
             (b["source"], b["target"])]
    lt = np.array([1, 2, 3], np.float32)
    kw = dict(solve=binding.SOLVE_KABSCH, last_translation=lt)
    T, st, rc, assoc = ctx.align_batch(pairs, associations=True, **kw)
    assert rc == binding.E_EMPTY_TARGET  # first negative status wins
    assert [s.status for s in st] == [0, binding.W_TOO_FEW_PAIRS, 0, binding.E_EMPTY_TARGET, 0]
    assert np.array_equal(T[3], np.eye(4, dtype=np.float32))
    good = [0, 1, 4]
    ref = _single(ctx, [pairs[i] for i in good], **kw)
    for i, (Ts, sts, rcs, idx, dist, _) in zip(good, ref):
        assert np.array_equal(T[i], Ts) and _same_stats(st[i], sts)
        assert np.array_equal(assoc[i][0], idx)
    assert st[0].iterations < 16 and st[0].final_mse <= 1e-4 and st[4].iterations >= st[0].iterations
    assert st[2].iterations == 0 and st[2].final_pairs == 0
    o = oracle.align(pairs[0][0], pairs[0][1], solve=1, sum_order=1, threads=4)
    assert st[0].iterations == o["iterations"]
    assert np.linalg.norm(T[0].astype(np.float64) - o["T"].astype(np.float64)) < 1e-5
    # n_pairs = 0 is a no-op
    T0, st0, rc0 = ctx.align_batch([])
    assert rc0 == 0 and T0.shape == (0, 4, 4)


class _Hip:
    """hipMalloc / hipMemcpy through the HIP runtime libicpk.so already mapped (no torch: it
    ships its own copy of the runtime, and two runtimes in one process do not see the GPU)."""

    def __init__(self):
        import ctypes as C

        self.C = C
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.ptrs = []

    def upload(self, a):
        a = np.ascontiguousarray(a, np.float32)
        p = self.C.c_void_p()
        assert self.lib.hipMalloc(self.C.byref(p), max(a.nbytes, 4)) == 0
        assert self.lib.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0  # hipMemcpyHostToDevice
        self.ptrs.append(p)
        return p.value

    def download(self, ptr, shape):
        out = np.empty(shape, np.float32)
        assert self.lib.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    def free(self):
        for p in self.ptrs:
            self.lib.hipFree(p)
        self.ptrs = []


def test_device_resident_pairs(ctx):
    """icpk_align_batch_device: the clouds are already in HBM (bench.py's config-4 path)."""
    hip = _Hip()
    try:
        pairs, dev = [], []
        for k in range(5):
            p = synth.kinect_pair(120, 160, valid=0.6, seed=700 + k)
            pairs.append((p["source"], p["target"]))
            dev.append((hip.upload(p["source"]), p["source"].shape[1], hip.upload(p["target"]), p["target"].shape[1]))
        kw = dict(max_iterations=7, fixed_iterations=1)
        Td, std, rcd = ctx.align_batch_device(dev, **kw)
        Th, sth, rch = ctx.align_batch(pairs, **kw)
        assert rcd == rch == 0 and np.array_equal(Td, Th)
        assert all(_same_stats(a, b) for a, b in zip(std, sth))
        for (sp, ns, tp, nt), (sh, th) in zip(dev, pairs):  # the caller's device buffers are read-only
            assert np.array_equal(hip.download(sp, (3, ns)), sh) and np.array_equal(hip.download(tp, (3, nt)), th)
    finally:
        hip.free()


@pytest.mark.parametrize("setup", ["0", "1", "2", "3"])  # 3: recorded, then replayed pair by pair (the fall-back)
def test_ragged_pairs_on_fresh_contexts_with_and_without_batched_setup(monkeypatch, setup):
    """The set-up launches of a lock-step group are recorded and issued once per step for all pairs
    (SetupRecorder, ICPK_BATCH_SETUP).  Recorded arguments hold pointers, so nothing they point at may be
    re-allocated between recording and the flush: fresh contexts (every capacity zero), sources LARGER than
    their targets (the queries' counting sort outgrowing the scratch the target's sort was recorded with
    is what a soak run caught), sizes that differ pair to pair, several groups, host and device buffers."""
    monkeypatch.setenv("ICPK_BATCH_SETUP", setup)
    monkeypatch.setenv("ICPK_BATCH_GROUP", "4")
    rng = np.random.default_rng(5)
    pairs = []
    for k in range(11):
        nt, nq = int(rng.integers(300, 4000)), int(rng.integers(300, 4000))
        if k % 2 == 0:
            nq = nt + int(rng.integers(2000, 9000)) + 1500 * k  # ever larger sources: re-allocation in every group
        tgt = (rng.uniform(-1, 1, (3, nt)) + 5).astype(np.float32)
        src = (tgt[:, rng.integers(0, nt, nq)] + rng.normal(0, 0.01, (3, nq))).astype(np.float32)
        pairs.append((src, tgt))
    kw = dict(max_iterations=5, fixed_iterations=1, max_nn_dist=0.2)
    with binding.Context(0) as single:
        want = _single(single, pairs, **kw)
    hip = _Hip()
    try:
        dev = [(hip.upload(s), s.shape[1], hip.upload(t), t.shape[1]) for s, t in pairs]
        with binding.Context(0) as a:  # device buffers first: every slot capacity starts at zero
            Td, std, rcd = a.align_batch_device(dev, **kw)
        with binding.Context(0) as b:
            Th, sth, rch, assoc = b.align_batch(pairs, associations=True, **kw)
    finally:
        hip.free()
    for k, (Tw, stw, rcw, idx, dist, _) in enumerate(want):
        for T, st in ((Td[k], std[k]), (Th[k], sth[k])):
            assert np.array_equal(T, Tw), k
            assert (st.iterations, st.status, st.final_pairs) == (stw.iterations, stw.status, stw.final_pairs), k
        assert np.array_equal(assoc[k][0], idx) and np.array_equal(assoc[k][1].view(np.uint32), dist.view(np.uint32)), k


def test_profile_in_the_frame_batch_mode_times_one_launch_per_group(monkeypatch):
    """params.profile = 1 in the lock-step path: one batched NN launch per group is bracketed by HIP events and
    booked on the group's first pair; results are the unprofiled ones."""
    monkeypatch.setenv("ICPK_BATCH_GROUP", "3")
    pairs = []
    for k in range(7):
        p = synth.kinect_pair(60, 80, valid=0.7, seed=900 + k)
        pairs.append((p["source"], p["target"]))
    with binding.Context(0) as c:
        T0, st0, rc0 = c.align_batch(pairs, max_iterations=4, fixed_iterations=1)
        T1, st1, rc1 = c.align_batch(pairs, max_iterations=4, fixed_iterations=1, profile=1)
    assert rc0 == rc1 == 0 and np.array_equal(T0, T1)
    for k in range(7):
        assert (st1[k].iterations, st1[k].final_pairs) == (st0[k].iterations, st0[k].final_pairs)
        if k % 3 == 0:  # first pair of a group of 3
            assert st1[k].nn_timed_launches == 1 and 0.0 < st1[k].nn_ms_total < 50.0
        else:
            assert st1[k].nn_timed_launches == 0 and st1[k].nn_ms_total == 0.0
        assert st0[k].nn_timed_launches == 0


def test_other_kernels_and_flavours_run_pair_by_pair(ctx):
    """Settings outside the lock-step path (another NN kernel, host loop) still give the same
    results through icpk_align_batch, one pair after the other."""
    pairs = []
    for k in range(3):
        p = synth.kinect_pair(90, 120, valid=0.7, seed=800 + k)
        pairs.append((p["source"], p["target"]))
    base, _, _ = ctx.align_batch(pairs, max_iterations=4, fixed_iterations=1)
    for kw in (dict(nn_mode=binding.NN_PRUNED), dict(nn_mode=binding.NN_EXACT), dict(host_loop=1)):
        T, st, rc = ctx.align_batch(pairs, max_iterations=4, fixed_iterations=1, **kw)
        assert rc == 0 and np.array_equal(T, base)
