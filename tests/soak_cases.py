"""Seeded soak cases shared by tools/soak_grid.py, tools/soak_align.py, tools/soak_batch.py (long runs, builder side)
and tests/test_gpu_soak.py (a bounded slice of each inside `-m gpu`, so that the driver runs them too).

Every case compares the product's default path (grid sweep, device loop, lock-step groups) with an independent one:
the exact kernel (literal arithmetic of icp.cpp:566-620 per pair, no spatial index) and the host loop, or the same
pairs one by one.  Bit for bit.  Each function returns (cases_run, mismatches) and stops early after `budget_s`."""
import os
import time

import numpy as np

from icp_slam_prototype_amd import binding, synth

KINDS = ["uniform", "clusters", "line", "plane_lattice", "duplicates", "tiny", "huge"]


def fuzz_cloud(rng, n, kind):
    if kind == "uniform":
        return rng.uniform(-3, 3, (3, n))
    if kind == "clusters":
        c = rng.uniform(-3, 3, (3, 12))
        return c[:, rng.integers(0, 12, n)] + rng.normal(0, 0.02, (3, n))
    if kind == "line":
        t = rng.uniform(0, 1, n)
        return np.stack([t * 4 - 2, 0.5 * t, np.full(n, 1.0)]) + rng.normal(0, 1e-4, (3, n))
    if kind == "plane_lattice":
        k = int(np.ceil(np.sqrt(n)))
        u, v = np.meshgrid(np.arange(k), np.arange(k))
        return np.stack([u.ravel()[:n] * 0.01, v.ravel()[:n] * 0.01, np.full(n, 2.0)])
    if kind == "duplicates":
        base = rng.uniform(-1, 1, (3, max(n // 7, 1)))
        return base[:, rng.integers(0, base.shape[1], n)]
    if kind == "tiny":
        return rng.uniform(-1, 1, (3, n)) * 1e-6
    return rng.uniform(-1, 1, (3, n)) * 1e4  # "huge"


def soak_grid(ctx, n_cases, seed0=5000, budget_s=None, oracle=None, log=None):
    """grid scan vs exact kernel (or the CPU oracle) over a first, unseeded sweep and two seeded ones, the source
    moving in between (icp.cpp:541-593)."""
    t0 = time.time()
    bad = done = 0
    for c in range(n_cases):
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        rng = np.random.default_rng(seed0 + c)
        kt, ks = KINDS[rng.integers(0, 7)], KINDS[rng.integers(0, 7)]
        nt, nq = int(rng.integers(1, 60000)), int(rng.integers(1, 40000))
        if oracle is not None:
            nt, nq = 1 + nt // 3, 1 + nq // 3
        off = rng.uniform(-10, 10, (3, 1))
        scale = float(10.0 ** rng.uniform(-2, 1))
        tgt = (fuzz_cloud(rng, nt, kt) * scale + off).astype(np.float32)
        src = (fuzz_cloud(rng, nq, ks) * scale + off + rng.normal(0, 0.01 * scale, (3, 1))).astype(np.float32)
        if rng.random() < 0.3:
            tgt = tgt[:, rng.permutation(nt)]
        ctx.set_target(tgt)
        ctx.set_source(src)
        cur = src
        for sweep in range(3):
            if oracle is not None:
                ie, de = oracle.nn_bruteforce(cur, tgt, threads=oracle.max_threads())
            else:
                ie, de = ctx.nn(binding.NN_EXACT)
            if sweep == 0:
                ctx.reset_source()  # first grid sweep unseeded (expanding search)
            ig, dg = ctx.nn(binding.NN_GRID)
            ok = np.array_equal(ie, ig) and np.array_equal(de.view(np.uint32), dg.view(np.uint32))
            if not ok:
                bad += 1
                if log:
                    log(f"MISMATCH case {c} {kt} {ks} {nt} {nq} sweep {sweep} {int((ie != ig).sum())}")
            R = synth.rot_xyz_deg(*rng.uniform(-2, 2, 3)).astype(np.float32)
            tr = (rng.normal(0, 0.02, 3) * scale).astype(np.float32)
            ctx.transform_source(R, tr)
            if oracle is not None:
                cur = oracle.transform_points(cur, R, tr)
        done += 1
        if log and c % 20 == 19:
            log(f"{c + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s")
    return done, bad


def soak_align(ctx, n_cases, seed0=9000, budget_s=None, log=None):
    """whole alignments (icp.cpp:155-258): grid scan + device loop vs exact kernel + host loop -- transform, status,
    iterations, pair count, mse, associations, moved source."""
    t0 = time.time()
    bad = done = 0
    for c in range(n_cases):
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        rng = np.random.default_rng(seed0 + c)
        if rng.random() < 0.5:
            p = synth.kinect_pair(rows=int(rng.integers(40, 200)), cols=int(rng.integers(60, 260)), valid=float(rng.uniform(0.2, 1.0)),
                                  seed=int(rng.integers(0, 1 << 30)), rot_deg=tuple(rng.uniform(-3, 3, 3)),
                                  shift=tuple(rng.uniform(-0.05, 0.05, 3)))
            src, tgt = p["source"], p["target"]
        else:
            nt, nq = int(rng.integers(50, 30000)), int(rng.integers(50, 20000))
            tgt = (fuzz_cloud(rng, nt, "clusters" if rng.random() < 0.5 else "uniform") + 5).astype(np.float32)
            src = (tgt[:, rng.integers(0, nt, nq)] + rng.normal(0, 0.02, (3, nq))).astype(np.float32)
        if min(src.shape[1], tgt.shape[1]) < 10:
            continue
        solve = int(rng.integers(0, 2))
        kw = dict(solve=solve, max_iterations=int(rng.integers(1, 12)), max_nn_dist=float(rng.choice([0.75, 0.1, 0.03])))
        if rng.random() < 0.5:
            kw["fixed_iterations"] = 1
        else:
            kw["threshold"] = float(10 ** rng.uniform(-6, -3))
        res = []
        for mode, host_loop in ((binding.NN_GRID, 0), (binding.NN_EXACT, 1)):
            ctx.set_target(tgt)
            ctx.set_source(src)
            T, st, rc = ctx.align(nn_mode=mode, host_loop=host_loop, **kw)
            idx, dist = ctx.get_associations()
            res.append((T.copy(), rc, st.iterations, st.final_pairs, st.final_mse, idx, dist, ctx.get_source()))
        a, b = res
        ok = (np.array_equal(a[0], b[0]) and a[1:5] == b[1:5] and np.array_equal(a[5], b[5]) and
              np.array_equal(a[6].view(np.uint32), b[6].view(np.uint32)) and np.array_equal(a[7].view(np.uint32), b[7].view(np.uint32)))
        if not ok:
            bad += 1
            if log:
                log(f"MISMATCH case {c} {kw} {src.shape} {tgt.shape} {a[1:5]} {b[1:5]}")
        done += 1
        if log and c % 50 == 49:
            log(f"{c + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s")
    return done, bad


def soak_batch(n_batches, seed0=12000, budget_s=None, device=False, log=None, groups=(1, 2, 3, 5, 8, 16)):
    """random batches through icpk_align_batch against the same pairs one by one through icpk_align: random group
    sizes, ragged pair sizes, both flavours, threshold exits, far-apart pairs (fallback), empty sources.
    Returns (pairs_run, mismatches)."""
    if device:
        import torch
    single = binding.Context(0)
    ctxs = {}
    t0 = time.time()
    bad = pairs_done = 0
    saved = os.environ.get("ICPK_BATCH_GROUP")
    try:
        for c in range(n_batches):
            if budget_s is not None and time.time() - t0 > budget_s:
                break
            rng = np.random.default_rng(seed0 + c)
            g = int(rng.choice(list(groups)))
            if g not in ctxs:
                os.environ["ICPK_BATCH_GROUP"] = str(g)
                ctxs[g] = binding.Context(0)
            ctx = ctxs[g]
            n = int(rng.integers(1, 3 * g + 2))
            pairs = []
            for k in range(n):
                u = rng.random()
                if u < 0.45:
                    p = synth.kinect_pair(rows=int(rng.integers(20, 160)), cols=int(rng.integers(30, 200)),
                                          valid=float(rng.uniform(0.2, 1.0)), seed=int(rng.integers(0, 1 << 30)),
                                          rot_deg=tuple(rng.uniform(-3, 3, 3)), shift=tuple(rng.uniform(-0.05, 0.05, 3)))
                    s, t = p["source"], p["target"]
                elif u < 0.9:
                    nt, nq = int(rng.integers(1, 20000)), int(rng.integers(1, 15000))
                    scale = float(10.0 ** rng.uniform(-2, 1))
                    off = rng.uniform(-10, 10, (3, 1))
                    t = (fuzz_cloud(rng, nt, KINDS[rng.integers(0, 7)]) * scale + off).astype(np.float32)
                    s = (fuzz_cloud(rng, nq, KINDS[rng.integers(0, 7)]) * scale + off + rng.normal(0, 0.01 * scale, (3, 1))).astype(np.float32)
                elif u < 0.95:  # far apart: < 3 pairs within reach -> fallback
                    q = synth.frustum_pair(int(rng.integers(3, 500)), seed=int(rng.integers(0, 1 << 30)))
                    s, t = q["source"] + np.float32(100), q["target"]
                else:
                    q = synth.frustum_pair(int(rng.integers(3, 500)), seed=int(rng.integers(0, 1 << 30)))
                    s, t = np.zeros((3, 0), np.float32), q["target"]
                pairs.append((np.ascontiguousarray(s, np.float32), np.ascontiguousarray(t, np.float32)))
            kw = dict(solve=int(rng.integers(0, 2)), max_iterations=int(rng.integers(0, 12)), fixed_iterations=int(rng.random() < 0.5),
                      max_nn_dist=float(rng.choice([0.75, 0.1, 0.02])), last_translation=rng.normal(0, 0.1, 3).astype(np.float32))
            if device:
                keep = [(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()) for s, t in pairs]
                torch.cuda.synchronize()
                args = [(a.data_ptr(), a.shape[1], b_.data_ptr(), b_.shape[1]) for a, b_ in keep]
                T, st, rc = ctx.align_batch_device(args, binding.default_params(**kw))
                assoc = None
            else:
                T, st, rc, assoc = ctx.align_batch(pairs, associations=True, **kw)
            for b, (s, t) in enumerate(pairs):
                single.set_target(t)
                single.set_source(s)
                Ts, sts, rcs = single.align(**kw)
                ok = np.array_equal(T[b].view(np.uint32), Ts.view(np.uint32)) and (st[b].iterations, st[b].status, st[b].final_pairs) == (
                    sts.iterations, sts.status, sts.final_pairs)
                if s.shape[1] > 0 and assoc is not None:
                    i1, d1 = single.get_associations()
                    ok = ok and np.array_equal(assoc[b][0], i1) and np.array_equal(assoc[b][1].view(np.uint32), d1.view(np.uint32))
                if not ok:
                    bad += 1
                    if log:
                        log(f"MISMATCH batch {c} pair {b} group {g} {s.shape} {t.shape} {kw}")
                pairs_done += 1
            if log and c % 10 == 9:
                log(f"{c + 1} batches, {pairs_done} pairs, {bad} mismatches, {time.time() - t0:.0f} s")
    finally:
        if saved is None:
            os.environ.pop("ICPK_BATCH_GROUP", None)
        else:
            os.environ["ICPK_BATCH_GROUP"] = saved
        single.close()
        for cx in ctxs.values():
            cx.close()
    return pairs_done, bad
