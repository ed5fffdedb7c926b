"""The C++ host mirror (icp_slam_prototype_amd/include/icp_align.hpp): a sequence
of depth frames through icp::Tracker -- the getTransformation-shaped entry point
-- compared with the same procedure restated in Python over the oracle."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from icp_slam_prototype_amd import build, synth

pytestmark = pytest.mark.gpu


def mul3f(A, B):
    A = A.astype(np.float64)
    B = B.astype(np.float64)
    return ((A[:, 0:1] * B[0:1, :] + A[:, 1:2] * B[1:2, :]) + A[:, 2:3] * B[2:3, :]).astype(np.float32)


def test_tracker_sequence_and_align(oracle):
    exe = build.build_cpp_test()
    rows, cols, max_iter, thr = 120, 160, 6, 1e-5
    rng = np.random.default_rng(0)
    frames = []
    for k in range(3):  # camera drifting by 0.5 degree / 1 cm per frame
        Rm = synth.rot_xyz_deg(0, 0.5 * k, 0)
        d = synth.render_room_depth(rows, cols, Rm, np.array([0.01 * k, 0, 0]), noise_sigma=0.001, rng=rng)
        d[rng.random(d.shape) > 0.5] = 0
        frames.append(d.astype(np.uint16))
    p = synth.frustum_pair(1500, seed=3, rot_deg=(0, 1, 0), shift=(0.01, 0, 0))
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<4if", rows, cols, len(frames), max_iter, thr))
            for d in frames:
                f.write(d.tobytes())
            f.write(struct.pack("<2i", p["source"].shape[1], p["target"].shape[1]))
            f.write(np.ascontiguousarray(p["source"], np.float32).tobytes())
            f.write(np.ascontiguousarray(p["target"], np.float32).tobytes())
        subprocess.check_call([exe, fin, fout])
        raw = open(fout, "rb").read()
    off = 0
    Rcam = np.eye(3, dtype=np.float32)
    pcam = np.full(3, 5, np.float32)
    lastR = np.eye(3, dtype=np.float32)
    lastT = np.zeros(3, np.float32)
    for i in range(1, len(frames)):
        rc, iters = struct.unpack_from("<2i", raw, off)
        off += 8
        vals = np.frombuffer(raw, np.float32, 16 + 9 + 3 + 3, off)
        off += 4 * 31
        T, camR, camP, eul = vals[:16].reshape(4, 4), vals[16:25].reshape(3, 3), vals[25:28], vals[28:31]
        # the same procedure over the oracle (icp.cpp:38-71, 98-268 frame-pair formulation)
        tgt = oracle.transform_points(oracle.backproject(frames[i - 1]), Rcam, pcam)
        src = oracle.transform_points(oracle.backproject(frames[i]), Rcam, pcam)
        o = oracle.align(src, tgt, max_iterations=max_iter, threshold=thr, solve=0, sum_order=1, threads=4,
                         last_rotation=lastR, last_translation=lastT)
        assert rc == o["status"] and iters == o["iterations"] and iters > 0
        assert np.linalg.norm(T.astype(np.float64) - o["T"].astype(np.float64)) < 1e-5
        assert np.linalg.norm(T.astype(np.float64) - o["T"].astype(np.float64)) < 1e-6  # Newton polar vs the oracle's Jacobi SVD
        for it in o["trace"]:
            Rcam = mul3f(Rcam, oracle.inv3(it["R"]))
            pcam = (pcam - it["t"]).astype(np.float32)
        lastT = -o["T"][:3, 3]
        assert np.allclose(camR, Rcam, rtol=0, atol=1e-6) and np.allclose(camP, pcam, rtol=0, atol=1e-6)
        assert np.array_equal(eul, oracle.to_euler(oracle.quaternion_from_matrix(camR)))  # same formulas on the same matrix
        assert np.allclose(eul, oracle.to_euler(oracle.quaternion_from_matrix(Rcam)), rtol=0, atol=1e-3)
    rc, iters = struct.unpack_from("<2i", raw, off)
    off += 8
    T = np.frombuffer(raw, np.float32, 16, off).reshape(4, 4)
    o = oracle.align(p["source"], p["target"], max_iterations=max_iter, solve=1, sum_order=1, fixed_iterations=True,
                     threads=4)
    assert rc == 0 and iters == max_iter and np.linalg.norm(T.astype(np.float64) - o["T"].astype(np.float64)) < 1e-6
    off += 4 * 16
    # icp::alignBatch x3 and the engine-less icp::align(source, target, params, &result) == icp::align
    agree, cagree = struct.unpack_from("<2i", raw, off)
    off += 8
    assert agree == 1, "frame-batch / default-engine results differ from icp::align"
    assert cagree == 1, "icp::Comm (RCCL, world of one) did not return the local rows"
    # icp::filterDepthImage (SLAM.cpp:553-574) vs the oracle
    img = np.frombuffer(raw, np.uint16, rows * cols, off).reshape(rows, cols)
    off += 2 * rows * cols
    assert np.array_equal(img, oracle.filter_depth_image(frames[0]))
    # icp::findGlobalKeyPointAssociations (icp.cpp:488-515) vs the oracle
    krc, na, nr = struct.unpack_from("<3i", raw, off)
    off += 12
    pairs = np.frombuffer(raw, np.int32, 2 * na, off).reshape(na, 2)
    off += 8 * na
    errors = np.frombuffer(raw, np.float32, na, off)
    off += 4 * na
    rejected = np.frombuffer(raw, np.int32, nr, off)
    off += 4 * nr
    want = oracle.keypoint_associations(p["source"], p["target"], 0.1)
    assert krc == 0 and np.array_equal(pairs[:, 0], want[0]) and np.array_equal(pairs[:, 1], want[1])
    assert np.array_equal(errors.view(np.uint32), want[2].view(np.uint32)) and np.array_equal(rejected, want[3])
    # icp::Tracker with filterFrames: filterDepthImage on both frames inside the call (icpk_backproject_pair)
    rc, iters = struct.unpack_from("<2i", raw, off)
    off += 8
    T = np.frombuffer(raw, np.float32, 16, off).reshape(4, 4)
    off += 4 * 16
    I3, p5 = np.eye(3, dtype=np.float32), np.full(3, 5, np.float32)
    tgt = oracle.transform_points(oracle.backproject(oracle.filter_depth_image(frames[0])), I3, p5)
    src = oracle.transform_points(oracle.backproject(oracle.filter_depth_image(frames[1])), I3, p5)
    o = oracle.align(src, tgt, max_iterations=max_iter, threshold=thr, solve=0, sum_order=1, threads=4)
    assert rc == o["status"] and iters == o["iterations"]
    assert np.linalg.norm(T.astype(np.float64) - o["T"].astype(np.float64)) < 1e-6
    assert off == len(raw)


@pytest.mark.parametrize("filt,resident", [(0, 1), (1, 1), (0, 0)])
def test_native_tracker_bench_matches_the_python_path(filt, resident):
    """tests/cpp/tracker_bench.cpp (the program behind bench.py's tracker_path.native_cpp) makes the call sequence of
    bench.py's Python loop: same frames in, the same transforms out, bit for bit (CRC-32 over all of them)."""
    import json
    import zlib

    from icp_slam_prototype_amd import binding

    exe = build.build_tracker_bench()
    rows, cols, rounds = 120, 160, 2
    rng = np.random.default_rng(1)
    frames = []
    for k in range(4):
        d = synth.render_room_depth(rows, cols, synth.rot_xyz_deg(0, 0.5 * k, 0), np.array([0.01 * k, 0, 0]),
                                    noise_sigma=0.002, rng=rng)
        d[rng.random(d.shape) > 0.5] = 0
        frames.append(d.astype(np.uint16))
    with tempfile.NamedTemporaryFile(suffix=".u16") as f:
        for d in frames:
            f.write(d.tobytes())
        f.flush()
        out = subprocess.run([exe, f.name, str(rows), str(cols), str(len(frames)), str(rounds), str(filt), str(resident), "0"],
                             capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip().splitlines()[-1])
    ctx = binding.Context(0)
    camR, camP = np.eye(3, dtype=np.float32), np.full(3, 5, np.float32)
    crc = its = n = 0
    for rnd in range(rounds + 1):  # (round 0 is the program's untimed warm-up pass)
        for i in range(1, len(frames)):
            ctx.backproject_pair(frames[i], frames[i - 1] if (i == 1 or not resident) else None, R=camR, t=camP, filter=bool(filt))
            T, st, rc = ctx.align(max_iterations=16, threshold=1e-4)
            if rnd:
                crc = zlib.crc32(np.ascontiguousarray(T, np.float32).tobytes(), crc)
                its += st.iterations
                n += 1
    assert got["pairs"] == n and got["transforms_crc32"] == f"{crc:08x}"
    assert abs(got["mean_iterations"] - its / n) < 1e-3 and got["points"] == [ctx.source_size, ctx.target_size]
    ctx.close()
