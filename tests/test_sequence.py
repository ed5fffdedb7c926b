"""The per-frame driver glue (icp_slam_prototype_amd/sequence.py): list/ground-truth
parsing and quaternion helpers on CPU; a synthetic 4-frame sequence on the GPU against
the same procedure over the oracle."""
import numpy as np
import pytest

from icp_slam_prototype_amd import sequence, synth


def test_parse_list_and_ground_truth():
    txt = "# depth maps\n# timestamp filename\n1.5 depth\\a.png\n2.25 depth/b.png \n"
    assert sequence.parse_list_file(txt, "/d/") == [(1.5, "/d/depth/a.png"), (2.25, "/d/depth/b.png")]
    gt = sequence.GroundTruth("# ground truth\n1.0 0 0 0 0 0 0 1\n2.0 1 2 3 0 0 0.7071068 0.7071068\n3.0 4 5 6 1 0 0 0\n")
    pos, q = gt.next(0.5)
    assert np.array_equal(pos, [0, 0, 0]) and np.array_equal(q, np.array([1, 0, 0, 0], np.float32))
    pos, q = gt.next(2.5)  # skips the 2.0 record, takes 3.0 (first with timestamp >= frame's)
    assert np.array_equal(pos, [4, 5, 6]) and np.array_equal(q, np.array([0, 1, 0, 0], np.float32))


def test_quaternion_helpers_match_scipy():
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(2)
    for _ in range(20):
        a, b = Rotation.random(random_state=rng.integers(1 << 30)), Rotation.random(random_state=rng.integers(1 << 30))
        qa, qb = a.as_quat()[[3, 0, 1, 2]], b.as_quat()[[3, 0, 1, 2]]
        prod = sequence.quat_mul(qa, qb).astype(np.float64)
        want = (a * b).as_quat()[[3, 0, 1, 2]]
        assert min(np.abs(prod - want).max(), np.abs(prod + want).max()) < 1e-6
        inv = sequence.quat_inverse(qa).astype(np.float64)
        assert np.allclose(sequence.quat_mul(qa, inv), [1, 0, 0, 0], atol=1e-6)


@pytest.mark.gpu
def test_sequence_runner_matches_oracle_procedure(oracle):
    from icp_slam_prototype_amd import binding

    rows, cols = 96, 128
    rng = np.random.default_rng(5)
    frames, gt_lines = [], []
    from scipy.spatial.transform import Rotation

    for k in range(4):
        Rm = synth.rot_xyz_deg(0, 0.4 * k, 0.1 * k)
        d = synth.render_room_depth(rows, cols, Rm, np.array([0.008 * k, 0, 0]), noise_sigma=0.001, rng=rng)
        d[rng.random(d.shape) > 0.6] = 0
        frames.append(d.astype(np.uint16))
        q = Rotation.from_matrix(Rm).as_quat()
        gt_lines.append("%f %f 0 0 %.7f %.7f %.7f %.7f" % (10.0 + k, 0.008 * k, q[0], q[1], q[2], q[3]))
    gt_text = "# gt\n" + "\n".join(gt_lines) + "\n"

    with binding.Context(0) as ctx:
        run = sequence.SequenceRunner(ctx, max_iterations=6, threshold=1e-6)
        gt = sequence.GroundTruth(gt_text)
        out = [run.step(f, 10.0 + k, gt) for k, f in enumerate(frames)]
    assert out[0] is None and all(o is not None for o in out[1:])

    # the same procedure over the oracle
    Rcam, pcam = np.eye(3, dtype=np.float32), np.full(3, 5, np.float32)
    lastT = np.zeros(3, np.float32)
    rot = np.eye(3, dtype=np.float32)
    gt2 = sequence.GroundTruth(gt_text)
    _, q0 = gt2.next(10.0)
    for k in range(1, 4):
        tgt = oracle.transform_points(oracle.backproject(frames[k - 1]), Rcam, pcam)
        src = oracle.transform_points(oracle.backproject(frames[k]), Rcam, pcam)
        o = oracle.align(src, tgt, max_iterations=6, threshold=1e-6, solve=0, sum_order=1, threads=4,
                         last_translation=lastT)
        for it in o["trace"]:
            Rcam = sequence._mul3f(Rcam, oracle.inv3(it["R"]))
            pcam = (pcam - it["t"]).astype(np.float32)
        lastT = -o["T"][:3, 3]
        rot = sequence._mul3f(rot, o["T"][:3, :3])
        _, qk = gt2.next(10.0 + k)
        r = out[k]
        assert np.array_equal(r["T"], o["T"]) and r["mse"] == o["final_mse"] and r["iterations"] == o["iterations"]
        assert np.array_equal(r["icp_euler"], oracle.to_euler(oracle.quaternion_from_matrix(rot)))
        assert np.array_equal(r["gt_euler"], oracle.to_euler(sequence.quat_mul(qk, sequence.quat_inverse(q0))))
        assert r["csv"].count(",") == 6
    # ground truth: about 0.4 degrees about y per frame in the reference's Euler convention
    assert abs(abs(out[3]["gt_euler"][1]) - 1.2) < 0.05
