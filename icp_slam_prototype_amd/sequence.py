"""Per-frame driver glue around the ICP path: the pose
bookkeeping and reporting SLAM.cpp does around its one call to
icp::getTransformation, restated over the C ABI so a depth sequence produces the
reference's CSV rows  `MSE,ICP rX,ICP rY,ICP rZ,GT rX,GT rY,GT rZ`  (SLAM.cpp:327).

Restated pieces (file:line of the reference):
  * list files `timestamp filename`, '#' comments          SLAM.cpp:374-410
  * ground truth `timestamp tx ty tz qx qy qz qw`, first
    record at or after the frame's timestamp               SLAM.cpp:432-490
  * quaternion product / inverse                           quaternion.cpp:188-195,325-328
  * deltaRotation = current * initial^-1                   SLAM.cpp:283
  * rotation = rotation * icpRotation, Euler of both       SLAM.cpp:285-293
  * per-frame ICP state (camera pose, last motion)         icp.cpp:22-26,47-71,237,246,260-261
Image decoding (cv::imread), filterDepthImage's dilate/erode, FAST key points and the
viewers stay out of scope: frames are passed in as uint16 arrays.
"""
import numpy as np

from . import binding


def parse_list_file(text, path=""):
    """SLAM.cpp:374-410 getNextImageFileName for a whole file: [(timestamp, path+filename)]."""
    out = []
    for line in text.splitlines():
        if not line or line[0] == "#":
            continue
        parts = line.split(" ")
        ts = float(parts[0])
        name = parts[1].replace("\\", "/").strip() if len(parts) > 1 else ""
        out.append((ts, path + name))
    return out


def quat_mul(a, b):
    """quaternion.cpp:188-195 operator* on (w, x, y, z), float32."""
    w, x, y, z = (np.float32(v) for v in a)
    qw, qx, qy, qz = (np.float32(v) for v in b)
    return np.array([w * qw - x * qx - y * qy - z * qz,
                     w * qx + x * qw + y * qz - z * qy,
                     w * qy + y * qw + z * qx - x * qz,
                     w * qz + z * qw + x * qy - y * qx], np.float32)


def quat_inverse(q):
    """quaternion.cpp:325-328: conjugate().scale(1 / norm()), norm() = w^2+x^2+y^2+z^2."""
    q = np.asarray(q, np.float32)
    n = np.float32(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    s = np.float32(1) / n
    return np.array([q[0] * s, -q[1] * s, -q[2] * s, -q[3] * s], np.float32)


class GroundTruth:
    """SLAM.cpp:432-490 getNextGroundTruth: a forward-only cursor over the records."""

    def __init__(self, text):
        self.rows = []
        for line in text.splitlines():
            if not line or line[0] == "#":
                continue
            v = line.split(" ")
            self.rows.append((float(v[0]), [np.float32(x) for x in v[1:8]]))
        self.k = 0

    def next(self, timestamp):
        """First unread record; while its timestamp is earlier than the frame's, read on.
        Returns (position (3,), quaternion (w,x,y,z))."""
        ts, rec = self.rows[self.k]
        self.k += 1
        while ts < timestamp:
            ts, rec = self.rows[self.k]
            self.k += 1
        tx, ty, tz, qx, qy, qz, qw = rec
        return np.array([tx, ty, tz], np.float32), np.array([qw, qx, qy, qz], np.float32)


def _mul3f(A, B):
    A = A.astype(np.float64)
    B = B.astype(np.float64)
    return ((A[:, 0:1] * B[0:1, :] + A[:, 1:2] * B[1:2, :]) + A[:, 2:3] * B[2:3, :]).astype(np.float32)


def _inv3f(R):
    m = R.astype(np.float64).reshape(9)
    a, b, c, d, e, f, g, h, i = m
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    s = 1.0 / det
    t = np.array([(e * i - f * h) * s, (c * h - b * i) * s, (b * f - c * e) * s,
                  (f * g - d * i) * s, (a * i - c * g) * s, (c * d - a * f) * s,
                  (d * h - e * g) * s, (b * g - a * h) * s, (a * e - b * d) * s])
    return t.astype(np.float32).reshape(3, 3)


class SequenceRunner:
    """Feeds frames one by one, like the while loop of SLAM.cpp:194-352."""

    HEADER = "MSE,ICP rX,ICP rY,ICP rZ,GT rX,GT rY,GT rZ"  # SLAM.cpp:327

    def __init__(self, ctx, max_iterations=16, threshold=1e-4, fx=468.60, cx=318.27, **params):
        self.ctx = ctx
        self.kw = dict(max_iterations=max_iterations, threshold=threshold, solve=binding.SOLVE_REFERENCE, **params)
        self.fx, self.cx = fx, cx
        self.camera_rotation = np.eye(3, dtype=np.float32)      # icp.cpp:49
        self.camera_position = np.full(3, 5, np.float32)        # icp.cpp:53
        self.last_rotation = np.eye(3, dtype=np.float32)
        self.last_translation = np.zeros(3, np.float32)
        self.rotation = np.eye(3, dtype=np.float32)             # SLAM.cpp `rotation` accumulator
        self.previous = None
        self.initial_q = None

    def step(self, depth, timestamp=None, ground_truth=None):
        """Returns None for the first frame (SLAM.cpp:306-327 only stores it and prints the
        header), else dict(mse, icp_euler, gt_euler, T, csv)."""
        depth = np.ascontiguousarray(depth, np.uint16)
        if self.previous is None:
            self.previous = depth.copy()
            if ground_truth is not None:
                _, self.initial_q = ground_truth.next(timestamp)
            return None
        c = self.ctx
        # icp.cpp:38-39 back-project both frames, :58-59 / :70-71 pose them; the posed source is the
        # starting point of the alignment (one call: icpk_backproject_pair)
        c.backproject_pair(depth, self.previous, R=self.camera_rotation, t=self.camera_position, fx=self.fx, cx=self.cx)
        T, st, rc = c.align(last_rotation=self.last_rotation, last_translation=self.last_translation, **self.kw)
        for it in c.get_trace(max(self.kw["max_iterations"], 1)):
            self.camera_rotation = _mul3f(self.camera_rotation, _inv3f(it["R"]))   # icp.cpp:235-237
            self.camera_position = (self.camera_position - it["t"]).astype(np.float32)  # icp.cpp:246
        self.last_translation = (-T[:3, 3]).astype(np.float32)                    # icp.cpp:260
        if rc != binding.W_TOO_FEW_PAIRS:
            self.last_rotation = np.eye(3, dtype=np.float32)                      # icp.cpp:261 (shadowed R)
        self.rotation = _mul3f(self.rotation, T[:3, :3])                         # SLAM.cpp:285
        icp_euler = binding.quaternion_to_euler(binding.matrix_to_quaternion(self.rotation))  # SLAM.cpp:289-290
        gt_euler = np.zeros(3, np.float32)
        if ground_truth is not None:
            _, cur_q = ground_truth.next(timestamp)
            delta = quat_mul(cur_q, quat_inverse(self.initial_q))                # SLAM.cpp:283
            gt_euler = binding.quaternion_to_euler(delta)                        # SLAM.cpp:292
        self.previous = depth.copy()                                              # SLAM.cpp:305
        vals = [st.final_mse, *icp_euler, *gt_euler]
        return dict(mse=np.float32(st.final_mse), icp_euler=icp_euler, gt_euler=gt_euler, T=T, status=rc,
                    iterations=st.iterations, csv=",".join("%g" % float(v) for v in vals))
