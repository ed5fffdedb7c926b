// icp_align.hpp -- C++ host side above the C ABI (include/icpk.h).
//
// Mirrors the reference's operator surface for the ICP path without OpenCV:
//   * icp::align(source, target, params, &result)      -- the inner loop on SoA views
//     (north_star's call surface; runs on the calling thread's default Engine, or pass one)
//   * icp::alignBatch(engine, pairs, params, results)  -- frame-batch mode
//   * icp::Comm                                        -- RCCL collectives of the multi-GPU modes
//   * icp::filterDepthImage / icp::findGlobalKeyPointAssociations -- SLAM.cpp:553-574,
//     icp.cpp:488-515 with the reference's names and argument meaning
//   * icp::Tracker                                     -- the per-frame state machine of
//     icp::getTransformation (icp.cpp:22-26 file-scope pose state, :38-71 cloud set-up,
//     :98-268 loop, :237/:246 pose update), fed with raw CV_16UC1 depth buffers
//   * icp::makeRotationMatrix / meanSquareError / toEulerianAngle helpers with the
//     reference's names and argument meaning (icp.hpp:25-27, SLAM.hpp:44-46)
// icp_opencv_adapter.hpp adds the exact cv::Mat signature when OpenCV is present.
//
// Header-only; link with -licpk.  Errors: status codes of icpk.h, never exceptions
// from the hot path (the Engine constructor throws std::runtime_error when no GPU
// is usable -- there is no CPU fallback).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "icpk.h"

namespace icp {

// xyz structure-of-arrays view of a point cloud in host memory (replaces
// point_list_t / color_point_t, pointcloud.hpp:13-23; colour weight is 0)
struct CloudView {
  const float* x = nullptr;
  const float* y = nullptr;
  const float* z = nullptr;
  int32_t n = 0;
};

struct AlignParams : icpk_params {
  AlignParams() { icpk_default_params(this); }
};

struct AlignResult {
  float T[16];        // row-major 4x4, same content as the cv::Mat of icp.cpp:29,284
  icpk_stats stats{};
  int status = ICPK_OK;
  float (*rotation())[4] { return reinterpret_cast<float(*)[4]>(T); }
};

// RAII owner of one icpk_ctx (one GPU, one HIP stream).  One Engine per host
// thread / per GPU; calls on one Engine are serialised.
class Engine {
 public:
  explicit Engine(int device = 0) {
    const int rc = icpk_create(&ctx_, device);
    if (rc != ICPK_OK) throw std::runtime_error("icpk_create failed (status " + std::to_string(rc) + "): no usable HIP device");
  }
  ~Engine() { icpk_destroy(ctx_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  icpk_ctx* ctx() const { return ctx_; }
  const char* last_error() const { return icpk_last_error(ctx_); }
  // pointcloud.cpp:27-30 (SUBSAMPLE_FACTOR, pointcloud.hpp:11) with a reproducible choice instead of rand(): every
  // cloud this engine back-projects from now on keeps one valid pixel in `factor` (0 / 1: all of them)
  int setSubsample(int factor = ICPK_SUBSAMPLE_FACTOR, uint64_t seed = 0) { return icpk_set_subsample(ctx_, factor, seed); }

 private:
  icpk_ctx* ctx_ = nullptr;
};

// The inner loop on explicit clouds (frame-pair formulation of icp.cpp:98-268).
inline int align(Engine& eng, const CloudView& source, const CloudView& target, const AlignParams& params,
                 AlignResult* out) {
  if (!out) return ICPK_E_ARG;
  int rc = icpk_set_target(eng.ctx(), target.x, target.y, target.z, target.n);
  if (rc == ICPK_OK) rc = icpk_set_source(eng.ctx(), source.x, source.y, source.z, source.n);
  if (rc != ICPK_OK) {
    for (int k = 0; k < 16; ++k) out->T[k] = (k % 5 == 0) ? 1.f : 0.f;
    out->status = rc;
    return rc;
  }
  out->status = icpk_align(eng.ctx(), &params, out->T, &out->stats);
  return out->status;
}

// north_star's call surface: align(source, target, params, &result) on the calling thread's
// default Engine (device ICPK_DEVICE or 0; created on first use; one per host thread, as one
// context serialises its calls).  Throws std::runtime_error on first use if no GPU is usable.
inline Engine& defaultEngine() {
  static thread_local Engine eng([] {
    const char* e = std::getenv("ICPK_DEVICE");
    return e ? std::atoi(e) : 0;
  }());
  return eng;
}
inline int align(const CloudView& source, const CloudView& target, const AlignParams& params, AlignResult* out) {
  return align(defaultEngine(), source, target, params, out);
}

// Frame-batch mode (BASELINE config 4): independent pairs, lock-step groups on the
// engine's GPU (icpk_align_batch).  results is resized to pairs.size(); returns the first
// negative status, else the largest.
struct FramePair {
  CloudView source, target;
};
inline int alignBatch(Engine& eng, const std::vector<FramePair>& pairs, const AlignParams& params,
                      std::vector<AlignResult>* results) {
  if (!results) return ICPK_E_ARG;
  const size_t n = pairs.size();
  std::vector<icpk_pair> raw(n);
  for (size_t k = 0; k < n; ++k) {
    raw[k] = icpk_pair{pairs[k].source.x, pairs[k].source.y, pairs[k].source.z, pairs[k].source.n,
                       pairs[k].target.x, pairs[k].target.y, pairs[k].target.z, pairs[k].target.n, nullptr, nullptr};
  }
  std::vector<float> T(16 * (n ? n : 1));
  std::vector<icpk_stats> st(n ? n : 1);
  const int rc = icpk_align_batch(eng.ctx(), (int32_t)n, raw.data(), &params, T.data(), st.data());
  results->resize(n);
  for (size_t k = 0; k < n; ++k) {
    std::memcpy((*results)[k].T, T.data() + 16 * k, sizeof((*results)[k].T));
    (*results)[k].stats = st[k];
    (*results)[k].status = st[k].status;
  }
  return rc;
}

// RCCL collectives of the multi-GPU modes on an Engine (icpk_comm_* of icpk.h): one process or
// host thread + one Engine per GPU; the 128-byte id made by rank 0 (Comm::uniqueId) reaches the
// other ranks by the host's own means.
class Comm {
 public:
  static int uniqueId(unsigned char id[ICPK_COMM_ID_BYTES]) { return icpk_comm_unique_id(id); }
  Comm(Engine& eng, const unsigned char id[ICPK_COMM_ID_BYTES], int rank, int world) : eng_(eng) {
    status = icpk_comm_init_rccl(eng.ctx(), id, rank, world);
  }
  ~Comm() { icpk_comm_destroy(eng_.ctx()); }
  Comm(const Comm&) = delete;
  Comm& operator=(const Comm&) = delete;
  int rank() const { return icpk_comm_rank(eng_.ctx()); }
  int world() const { return icpk_comm_world(eng_.ctx()); }
  // this rank's block [start, start + count) of n_items
  void partition(int32_t n_items, int32_t* start, int32_t* count) const {
    icpk_comm_partition(n_items, world(), rank(), start, count);
  }
  int broadcastTarget(int root = 0) { return icpk_comm_broadcast_target(eng_.ctx(), root); }
  // ONE large pair, queries sharded over the ranks (collective call): the engine's source is this rank's slice of
  // the queries, the target the same on every rank (broadcastTarget); one in-stream all-reduce per iteration, no host
  // round trip.  Every rank gets the same result.
  int alignQuerySharded(const AlignParams& params, AlignResult* result) {
    if (!result) return ICPK_E_ARG;
    result->status = icpk_align_query_sharded(eng_.ctx(), &params, result->T, &result->stats);
    return result->status;
  }
  // local: this rank's block of results; all: resized to n_total, global pair order
  int gatherResults(const std::vector<AlignResult>& local, int32_t n_total, std::vector<AlignResult>* all) {
    if (!all) return ICPK_E_ARG;
    std::vector<float> Tl(16 * (local.size() ? local.size() : 1)), Ta(16 * (size_t)(n_total > 0 ? n_total : 1)),
        Sa(4 * (size_t)(n_total > 0 ? n_total : 1));
    std::vector<icpk_stats> sl(local.size() ? local.size() : 1);
    for (size_t k = 0; k < local.size(); ++k) {
      std::memcpy(Tl.data() + 16 * k, local[k].T, 16 * sizeof(float));
      sl[k] = local[k].stats;
      sl[k].status = local[k].status;
    }
    const int rc = icpk_comm_gather_results(eng_.ctx(), Tl.data(), sl.data(), (int32_t)local.size(), n_total, Ta.data(),
                                            Sa.data());
    if (rc != ICPK_OK) return rc;
    all->assign((size_t)n_total, AlignResult{});
    for (int32_t k = 0; k < n_total; ++k) {
      AlignResult& r = (*all)[(size_t)k];
      std::memcpy(r.T, Ta.data() + 16 * (size_t)k, 16 * sizeof(float));
      int32_t iv[3];  // (int32 bit patterns in the first three slots of a row)
      std::memcpy(iv, Sa.data() + 4 * (size_t)k, sizeof(iv));
      r.stats.iterations = iv[0];
      r.status = r.stats.status = iv[1];
      r.stats.final_pairs = iv[2];
      r.stats.final_mse = Sa[4 * (size_t)k + 3];
    }
    return ICPK_OK;
  }
  int status = ICPK_OK;

 private:
  Engine& eng_;
};

// SLAM.cpp:553-574 (SLAM.hpp:34): filterDepthImage(image, rgbImage, maxDistance, minDistance) on a
// CV_16UC1 buffer, in place; the colour image is not touched by the reference either
inline int filterDepthImage(Engine& eng, uint16_t* image, int rows, int cols, int maxDistance = 25000,
                            int minDistance = 1000) {
  return icpk_filter_depth_image(eng.ctx(), image, image, rows, cols, maxDistance, minDistance, 1, -1, -1);
}

// pointcloud.cpp:60-98: the 3-D points of a frame's key points (cv::KeyPoint::pt as (x, y) pairs), the cloud that
// findGlobalKeyPointAssociations below takes as dataKeypoints.  Host only.  out_xyz: n x 3 floats, kept: n ints or null.
inline int backprojectKeyPoints(const uint16_t* depth, int rows, int cols, const float* kp_xy, int n, float* out_xyz,
                                int32_t* kept = nullptr, float fx = ICPK_FX, float cx = ICPK_CX) {
  return icpk_backproject_keypoints(depth, rows, cols, kp_xy, n, fx, cx, out_xyz, kept);
}

// icp.cpp:488-515: data key points vs map key points.  errors / associations are rebuilt (pairs of
// (query index, nearest index) in query order), nonAssociations is appended to; an empty map
// returns ICPK_W_EMPTY_MAP and touches nothing (icp.cpp:490-491).
inline int findGlobalKeyPointAssociations(Engine& eng, const CloudView& dataKeypoints, const CloudView& mapKeypoints,
                                          std::vector<float>& errors,
                                          std::vector<std::pair<int32_t, int32_t>>& associations,
                                          std::vector<int32_t>& nonAssociations,
                                          float maxDistance = ICPK_MAX_NN_KEYPOINT_DISTANCE) {
  if (mapKeypoints.n <= 0) return ICPK_W_EMPTY_MAP;
  int rc = icpk_set_target(eng.ctx(), mapKeypoints.x, mapKeypoints.y, mapKeypoints.z, mapKeypoints.n);
  if (rc == ICPK_OK) rc = icpk_set_source(eng.ctx(), dataKeypoints.x, dataKeypoints.y, dataKeypoints.z, dataKeypoints.n);
  if (rc != ICPK_OK) return rc;
  const size_t nq = (size_t)(dataKeypoints.n > 0 ? dataKeypoints.n : 0), had = nonAssociations.size();
  std::vector<int32_t> q(nq + 1), t(nq + 1);
  std::vector<float> d(nq + 1);
  nonAssociations.resize(had + nq);
  int32_t na = 0, nr = (int32_t)had;
  rc = icpk_associate_keypoints(eng.ctx(), ICPK_NN_GRID, maxDistance, q.data(), t.data(), d.data(), &na,
                                nonAssociations.data(), (int32_t)nonAssociations.size(), &nr);
  if (rc != ICPK_OK) {
    nonAssociations.resize(had);
    return rc;
  }
  nonAssociations.resize((size_t)nr);
  errors.assign(d.begin(), d.begin() + na);
  associations.clear();
  for (int32_t k = 0; k < na; ++k) associations.emplace_back(q[(size_t)k], t[(size_t)k]);
  return ICPK_OK;
}

// icp.cpp:640-653
inline void makeRotationMatrix(float x, float y, float z, float out[9]) { icpk_make_rotation_matrix(x, y, z, out); }

// icp.cpp:622-638: (mean error)^2, sequential float accumulation like the reference
inline float meanSquareError(const std::vector<float>& errors) {
  float s = 0.f;
  for (float e : errors) s += e;
  if (!errors.empty()) {
    s /= (float)errors.size();
    s = (float)((double)s * (double)s);
  }
  return s;
}

// quaternion.cpp:23-79 + SLAM.cpp:613-636: rotation matrix -> roll/pitch/yaw degrees
inline void toEulerianAngle(const float rotation[9], float& x, float& y, float& z) {
  float q[4], e[3];
  icpk_matrix_to_quaternion(rotation, q);
  icpk_quaternion_to_euler(q, e);
  x = e[0];
  y = e[1];
  z = e[2];
}

// The state icp.cpp keeps at file scope (cameraRotation, lastRotation,
// cameraPosition, lastTranslation, icp.cpp:22-25) plus the per-frame procedure of
// getTransformation, with the accumulated key-point map replaced by the previous
// frame's full cloud (the association the reference keeps commented at
// icp.cpp:253).  Depth buffers are CV_16UC1 row-major (rows x cols uint16).
class Tracker {
 public:
  explicit Tracker(Engine& eng, float fx = ICPK_FX, float cx = ICPK_CX) : eng_(eng), fx_(fx), cx_(cx) { reset(); }
  Engine& engine() const { return eng_; }

  void reset() {
    static const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(cameraRotation, I, sizeof(I));  // icp.cpp:49
    std::memcpy(lastRotation, I, sizeof(I));    // icp.cpp:50
    cameraPosition[0] = cameraPosition[1] = cameraPosition[2] = 5.f;   // icp.cpp:53
    lastTranslation[0] = lastTranslation[1] = lastTranslation[2] = 0.f;  // icp.cpp:54
    params = AlignParams();
    params.solve = ICPK_SOLVE_REFERENCE;
  }

  // Same meaning as icp::getTransformation(data, previous, ..., maxIterations, threshold):
  // returns the status, writes the 4x4 into T (row-major).  `data` is the current
  // frame, `previous` the frame before it.  previous == nullptr: the frame that was `data` in the
  // last call -- what SLAM.cpp hands back (SLAM.cpp:305, previous = filtered.clone()); it has stayed
  // on the device, so only the new frame is uploaded (same results as passing it again).
  int getTransformation(const uint16_t* data, const uint16_t* previous, int rows, int cols, int maxIterations,
                        float threshold, float T[16]) {
    icpk_ctx* c = eng_.ctx();
    // icp.cpp:38-39: back-project both frames; :58-59 / :70-71: rotate by the camera
    // rotation, translate by the camera position (both clouds get the current pose)
    // -- one call: both uploads, 3 launches (5 with filterFrames), one host wait; the posed source is the
    // starting point of the alignment
    int rc = icpk_backproject_pair(c, data, previous, rows, cols, fx_, cx_, nullptr, cameraRotation, cameraPosition,
                                   filterFrames ? 1 : 0, maxDistance, minDistance, 1, -1, -1, nullptr, nullptr);
    if (rc != ICPK_OK) return rc;
    params.max_iterations = maxIterations;
    params.threshold = threshold;
    std::memcpy(params.last_rotation, lastRotation, sizeof(lastRotation));
    std::memcpy(params.last_translation, lastTranslation, sizeof(lastTranslation));
    icpk_stats st;
    rc = icpk_align(c, &params, T, &st);
    if (rc < 0) return rc;
    lastStats = st;
    // pose bookkeeping exactly as the loop does it (icp.cpp:235-237, 245-246)
    int32_t niter = 0;
    std::vector<float> R((size_t)maxIterations * 9 + 9), t((size_t)maxIterations * 3 + 3);
    icpk_get_trace(c, &niter, R.data(), t.data(), nullptr, nullptr);
    for (int i = 0; i < niter; ++i) {
      float Rinv[9], prod[9];
      invert3(R.data() + 9 * i, Rinv);
      mul3(cameraRotation, Rinv, prod);  // cameraRotation *= R
      std::memcpy(cameraRotation, prod, sizeof(prod));
      for (int k = 0; k < 3; ++k) cameraPosition[k] -= t[3 * i + k];  // cameraPosition -= offset
    }
    // icp.cpp:260-261: lastTranslation = -offset; lastRotation = R (the outer R: identity
    // on the normal path because the inner R shadows it)
    for (int k = 0; k < 3; ++k) lastTranslation[k] = -T[4 * k + 3];
    if (rc != ICPK_W_TOO_FEW_PAIRS) {
      static const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      std::memcpy(lastRotation, I, sizeof(I));
    }
    return rc;
  }

  // SLAM.cpp:229 filters every frame before it reaches getTransformation; set this to have the filter run
  // on the device inside the same call instead (SLAM.hpp:15-16 limits)
  bool filterFrames = false;
  int maxDistance = 25000, minDistance = 1000;
  float cameraRotation[9];
  float lastRotation[9];
  float cameraPosition[3];
  float lastTranslation[3];
  AlignParams params;
  icpk_stats lastStats{};

 private:
  static void mul3(const float A[9], const float B[9], float C[9]) {  // CV_32F product, double accumulate
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += (double)A[3 * r + k] * (double)B[3 * k + cc];
        C[3 * r + cc] = (float)s;
      }
  }
  static void invert3(const float m[9], float out[9]) {  // icp.cpp:235 Mat::inv on a 3x3
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double s = det != 0.0 ? 1.0 / det : 0.0;
    const double t[9] = {(e * i - f * h) * s, (c * h - b * i) * s, (b * f - c * e) * s,
                         (f * g - d * i) * s, (a * i - c * g) * s, (c * d - a * f) * s,
                         (d * h - e * g) * s, (b * g - a * h) * s, (a * e - b * d) * s};
    for (int k = 0; k < 9; ++k) out[k] = (float)t[k];
  }

  Engine& eng_;
  float fx_, cx_;
};

}  // namespace icp
