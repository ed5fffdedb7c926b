// icp_opencv_adapter.hpp -- drop-in replacement for the reference's
//   cv::Mat icp::getTransformation(cv::Mat& data, cv::Mat& previous, cv::Mat color,
//       std::vector<cv::KeyPoint> keypoints, cv::Mat& rotation, int maxIterations,
//       float threshold, cv::viz::Viz3d& depthWindow)            (icp.hpp:23, icp.cpp:28)
// so that the call at SLAM.cpp:277 compiles unchanged.  Needs OpenCV (core + viz
// headers for the signature only) -- NOT available in this build image, so this
// header is not compiled by the tests here; all of its logic lives in
// icp::Tracker (icp_align.hpp), which IS tested (tests/test_gpu_cpp_mirror.py).
//
// Differences from the reference, all on the caller's side of the boundary:
//   * the target of the association is the previous frame's full cloud (the variant
//     the reference keeps commented at icp.cpp:253), not the accumulated key-point
//     map (map.cpp is out of scope); `color`, `keypoints`, `rotation` are accepted
//     and ignored exactly as `rotation` already is in the reference;
//   * nothing is drawn into depthWindow (icp.cpp:41, 273-282 are UI);
//   * rows 3 of the returned matrix is (0,0,0,1) instead of uninitialised memory.
#pragma once
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/viz/vizcore.hpp>

#include "icp_align.hpp"

namespace icp {

inline Tracker& default_tracker() {
  static Engine engine(0);          // one GPU context per process, like the file-scope state of icp.cpp:22-26
  static Tracker tracker(engine);
  return tracker;
}

// SLAM.cpp:305 hands every frame back as the next call's `previous` (previous = filtered.clone()).  With this switch
// on, calls after the first take `previous` from the device, where the last call's `data` has stayed, instead of
// uploading it again (icp::Tracker::getTransformation(data, nullptr, ...): same results).  Leave it off if the caller
// may hand over a `previous` that is NOT the last call's `data`.
inline bool& sequential_frames() {
  static bool on = false;
  return on;
}

inline cv::Mat getTransformation(cv::Mat& data, cv::Mat& previous, cv::Mat /*color*/,
                                 std::vector<cv::KeyPoint> /*keypoints*/, cv::Mat& /*rotation*/, int maxIterations,
                                 float threshold, cv::viz::Viz3d& /*depthWindow*/) {
  CV_Assert(data.type() == CV_16UC1 && previous.type() == CV_16UC1 && data.size() == previous.size());
  cv::Mat d = data.isContinuous() ? data : data.clone();
  cv::Mat p = previous.isContinuous() ? previous : previous.clone();
  cv::Mat rigidTransformation(4, 4, CV_32FC1);
  float T[16];
  static bool have_previous = false;  // a frame of this size is resident from the last call
  static int last_rows = 0, last_cols = 0;
  const bool resident = sequential_frames() && have_previous && last_rows == d.rows && last_cols == d.cols;
  int rc = default_tracker().getTransformation(d.ptr<uint16_t>(), resident ? nullptr : p.ptr<uint16_t>(), d.rows, d.cols,
                                               maxIterations, threshold, T);
  if (rc == ICPK_E_NOT_SET && resident)  // (the image buffers were used by another call in between: upload both)
    rc = default_tracker().getTransformation(d.ptr<uint16_t>(), p.ptr<uint16_t>(), d.rows, d.cols, maxIterations, threshold, T);
  have_previous = rc >= 0;
  last_rows = d.rows;
  last_cols = d.cols;
  std::memcpy(rigidTransformation.ptr<float>(), T, sizeof(T));
  return rigidTransformation;
}

// SLAM.cpp:553-574 (SLAM.hpp:34): same signature as the reference's global filterDepthImage, in
// place on the CV_16UC1 image; the colour image is not touched (the reference does not touch it either)
inline void filterDepthImage(cv::Mat& image, cv::Mat& /*rgbImage*/, int maxDistance, int minDistance) {
  CV_Assert(image.type() == CV_16UC1);
  cv::Mat d = image.isContinuous() ? image : image.clone();
  const int rc = filterDepthImage(default_tracker().engine(), d.ptr<uint16_t>(), d.rows, d.cols, maxDistance, minDistance);
  CV_Assert(rc == ICPK_OK);
  if (d.data != image.data) d.copyTo(image);
}

inline cv::Mat makeRotationMatrix(float x, float y, float z) {  // icp.cpp:640-653
  cv::Mat m(3, 3, CV_32FC1);
  icpk_make_rotation_matrix(x, y, z, m.ptr<float>());
  return m;
}

}  // namespace icp
