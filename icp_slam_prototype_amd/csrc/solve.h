// solve.h -- host-side 3x3 solves of the ICP step (float64 arithmetic).
#pragma once
#include <stdint.h>

namespace icpk {

struct Mat3 {
  double m[3][3];
};

// A = U diag(S) V^T, S descending, U and V orthogonal.
void svd3(const Mat3& A, Mat3& U, double S[3], Mat3& V);
double det3(const Mat3& A);

// icp.cpp:215-223
void solve_reference(const float M[9], float R[9]);
// icp.cpp:235 (cv::Mat::inv on a 3x3 CV_32F)
bool invert3f(const float R[9], float out[9]);
// icp.cpp:231 / :652 (3x3 CV_32F product)
void mul3f(const float A[9], const float B[9], float C[9]);
// rigid_transform_3D.py:9-40 from raw sums
void solve_kabsch(int64_t n, const double sa[3], const double sb[3], const double sab[9], double R[9], double t[3]);

// point-to-plane step (extension): sums = 21 upper-triangle J J^T, 6 J r, (1 unused);
// solves A x = -b by Cholesky, x = (alpha, t), R = exp([alpha]x).  false if A is not SPD.
bool solve_p2l(const double sums[28], double R[9], double t[3]);

// icp.cpp:640-653, quaternion.cpp:23-79, SLAM.cpp:613-636
void make_rotation_matrix(float x, float y, float z, float out[9]);
void matrix_to_quaternion(const float m[9], float q[4]);
void quaternion_to_euler(const float q[4], float e[3]);

}  // namespace icpk
