// solve_impl.h -- the O(1) dense algebra of the ICP step (what the reference delegates
// to OpenCV: cv::SVD, cv::determinant, Mat::inv, 3x3 GEMM) and its small pose helpers,
// float64 internally.  Header-inline and __host__ __device__: the same source runs in
// the host loop (icpk_api.cpp) and in the device-side loop (kernels_loop.hip), compiled
// with -ffp-contract=off on both sides so the two agree bit for bit wherever the
// hardware's +,-,*,/ and sqrt are correctly rounded.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ICPK_HD __host__ __device__ inline
#else
#define ICPK_HD inline
#endif

namespace icpk {

struct Mat3 {
  double m[3][3];
};

namespace detail {

ICPK_HD double col_dot(const Mat3& W, int p, int q) {
  return W.m[0][p] * W.m[0][q] + W.m[1][p] * W.m[1][q] + W.m[2][p] * W.m[2][q];
}

ICPK_HD void rotate_cols(Mat3& W, int p, int q, double c, double s) {
  for (int r = 0; r < 3; ++r) {
    const double wp = W.m[r][p], wq = W.m[r][q];
    W.m[r][p] = c * wp - s * wq;
    W.m[r][q] = s * wp + c * wq;
  }
}

ICPK_HD void cross_cols(const Mat3& U, int a, int b, double out[3]) {
  out[0] = U.m[1][a] * U.m[2][b] - U.m[2][a] * U.m[1][b];
  out[1] = U.m[2][a] * U.m[0][b] - U.m[0][a] * U.m[2][b];
  out[2] = U.m[0][a] * U.m[1][b] - U.m[1][a] * U.m[0][b];
}

// R = V U^T
ICPK_HD void v_ut(const Mat3& V, const Mat3& U, double R[9]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R[3 * r + c] = V.m[r][0] * U.m[c][0] + V.m[r][1] * U.m[c][1] + V.m[r][2] * U.m[c][2];
}

ICPK_HD double det9(const double m[9]) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

}  // namespace detail
using namespace detail;

ICPK_HD double det3(const Mat3& A) {
  return A.m[0][0] * (A.m[1][1] * A.m[2][2] - A.m[1][2] * A.m[2][1]) -
         A.m[0][1] * (A.m[1][0] * A.m[2][2] - A.m[1][2] * A.m[2][0]) +
         A.m[0][2] * (A.m[1][0] * A.m[2][1] - A.m[1][1] * A.m[2][0]);
}

// One-sided Jacobi: orthogonalise the columns of W = A V by plane rotations.
ICPK_HD void svd3(const Mat3& A, Mat3& U, double S[3], Mat3& V) {
  Mat3 W = A;
  Mat3 Q = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
  const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
  for (int sweep = 0; sweep < 64; ++sweep) {
    bool any = false;
    for (const auto& pq : pairs) {
      const int p = pq[0], q = pq[1];
      const double app = col_dot(W, p, p), aqq = col_dot(W, q, q), apq = col_dot(W, p, q);
      if (apq * apq <= 1e-30 * (app * aqq)) continue;  // columns orthogonal to 1e-15: R is then exact to ~1e-15
      any = true;
      const double zeta = (aqq - app) / (2.0 * apq);
      const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
      const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
      rotate_cols(W, p, q, c, s);
      rotate_cols(Q, p, q, c, s);
    }
    if (!any) break;
  }
  double sv[3];
  int ord[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) sv[j] = std::sqrt(col_dot(W, j, j));
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (sv[ord[b]] > sv[ord[a]]) { const int t_ = ord[a]; ord[a] = ord[b]; ord[b] = t_; }
  bool good[3];
  for (int k = 0; k < 3; ++k) {
    const int j = ord[k];
    S[k] = sv[j];
    good[k] = sv[j] > 1e-300 && sv[j] > 1e-15 * sv[ord[0]];
    for (int r = 0; r < 3; ++r) {
      V.m[r][k] = Q.m[r][j];
      U.m[r][k] = good[k] ? W.m[r][j] / sv[j] : 0.0;
    }
  }
  if (!good[0]) {  // zero matrix
    U = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
    return;
  }
  if (!good[1]) {  // rank 1: any unit vector orthogonal to u0
    int mi = 0;
    for (int r = 1; r < 3; ++r)
      if (std::fabs(U.m[r][0]) < std::fabs(U.m[mi][0])) mi = r;
    double v[3];
    for (int r = 0; r < 3; ++r) v[r] = (r == mi ? 1.0 : 0.0) - U.m[mi][0] * U.m[r][0];
    const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    for (int r = 0; r < 3; ++r) U.m[r][1] = v[r] / n;
  }
  if (!good[1] || !good[2]) {
    double c[3];
    cross_cols(U, 0, 1, c);
    for (int r = 0; r < 3; ++r) U.m[r][2] = c[r];
  }
}

// ---- orthogonal polar factor by Newton's iteration ------------------------------------------
// The rotation both solve flavours want is the orthogonal polar factor of the 3x3 moment:
// A = U S V^T  =>  Q = U V^T (A = Q P, P symmetric positive semi-definite), independent of the
// SVD algorithm and its sign conventions whenever A is non-singular.  Newton's iteration
//     Q <- (g Q + (g Q)^-T) / 2,   (g Q)^-T = cof(Q) / (g det Q),
// with determinantal scaling g ~ |det Q|^(-1/3) rounded to a power of two (exact), reaches it in
// 6-9 steps of ONE division each, against ~15 plane rotations of 3 divisions + 2 square roots for
// the Jacobi SVD above (4 us of the device loop's single-lane step at 92k points).  After an
// unscaled step every singular value is >= 1, so det - 1 bounds the distance to orthogonality:
// once g == 1 and |det| - 1 <= 1e-8 the next iterate is orthogonal to ~1e-16.
// Only +, -, *, /, fma and exponent-field arithmetic: bit-identical on host and device.
// Returns false (caller falls back to svd3) when A is not safely invertible: a non-finite or
// tiny / huge entry, |det| < 1e-7 after scaling the largest entry to [1, 2) (cond ~> 1e7:
// rank-deficient moments such as an exactly planar centred cloud), or no convergence.
ICPK_HD int exp2_floor(double x) {  // floor(log2 x) of a positive normal double
  uint64_t u;
  std::memcpy(&u, &x, sizeof(u));
  return (int)((u >> 52) & 0x7ffu) - 1023;
}
ICPK_HD double pow2i(int e) {  // 2^e, -1022 <= e <= 1023
  const uint64_t u = (uint64_t)(e + 1023) << 52;
  double d;
  std::memcpy(&d, &u, sizeof(d));
  return d;
}

ICPK_HD bool polar3(const double A[9], double Q[9]) {
  double amax = 0.0;
  for (int k = 0; k < 9; ++k) {
    const double v = std::fabs(A[k]);
    if (!(v <= 1e290)) return false;  // inf / NaN / absurd
    amax = v > amax ? v : amax;
  }
  if (!(amax >= 1e-290)) return false;
  const double s0 = pow2i(-exp2_floor(amax));  // largest entry -> [1, 2)
  for (int k = 0; k < 9; ++k) Q[k] = A[k] * s0;
  for (int it = 0; it < 40; ++it) {
    double c[9];
    c[0] = __builtin_fma(Q[4], Q[8], -(Q[5] * Q[7]));
    c[1] = __builtin_fma(Q[5], Q[6], -(Q[3] * Q[8]));
    c[2] = __builtin_fma(Q[3], Q[7], -(Q[4] * Q[6]));
    c[3] = __builtin_fma(Q[7], Q[2], -(Q[8] * Q[1]));
    c[4] = __builtin_fma(Q[8], Q[0], -(Q[6] * Q[2]));
    c[5] = __builtin_fma(Q[6], Q[1], -(Q[7] * Q[0]));
    c[6] = __builtin_fma(Q[1], Q[5], -(Q[2] * Q[4]));
    c[7] = __builtin_fma(Q[2], Q[3], -(Q[0] * Q[5]));
    c[8] = __builtin_fma(Q[0], Q[4], -(Q[1] * Q[3]));
    const double det = __builtin_fma(Q[2], c[2], __builtin_fma(Q[1], c[1], Q[0] * c[0]));
    const double ad = std::fabs(det);
    if (it == 0 && !(ad >= 1e-7)) return false;
    if (!(ad >= 1e-30) || !(ad <= 1e30)) return false;
    const int ed = exp2_floor(ad);                                  // |det| in [2^ed, 2^(ed+1))
    const int eg = ed >= 0 ? -((ed + 1) / 3) : ((1 - ed) / 3);     // ~ -ed/3; 0 for |det| in [1/2, 4)
    // (it > 0: Q is then a Newton iterate, every singular value is >= 1 and |det| - 1 bounds their excess e;
    // the step below maps 1 + e to 1 + e^2 / 2: 5e-11 from 1e-5, three decades below the float32 the rotation
    // is delivered in and below the 1e-8 the pinned Kabsch vectors are held to)
    const bool last = it > 0 && eg == 0 && ad - 1.0 <= 1e-5;
    const double hg = pow2i(eg - 1);                 // g / 2
    const double hinv = 0.5 / (det * pow2i(eg));     // 1 / (2 g det)
    for (int k = 0; k < 9; ++k) Q[k] = __builtin_fma(hg, Q[k], hinv * c[k]);
    if (last) return true;
  }
  return false;
}

ICPK_HD void solve_reference(const float M[9], float R[9]) {
  double Ad[9], Qd[9], Rd[9];
  for (int k = 0; k < 9; ++k) Ad[k] = M[k];
  if (polar3(Ad, Qd)) {
    // icp.cpp:215-218  R = svd.vt.t() * svd.u.t() = V U^T = (U V^T)^T
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Rd[3 * r + c] = Qd[3 * c + r];
  } else {  // (near-)singular moment: the SVD's completion of the null directions
    Mat3 A, U, V;
    double S[3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) A.m[r][c] = M[3 * r + c];
    svd3(A, U, S, V);
    v_ut(V, U, Rd);
  }
  double Rf[9];
  for (int k = 0; k < 9; ++k) {
    R[k] = (float)Rd[k];
    Rf[k] = R[k];
  }
  if (det9(Rf) < 0) {  // icp.cpp:220-223: negate column 2 of R (not of V)
    R[2] = -R[2];
    R[5] = -R[5];
    R[8] = -R[8];
  }
}

ICPK_HD bool invert3f(const float Rin[9], float out[9]) {
  double m[9];
  for (int k = 0; k < 9; ++k) m[k] = Rin[k];
  double d = det9(m);
  if (d == 0.0) {
    for (int k = 0; k < 9; ++k) out[k] = 0.f;
    return false;
  }
  d = 1.0 / d;
  const double t[9] = {(m[4] * m[8] - m[5] * m[7]) * d, (m[2] * m[7] - m[1] * m[8]) * d, (m[1] * m[5] - m[2] * m[4]) * d,
                       (m[5] * m[6] - m[3] * m[8]) * d, (m[0] * m[8] - m[2] * m[6]) * d, (m[2] * m[3] - m[0] * m[5]) * d,
                       (m[3] * m[7] - m[4] * m[6]) * d, (m[1] * m[6] - m[0] * m[7]) * d, (m[0] * m[4] - m[1] * m[3]) * d};
  for (int k = 0; k < 9; ++k) out[k] = (float)t[k];
  return true;
}

ICPK_HD void mul3f(const float A[9], const float B[9], float C[9]) {
  float t[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += (double)A[3 * r + k] * (double)B[3 * k + c];
      t[3 * r + c] = (float)s;
    }
  for (int k = 0; k < 9; ++k) C[k] = t[k];
}

ICPK_HD void solve_kabsch(int64_t n, const double sa[3], const double sb[3], const double sab[9], double R[9], double t[3]) {
  double ca[3], cb[3];
  for (int k = 0; k < 3; ++k) {
    ca[k] = sa[k] / (double)n;  // rigid_transform_3D.py:14-15
    cb[k] = sb[k] / (double)n;
  }
  Mat3 H, U, V;
  double S[3], Hd[9], Qd[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Hd[3 * r + c] = H.m[r][c] = sab[3 * r + c] - (double)n * ca[r] * cb[c];  // :18-22 AA^T BB
  // :26-28  U,S,Vt = svd(H); R = Vt.T * U.T = (U V^T)^T: the polar factor transposed, when it is
  // a proper rotation (det H > 0); the reflection branch (:31-34) flips the SMALLEST singular
  // direction, which only the SVD knows
  if (det9(Hd) > 0 && polar3(Hd, Qd)) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R[3 * r + c] = Qd[3 * c + r];
  } else {
    svd3(H, U, S, V);
    v_ut(V, U, R);        // :28  R = Vt.T * U.T
    if (det9(R) < 0) {    // :31-34 negate the last row of Vt
      for (int r = 0; r < 3; ++r) V.m[r][2] = -V.m[r][2];
      v_ut(V, U, R);
    }
  }
  for (int r = 0; r < 3; ++r) t[r] = -(R[3 * r] * ca[0] + R[3 * r + 1] * ca[1] + R[3 * r + 2] * ca[2]) + cb[r];  // :36
}

ICPK_HD bool solve_p2l(const double sums[28], double R[9], double t[3]) {
  double A[6][6], L[6][6] = {}, y[6], x[6];
  for (int a = 0, k = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b, ++k) A[a][b] = A[b][a] = sums[k];
  double dmax = 0.0;
  for (int a = 0; a < 6; ++a) dmax = A[a][a] > dmax ? A[a][a] : dmax;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i][j];
      for (int m = 0; m < j; ++m) s -= L[i][m] * L[j][m];
      if (i == j) {
        if (!(s > 1e-12 * dmax)) return false;  // not positive definite: degenerate geometry
        L[i][i] = std::sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  for (int i = 0; i < 6; ++i) {  // L y = -b
    double s = -sums[21 + i];
    for (int m = 0; m < i; ++m) s -= L[i][m] * y[m];
    y[i] = s / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {  // L^T x = y
    double s = y[i];
    for (int m = i + 1; m < 6; ++m) s -= L[m][i] * x[m];
    x[i] = s / L[i][i];
  }
  // Rodrigues: R = I + A1 K + B1 K^2, K = [alpha]x
  const double a0 = x[0], a1 = x[1], a2 = x[2];
  const double th2 = (a0 * a0 + a1 * a1) + a2 * a2, th = std::sqrt(th2);
  double A1, B1;
  if (th < 1e-9) {
    A1 = 1.0 - th2 / 6.0;
    B1 = 0.5 - th2 / 24.0;
  } else {
    A1 = std::sin(th) / th;
    B1 = (1.0 - std::cos(th)) / th2;
  }
  const double K[9] = {0, -a2, a1, a2, 0, -a0, -a1, a0, 0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      const double k2 = (K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c]) + K[3 * r + 2] * K[6 + c];
      R[3 * r + c] = (r == c ? 1.0 : 0.0) + (A1 * K[3 * r + c] + B1 * k2);
    }
  t[0] = x[3];
  t[1] = x[4];
  t[2] = x[5];
  return true;
}

// icp.cpp:595-602 distance(cv::Point3f, cv::Point3f): float differences, pow(float, int)
// promotes to double, the sum and the sqrt are double, the result is narrowed on return.
// (Not on the loop's path in the reference -- its one call site, pointcloud.cpp:246, is dead --
// restated for SURVEY.md 8a row 7.)  The products of floats are exact in double, so the
// explicit fma rounds like the reference's separate multiply and add.
ICPK_HD float distance3(float ax, float ay, float az, float bx, float by, float bz) {
  const float x = ax - bx, y = ay - by, z = az - bz;
  const double s = ((double)x * (double)x + (double)y * (double)y) + (double)z * (double)z;
  return (float)std::sqrt(s);
}

ICPK_HD void make_rotation_matrix(float x, float y, float z, float out[9]) {
  const float PI = 3.14159265358979f;  // icp.hpp:4
  const double rx = x * PI / 180, ry = y * PI / 180, rz = z * PI / 180;
  const float d[9] = {1, 0, 0, 0, (float)std::cos(rx), (float)std::sin(rx), 0, (float)-std::sin(rx), (float)std::cos(rx)};
  const float f[9] = {(float)std::cos(ry), 0, (float)-std::sin(ry), 0, 1, 0, (float)std::sin(ry), 0, (float)std::cos(ry)};
  const float g[9] = {(float)std::cos(rz), (float)std::sin(rz), 0, (float)-std::sin(rz), (float)std::cos(rz), 0, 0, 0, 1};
  float ab[9];
  mul3f(d, f, ab);
  mul3f(ab, g, out);
}

ICPK_HD void matrix_to_quaternion(const float m[9], float q[4]) {
  struct { ICPK_HD float operator()(float v) const { return v >= 0.0f ? 1.0f : -1.0f; } } sgn;  // quaternion.hpp:22
  const float r11 = m[0], r12 = m[1], r13 = m[2], r21 = m[3], r22 = m[4], r23 = m[5], r31 = m[6], r32 = m[7], r33 = m[8];
  float w = (r11 + r22 + r33 + 1.0f) / 4.0f;
  float x = (r11 - r22 - r33 + 1.0f) / 4.0f;
  float y = (-r11 + r22 - r33 + 1.0f) / 4.0f;
  float z = (-r11 - r22 + r33 + 1.0f) / 4.0f;
  w = std::sqrt(w < 0.0f ? 0.0f : w);
  x = std::sqrt(x < 0.0f ? 0.0f : x);
  y = std::sqrt(y < 0.0f ? 0.0f : y);
  z = std::sqrt(z < 0.0f ? 0.0f : z);
  if (w >= x && w >= y && w >= z) {
    x *= sgn(r32 - r23);
    y *= sgn(r13 - r31);
    z *= sgn(r21 - r12);
  } else if (x >= w && x >= y && x >= z) {
    w *= sgn(r32 - r23);
    y *= sgn(r21 + r12);
    z *= sgn(r13 + r31);
  } else if (y >= w && y >= x && y >= z) {
    w *= sgn(r13 - r31);
    x *= sgn(r21 + r12);
    z *= sgn(r32 + r23);
  } else if (z >= w && z >= x && z >= y) {
    w *= sgn(r21 - r12);
    x *= sgn(r31 + r13);
    y *= sgn(r32 + r23);
  }
  const float r = std::sqrt(w * w + x * x + y * y + z * z);  // quaternion.hpp:23
  q[0] = w / r;
  q[1] = x / r;
  q[2] = y / r;
  q[3] = z / r;
}

ICPK_HD void quaternion_to_euler(const float q[4], float e[3]) {
  const float PI = 3.14159265358979f;
  const float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const float ysqr = qy * qy;
  const float t0 = 2.0f * (qw * qx + qy * qz);
  const float t1 = 1.0f - 2.0f * (qx * qx + ysqr);
  const float ex = std::atan2(t0, t1);
  float t2 = 2.0f * (qw * qy - qz * qx);
  t2 = t2 > 1.0f ? 1.0f : t2;
  t2 = t2 < -1.0f ? -1.0f : t2;
  const float ey = std::asin(t2);
  const float t3 = 2.0f * (qw * qz + qx * qy);
  const float t4 = 1.0f - 2.0f * (ysqr + qz * qz);
  const float ez = std::atan2(t3, t4);
  e[0] = ex * 180.0f / PI;
  e[1] = ey * 180.0f / PI;
  e[2] = ez * 180.0f / PI;
}

}  // namespace icpk
