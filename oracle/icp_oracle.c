/*
 * oracle/icp_oracle.c -- CPU restatement of the ICP inner loop of
 * BenniG123/icp-slam-prototype.
 *
 * ***  TEST INFRASTRUCTURE ONLY.  ***
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.  The product (libicpk.so and everything in
 * icp_slam_prototype_amd/) never links, loads or calls it.
 *
 * PARITY STATUS
 *   - The C++ hot path of the reference (icp.cpp / pointcloud.cpp) cannot be
 *     built in this image: every translation unit includes OpenCV 3.2
 *     (core + viz), which is absent, and the rules of this build forbid
 *     stand-ins for missing libraries.  The reference ships no golden vectors
 *     for that path.  Hence the nearest-neighbour, reduction and
 *     reference-flavour solve restated below are "PARITY UNPINNED": they
 *     follow the cited source lines and C++ language rules, and are checked
 *     only against hand-computable known answers (tests/test_oracle_*.py).
 *   - The centred Kabsch solve IS pinned: tests/golden/kabsch_*.npz were
 *     produced by executing the reference's own rigid_transform_3D.py
 *     (tests/golden/make_kabsch_golden.py) and orc_solve_kabsch reproduces
 *     them.
 *
 * All citations "file:line" are relative to /root/reference.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ---- constants restated from the reference's headers -------------------- */
#define ORC_PI_F 3.14159265358979f /* icp.hpp:4  (a float literal)           */
#define ORC_FX 468.60f             /* pointcloud.hpp:7                       */
#define ORC_FY 468.61f             /* pointcloud.hpp:8  (unused by the ref)  */
#define ORC_CX 318.27f             /* pointcloud.hpp:9                       */
#define ORC_CY 243.99f             /* pointcloud.hpp:10 (unused by the ref)  */

/* canonical reduction geometry -- must equal include/icpk.h ICPK_RED_*      */
#define ORC_RED_THREADS 256
#define ORC_RED_MAX_BLOCKS 256
#define ORC_NSUM 19 /* 9 M + 3 S + 1 E + 3 A + 3 B */

/* ------------------------------------------------------------------------ */
/* icp.cpp:606-620  distance(color_point_t, color_point_t)                   */
/*   x,y,z are float differences; pow(float,int) promotes to double under    */
/*   C++11, so the three squares and both additions are double; the sum is   */
/*   rounded once to the float `xyz`; colour weight is 0 (icp.hpp:6) and     */
/*   `xyz * 1.0f - 0.0f + rgb * 0.0f` == xyz for finite rgb; sqrt(float) is  */
/*   the float overload.                                                     */
/* ------------------------------------------------------------------------ */
static inline float orc_dist3(float ax, float ay, float az, float bx, float by,
                              float bz) {
  float x = ax - bx;
  float y = ay - by;
  float z = az - bz;
  float xyz = (float)(((double)x * (double)x + (double)y * (double)y) +
                      (double)z * (double)z);
  return sqrtf(xyz);
}

ORC_API float orc_distance(float ax, float ay, float az, float bx, float by,
                           float bz) {
  return orc_dist3(ax, ay, az, bx, by, bz);
}

/* ------------------------------------------------------------------------ */
/* icp.cpp:541-563 findGlobalNearestNeighborAssociations +                   */
/* icp.cpp:566-593 getNearestPoint: linear scan seeded with element 0,       */
/* strict '<' on the float distance => lowest index wins ties.               */
/* Output is per query (index of the element the scan would copy, and its    */
/* distance); the `d < max` acceptance (icp.cpp:553) is applied by callers.  */
/* threads<=1: single thread like the reference; >1: OpenMP over queries     */
/* (each query's scan is still sequential, results identical).               */
/* Returns 0, or -1 if the target is empty (icp.cpp:572 dereferences         */
/* begin() unconditionally: UB in the reference).                            */
/* ------------------------------------------------------------------------ */
ORC_API int orc_nn_bruteforce(const float *qx, const float *qy, const float *qz,
                              int nq, const float *tx, const float *ty,
                              const float *tz, int nt, int32_t *idx,
                              float *dist, int threads) {
  if (nt <= 0) return -1;
#ifdef _OPENMP
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
  for (int i = 0; i < nq; i++) {
    float px = qx[i], py = qy[i], pz = qz[i];
    int32_t bi = 0;
    float best = orc_dist3(px, py, pz, tx[0], ty[0], tz[0]);
    for (int j = 1; j < nt; j++) {
      float d = orc_dist3(px, py, pz, tx[j], ty[j], tz[j]);
      if (d < best) {
        best = d;
        bi = j;
      }
    }
    idx[i] = bi;
    dist[i] = best;
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* icp.cpp:314-344 calculateOffset: sequential float accumulation of         */
/* (a - b) (cv::Point3f operator- is component-wise float) over accepted     */
/* pairs in query order, then float division by the count.                   */
/* ------------------------------------------------------------------------ */
/* icp.cpp:595-602 distance(cv::Point3f, cv::Point3f): x, y, z float            */
/* differences; pow(float, int) -> double; double sum, double sqrt; the result  */
/* is narrowed to float by the return type.  (Call site pointcloud.cpp:246 is   */
/* dead code in the reference.)                                                 */
/* ------------------------------------------------------------------------ */
ORC_API float orc_distance3(float ax, float ay, float az, float bx, float by, float bz) {
  float x = ax - bx;
  float y = ay - by;
  float z = az - bz;
  return (float)sqrt(((double)x * (double)x + (double)y * (double)y) + (double)z * (double)z);
}

/* ------------------------------------------------------------------------ */
/* icp.cpp:488-515 findGlobalKeyPointAssociations + :517-539                   */
/* getNearestKeyPoint.  Same scan as getNearestPoint over the key-point lists; */
/* a query is accepted when d < max_dist (MAX_NN_KEYPOINT_DISTANCE 0.1f,       */
/* icp.hpp:10), else APPENDED to nonAssociations (:507-509; the caller's list  */
/* is never cleared, so *n_rej is in/out); an empty map returns BEFORE         */
/* errors/associations are cleared (:490-491): nothing is touched, return 1.   */
/* assoc_q / assoc_t: indices of the accepted (query, nearest) pairs in query  */
/* order; assoc_d their distances (the `errors` vector).                       */
/* ------------------------------------------------------------------------ */
ORC_API int orc_keypoint_associations(const float *qx, const float *qy, const float *qz, int nq,
                                      const float *tx, const float *ty, const float *tz, int nt,
                                      float max_dist, int32_t *assoc_q, int32_t *assoc_t,
                                      float *assoc_d, int32_t *n_assoc, int32_t *rej_q,
                                      int32_t *n_rej) {
  if (nt == 0) return 1;
  int na = 0, nr = *n_rej;
  for (int i = 0; i < nq; i++) {
    int best = 0;
    float sd = orc_dist3(qx[i], qy[i], qz[i], tx[0], ty[0], tz[0]);
    for (int j = 1; j < nt; j++) {
      float d = orc_dist3(qx[i], qy[i], qz[i], tx[j], ty[j], tz[j]);
      if (d < sd) {
        sd = d;
        best = j;
      }
    }
    if (sd < max_dist) {
      assoc_q[na] = i;
      assoc_t[na] = best;
      assoc_d[na] = sd;
      na++;
    } else {
      rej_q[nr++] = i;
    }
  }
  *n_assoc = na;
  *n_rej = nr;
  return 0;
}

/* ------------------------------------------------------------------------ */
ORC_API int orc_calculate_offset_seq(const float *ax, const float *ay,
                                     const float *az, int nq, const float *tx,
                                     const float *ty, const float *tz,
                                     const int32_t *idx, const float *dist,
                                     float max_dist, float out[3]) {
  float ox = 0.f, oy = 0.f, oz = 0.f;
  int count = 0;
  for (int i = 0; i < nq; i++) {
    if (!(dist[i] < max_dist)) continue; /* icp.cpp:553 */
    int j = idx[i];
    float dx = ax[i] - tx[j];
    float dy = ay[i] - ty[j];
    float dz = az[i] - tz[j];
    ox += dx;
    oy += dy;
    oz += dz;
    count++;
  }
  if (count > 0) {
    ox /= (float)count; /* icp.cpp:338 float /= int */
    oy /= (float)count;
    oz /= (float)count;
  }
  out[0] = ox;
  out[1] = oy;
  out[2] = oz;
  return count;
}

/* ------------------------------------------------------------------------ */
/* icp.cpp:622-638 meanSquareError: (mean of distances)^2, float sequential; */
/* pow(float,2) is double, stored back into the float.                       */
/* ------------------------------------------------------------------------ */
ORC_API float orc_mse_seq(const float *dist, int nq, float max_dist) {
  float sum = 0.f;
  int n = 0;
  for (int i = 0; i < nq; i++) {
    if (!(dist[i] < max_dist)) continue;
    sum += dist[i];
    n++;
  }
  if (n > 0) {
    sum /= (float)n;
    sum = (float)((double)sum * (double)sum);
  }
  return sum;
}

/* ------------------------------------------------------------------------ */
/* icp.cpp:199-212: M = previousMat.t() * dataMat, both N x 3 CV_32F built   */
/* by PointCloud::centered_matrix (pointcloud.cpp:361-371, which does NOT    */
/* centre).  M[r][c] = sum_i b_i[r] * a_i[c]  (b = matched target, a =       */
/* data).  OpenCV's 32F GEMM accumulates in double and stores float          */
/* (believed; OpenCV is absent => unpinned).  Sequential over pairs.         */
/* ------------------------------------------------------------------------ */
ORC_API int orc_cross_moment_seq(const float *ax, const float *ay,
                                 const float *az, int nq, const float *tx,
                                 const float *ty, const float *tz,
                                 const int32_t *idx, const float *dist,
                                 float max_dist, float M[9]) {
  double m[9] = {0};
  int count = 0;
  for (int i = 0; i < nq; i++) {
    if (!(dist[i] < max_dist)) continue;
    int j = idx[i];
    double a[3] = {ax[i], ay[i], az[i]};
    double b[3] = {tx[j], ty[j], tz[j]};
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) m[3 * r + c] += b[r] * a[c];
    count++;
  }
  for (int k = 0; k < 9; k++) M[k] = (float)m[k];
  return count;
}

#define ORC_NS_MAX 28
/* Canonical tree, shared by both reductions.  acc: [B*256][ns] per-virtual-thread
 * sums.  Stage 1 (per block of 256): 64-lane xor butterfly (32,16,...,1) in each of
 * the 4 waves, then ((w0+w1)+w2)+w3.  Stage 2: the B block sums fill 256 slots (rest
 * +0.0) and are combined by exactly the same 4-wave butterfly tree.                 */
static void orc_tree256(const double *v /*[256][ns]*/, int ns, double *out) {
  double w[4][ORC_NS_MAX];
  double lane[64][ORC_NS_MAX], nxt[64][ORC_NS_MAX];
  for (int wv = 0; wv < 4; wv++) {
    for (int l = 0; l < 64; l++) memcpy(lane[l], v + (size_t)(wv * 64 + l) * ns, sizeof(double) * ns);
    for (int m = 32; m >= 1; m >>= 1) {
      for (int l = 0; l < 64; l++)
        for (int s = 0; s < ns; s++) nxt[l][s] = lane[l][s] + lane[l ^ m][s];
      memcpy(lane, nxt, sizeof(lane));
    }
    memcpy(w[wv], lane[0], sizeof(double) * ns);
  }
  for (int s = 0; s < ns; s++) out[s] = ((w[0][s] + w[1][s]) + w[2][s]) + w[3][s];
}

static void orc_tree_finish(const double *acc, int B, int ns, double *sums) {
  double *slots = (double *)calloc((size_t)256 * ns, sizeof(double));
  for (int b = 0; b < B; b++) orc_tree256(acc + (size_t)b * 256 * ns, ns, slots + (size_t)b * ns);
  orc_tree256(slots, ns, sums);
  free(slots);
}

/* ------------------------------------------------------------------------ */
/* Canonical (order-defined) double reduction.  This is NOT in the           */
/* reference; it defines a summation tree that a parallel machine can        */
/* reproduce bit for bit, so the HIP path and this oracle agree exactly.     */
/*   B = clamp(ceil(nq/256), 1, 256) blocks of 256 virtual threads;          */
/*   thread g accumulates elements g, g+P, g+2P ... (P = 256 B) in order;    */
/*   64-lane xor butterfly (32,16,8,4,2,1); the 4 wave sums of a block are   */
/*   added ((w0+w1)+w2)+w3; the B block sums (padded to 256 with +0.0) go    */
/*   through the same 4-wave butterfly tree once more.                       */
/* sums layout: [0..8] M (row-major, M[r][c] = sum b_r a_c), [9..11] S =     */
/* sum (float)(a-b), [12] E = sum dist, [13..15] A = sum a, [16..18] B =     */
/* sum b.  Returns the accepted count.                                       */
/* ------------------------------------------------------------------------ */
ORC_API int64_t orc_sums_canonical(const float *ax, const float *ay,
                                   const float *az, int nq, const float *tx,
                                   const float *ty, const float *tz,
                                   const int32_t *idx, const float *dist,
                                   float max_dist, double sums[ORC_NSUM]) {
  int B = (nq + ORC_RED_THREADS - 1) / ORC_RED_THREADS;
  if (B < 1) B = 1;
  if (B > ORC_RED_MAX_BLOCKS) B = ORC_RED_MAX_BLOCKS;
  const int P = B * ORC_RED_THREADS;
  double *acc = (double *)calloc((size_t)P * ORC_NSUM, sizeof(double));
  int64_t count = 0;
  for (int g = 0; g < P; g++) {
    double *v = acc + (size_t)g * ORC_NSUM;
    for (int i = g; i < nq; i += P) {
      if (!(dist[i] < max_dist)) continue;
      int j = idx[i];
      float a[3] = {ax[i], ay[i], az[i]};
      float b[3] = {tx[j], ty[j], tz[j]};
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) v[3 * r + c] += (double)b[r] * (double)a[c];
      for (int k = 0; k < 3; k++) v[9 + k] += (double)(float)(a[k] - b[k]);
      v[12] += (double)dist[i];
      for (int k = 0; k < 3; k++) v[13 + k] += (double)a[k];
      for (int k = 0; k < 3; k++) v[16 + k] += (double)b[k];
      count++;
    }
  }
  orc_tree_finish(acc, B, ORC_NSUM, sums);
  free(acc);
  return count;
}

/* ======================================================================== */
/* small dense helpers (all "what OpenCV would do" parts are unpinned)       */
/* ======================================================================== */

/* 3x3 float product, double accumulation, float store (cv::gemm 32F).       */
static void mat3_mul_f(const float A[9], const float B[9], float C[9]) {
  float t[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += (double)A[3 * r + k] * (double)B[3 * k + c];
      t[3 * r + c] = (float)s;
    }
  memcpy(C, t, sizeof(t));
}

static double det3_d(const double m[9]) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
         m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* icp.cpp:235  R = R.inv(): cv::invert(DECOMP_LU) on a 3x3 CV_32F uses the  */
/* closed-form adjugate / determinant evaluated in double, stored as float.  */
ORC_API int orc_inv3_f(const float Rin[9], float out[9]) {
  double m[9];
  for (int k = 0; k < 9; k++) m[k] = Rin[k];
  double d = det3_d(m);
  if (d == 0.0) {
    for (int k = 0; k < 9; k++) out[k] = 0.f;
    return -1;
  }
  d = 1.0 / d;
  double t[9];
  t[0] = (m[4] * m[8] - m[5] * m[7]) * d;
  t[1] = (m[2] * m[7] - m[1] * m[8]) * d;
  t[2] = (m[1] * m[5] - m[2] * m[4]) * d;
  t[3] = (m[5] * m[6] - m[3] * m[8]) * d;
  t[4] = (m[0] * m[8] - m[2] * m[6]) * d;
  t[5] = (m[2] * m[3] - m[0] * m[5]) * d;
  t[6] = (m[3] * m[7] - m[4] * m[6]) * d;
  t[7] = (m[1] * m[6] - m[0] * m[7]) * d;
  t[8] = (m[0] * m[4] - m[1] * m[3]) * d;
  for (int k = 0; k < 9; k++) out[k] = (float)t[k];
  return 0;
}

/* One-sided (Hestenes) Jacobi SVD of a 3x3 double matrix: A = U diag(S) V^T,
 * S sorted descending, U and V orthogonal (U completed by cross product when
 * a singular value vanishes).                                               */
ORC_API void orc_svd3(const double A[9], double U[9], double S[3], double V[9]) {
  double W[9], Vt[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(W, A, sizeof(W));
  for (int sweep = 0; sweep < 60; sweep++) {
    int rotated = 0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; r++) {
          alpha += W[3 * r + p] * W[3 * r + p];
          beta += W[3 * r + q] * W[3 * r + q];
          gamma += W[3 * r + p] * W[3 * r + q];
        }
        if (gamma * gamma <= 1e-30 * (alpha * beta)) continue; /* orthogonal to 1e-15 */
        rotated = 1;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; r++) {
          double wp = W[3 * r + p], wq = W[3 * r + q];
          W[3 * r + p] = c * wp - s * wq;
          W[3 * r + q] = s * wp + c * wq;
          double vp = Vt[3 * r + p], vq = Vt[3 * r + q];
          Vt[3 * r + p] = c * vp - s * vq;
          Vt[3 * r + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double sv[3];
  for (int j = 0; j < 3; j++) {
    double n2 = 0;
    for (int r = 0; r < 3; r++) n2 += W[3 * r + j] * W[3 * r + j];
    sv[j] = sqrt(n2);
  }
  int ord[3] = {0, 1, 2};
  for (int a = 0; a < 2; a++)
    for (int b = a + 1; b < 3; b++)
      if (sv[ord[b]] > sv[ord[a]]) {
        int t = ord[a];
        ord[a] = ord[b];
        ord[b] = t;
      }
  double smax = sv[ord[0]];
  int good[3];
  for (int k = 0; k < 3; k++) {
    int j = ord[k];
    S[k] = sv[j];
    good[k] = (sv[j] > 1e-300 && sv[j] > 1e-15 * smax);
    for (int r = 0; r < 3; r++) {
      V[3 * r + k] = Vt[3 * r + j];
      U[3 * r + k] = good[k] ? W[3 * r + j] / sv[j] : 0.0;
    }
  }
  /* complete U for vanishing singular values */
  if (!good[0]) {
    for (int k = 0; k < 9; k++) U[k] = (k % 4 == 0);
  } else {
    if (!good[1]) {
      /* any unit vector orthogonal to u0 */
      double u0[3] = {U[0], U[3], U[6]};
      int m = 0;
      if (fabs(u0[1]) < fabs(u0[m])) m = 1;
      if (fabs(u0[2]) < fabs(u0[m])) m = 2;
      double e[3] = {0, 0, 0};
      e[m] = 1.0;
      double dot = u0[m];
      double v[3] = {e[0] - dot * u0[0], e[1] - dot * u0[1], e[2] - dot * u0[2]};
      double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      U[1] = v[0] / n;
      U[4] = v[1] / n;
      U[7] = v[2] / n;
    }
    if (!good[2] || !good[1]) {
      double a0 = U[0], a1 = U[3], a2 = U[6], b0 = U[1], b1 = U[4], b2 = U[7];
      U[2] = a1 * b2 - a2 * b1;
      U[5] = a2 * b0 - a0 * b2;
      U[8] = a0 * b1 - a1 * b0;
    }
  }
}

/* ------------------------------------------------------------------------ */
/* icp.cpp:215-223: cv::SVD svd(M); R = svd.vt.t() * svd.u.t();              */
/* if (cv::determinant(R) < 0) R.col(2) *= -1.   M and R are CV_32F.         */
/* Restated with a float64 SVD; R rounded to float (unpinned: OpenCV's float */
/* Jacobi differs in the last bits; R = V U^T is the orthogonal polar factor */
/* of M^T and is convention independent for non-singular M).                 */
/* ------------------------------------------------------------------------ */
ORC_API void orc_solve_reference(const float M[9], float R[9]) {
  double A[9], U[9], S[3], V[9], Rd[9];
  for (int k = 0; k < 9; k++) A[k] = M[k];
  orc_svd3(A, U, S, V);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += V[3 * r + k] * U[3 * c + k];
      Rd[3 * r + c] = s;
    }
  for (int k = 0; k < 9; k++) R[k] = (float)Rd[k];
  double Rf[9];
  for (int k = 0; k < 9; k++) Rf[k] = R[k];
  if (det3_d(Rf) < 0) {
    R[2] = -R[2];
    R[5] = -R[5];
    R[8] = -R[8];
  }
}

/* ------------------------------------------------------------------------ */
/* rigid_transform_3D.py:9-40 (centred Kabsch), all float64:                 */
/*   H = AA^T BB (H[r][c] = sum aa_r bb_c), U,S,Vt = svd(H), R = Vt^T U^T,   */
/*   if det(R) < 0: Vt[2,:] *= -1, R = Vt^T U^T;  t = -R cA + cB.            */
/* Input: raw sums (n, sum a, sum b, sum a b^T as Mt[r][c] = sum a_r b_c).   */
/* ------------------------------------------------------------------------ */
ORC_API void orc_solve_kabsch_from_sums(int64_t n, const double sa[3],
                                        const double sb[3], const double sab[9],
                                        double R[9], double t[3]) {
  double ca[3], cb[3], H[9], U[9], S[3], V[9];
  for (int k = 0; k < 3; k++) {
    ca[k] = sa[k] / (double)n;
    cb[k] = sb[k] / (double)n;
  }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) H[3 * r + c] = sab[3 * r + c] - (double)n * ca[r] * cb[c];
  orc_svd3(H, U, S, V);
  for (int pass = 0; pass < 2; pass++) {
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += V[3 * r + k] * U[3 * c + k];
        R[3 * r + c] = s;
      }
    if (pass == 0 && det3_d(R) < 0) {
      V[2] = -V[2];
      V[5] = -V[5];
      V[8] = -V[8];
    } else
      break;
  }
  for (int r = 0; r < 3; r++)
    t[r] = -(R[3 * r] * ca[0] + R[3 * r + 1] * ca[1] + R[3 * r + 2] * ca[2]) + cb[r];
}

/* Direct form: A (source) and B (target) as n x 3 row-major float64, exactly
 * the call shape of rigid_transform_3D(A, B); centred sums formed the way the
 * script does (mean, subtract, product).                                    */
ORC_API void orc_rigid_transform_3D(const double *A, const double *B, int n,
                                    double R[9], double t[3]) {
  double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0}, H[9] = {0}, U[9], S[3], V[9];
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) {
      ca[k] += A[3 * i + k];
      cb[k] += B[3 * i + k];
    }
  for (int k = 0; k < 3; k++) {
    ca[k] /= n;
    cb[k] /= n;
  }
  for (int i = 0; i < n; i++)
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
        H[3 * r + c] += (A[3 * i + r] - ca[r]) * (B[3 * i + c] - cb[c]);
  orc_svd3(H, U, S, V);
  for (int pass = 0; pass < 2; pass++) {
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += V[3 * r + k] * U[3 * c + k];
        R[3 * r + c] = s;
      }
    if (pass == 0 && det3_d(R) < 0) {
      V[2] = -V[2];
      V[5] = -V[5];
      V[8] = -V[8];
    } else
      break;
  }
  for (int r = 0; r < 3; r++)
    t[r] = -(R[3 * r] * ca[0] + R[3 * r + 1] * ca[1] + R[3 * r + 2] * ca[2]) + cb[r];
}

/* ------------------------------------------------------------------------ */
/* pointcloud.cpp:321-346 PointCloud::rotate (p <- R p about the world       */
/* origin via N x 3 -> 3 x N GEMM, 32F in/out, double accumulation) followed */
/* by pointcloud.cpp:349-359 translate (float +=).  One fused helper:        */
/*   p' = fl32( fl32(R p) + t ).                                             */
/* ------------------------------------------------------------------------ */
ORC_API void orc_transform_points(float *x, float *y, float *z, int n,
                                  const float R[9], const float t[3]) {
  for (int i = 0; i < n; i++) {
    double px = x[i], py = y[i], pz = z[i];
    float rx = (float)(((double)R[0] * px + (double)R[1] * py) + (double)R[2] * pz);
    float ry = (float)(((double)R[3] * px + (double)R[4] * py) + (double)R[5] * pz);
    float rz = (float)(((double)R[6] * px + (double)R[7] * py) + (double)R[8] * pz);
    x[i] = rx + t[0];
    y[i] = ry + t[1];
    z[i] = rz + t[2];
  }
}

/* ------------------------------------------------------------------------ */
/* icp.cpp:640-653 makeRotationMatrix(x deg, y deg, z deg) = Rx * Ry * Rz    */
/* with the reference's sign convention; `x * PI / 180` is float arithmetic  */
/* (PI is a float literal) widened to double for cos/sin, entries narrowed   */
/* to float, products via 32F GEMM.                                          */
/* ------------------------------------------------------------------------ */
ORC_API void orc_make_rotation_matrix(float x, float y, float z, float out[9]) {
  double rotX = x * ORC_PI_F / 180;
  double rotY = y * ORC_PI_F / 180;
  double rotZ = z * ORC_PI_F / 180;
  float d[9] = {1, 0, 0, 0, (float)cos(rotX), (float)sin(rotX), 0, (float)-sin(rotX), (float)cos(rotX)};
  float f[9] = {(float)cos(rotY), 0, (float)-sin(rotY), 0, 1, 0, (float)sin(rotY), 0, (float)cos(rotY)};
  float g[9] = {(float)cos(rotZ), (float)sin(rotZ), 0, (float)-sin(rotZ), (float)cos(rotZ), 0, 0, 0, 1};
  float ab[9];
  mat3_mul_f(d, f, ab);
  mat3_mul_f(ab, g, out);
}

/* ------------------------------------------------------------------------ */
/* quaternion.cpp:23-79 Quaternion(cv::Mat) -> (w,x,y,z), float arithmetic.  */
/* ------------------------------------------------------------------------ */
static inline float orc_sign(float v) { return (v >= 0.0f) ? +1.0f : -1.0f; } /* quaternion.hpp:22 */

ORC_API void orc_quaternion_from_matrix(const float m[9], float q[4]) {
  float r11 = m[0], r12 = m[1], r13 = m[2], r21 = m[3], r22 = m[4], r23 = m[5],
        r31 = m[6], r32 = m[7], r33 = m[8];
  float w = (r11 + r22 + r33 + 1.0f) / 4.0f;
  float x = (r11 - r22 - r33 + 1.0f) / 4.0f;
  float y = (-r11 + r22 - r33 + 1.0f) / 4.0f;
  float z = (-r11 - r22 + r33 + 1.0f) / 4.0f;
  if (w < 0.0f) w = 0.0f;
  if (x < 0.0f) x = 0.0f;
  if (y < 0.0f) y = 0.0f;
  if (z < 0.0f) z = 0.0f;
  w = sqrtf(w);
  x = sqrtf(x);
  y = sqrtf(y);
  z = sqrtf(z);
  if (w >= x && w >= y && w >= z) {
    x *= orc_sign(r32 - r23);
    y *= orc_sign(r13 - r31);
    z *= orc_sign(r21 - r12);
  } else if (x >= w && x >= y && x >= z) {
    w *= orc_sign(r32 - r23);
    y *= orc_sign(r21 + r12);
    z *= orc_sign(r13 + r31);
  } else if (y >= w && y >= x && y >= z) {
    w *= orc_sign(r13 - r31);
    x *= orc_sign(r21 + r12);
    z *= orc_sign(r32 + r23);
  } else if (z >= w && z >= x && z >= y) {
    w *= orc_sign(r21 - r12);
    x *= orc_sign(r31 + r13);
    y *= orc_sign(r32 + r23);
  }
  float r = sqrtf(w * w + x * x + y * y + z * z); /* quaternion.hpp:23 NORM */
  q[0] = w / r;
  q[1] = x / r;
  q[2] = y / r;
  q[3] = z / r;
}

/* SLAM.cpp:613-636 toEulerianAngle(Quaternion) -> degrees (float math).     */
ORC_API void orc_to_euler(const float q[4], float e[3]) {
  float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  float ysqr = qy * qy;
  float t0 = 2.0f * (qw * qx + qy * qz);
  float t1 = 1.0f - 2.0f * (qx * qx + ysqr);
  float x = atan2f(t0, t1);
  float t2 = +2.0f * (qw * qy - qz * qx);
  t2 = t2 > 1.0f ? 1.0f : t2;
  t2 = t2 < -1.0f ? -1.0f : t2;
  float y = asinf(t2);
  float t3 = +2.0f * (qw * qz + qx * qy);
  float t4 = +1.0f - 2.0f * (ysqr + qz * qz);
  float z = atan2f(t3, t4);
  e[0] = x * 180.0f / ORC_PI_F;
  e[1] = y * 180.0f / ORC_PI_F;
  e[2] = z * 180.0f / ORC_PI_F;
}

/* ------------------------------------------------------------------------ */
/* pointcloud.cpp:19-58 back-projection of a CV_16UC1 depth image, row-major */
/* scan, zero depth skipped, p_z = d/5000, p_x = (x - CX) p_z / FX and       */
/* p_y = (y - CX) p_z / FX (the reference uses CX and FX for y too,          */
/* pointcloud.cpp:39).  The unseeded rand()%40 subsample (pointcloud.cpp:28) */
/* is replaced by an optional caller-supplied keep mask (NULL = keep all).   */
/* Returns the number of points written.                                     */
/* ------------------------------------------------------------------------ */
ORC_API int orc_backproject(const uint16_t *depth, int rows, int cols,
                            const uint8_t *keep, float fx, float cx, float *x,
                            float *y, float *z) {
  int n = 0;
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      uint16_t d = depth[(size_t)r * cols + c];
      if (d == 0) continue;
      if (keep && !keep[(size_t)r * cols + c]) continue;
      float pz = ((float)d) / 5000.0f;
      float px = (c - cx) * pz / fx;
      float py = (r - cx) * pz / fx;
      x[n] = px;
      y[n] = py;
      z[n] = pz;
      n++;
    }
  return n;
}

/* SLAM.cpp:553-574 filterDepthImage, range clamp part only (the 5x5         */
/* dilate/erode that follows is OpenCV imgproc: out of scope).               */
ORC_API void orc_depth_range_filter(uint16_t *depth, int n, int max_d, int min_d) {
  for (int i = 0; i < n; i++) {
    if (depth[i] > max_d)
      depth[i] = 0;
    else if (depth[i] < min_d)
      depth[i] = 0;
  }
}

/* SLAM.cpp:568-573: cv::getStructuringElement(MORPH_RECT, Size(5,5), Point(3,3)), then   */
/* cv::dilate(image, image, element); cv::erode(image, image, element) on the CV_16UC1    */
/* depth image.  OpenCV (3.2, absent here) is third-party: PARITY UNPINNED for two rules  */
/* restated from its documentation: (1) the anchor handed to getStructuringElement only   */
/* shapes MORPH_CROSS elements -- a MORPH_RECT element is 5x5 ones, and dilate/erode are  */
/* called with their default anchor (-1,-1) = the element centre (2,2); the anchor is a   */
/* parameter here so that either reading can be run; (2) the default border is            */
/* BORDER_CONSTANT with morphologyDefaultBorderValue(): out-of-image pixels never win,    */
/* i.e. the max (min) runs over the in-image part of the window; a window wholly outside  */
/* (impossible with an in-range anchor) would give 0 (65535).                             */
/* Output pixel (y, x) looks at rows y - ay .. y - ay + 4, columns x - ax .. x - ax + 4.  */
static void orc_morph5(const uint16_t *in, uint16_t *out, int rows, int cols, int ax, int ay,
                       int erode) {
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      unsigned v = erode ? 65535u : 0u;
      for (int ky = 0; ky < 5; ky++)
        for (int kx = 0; kx < 5; kx++) {
          const int yy = y - ay + ky, xx = x - ax + kx;
          if (yy < 0 || yy >= rows || xx < 0 || xx >= cols) continue;
          const unsigned p = in[(size_t)yy * cols + xx];
          v = erode ? (p < v ? p : v) : (p > v ? p : v);
        }
      out[(size_t)y * cols + x] = (uint16_t)v;
    }
}

/* SLAM.cpp:553-574 filterDepthImage as a whole: range clamp, 5x5 dilate, 5x5 erode. */
ORC_API void orc_filter_depth_image(const uint16_t *in, uint16_t *out, int rows, int cols,
                                    int max_d, int min_d, int ax, int ay) {
  const size_t n = (size_t)rows * cols;
  uint16_t *a = (uint16_t *)malloc(n * sizeof(uint16_t) + 2);
  uint16_t *b = (uint16_t *)malloc(n * sizeof(uint16_t) + 2);
  memcpy(a, in, n * sizeof(uint16_t));
  orc_depth_range_filter(a, (int)n, max_d, min_d);
  orc_morph5(a, b, rows, cols, ax, ay, 0);
  orc_morph5(b, out, rows, cols, ax, ay, 1);
  free(a);
  free(b);
}

/* ======================================================================== */
/* Point-to-plane extension (BASELINE config 3).  NOT in the reference        */
/* (TODO:9 "Look into Point to Plane"); PARITY UNPINNED by construction: this */
/* float64 restatement is the only oracle.  The one reference artefact is the */
/* dead getNormalMap (SLAM.cpp:412-430), restated as normals mode 1.          */
/* ======================================================================== */
#define ORC_NP2L 28 /* 21 upper-triangle J J^T + 6 J r + 1 sum dist */

/* Back-projection (as orc_backproject, no keep mask) plus one normal per      */
/* valid pixel, compacted in the same order.                                   */
/* mode 0: n = normalize((P(r,c+1)-P(r,c-1)) x (P(r+1,c)-P(r-1,c))), float     */
/*         arithmetic, all four neighbours must be valid, else n = 0;          */
/* mode 1: SLAM.cpp:421-425 literally: central differences of the RAW depth    */
/*         (as float) along rows ("x") and columns ("y"), d = (-dzdx,-dzdy,1), */
/*         cv::normalize (norm in double, components scaled by 1/norm);        */
/*         interior pixels only (the reference reads out of bounds at the      */
/*         last row/column), border n = 0.                                     */
ORC_API int orc_backproject_normals(const uint16_t *depth, int rows, int cols, float fx,
                                    float cx, int mode, float *x, float *y, float *z,
                                    float *nx, float *ny, float *nz) {
#define ORC_D(r, c) depth[(size_t)(r) * cols + (c)]
  int n = 0;
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      uint16_t d = ORC_D(r, c);
      if (d == 0) continue;
      float pz = ((float)d) / 5000.0f;
      x[n] = (c - cx) * pz / fx;
      y[n] = (r - cx) * pz / fx;
      z[n] = pz;
      float ox = 0.f, oy = 0.f, oz = 0.f;
      int interior = (r > 0 && r < rows - 1 && c > 0 && c < cols - 1);
      if (mode == 0) {
        if (interior && ORC_D(r, c + 1) && ORC_D(r, c - 1) && ORC_D(r + 1, c) && ORC_D(r - 1, c)) {
          float P[4][3];
          const int rr[4] = {r, r, r + 1, r - 1}, cc[4] = {c + 1, c - 1, c, c};
          for (int k = 0; k < 4; k++) {
            float qz = ((float)ORC_D(rr[k], cc[k])) / 5000.0f;
            P[k][0] = (cc[k] - cx) * qz / fx;
            P[k][1] = (rr[k] - cx) * qz / fx;
            P[k][2] = qz;
          }
          float ax = P[0][0] - P[1][0], ay = P[0][1] - P[1][1], az = P[0][2] - P[1][2];
          float bx = P[2][0] - P[3][0], by = P[2][1] - P[3][1], bz = P[2][2] - P[3][2];
          float m0 = ay * bz, m1 = az * by, m2 = az * bx, m3 = ax * bz, m4 = ax * by, m5 = ay * bx;
          float vx = m0 - m1, vy = m2 - m3, vz = m4 - m5;
          float s0 = vx * vx, s1 = vy * vy, s2 = vz * vz;
          float l2 = (s0 + s1) + s2;
          float l = sqrtf(l2);
          if (l > 0.f) {
            ox = vx / l;
            oy = vy / l;
            oz = vz / l;
          }
        }
      } else if (interior) {
        float dzdx = ((float)ORC_D(r + 1, c) - (float)ORC_D(r - 1, c)) / 2.0f; /* SLAM.cpp:421 */
        float dzdy = ((float)ORC_D(r, c + 1) - (float)ORC_D(r, c - 1)) / 2.0f; /* SLAM.cpp:422 */
        float dv[3] = {-dzdx, -dzdy, 1.0f};                                   /* SLAM.cpp:424 */
        double nv = sqrt((double)dv[0] * dv[0] + (double)dv[1] * dv[1] + (double)dv[2] * dv[2]);
        double sc = nv != 0.0 ? 1.0 / nv : 0.0; /* cv::normalize(Vec3f) */
        ox = (float)(dv[0] * sc);
        oy = (float)(dv[1] * sc);
        oz = (float)(dv[2] * sc);
      }
      nx[n] = ox;
      ny[n] = oy;
      nz[n] = oz;
      n++;
    }
#undef ORC_D
  return n;
}

/* normals rotate with the cloud: n' = fl32(R n) (double accumulation)        */
ORC_API void orc_rotate_normals(float *nx, float *ny, float *nz, int n, const float R[9]) {
  const float zero[3] = {0.f, 0.f, 0.f};
  orc_transform_points(nx, ny, nz, n, R, zero);
}

/* canonical sums of the linearised point-to-plane normal equations over the   */
/* accepted pairs (dist < max_dist and a non-zero target normal):              */
/*   J = [p x n ; n] (6), r = (p - q).n, sums[0..20] upper triangle of J J^T   */
/*   row-major, sums[21..26] = J r, sums[27] = dist.  float64, same tree as    */
/*   orc_sums_canonical.                                                        */
ORC_API int64_t orc_sums_p2l_canonical(const float *ax, const float *ay, const float *az, int nq,
                                       const float *tx, const float *ty, const float *tz,
                                       const float *nx, const float *ny, const float *nz,
                                       const int32_t *idx, const float *dist, float max_dist,
                                       double sums[ORC_NP2L]) {
  int B = (nq + ORC_RED_THREADS - 1) / ORC_RED_THREADS;
  if (B < 1) B = 1;
  if (B > ORC_RED_MAX_BLOCKS) B = ORC_RED_MAX_BLOCKS;
  const int P = B * ORC_RED_THREADS;
  double *acc = (double *)calloc((size_t)P * ORC_NP2L, sizeof(double));
  int64_t count = 0;
  for (int g = 0; g < P; g++) {
    double *v = acc + (size_t)g * ORC_NP2L;
    for (int i = g; i < nq; i += P) {
      if (!(dist[i] < max_dist)) continue;
      int j = idx[i];
      double n[3] = {nx[j], ny[j], nz[j]};
      if (n[0] == 0.0 && n[1] == 0.0 && n[2] == 0.0) continue;
      double p[3] = {ax[i], ay[i], az[i]}, q[3] = {tx[j], ty[j], tz[j]};
      double J[6];
      J[0] = p[1] * n[2] - p[2] * n[1];
      J[1] = p[2] * n[0] - p[0] * n[2];
      J[2] = p[0] * n[1] - p[1] * n[0];
      J[3] = n[0];
      J[4] = n[1];
      J[5] = n[2];
      double r = ((p[0] - q[0]) * n[0] + (p[1] - q[1]) * n[1]) + (p[2] - q[2]) * n[2];
      int k = 0;
      for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) v[k++] += J[a] * J[b];
      for (int a = 0; a < 6; a++) v[21 + a] += J[a] * r;
      v[27] += (double)dist[i];
      count++;
    }
  }
  orc_tree_finish(acc, B, ORC_NP2L, sums);
  free(acc);
  return count;
}

/* Solve A x = -b (6x6 SPD, Cholesky, float64), x = (alpha, t); R = exp([alpha]x)
 * (Rodrigues).  Returns 0, or -1 if A is not positive definite.              */
ORC_API int orc_solve_p2l(const double sums[ORC_NP2L], double R[9], double t[3]) {
  double A[6][6], L[6][6] = {{0}}, x[6], yv[6];
  int k = 0;
  for (int a = 0; a < 6; a++)
    for (int b = a; b < 6; b++) A[a][b] = A[b][a] = sums[k++];
  double dmax = 0;
  for (int a = 0; a < 6; a++)
    if (A[a][a] > dmax) dmax = A[a][a];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i][j];
      for (int m = 0; m < j; m++) s -= L[i][m] * L[j][m];
      if (i == j) {
        if (!(s > 1e-12 * dmax)) return -1;
        L[i][i] = sqrt(s);
      } else
        L[i][j] = s / L[j][j];
    }
  for (int i = 0; i < 6; i++) {
    double s = -sums[21 + i];
    for (int m = 0; m < i; m++) s -= L[i][m] * yv[m];
    yv[i] = s / L[i][i];
  }
  for (int i = 5; i >= 0; i--) {
    double s = yv[i];
    for (int m = i + 1; m < 6; m++) s -= L[m][i] * x[m];
    x[i] = s / L[i][i];
  }
  double a0 = x[0], a1 = x[1], a2 = x[2];
  double th2 = (a0 * a0 + a1 * a1) + a2 * a2, th = sqrt(th2);
  double A1, B1;
  if (th < 1e-9) {
    A1 = 1.0 - th2 / 6.0;
    B1 = 0.5 - th2 / 24.0;
  } else {
    A1 = sin(th) / th;
    B1 = (1.0 - cos(th)) / th2;
  }
  double K[9] = {0, -a2, a1, a2, 0, -a0, -a1, a0, 0}, K2[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) K2[3 * r + c] = (K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c]) + K[3 * r + 2] * K[6 + c];
  for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + (A1 * K[i] + B1 * K2[i]);
  t[0] = x[3];
  t[1] = x[4];
  t[2] = x[5];
  return 0;
}

/* ======================================================================== */
/* Full loop.  Frame-pair formulation of icp.cpp:98-268 with the live        */
/* keypoint association replaced by the full-cloud association the reference */
/* keeps commented at icp.cpp:253 (findGlobalNearestNeighborAssociations).   */
/* ======================================================================== */
typedef struct {
  int32_t max_iterations; /* SLAM.cpp:277: 16            */
  float threshold;        /* SLAM.cpp:277: 0.0001f       */
  float max_nn_dist;      /* icp.hpp:8: 0.75f            */
  int32_t min_pairs;      /* icp.cpp:163: 3              */
  int32_t solve;          /* 0 reference, 1 kabsch       */
  int32_t sum_order;      /* 0 sequential (reference order), 1 canonical tree */
  int32_t fixed_iterations; /* 1: ignore threshold (benchmark mode) */
  int32_t threads;        /* OpenMP threads for the NN sweep */
  float last_rotation[9];    /* icp.cpp:23,176 fallback motion */
  float last_translation[3]; /* icp.cpp:25,177 */
} orc_params;

typedef struct {
  int32_t n_pairs;
  float mse;      /* value tested by the while at icp.cpp:155 for THIS iteration */
  float M[9];     /* reference solve: float cross moment; kabsch: centred H as float */
  float R[9];     /* rotation found this iteration (before inversion) */
  float t[3];     /* reference: offset (icp.cpp:240); kabsch: translation */
} orc_iter_trace;

typedef struct {
  int32_t iterations;  /* loop bodies completed */
  int32_t status;      /* 0 ok, 1 fell back (<min_pairs), -1 empty target */
  int32_t final_pairs;
  float final_mse;
} orc_result;

static void nn_and_stats(float *sx, float *sy, float *sz, int ns, const float *tx,
                         const float *ty, const float *tz, int nt, const float *nx,
                         const float *ny, const float *nz, int32_t *idx,
                         float *dist, const orc_params *p, float *mse, int *npairs,
                         double sums[ORC_NP2L]) {
  orc_nn_bruteforce(sx, sy, sz, ns, tx, ty, tz, nt, idx, dist, p->threads);
  if (p->solve == 2) { /* point-to-plane: pairs also need a valid target normal */
    int64_t n = orc_sums_p2l_canonical(sx, sy, sz, ns, tx, ty, tz, nx, ny, nz, idx, dist,
                                       p->max_nn_dist, sums);
    *npairs = (int)n;
    if (n > 0) {
      float m = (float)(sums[27] / (double)n);
      *mse = (float)((double)m * (double)m);
    } else
      *mse = 0.f;
    return;
  }
  if (p->sum_order == 0) {
    *mse = orc_mse_seq(dist, ns, p->max_nn_dist);
    int n = 0;
    for (int i = 0; i < ns; i++) n += (dist[i] < p->max_nn_dist);
    *npairs = n;
  } else {
    int64_t n = orc_sums_canonical(sx, sy, sz, ns, tx, ty, tz, idx, dist,
                                   p->max_nn_dist, sums);
    *npairs = (int)n;
    if (n > 0) {
      float m = (float)(sums[12] / (double)n);
      *mse = (float)((double)m * (double)m);
    } else
      *mse = 0.f;
  }
}

/* source arrays are transformed IN PLACE (like dataCloud in the reference). */
/* nx/ny/nz: target normals (solve == 2 only, may be NULL otherwise).          */
ORC_API int orc_align2(float *sx, float *sy, float *sz, int ns, const float *tx,
                       const float *ty, const float *tz, int nt, const float *nx,
                       const float *ny, const float *nz,
                       const orc_params *p, float T[16], int32_t *idx, float *dist,
                       orc_iter_trace *trace, orc_result *res) {
  static const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  float Trot[9];
  float offset[3] = {0, 0, 0};
  double Tk[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}; /* kabsch accumulated [R|t] */
  memcpy(Trot, I3, sizeof(I3));
  for (int k = 0; k < 16; k++) T[k] = (k % 5 == 0) ? 1.f : 0.f;
  res->iterations = 0;
  res->status = 0;
  res->final_pairs = 0;
  res->final_mse = 0.f;
  if (nt <= 0) {
    res->status = -1;
    return -1;
  }
  float mse;
  int npairs;
  double sums[ORC_NP2L];
  if (p->solve == 2 && (!nx || !ny || !nz)) {
    res->status = -1;
    return -1;
  }
  nn_and_stats(sx, sy, sz, ns, tx, ty, tz, nt, nx, ny, nz, idx, dist, p, &mse, &npairs, sums); /* icp.cpp:98 */
  int i = 0;
  while ((p->fixed_iterations || mse > p->threshold) && i < p->max_iterations) { /* icp.cpp:155 */
    if (npairs < p->min_pairs) { /* icp.cpp:163-182 */
      float lt[3] = {p->last_translation[0], p->last_translation[1], p->last_translation[2]};
      orc_transform_points(sx, sy, sz, ns, p->last_rotation, lt);
      offset[0] = -lt[0];
      offset[1] = -lt[1];
      offset[2] = -lt[2];
      res->status = 1;
      break;
    }
    orc_iter_trace tr;
    memset(&tr, 0, sizeof(tr));
    tr.n_pairs = npairs;
    tr.mse = mse;
    if (p->solve == 0) {
      float M[9], R[9], Rinv[9], neg[3];
      if (p->sum_order == 0) {
        orc_cross_moment_seq(sx, sy, sz, ns, tx, ty, tz, idx, dist, p->max_nn_dist, M); /* icp.cpp:212 */
        orc_calculate_offset_seq(sx, sy, sz, ns, tx, ty, tz, idx, dist, p->max_nn_dist,
                                 offset); /* icp.cpp:240: pre-rotation copies */
      } else {
        for (int k = 0; k < 9; k++) M[k] = (float)sums[k];
        for (int k = 0; k < 3; k++) offset[k] = (float)(sums[9 + k] / (double)npairs);
      }
      orc_solve_reference(M, R); /* icp.cpp:215-223 */
      if (i == 0)
        memcpy(Trot, R, sizeof(Trot)); /* icp.cpp:227-229 */
      else
        mat3_mul_f(R, Trot, Trot); /* icp.cpp:231 */
      orc_inv3_f(R, Rinv);         /* icp.cpp:235 */
      neg[0] = -offset[0];
      neg[1] = -offset[1];
      neg[2] = -offset[2];
      orc_transform_points(sx, sy, sz, ns, Rinv, neg); /* icp.cpp:236,245 */
      memcpy(tr.M, M, sizeof(M));
      memcpy(tr.R, R, sizeof(R));
      memcpy(tr.t, offset, sizeof(offset));
    } else if (p->solve == 2) {
      double Rd[9], td[3];
      if (orc_solve_p2l(sums, Rd, td) != 0) { /* degenerate normal equations */
        res->status = 2;
        break;
      }
      float Rf[9], tf[3];
      for (int k = 0; k < 9; k++) Rf[k] = (float)Rd[k];
      for (int k = 0; k < 3; k++) tf[k] = (float)td[k];
      orc_transform_points(sx, sy, sz, ns, Rf, tf);
      double Tn[12];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) {
          double sacc = 0;
          for (int k = 0; k < 3; k++) sacc += (double)Rf[3 * r + k] * Tk[4 * k + c];
          Tn[4 * r + c] = sacc + (c == 3 ? (double)tf[r] : 0.0);
        }
      memcpy(Tk, Tn, sizeof(Tk));
      memcpy(tr.R, Rf, sizeof(Rf));
      memcpy(tr.t, tf, sizeof(tf));
    } else {
      double sa[3], sb[3], sab[9], Rd[9], td[3];
      if (p->sum_order == 0) {
        for (int k = 0; k < 3; k++) sa[k] = sb[k] = 0;
        for (int k = 0; k < 9; k++) sab[k] = 0;
        for (int q = 0; q < ns; q++) {
          if (!(dist[q] < p->max_nn_dist)) continue;
          int j = idx[q];
          double a[3] = {sx[q], sy[q], sz[q]}, b[3] = {tx[j], ty[j], tz[j]};
          for (int r = 0; r < 3; r++) {
            sa[r] += a[r];
            sb[r] += b[r];
            for (int c = 0; c < 3; c++) sab[3 * r + c] += a[r] * b[c];
          }
        }
      } else {
        for (int k = 0; k < 3; k++) {
          sa[k] = sums[13 + k];
          sb[k] = sums[16 + k];
        }
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) sab[3 * r + c] = sums[3 * c + r]; /* M^T */
      }
      orc_solve_kabsch_from_sums(npairs, sa, sb, sab, Rd, td);
      float Rf[9], tf[3];
      for (int k = 0; k < 9; k++) Rf[k] = (float)Rd[k];
      for (int k = 0; k < 3; k++) tf[k] = (float)td[k];
      orc_transform_points(sx, sy, sz, ns, Rf, tf);
      /* compose with the float-rounded step actually applied */
      double Tn[12];
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 4; c++) {
          double s = 0;
          for (int k = 0; k < 3; k++) s += (double)Rf[3 * r + k] * Tk[4 * k + c];
          Tn[4 * r + c] = s + (c == 3 ? (double)tf[r] : 0.0);
        }
      }
      memcpy(Tk, Tn, sizeof(Tk));
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
          tr.M[3 * r + c] = (float)(sab[3 * r + c] - sa[r] * sb[c] / (double)npairs);
      memcpy(tr.R, Rf, sizeof(Rf));
      memcpy(tr.t, tf, sizeof(tf));
    }
    if (trace) trace[i] = tr;
    nn_and_stats(sx, sy, sz, ns, tx, ty, tz, nt, nx, ny, nz, idx, dist, p, &mse, &npairs, sums); /* icp.cpp:255 */
    i++; /* icp.cpp:257 */
  }
  res->iterations = i;
  res->final_pairs = npairs;
  res->final_mse = mse;
  if (p->solve == 0) {
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) T[4 * r + c] = Trot[3 * r + c];
      T[4 * r + 3] = offset[r]; /* icp.cpp:266-268: LAST offset only */
    }
  } else {
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) T[4 * r + c] = (float)Tk[4 * r + c];
  }
  return res->status;
}

ORC_API int orc_align(float *sx, float *sy, float *sz, int ns, const float *tx,
                      const float *ty, const float *tz, int nt,
                      const orc_params *p, float T[16], int32_t *idx, float *dist,
                      orc_iter_trace *trace, orc_result *res) {
  return orc_align2(sx, sy, sz, ns, tx, ty, tz, nt, NULL, NULL, NULL, p, T, idx, dist, trace, res);
}

ORC_API int orc_sizeof_params(void) { return (int)sizeof(orc_params); }
ORC_API int orc_sizeof_trace(void) { return (int)sizeof(orc_iter_trace); }
ORC_API int orc_sizeof_result(void) { return (int)sizeof(orc_result); }
ORC_API int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
