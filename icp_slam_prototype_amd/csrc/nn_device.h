// nn_device.h -- device functions shared by the NN kernels (kernels_nn.hip,
// kernels_nn_pruned.hip): the exact pair distance of icp.cpp:606-620, the fp32 filter
// threshold and the per-group fast / slow paths of the filtered scan.
#pragma once
#include "icpk_internal.h"

namespace icpk {

// ---- exact pair distance ----------------------------------------------------
// The products are exact in double (24x24 bits), so fma(y,y,x*x) rounds exactly
// like the reference's (x*x + y*y); same for the second addition.
__device__ __forceinline__ float pair_dist(float qx, float qy, float qz, float tx, float ty, float tz) {
  const float dx = qx - tx;
  const float dy = qy - ty;
  const float dz = qz - tz;
  const double ddx = (double)dx, ddy = (double)dy, ddz = (double)dz;
  const double s = __builtin_fma(ddz, ddz, __builtin_fma(ddy, ddy, ddx * ddx));
  // correctly rounded float sqrt (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt;
  // checked on the device by tests/test_gpu_parity.py::test_pair_distance_bits)
  return __builtin_sqrtf((float)s);
}

constexpr int NNF_G = 8;  // targets per group: one s_load_dwordx8 per plane

__device__ __forceinline__ float filt_threshold(float d) {
  const float t = __builtin_fmaf(d * d, 1.0f + 0x1p-20f, 0x1p-120f);
  return (t == t) ? t : __builtin_inff();  // NaN best (garbage input): pass everything
}

// fast path for one group of NNF_G wave-uniform targets: the fp32 estimates e[k][u]
// of every (target k, query u of this lane) and their minimum per query
// (v_min3_f32 folds two estimates per instruction)
template <int Q>
__device__ __forceinline__ void group_estimates(const float (&qx)[Q], const float (&qy)[Q], const float (&qz)[Q],
                                                const float (&X)[NNF_G], const float (&Y)[NNF_G],
                                                const float (&Z)[NNF_G], float (&e)[NNF_G][Q], float (&m)[Q]) {
#pragma unroll
  for (int u = 0; u < Q; ++u) m[u] = __builtin_inff();
#pragma unroll
  for (int k = 0; k < NNF_G; k += 2) {
#pragma unroll
    for (int u = 0; u < Q; ++u) {
      const float dx0 = qx[u] - X[k], dy0 = qy[u] - Y[k], dz0 = qz[u] - Z[k];
      const float dx1 = qx[u] - X[k + 1], dy1 = qy[u] - Y[k + 1], dz1 = qz[u] - Z[k + 1];
      e[k][u] = __builtin_fmaf(dz0, dz0, __builtin_fmaf(dy0, dy0, dx0 * dx0));
      e[k + 1][u] = __builtin_fmaf(dz1, dz1, __builtin_fmaf(dy1, dy1, dx1 * dx1));
      m[u] = __builtin_fminf(m[u], __builtin_fminf(e[k][u], e[k + 1][u]));
    }
  }
}

// slow path, entered when some lane's minimum passed its threshold: walk the group
// and re-evaluate exactly only those targets some lane still passes (wave-uniform
// branch per target); lexicographic (d, j) merge; thresholds tighten as we go
// PERM: the scanned planes are a permutation of the caller's cloud; tperm[j] is the
// original index, which is what the lowest-index tie rule and the result refer to.
template <int Q, bool PERM>
__device__ __forceinline__ void group_exact(const float (&qx)[Q], const float (&qy)[Q], const float (&qz)[Q],
                                            const float (&X)[NNF_G], const float (&Y)[NNF_G], const float (&Z)[NNF_G],
                                            const float (&e)[NNF_G][Q], int j, const int* __restrict__ tperm,
                                            float (&bd)[Q], int (&bj)[Q], float (&T)[Q]) {
#pragma unroll
  for (int k = 0; k < NNF_G; ++k) {
    bool hit = false;
#pragma unroll
    for (int u = 0; u < Q; ++u) hit |= (e[k][u] <= T[u]);
    if (__builtin_amdgcn_ballot_w64(hit) != 0) {
#pragma unroll
      for (int u = 0; u < Q; ++u) {
        const float d = pair_dist(qx[u], qy[u], qz[u], X[k], Y[k], Z[k]);
        const int jj = PERM ? tperm[j + k] : j + k;  // uniform address: scalar load
        const bool up = (d < bd[u]) | ((d == bd[u]) & (jj < bj[u]));
        bd[u] = up ? d : bd[u];
        bj[u] = up ? jj : bj[u];
        T[u] = up ? filt_threshold(d) : T[u];
      }
    }
  }
}

// Same slow path when the targets differ per lane (the sliced pruned scan): X/Y/Z are
// per-lane values, j the per-lane position of the group in the scanned planes.
template <int Q>
__device__ __forceinline__ void group_exact_lanes(const float (&qx)[Q], const float (&qy)[Q], const float (&qz)[Q],
                                                  const float (&X)[NNF_G], const float (&Y)[NNF_G],
                                                  const float (&Z)[NNF_G], const float (&e)[NNF_G][Q], int j,
                                                  const int* __restrict__ tperm, float (&bd)[Q], int (&bj)[Q],
                                                  float (&T)[Q]) {
#pragma unroll
  for (int k = 0; k < NNF_G; ++k) {
    bool hit = false;
#pragma unroll
    for (int u = 0; u < Q; ++u) hit |= (e[k][u] <= T[u]);
    if (__builtin_amdgcn_ballot_w64(hit) != 0) {
      const int jj = tperm[j + k];  // per-lane gather (rare path)
#pragma unroll
      for (int u = 0; u < Q; ++u) {
        const float d = pair_dist(qx[u], qy[u], qz[u], X[k], Y[k], Z[k]);
        const bool up = (d < bd[u]) | ((d == bd[u]) & (jj < bj[u]));
        bd[u] = up ? d : bd[u];
        bj[u] = up ? jj : bj[u];
        T[u] = up ? filt_threshold(d) : T[u];
      }
    }
  }
}

// Lower bound of the squared distance from the lane's queries to an axis-aligned
// box; true if some query of this lane may still find a passing target inside.
// lb (fp32) <= s_j (1 + 6*2^-24) for every target j in the box, and a target passes
// the filter only if e_j <= T with e_j >= s_j (1 - 3*2^-24), so lb <= T (1 + 2^-19)
// is a safe (never wrongly skipping) test.  Empty boxes (lo = +inf, hi = -inf) never pass.
template <int Q>
__device__ __forceinline__ bool box_may_hit(const float (&qx)[Q], const float (&qy)[Q], const float (&qz)[Q], float lox,
                                            float loy, float loz, float hix, float hiy, float hiz,
                                            const float (&T)[Q]) {
  bool hit = false;
#pragma unroll
  for (int u = 0; u < Q; ++u) {
    const float dx = __builtin_fmaxf(__builtin_fmaxf(lox - qx[u], qx[u] - hix), 0.f);
    const float dy = __builtin_fmaxf(__builtin_fmaxf(loy - qy[u], qy[u] - hiy), 0.f);
    const float dz = __builtin_fmaxf(__builtin_fmaxf(loz - qz[u], qz[u] - hiz), 0.f);
    const float lb = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    hit |= (lb <= T[u] * (1.0f + 0x1p-19f));
  }
  return hit;
}

}  // namespace icpk
