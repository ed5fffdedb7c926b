// wave_sum.h -- the partner exchange of the canonical wave64 butterfly (include/icpk.h,
// ICPK_RED_*): lane i adds the value of lane i ^ m, m = 32, 16, 8, 4, 2, 1.  The tree is fixed
// by the ABI; HOW a lane gets its partner's value is free.  __shfl_xor compiles to
// ds_bpermute_b32 (two per double, through the LDS crossbar: 2.4 us for the 19 sums of K2);
// for m <= 8 the partner is in the same row of 16 lanes and a DPP move does it in the VALU;
// m = 16 and 32 use gfx950's v_permlane16_swap / v_permlane32_swap.
#pragma once
#include <hip/hip_runtime.h>

namespace icpk {

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

// value of lane (i ^ M) for every lane i of a full wave64
template <int M>
__device__ __forceinline__ int xor_lane(int v) {
  if constexpr (M == 1) return dpp_mov<0xB1>(v);                       // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return dpp_mov<0x4E>(v);                  // quad_perm [2,3,0,1]
  else if constexpr (M == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));  // row_half_mirror (i^7), then quad_perm [3,2,1,0] (i^3)
  else if constexpr (M == 8) return dpp_mov<0x128>(v);                 // row_ror:8
  else return __shfl_xor(v, M, 64);
}

template <int M>
__device__ __forceinline__ double xor_lane(double v) {
  const int lo = xor_lane<M>(__double2loint(v));
  const int hi = xor_lane<M>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// x + (value of lane i ^ M) for M = 16, 32 through gfx950's v_permlane{16,32}_swap: with both
// operands = x the instruction leaves A = [even rows | even rows], B = [odd rows | odd rows]
// (rows of 16 resp. 32 lanes), so A + B is self + partner on the even rows and partner + self
// on the odd ones -- the same IEEE sum either way (addition commutes).
template <int M>
__device__ __forceinline__ double add_xor_swap(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  if constexpr (M == 32) {
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  } else {
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  }
}
template <int M>
__device__ __forceinline__ int add_xor_swap(int v) {
  if constexpr (M == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)(r[0] + r[1]);
  } else {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)(r[0] + r[1]);
  }
}

// ---- the same tree as a reduce-scatter --------------------------------------------------------
// Every lane of the butterfly above ends with every total: 6 stages x NS sums.  Nobody needs 64 copies.
// With TWO different operands the swap instructions pair two sums per exchange: permlane32_swap(a, b)
// leaves [a_lo | b_lo] and [a_hi | b_hi], whose sum is stage 32 of `a` on the low 32 lanes and of `b` on the
// high ones (self + partner or partner + self: the same IEEE sum); permlane16_swap does the same with the
// even / odd rows of 16 lanes.  After the two cross-row stages a lane carries a quarter of the sums, and
// only those go through the four in-row DPP stages: 7 + 4 + 4 x 4 double additions instead of 6 x 13 for
// the reference flavour's 13 sums (10 + 5 + 4 x 5 instead of 114 for all 19), no operand copies.  Same
// pairs, same order, same bits.
// Layout afterwards: with H1 = ceil(NS / 2), H2 = ceil(H1 / 2), lane l of row r = l / 16 holds in u[k] the
// wave's total of sum  k + (r & 1) H2 + (r >> 1) H1  (when k + (r & 1) H2 < H1 and the index is < NS).
template <int M>
__device__ __forceinline__ double swap_add2(double a, double b) {
  const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
  const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
  if constexpr (M == 32) {
    const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  } else {
    const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  }
}

template <int NS>
struct WaveScatter {
  static constexpr int H1 = (NS + 1) / 2, H2 = (H1 + 1) / 2;
  // sum index carried in u[k] by the lanes of row r, or -1
  static __device__ __forceinline__ int index(int r, int k) {
    const int h = k + ((r & 1) ? H2 : 0);
    const int idx = h + ((r & 2) ? H1 : 0);
    return (h < H1 && idx < NS) ? idx : -1;
  }
};

template <int NS>
__device__ __forceinline__ void wave_reduce_scatter(const double (&v)[NS], double (&u)[WaveScatter<NS>::H2], int& cnt) {
  constexpr int H1 = WaveScatter<NS>::H1, H2 = WaveScatter<NS>::H2;
  double s1[H1];
#pragma unroll
  for (int j = 0; j < H1; ++j) s1[j] = swap_add2<32>(v[j], j + H1 < NS ? v[j + H1] : 0.0);
#pragma unroll
  for (int k = 0; k < H2; ++k) u[k] = swap_add2<16>(s1[k], k + H2 < H1 ? s1[k + H2] : 0.0);
  cnt = add_xor_swap<32>(cnt);
  cnt = add_xor_swap<16>(cnt);
#define ICPK_STAGE(M)                                                       \
  _Pragma("unroll") for (int k = 0; k < H2; ++k) u[k] += xor_lane<M>(u[k]); \
  cnt += xor_lane<M>(cnt);
  ICPK_STAGE(8)
  ICPK_STAGE(4)
  ICPK_STAGE(2)
  ICPK_STAGE(1)
#undef ICPK_STAGE
}

// the wave's NS totals into ws[0..NS) (one lane per row of 16 stores the sums its row carries)
template <int NS>
__device__ __forceinline__ void wave_scatter_store(const double (&u)[WaveScatter<NS>::H2], int lane, double* ws) {
  if ((lane & 15) != 0) return;
  const int r = lane >> 4;
#pragma unroll
  for (int k = 0; k < WaveScatter<NS>::H2; ++k) {
    const int idx = WaveScatter<NS>::index(r, k);
    if (idx >= 0) ws[idx] = u[k];
  }
}

}  // namespace icpk
