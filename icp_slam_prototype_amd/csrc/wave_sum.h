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

// v[s] += partner's v[s] for all six stages, in the canonical order
template <int NS>
__device__ __forceinline__ void wave_butterfly(double (&v)[NS], int& cnt) {
#define ICPK_STAGE_SWAP(M)                                                 \
  _Pragma("unroll") for (int s = 0; s < NS; ++s) v[s] = add_xor_swap<M>(v[s]); \
  cnt = add_xor_swap<M>(cnt);
#define ICPK_STAGE(M)                                                      \
  _Pragma("unroll") for (int s = 0; s < NS; ++s) v[s] += xor_lane<M>(v[s]); \
  cnt += xor_lane<M>(cnt);
  ICPK_STAGE_SWAP(32)
  ICPK_STAGE_SWAP(16)
  ICPK_STAGE(8)
  ICPK_STAGE(4)
  ICPK_STAGE(2)
  ICPK_STAGE(1)
#undef ICPK_STAGE
#undef ICPK_STAGE_SWAP
}

}  // namespace icpk
