// kernels_sort.hip -- spatial (Morton) ordering used by the pruned NN scan.
//
// the reference's own planned replacement for the brute-force
// scan is a voxel lookup (icp.cpp:347-486, map.hpp:9-17).  Here the same idea serves
// an EXACT search: targets are re-ordered along a 30-bit Morton curve so that every
// run of 128 consecutive points is a compact 3-D cluster with a tight bounding box;
// the NN kernel skips boxes that are out of reach and still returns the brute-force
// result (original indices are carried along for the lowest-index tie rule).
//
// The sort itself is a set-up step (once per target cloud / per alignment), done with
// rocPRIM's device radix sort; everything on the per-iteration path is hand-written.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "icpk_internal.h"

namespace icpk {

// cloud bounds from the per-tile boxes (lo x,y,z then hi x,y,z)
__global__ __launch_bounds__(256) void bounds_kernel(const float* __restrict__ tbox, int tbox_stride, int ntiles,
                                                     float* __restrict__ bounds) {
  __shared__ float red[6][256];
  const int tid = threadIdx.x;
  for (int c = 0; c < 6; ++c) {
    float v = c < 3 ? __builtin_inff() : -__builtin_inff();
    for (int t = tid; t < ntiles; t += 256) {
      const float b = tbox[c * tbox_stride + t];
      v = c < 3 ? __builtin_fminf(v, b) : __builtin_fmaxf(v, b);
    }
    red[c][tid] = v;
  }
  __syncthreads();
  if (tid < 6) {
    float v = red[tid][0];
    for (int k = 1; k < 256; ++k) v = tid < 3 ? __builtin_fminf(v, red[tid][k]) : __builtin_fmaxf(v, red[tid][k]);
    bounds[tid] = v;
  }
}

__device__ __forceinline__ unsigned spread10(unsigned v) {  // 10 bits -> every third bit
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// key = 30-bit Morton code of the point inside `bounds`; non-finite points (padding)
// get 0xffffffff and sort last.  vals = identity.
__global__ void morton_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                              int n, const float* __restrict__ bounds, unsigned* __restrict__ keys,
                              int* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p[3] = {x[i], y[i], z[i]};
  unsigned q[3];
  bool finite = true;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float lo = bounds[c], hi = bounds[3 + c];
    const float ext = hi - lo;
    const float s = ext > 0.f ? 1023.0f / ext : 0.f;
    float f = (p[c] - lo) * s;
    finite = finite && (p[c] - p[c] == 0.f);
    f = __builtin_fminf(__builtin_fmaxf(f, 0.f), 1023.f);  // NaN -> 0
    q[c] = (unsigned)f;
  }
  keys[i] = finite ? (spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2)) : 0xffffffffu;
  vals[i] = i;
}

void launch_bounds(const float* tbox, int tbox_stride, int ntiles, float* bounds, hipStream_t s) {
  hipLaunchKernelGGL(bounds_kernel, dim3(1), dim3(256), 0, s, tbox, tbox_stride, ntiles, bounds);
}

void launch_morton(const float* x, const float* y, const float* z, int n, const float* bounds, unsigned* keys, int* vals,
                   hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(morton_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, z, n, bounds, keys, vals);
}

size_t sort_temp_bytes(int n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned*)nullptr, (unsigned*)nullptr, (int*)nullptr, (int*)nullptr,
                                  (size_t)n, 0, 32, (hipStream_t) nullptr);
  return bytes;
}

// stable LSD radix sort: equal Morton codes keep their original relative order
int launch_sort_pairs(void* temp, size_t temp_bytes, const unsigned* keys_in, unsigned* keys_out, const int* vals_in,
                      int* vals_out, int n, hipStream_t s) {
  return (int)rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 32, s);
}



// out[k] = in[perm[k]] for k < n, pad beyond (planes are NN_TILE-padded)
__global__ void gather_planes_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                     const float* __restrict__ z, const int* __restrict__ perm, int n, int n_pad,
                                     float pad, float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                                     int* __restrict__ perm_pad) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_pad) return;
  if (k < n) {
    const int j = perm[k];
    ox[k] = x[j];
    oy[k] = y[j];
    oz[k] = z[j];
  } else {
    ox[k] = pad;
    oy[k] = pad;
    oz[k] = pad;
    perm_pad[k] = 0x7fffffff;
  }
}

void launch_gather_planes(const float* x, const float* y, const float* z, const int* perm, int n, int n_pad, float pad,
                          float* ox, float* oy, float* oz, int* perm_pad, hipStream_t s) {
  hipLaunchKernelGGL(gather_planes_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, s, x, y, z, perm, n, n_pad, pad, ox,
                     oy, oz, perm_pad);
}

}  // namespace icpk
