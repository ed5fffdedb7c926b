// kernels_sort.hip -- spatial (Morton) ordering used by the pruned NN scan.
//
// the reference's own planned replacement for the brute-force
// scan is a voxel lookup (icp.cpp:347-486, map.hpp:9-17).  Here the same idea serves
// an EXACT search: targets are re-ordered along a Morton curve so that every run of
// 128 consecutive points is a compact 3-D cluster with a tight bounding box; the NN
// kernel skips boxes that are out of reach and still returns the brute-force result
// (original indices are carried along for the lowest-index tie rule).
//
// The sort is the same hand-written counting sort the grid scan uses (kernels_grid.hip: slot inside
// the bin by atomics, device-sized exclusive scan of the counts, scatter), over the cells of a
// 2^b x 2^b x 2^b lattice of the cloud's bounding box taken in Morton order, b = 7 or 6 bits per
// axis (whatever the context's count table holds), non-finite points in a last bin of their own.
// The order INSIDE a cell is whatever the atomics make it: only the tightness of the boxes, never
// a result, depends on it.  (Rounds 1-2 called rocPRIM's radix sort on 30-bit codes here: the one
// library call of the code base, in a set-up step of a non-default mode; gone in round 3.)
#include <cstring>

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

// key = Morton code (bits per axis: `bits`) of the point's lattice cell inside `bounds`; non-finite
// points (padding) get the bin after the last cell, 8^bits, and sort last.  slot = arrival number
// inside the bin.
__global__ void morton_slot_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                   int n, const float* __restrict__ bounds, int bits, unsigned* __restrict__ keys,
                                   int* __restrict__ slot, int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p[3] = {x[i], y[i], z[i]};
  const float top = (float)((1 << bits) - 1);
  unsigned q[3];
  bool finite = true;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float lo = bounds[c], hi = bounds[3 + c];
    const float ext = hi - lo;
    const float s = ext > 0.f ? top / ext : 0.f;
    float f = (p[c] - lo) * s;
    finite = finite && (p[c] - p[c] == 0.f);
    f = __builtin_fminf(__builtin_fmaxf(f, 0.f), top);  // NaN -> 0
    q[c] = (unsigned)f;
  }
  const unsigned k = finite ? (spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2)) : (1u << (3 * bits));
  keys[i] = k;
  slot[i] = atomicAdd(&count[k], 1);
}

__global__ void morton_scatter_kernel(const unsigned* __restrict__ keys, const int* __restrict__ slot,
                                      const int* __restrict__ start, int n, unsigned* __restrict__ keys_out,
                                      int* __restrict__ perm_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned k = keys[i];
  const int pos = start[k] + slot[i];
  keys_out[pos] = k;
  perm_out[pos] = i;
}

// the table size launch_grid_scan reads from a GridInfo on the device: 8^bits cells + the bin of the non-finite points
__global__ void morton_table_kernel(GridInfo* __restrict__ g, int bits) {
  if (threadIdx.x == 0) {
    g->ncells = (1 << (3 * bits)) + 1;
    g->ncells_q = g->ncells;
  }
}

void launch_bounds(const float* tbox, int tbox_stride, int ntiles, float* bounds, hipStream_t s) {
  hipLaunchKernelGGL(bounds_kernel, dim3(1), dim3(256), 0, s, tbox, tbox_stride, ntiles, bounds);
}

// keys_out / perm_out: the points by Morton cell; keys: the unsorted keys (kept: the queries' keys seed the first
// sweep); count: the context's zeroed count table (handed back zeroed), start / bsum: scan outputs / scratch
void launch_morton_order(const float* x, const float* y, const float* z, int n, const float* bounds, int bits,
                         unsigned* keys, int* slot, int* count, int* start, int* bsum, GridInfo* table,
                         unsigned* keys_out, int* perm_out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(morton_table_kernel, dim3(1), dim3(64), 0, s, table, bits);
  hipLaunchKernelGGL(morton_slot_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, z, n, bounds, bits, keys, slot, count);
  launch_grid_scan(count, start, bsum, table, 0, s);
  hipLaunchKernelGGL(morton_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, keys, slot, start, n, keys_out, perm_out);
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
