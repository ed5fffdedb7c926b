// kernels_backproject.hip -- K4: depth image -> compacted xyz-SoA cloud.
//
// pointcloud.cpp:19-58: row-major scan, zero depth skipped,
//   p_z = (float)d / 5000.0f; p_x = (x - CX) * p_z / FX; p_y = (y - CX) * p_z / FX
// (CX and FX are used for the y axis too, pointcloud.cpp:39).  The unseeded
// rand()%40 subsample of pointcloud.cpp:28 is not reproduced.  Output order is
// the row-major order of the valid pixels (order-preserving compaction):
//   pass 1: per-1024-pixel block count (wave64 ballot + popcount)
//   pass 2: exclusive scan of the block counts (one workgroup)
//   pass 3: in-block rank from ballots, scatter.
#include "icpk_internal.h"

namespace icpk {

constexpr int BP_THREADS = 256;
constexpr int BP_PER_THREAD = 4;
constexpr int BP_BLOCK = BP_THREADS * BP_PER_THREAD;  // 1024 pixels per workgroup

__global__ __launch_bounds__(BP_THREADS) void bp_count_kernel(const uint16_t* __restrict__ depth, int npix,
                                                              int* __restrict__ block_counts) {
  const int base = blockIdx.x * BP_BLOCK;
  int c = 0;
#pragma unroll
  for (int k = 0; k < BP_PER_THREAD; ++k) {
    const int p = base + k * BP_THREADS + threadIdx.x;
    const bool valid = p < npix && depth[p] != 0;
    c += __popcll(__ballot(valid));
  }
  __shared__ int wc[BP_THREADS / 64];
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// exclusive scan in place; block_counts[nblocks] receives the total
__global__ void bp_scan_kernel(int* __restrict__ block_counts, int nblocks, int* __restrict__ n_out) {
  __shared__ int carry;
  __shared__ int wsum[4];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? block_counts[i] : 0;
    int incl = v;  // wave64 inclusive scan
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d, 64);
      if ((threadIdx.x & 63) >= d) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
    const int c = carry;
    if (i < nblocks) block_counts[i] = c + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = c + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    block_counts[nblocks] = carry;
    *n_out = carry;
  }
}

__global__ __launch_bounds__(BP_THREADS) void bp_scatter_kernel(const uint16_t* __restrict__ depth, int npix, int cols,
                                                                float fx, float cx, float ox, float oy, float oz,
                                                                const int* __restrict__ block_offsets,
                                                                float* __restrict__ x, float* __restrict__ y,
                                                                float* __restrict__ z) {
  __shared__ int wcount[BP_PER_THREAD][BP_THREADS / 64];
  const int base = blockIdx.x * BP_BLOCK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long mask[BP_PER_THREAD];
  uint16_t dv[BP_PER_THREAD];
#pragma unroll
  for (int k = 0; k < BP_PER_THREAD; ++k) {
    const int p = base + k * BP_THREADS + threadIdx.x;
    dv[k] = p < npix ? depth[p] : (uint16_t)0;
    mask[k] = __ballot(dv[k] != 0);
    if (lane == 0) wcount[k][wave] = __popcll(mask[k]);
  }
  __syncthreads();
  int off = block_offsets[blockIdx.x];
#pragma unroll
  for (int k = 0; k < BP_PER_THREAD; ++k) {
    // pixels of sub-row k, waves in order: rank = earlier sub-rows + earlier waves + earlier lanes
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wcount[k][w];
    const int rank = off + before + __popcll(mask[k] & ((1ull << lane) - 1ull));
    if (dv[k] != 0) {
      const int p = base + k * BP_THREADS + threadIdx.x;
      const int r = p / cols, c = p - r * cols;
      const float pz = ((float)dv[k]) / 5000.0f;       // pointcloud.cpp:37
      const float px = ((float)c - cx) * pz / fx;       // pointcloud.cpp:38
      const float py = ((float)r - cx) * pz / fx;       // pointcloud.cpp:39 (CX, FX)
      x[rank] = px + ox;                                // pointcloud.cpp:349-359 translate
      y[rank] = py + oy;
      z[rank] = pz + oz;
    }
    off += wcount[k][0] + wcount[k][1] + wcount[k][2] + wcount[k][3];
  }
}

void launch_backproject(const uint16_t* depth, int rows, int cols, float fx, float cx, float ox, float oy, float oz,
                        float* x, float* y, float* z, int* block_counts, int* n_out, hipStream_t s) {
  const int npix = rows * cols;
  const int nblocks = (npix + BP_BLOCK - 1) / BP_BLOCK;
  hipLaunchKernelGGL(bp_count_kernel, dim3(nblocks), dim3(BP_THREADS), 0, s, depth, npix, block_counts);
  hipLaunchKernelGGL(bp_scan_kernel, dim3(1), dim3(256), 0, s, block_counts, nblocks, n_out);
  hipLaunchKernelGGL(bp_scatter_kernel, dim3(nblocks), dim3(BP_THREADS), 0, s, depth, npix, cols, fx, cx, ox, oy, oz,
                     block_counts, x, y, z);
}

}  // namespace icpk
