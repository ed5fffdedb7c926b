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

// Surface normal of pixel (r, c) -- point-to-plane extension (not in the reference,
// TODO:9).  mode 0: normalised cross product of the central differences of the
// back-projected 4-neighbours (all must be valid), float arithmetic in a fixed order;
// mode 1: SLAM.cpp:421-425 literally (raw-depth central differences along rows/cols,
// d = (-dzdx, -dzdy, 1), cv::normalize with a double norm), interior pixels only.
__device__ __forceinline__ void pixel_normal(const uint16_t* __restrict__ depth, int rows, int cols, int r, int c,
                                             float fx, float cx, int mode, float& nx, float& ny, float& nz) {
  nx = ny = nz = 0.f;
  if (!(r > 0 && r < rows - 1 && c > 0 && c < cols - 1)) return;
  const uint16_t dE = depth[r * cols + c + 1], dW = depth[r * cols + c - 1];
  const uint16_t dS = depth[(r + 1) * cols + c], dN = depth[(r - 1) * cols + c];
  if (mode == 0) {
    if (!(dE && dW && dS && dN)) return;
    const float zE = ((float)dE) / 5000.0f, zW = ((float)dW) / 5000.0f;
    const float zS = ((float)dS) / 5000.0f, zN = ((float)dN) / 5000.0f;
    const float xE = ((float)(c + 1) - cx) * zE / fx, yE = ((float)r - cx) * zE / fx;
    const float xW = ((float)(c - 1) - cx) * zW / fx, yW = ((float)r - cx) * zW / fx;
    const float xS = ((float)c - cx) * zS / fx, yS = ((float)(r + 1) - cx) * zS / fx;
    const float xN = ((float)c - cx) * zN / fx, yN = ((float)(r - 1) - cx) * zN / fx;
    const float ax = xE - xW, ay = yE - yW, az = zE - zW;
    const float bx = xS - xN, by = yS - yN, bz = zS - zN;
    const float m0 = ay * bz, m1 = az * by, m2 = az * bx, m3 = ax * bz, m4 = ax * by, m5 = ay * bx;
    const float vx = m0 - m1, vy = m2 - m3, vz = m4 - m5;
    const float s0 = vx * vx, s1 = vy * vy, s2 = vz * vz;
    const float l = __builtin_sqrtf((s0 + s1) + s2);
    if (l > 0.f) {
      nx = vx / l;
      ny = vy / l;
      nz = vz / l;
    }
  } else {
    const float dzdx = ((float)dS - (float)dN) / 2.0f;  // SLAM.cpp:421 ("x" runs over rows)
    const float dzdy = ((float)dE - (float)dW) / 2.0f;  // SLAM.cpp:422
    const float d0 = -dzdx, d1 = -dzdy, d2 = 1.0f;      // SLAM.cpp:424
    const double nv = __builtin_sqrt(((double)d0 * d0 + (double)d1 * d1) + (double)d2 * d2);
    const double sc = nv != 0.0 ? 1.0 / nv : 0.0;
    nx = (float)(d0 * sc);
    ny = (float)(d1 * sc);
    nz = (float)(d2 * sc);
  }
}

template <bool NORMALS>
__global__ __launch_bounds__(BP_THREADS) void bp_scatter_kernel(const uint16_t* __restrict__ depth, int npix, int cols,
                                                                float fx, float cx, float ox, float oy, float oz,
                                                                const int* __restrict__ block_offsets,
                                                                float* __restrict__ x, float* __restrict__ y,
                                                                float* __restrict__ z, float* __restrict__ nxp,
                                                                float* __restrict__ nyp, float* __restrict__ nzp,
                                                                int normals_mode) {
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
      if (NORMALS) {
        float a, b, cc;
        pixel_normal(depth, npix / cols, cols, r, c, fx, cx, normals_mode, a, b, cc);
        nxp[rank] = a;
        nyp[rank] = b;
        nzp[rank] = cc;
      }
    }
    off += wcount[k][0] + wcount[k][1] + wcount[k][2] + wcount[k][3];
  }
}

void launch_backproject(const uint16_t* depth, int rows, int cols, float fx, float cx, float ox, float oy, float oz,
                        float* x, float* y, float* z, float* nx, float* ny, float* nz, int normals_mode,
                        int* block_counts, int* n_out, hipStream_t s) {
  const int npix = rows * cols;
  const int nblocks = (npix + BP_BLOCK - 1) / BP_BLOCK;
  hipLaunchKernelGGL(bp_count_kernel, dim3(nblocks), dim3(BP_THREADS), 0, s, depth, npix, block_counts);
  hipLaunchKernelGGL(bp_scan_kernel, dim3(1), dim3(256), 0, s, block_counts, nblocks, n_out);
  if (nx)
    hipLaunchKernelGGL(bp_scatter_kernel<true>, dim3(nblocks), dim3(BP_THREADS), 0, s, depth, npix, cols, fx, cx, ox, oy,
                       oz, block_counts, x, y, z, nx, ny, nz, normals_mode);
  else
    hipLaunchKernelGGL(bp_scatter_kernel<false>, dim3(nblocks), dim3(BP_THREADS), 0, s, depth, npix, cols, fx, cx, ox,
                       oy, oz, block_counts, x, y, z, nx, ny, nz, normals_mode);
}

}  // namespace icpk
