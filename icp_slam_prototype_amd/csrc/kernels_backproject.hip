// kernels_backproject.hip -- K4: depth image -> compacted xyz-SoA cloud.
//
// pointcloud.cpp:19-58: row-major scan, zero depth skipped,
//   p_z = (float)d / 5000.0f; p_x = (x - CX) * p_z / FX; p_y = (y - CX) * p_z / FX
// (CX and FX are used for the y axis too, pointcloud.cpp:39).  The unseeded
// rand()%40 subsample of pointcloud.cpp:28 has a seeded stand-in (bp_keep below; off by default).  Output order is
// the row-major order of the valid pixels (order-preserving compaction):
//   pass 1: per-1024-pixel block count (wave64 ballot + popcount)
//   pass 2: exclusive scan of the block counts (one workgroup)
//   pass 3: in-block rank from ballots, scatter.
#include "icpk_internal.h"

namespace icpk {

// The reference keeps one valid pixel in SUBSAMPLE_FACTOR at random (pointcloud.cpp:27-30, `rand() % 40`, never
// seeded).  Here, when a factor > 1 is set (icpk_set_subsample): a counter-based choice -- pixel p of image stream k
// is kept iff the upper half of splitmix64(seed + (k + 1) * golden + p * odd) is a multiple of the factor -- the same
// for the counting and the scatter pass, reproducible, restated in oracle/icp_oracle.py:subsample_keep.
// key = seed + (k + 1) * 0x9E3779B97F4A7C15 is formed on the host; factor <= 1: every valid pixel.
__host__ __device__ __forceinline__ bool bp_keep(unsigned long long key, int factor, int p) {
  if (factor <= 1) return true;
  unsigned long long z = key + (unsigned long long)(unsigned)p * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 32) % (unsigned)factor == 0u;
}

constexpr int BP_THREADS = 256;
constexpr int BP_PER_THREAD = 4;
constexpr int BP_BLOCK = BP_THREADS * BP_PER_THREAD;  // 1024 pixels per workgroup

__device__ __forceinline__ void bp_count_body(const uint16_t* __restrict__ depth, int npix,
                                              int* __restrict__ block_counts, const int block,
                                              const uint16_t* __restrict__ host_src = nullptr,
                                              uint16_t* __restrict__ raw_out = nullptr, unsigned long long sub_key = 0,
                                              int sub_factor = 0) {
  const int base = block * BP_BLOCK;
  int c = 0;
  if (host_src) {  // (all four loads in flight before the first is looked at: they cross PCIe)
    uint16_t d[BP_PER_THREAD];
#pragma unroll
    for (int k = 0; k < BP_PER_THREAD; ++k) {
      const int p = base + k * BP_THREADS + threadIdx.x;
      d[k] = p < npix ? host_src[p] : (uint16_t)0;
    }
#pragma unroll
    for (int k = 0; k < BP_PER_THREAD; ++k) {
      const int p = base + k * BP_THREADS + threadIdx.x;
      if (p < npix) raw_out[p] = d[k];
      c += __popcll(__ballot(d[k] != 0 && bp_keep(sub_key, sub_factor, p)));
    }
  } else {
#pragma unroll
    for (int k = 0; k < BP_PER_THREAD; ++k) {
      const int p = base + k * BP_THREADS + threadIdx.x;
      const bool valid = p < npix && depth[p] != 0 && bp_keep(sub_key, sub_factor, p);
      c += __popcll(__ballot(valid));
    }
  }
  __shared__ int wc[BP_THREADS / 64];
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[block] = wc[0] + wc[1] + wc[2] + wc[3];
}

__global__ __launch_bounds__(BP_THREADS) void bp_count_kernel(const uint16_t* __restrict__ depth, int npix,
                                                              int* __restrict__ block_counts, unsigned long long sub_key,
                                                              int sub_factor) {
  bp_count_body(depth, npix, block_counts, blockIdx.x, nullptr, nullptr, sub_key, sub_factor);
}

// frame-pair entry (icpk_backproject_pair): blockIdx.y = image (0: current frame / source, 1: previous / target)
__global__ __launch_bounds__(BP_THREADS) void bp_count_pair_kernel(const BpPair b, int npix) {
  const BpImage& im = b.im[blockIdx.y];
  bp_count_body(im.depth, npix, im.counts, blockIdx.x, im.host_src, im.raw_out, im.sub_key, im.sub_factor);
}

// exclusive scan in place; block_counts[nblocks] receives the total
__device__ __forceinline__ void bp_scan_body(int* __restrict__ block_counts, int nblocks, int* __restrict__ n_out) {
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

__global__ void bp_scan_kernel(int* __restrict__ block_counts, int nblocks, int* __restrict__ n_out) {
  bp_scan_body(block_counts, nblocks, n_out);
}

// n_host: pinned, mapped words the host spins on (icpk_backproject_pair returns as soon as both totals are known,
// while the scatter still runs) or nullptr
__global__ void bp_scan_pair_kernel(const BpPair b, int nblocks, int* __restrict__ n_out, int* __restrict__ n_host) {
  bp_scan_body(b.im[blockIdx.x].counts, nblocks, n_out + blockIdx.x);
  if (n_host && threadIdx.x == 0)  // (the thread that wrote the total)
    __hip_atomic_store(n_host + blockIdx.x, b.im[blockIdx.x].counts[nblocks], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
                                                                int normals_mode, unsigned long long sub_key,
                                                                int sub_factor) {
  __shared__ int wcount[BP_PER_THREAD][BP_THREADS / 64];
  const int base = blockIdx.x * BP_BLOCK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long mask[BP_PER_THREAD];
  uint16_t dv[BP_PER_THREAD];
#pragma unroll
  for (int k = 0; k < BP_PER_THREAD; ++k) {
    const int p = base + k * BP_THREADS + threadIdx.x;
    dv[k] = p < npix ? depth[p] : (uint16_t)0;
    if (!bp_keep(sub_key, sub_factor, p)) dv[k] = 0;  // (a pixel the subsample drops is an empty pixel from here on)
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

// frame-pair entry: scatter of both images with the camera pose applied on the way (icp.cpp:58-59, 70-71:
// rotate, then translate -- the arithmetic of K3, kernels_transform.hip, on the offset point), the source
// written to its pristine and its working copy at once, and every plane padded up to the next multiple of
// NN_TILE (workgroup 0 of each image: the total is on the device by now).  Replaces, bit for bit,
// icpk_backproject x 2 + icpk_transform_target / _source + icpk_commit_source: 3 launches instead of 25.
__global__ __launch_bounds__(BP_THREADS) void bp_scatter_pair_kernel(const BpPair b, int npix, int cols, int nblocks,
                                                                     float fx, float cx, float ox, float oy, float oz,
                                                                     const Rt rt, int posed) {
  const BpImage& im = b.im[blockIdx.y];
  const uint16_t* __restrict__ depth = im.depth;
  float* __restrict__ x = im.x;
  float* __restrict__ y = im.y;
  float* __restrict__ z = im.z;
  float* __restrict__ x2 = im.x2;
  float* __restrict__ y2 = im.y2;
  float* __restrict__ z2 = im.z2;
  __shared__ int wcount[BP_PER_THREAD][BP_THREADS / 64];
  const int base = blockIdx.x * BP_BLOCK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long mask[BP_PER_THREAD];
  uint16_t dv[BP_PER_THREAD];
#pragma unroll
  for (int k = 0; k < BP_PER_THREAD; ++k) {
    const int p = base + k * BP_THREADS + threadIdx.x;
    dv[k] = p < npix ? depth[p] : (uint16_t)0;
    if (!bp_keep(im.sub_key, im.sub_factor, p)) dv[k] = 0;
    mask[k] = __ballot(dv[k] != 0);
    if (lane == 0) wcount[k][wave] = __popcll(mask[k]);
  }
  __syncthreads();
  int off = im.counts[blockIdx.x];
#pragma unroll
  for (int k = 0; k < BP_PER_THREAD; ++k) {
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wcount[k][w];
    const int rank = off + before + __popcll(mask[k] & ((1ull << lane) - 1ull));
    if (im.point_of_pixel) {
      const int p = base + k * BP_THREADS + threadIdx.x;
      if (p < npix) im.point_of_pixel[p] = dv[k] != 0 ? rank : -1;
    }
    if (dv[k] != 0) {
      const int p = base + k * BP_THREADS + threadIdx.x;
      if (im.pixel_of_point) im.pixel_of_point[rank] = p;
      const int r = p / cols, c = p - r * cols;
      const float pz = ((float)dv[k]) / 5000.0f;  // pointcloud.cpp:37-39
      const float px = ((float)c - cx) * pz / fx;
      const float py = ((float)r - cx) * pz / fx;
      float vx = px + ox, vy = py + oy, vz = pz + oz;
      if (posed) {  // pointcloud.cpp:321-359: p <- fl32(fl32(R p) + t)
        const double dx = vx, dy = vy, dz = vz;
        vx = (float)__builtin_fma((double)rt.R[2], dz, __builtin_fma((double)rt.R[1], dy, (double)rt.R[0] * dx)) + rt.t[0];
        vy = (float)__builtin_fma((double)rt.R[5], dz, __builtin_fma((double)rt.R[4], dy, (double)rt.R[3] * dx)) + rt.t[1];
        vz = (float)__builtin_fma((double)rt.R[8], dz, __builtin_fma((double)rt.R[7], dy, (double)rt.R[6] * dx)) + rt.t[2];
      }
      x[rank] = vx;
      y[rank] = vy;
      z[rank] = vz;
      if (x2) {
        x2[rank] = vx;
        y2[rank] = vy;
        z2[rank] = vz;
      }
    }
    off += wcount[k][0] + wcount[k][1] + wcount[k][2] + wcount[k][3];
  }
  if (blockIdx.x == 0) {
    const int n = im.counts[nblocks];
    const int padded = ((n < 1 ? 1 : n) + NN_TILE - 1) / NN_TILE * NN_TILE;
    for (int i = n + (int)threadIdx.x; i < padded; i += BP_THREADS) {
      x[i] = y[i] = z[i] = im.pad;
      if (x2) x2[i] = y2[i] = z2[i] = im.pad;
    }
  }
}

void launch_backproject_pair(const BpPair& b, int rows, int cols, float fx, float cx, float ox, float oy, float oz,
                             const Rt& rt, int posed, int* n_out, int* n_host, hipStream_t s) {
  const int npix = rows * cols;
  const int nblocks = (npix + BP_BLOCK - 1) / BP_BLOCK;
  hipLaunchKernelGGL(bp_count_pair_kernel, dim3(nblocks, 2), dim3(BP_THREADS), 0, s, b, npix);
  hipLaunchKernelGGL(bp_scan_pair_kernel, dim3(2), dim3(256), 0, s, b, nblocks, n_out, n_host);
  hipLaunchKernelGGL(bp_scatter_pair_kernel, dim3(nblocks, 2), dim3(BP_THREADS), 0, s, b, npix, cols, nblocks, fx, cx, ox,
                     oy, oz, rt, posed);
}

void launch_backproject(const uint16_t* depth, int rows, int cols, float fx, float cx, float ox, float oy, float oz,
                        float* x, float* y, float* z, float* nx, float* ny, float* nz, int normals_mode,
                        int* block_counts, int* n_out, unsigned long long sub_key, int sub_factor, hipStream_t s) {
  const int npix = rows * cols;
  const int nblocks = (npix + BP_BLOCK - 1) / BP_BLOCK;
  hipLaunchKernelGGL(bp_count_kernel, dim3(nblocks), dim3(BP_THREADS), 0, s, depth, npix, block_counts, sub_key, sub_factor);
  hipLaunchKernelGGL(bp_scan_kernel, dim3(1), dim3(256), 0, s, block_counts, nblocks, n_out);
  if (nx)
    hipLaunchKernelGGL(bp_scatter_kernel<true>, dim3(nblocks), dim3(BP_THREADS), 0, s, depth, npix, cols, fx, cx, ox, oy,
                       oz, block_counts, x, y, z, nx, ny, nz, normals_mode, sub_key, sub_factor);
  else
    hipLaunchKernelGGL(bp_scatter_kernel<false>, dim3(nblocks), dim3(BP_THREADS), 0, s, depth, npix, cols, fx, cx, ox,
                       oy, oz, block_counts, x, y, z, nx, ny, nz, normals_mode, sub_key, sub_factor);
}

}  // namespace icpk
