// kernels_nn.hip -- K1: brute-force nearest neighbour for gfx950 (MI355X).
//
// Replaces the nested scan of icp.cpp:550-559 x icp.cpp:576-584
// (findGlobalNearestNeighborAssociations -> getNearestPoint -> distance).
//
// Results are bit-identical to that scan:
//   d(q,t) = sqrtf( (float)( (double)dx*dx + (double)dy*dy + (double)dz*dz ) )
//   (icp.cpp:606-620: pow(float,int) is double, one rounding to float, float
//   sqrt), compared with strict '<' on the float d, lowest target index wins.
//
// Mapping: one query per lane (registers), 256 lanes per workgroup; the target
// cloud streams through LDS in 1024-point xyz-SoA tiles (coalesced 16-byte
// global loads, broadcast ds_read_b128 in the inner loop).  The target range is
// also split over blockIdx.y so small clouds still fill 256 CUs; partial
// results merge with a 64-bit atomic min on (distance bits, index), which is
// the lexicographic min and therefore independent of scheduling order.
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

__global__ void fill_u64_kernel(nn_key_t* __restrict__ p, int n, nn_key_t v) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

void launch_fill_u64(nn_key_t* p, int n, nn_key_t v, hipStream_t s) {
  if (n <= 0) return;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_u64_kernel, dim3(blocks), dim3(256), 0, s, p, n, v);
}

// ---- K1 exact -----------------------------------------------------------------
__global__ __launch_bounds__(NN_THREADS) void nn_exact_kernel(NnArgs a) {
  __shared__ float4 sx[NN_TILE / 4];
  __shared__ float4 sy[NN_TILE / 4];
  __shared__ float4 sz[NN_TILE / 4];

  const int tid = threadIdx.x;
  const int i = blockIdx.x * NN_THREADS + tid;
  const bool live = i < a.nq;
  const float qx = live ? a.qx[i] : 0.f;
  const float qy = live ? a.qy[i] : 0.f;
  const float qz = live ? a.qz[i] : 0.f;

  const int ntiles = a.nt_pad / NN_TILE;
  const int tile0 = blockIdx.y * a.tiles_per_chunk;
  int tile1 = tile0 + a.tiles_per_chunk;
  if (tile1 > ntiles) tile1 = ntiles;

  float bestd = __builtin_inff();
  int besti = 0x7fffffff;
  if (blockIdx.y == 0) {  // icp.cpp:572-573: the scan is seeded with element 0
    bestd = pair_dist(qx, qy, qz, a.tx[0], a.ty[0], a.tz[0]);
    besti = 0;
  }

  const float4* __restrict__ gx = reinterpret_cast<const float4*>(a.tx);
  const float4* __restrict__ gy = reinterpret_cast<const float4*>(a.ty);
  const float4* __restrict__ gz = reinterpret_cast<const float4*>(a.tz);

  float4 px = gx[tile0 * (NN_TILE / 4) + tid];
  float4 py = gy[tile0 * (NN_TILE / 4) + tid];
  float4 pz = gz[tile0 * (NN_TILE / 4) + tid];

  for (int t = tile0; t < tile1; ++t) {
    __syncthreads();  // everyone finished reading the previous tile
    sx[tid] = px;
    sy[tid] = py;
    sz[tid] = pz;
    __syncthreads();
    if (t + 1 < tile1) {  // prefetch the next tile while this one is scanned
      px = gx[(t + 1) * (NN_TILE / 4) + tid];
      py = gy[(t + 1) * (NN_TILE / 4) + tid];
      pz = gz[(t + 1) * (NN_TILE / 4) + tid];
    }
    const int jbase = t * NN_TILE;
#pragma unroll 2
    for (int v = 0; v < NN_TILE / 4; ++v) {
      const float4 X = sx[v];  // same address in every lane: LDS broadcast
      const float4 Y = sy[v];
      const float4 Z = sz[v];
      const int j = jbase + 4 * v;
      float d;
      bool up;
      d = pair_dist(qx, qy, qz, X.x, Y.x, Z.x);
      up = d < bestd;  // icp.cpp:578 strict '<': ascending j keeps the lowest index
      bestd = up ? d : bestd;
      besti = up ? j : besti;
      d = pair_dist(qx, qy, qz, X.y, Y.y, Z.y);
      up = d < bestd;
      bestd = up ? d : bestd;
      besti = up ? j + 1 : besti;
      d = pair_dist(qx, qy, qz, X.z, Y.z, Z.z);
      up = d < bestd;
      bestd = up ? d : bestd;
      besti = up ? j + 2 : besti;
      d = pair_dist(qx, qy, qz, X.w, Y.w, Z.w);
      up = d < bestd;
      bestd = up ? d : bestd;
      besti = up ? j + 3 : besti;
    }
  }

  // chunk 0 always publishes (it holds the reference's seed element, also when
  // the distance is inf/NaN); other chunks only when they found something.
  if (live && (blockIdx.y == 0 || bestd < __builtin_inff())) {
    const nn_key_t key = ((nn_key_t)__float_as_uint(bestd) << 32) | (nn_key_t)(unsigned)besti;
    atomicMin(&a.best[i], key);
  }
}

void launch_nn_exact(const NnArgs& a, hipStream_t s) {
  const int ntiles = a.nt_pad / NN_TILE;
  const int nchunks = (ntiles + a.tiles_per_chunk - 1) / a.tiles_per_chunk;
  dim3 grid((a.nq + NN_THREADS - 1) / NN_THREADS, nchunks);
  hipLaunchKernelGGL(nn_exact_kernel, grid, dim3(NN_THREADS), 0, s, a);
}

// ---- test hook: the pair distance on its own --------------------------------
__global__ void pair_distance_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                     int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = pair_dist(a[i], a[n + i], a[2 * n + i], b[i], b[n + i], b[2 * n + i]);
}

void launch_pair_distance(const float* a, const float* b, float* out, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(pair_distance_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, out, n);
}

}  // namespace icpk
