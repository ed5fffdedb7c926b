// kernels_transform.hip -- K3: rigid transform of the working source cloud.
//
// pointcloud.cpp:321-346 PointCloud::rotate (p <- R p about the world origin;
// the reference goes through N x 3 -> 3 x N transposes and a 32F GEMM with
// double accumulation) followed by pointcloud.cpp:349-359 translate (float +=):
//     p' = fl32( fl32(R p) + t )
// In place on xyz-SoA planes, 16 bytes per lane per access.
#include "icpk_internal.h"

namespace icpk {

__device__ __forceinline__ float rot_row(float r0, float r1, float r2, float x, float y, float z) {
  // products of two floats are exact in double; two double roundings, then one to float
  return (float)__builtin_fma((double)r2, (double)z, __builtin_fma((double)r1, (double)y, (double)r0 * (double)x));
}

__global__ __launch_bounds__(256) void transform_kernel(float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ z, int n4, Rt rt) {
  float4* x4 = reinterpret_cast<float4*>(x);
  float4* y4 = reinterpret_cast<float4*>(y);
  float4* z4 = reinterpret_cast<float4*>(z);
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 X = x4[i], Y = y4[i], Z = z4[i];
    float4 ox, oy, oz;
#define ICPK_ROW(c)                                                               \
  ox.c = rot_row(rt.R[0], rt.R[1], rt.R[2], X.c, Y.c, Z.c) + rt.t[0];             \
  oy.c = rot_row(rt.R[3], rt.R[4], rt.R[5], X.c, Y.c, Z.c) + rt.t[1];             \
  oz.c = rot_row(rt.R[6], rt.R[7], rt.R[8], X.c, Y.c, Z.c) + rt.t[2];
    ICPK_ROW(x) ICPK_ROW(y) ICPK_ROW(z) ICPK_ROW(w)
#undef ICPK_ROW
    x4[i] = ox;
    y4[i] = oy;
    z4[i] = oz;
  }
}

void launch_transform(float* x, float* y, float* z, int n, const Rt& rt, hipStream_t s) {
  if (n <= 0) return;
  const int n4 = (n + 3) / 4;  // planes are padded to a multiple of NN_TILE floats
  int blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(transform_kernel, dim3(blocks), dim3(256), 0, s, x, y, z, n4, rt);
}

__global__ void fill_f32_kernel(float* __restrict__ p, int n, float v) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

void launch_fill_f32(float* p, int n, float v, hipStream_t s) {
  if (n <= 0) return;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, s, p, n, v);
}

// Ingest of a device-resident cloud into a context (frame-batch set-up): the three planes, which
// may lie anywhere, into the context's plane layout, padded up to n_pad with `pad`, and, for a source
// cloud, into the working copy as well -- ONE launch instead of three copies, three fills and the
// copy of the working source.  d2 == nullptr: one destination.
__device__ __forceinline__ void ingest_cloud_body(const IngestArgs& a, int block, int nblocks) {
  const int stride = nblocks * 256;
  for (int i = block * 256 + threadIdx.x; i < a.n_pad; i += stride) {
    const bool in = i < a.n;
    const float vx = in ? a.x[i] : a.pad, vy = in ? a.y[i] : a.pad, vz = in ? a.z[i] : a.pad;
    a.d1[i] = vx;
    a.d1[(size_t)a.cap1 + i] = vy;
    a.d1[2 * (size_t)a.cap1 + i] = vz;
    if (a.d2) {
      a.d2[i] = vx;
      a.d2[(size_t)a.cap2 + i] = vy;
      a.d2[2 * (size_t)a.cap2 + i] = vz;
    }
  }
}

__global__ __launch_bounds__(256) void ingest_cloud_kernel(const IngestArgs a) { ingest_cloud_body(a, blockIdx.x, gridDim.x); }

// frame-batch set-up: blockIdx.y = cloud of the group
__global__ __launch_bounds__(256) void ingest_cloud_batch_kernel(const SetupBatchOf<IngestArgs> b) {
  ingest_cloud_body(b.p[blockIdx.y], blockIdx.x, gridDim.x);
}

static inline int ingest_blocks(int n_pad) {
  int blocks = (n_pad + 255) / 256;
  return blocks > 2048 ? 2048 : blocks;
}

void launch_ingest_cloud(const float* x, const float* y, const float* z, int n, int n_pad, float pad, float* d1, int cap1,
                         float* d2, int cap2, hipStream_t s) {
  if (n_pad <= 0) return;
  IngestArgs a{x, y, z, n, n_pad, pad, cap1, d1, d2, cap2, 0};
  if (SetupRecorder* r = setup_recorder()) {
    if (r->n < SETUP_MAX_CALLS) {
      r->calls[r->n].kind = SK_INGEST;
      r->calls[r->n++].ingest = a;
    } else {
      r->overflow = true;
    }
    return;
  }
  hipLaunchKernelGGL(ingest_cloud_kernel, dim3(ingest_blocks(n_pad)), dim3(256), 0, s, a);
}

void launch_ingest_batch(const SetupBatchOf<IngestArgs>& b, int count, hipStream_t s) {
  int m = 0;
  for (int k = 0; k < count; ++k) m = b.p[k].n_pad > m ? b.p[k].n_pad : m;
  if (count <= 0 || m <= 0) return;
  int blocks = ingest_blocks(m);
  if (blocks > 512) blocks = 512;  // (x count clouds: plenty of workgroups; the body strides)
  hipLaunchKernelGGL(ingest_cloud_batch_kernel, dim3(blocks, count), dim3(256), 0, s, b);
}

}  // namespace icpk
