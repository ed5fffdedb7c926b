// kernels_frontend.hip -- the two remaining pieces either side of the loop (SURVEY.md 8a row 8,
// 8f rank 1):
//   * key-point association lists: icp.cpp:488-515 findGlobalKeyPointAssociations builds, in
//     query order, the accepted (query, nearest) pairs with their distances (`associations`,
//     `errors`) and APPENDS the rejected queries to `nonAssociations` (:507-509).  The NN sweep is
//     the ordinary K1 (getNearestKeyPoint, :517-539, is the same scan as getNearestPoint); what is
//     new is the order-preserving split of its result: count per 1024-query block (wave64
//     ballot + popcount), exclusive scan of the block counts, scatter by rank.
//   * depth filter: SLAM.cpp:553-574 filterDepthImage = range clamp to [min, max] (else 0), then
//     cv::dilate and cv::erode with a 5x5 rectangle, fused into ONE LDS-tiled pass over the
//     uint16 image (separable max then min, halo of 4 pixels).
#include "icpk_internal.h"

namespace icpk {

// ---- association split ----------------------------------------------------------------------
constexpr int AS_THREADS = 256;
constexpr int AS_PER_THREAD = 4;
constexpr int AS_BLOCK = AS_THREADS * AS_PER_THREAD;

__device__ __forceinline__ bool as_accept(nn_key_t key, float max_dist) {
  return __uint_as_float((unsigned)(key >> 32)) < max_dist;  // icp.cpp:503 (false for NaN)
}

__global__ __launch_bounds__(AS_THREADS) void as_count_kernel(const nn_key_t* __restrict__ best, int nq, float max_dist,
                                                              int* __restrict__ block_counts) {
  const int base = blockIdx.x * AS_BLOCK;
  int c = 0;
#pragma unroll
  for (int k = 0; k < AS_PER_THREAD; ++k) {
    const int i = base + k * AS_THREADS + threadIdx.x;
    c += __popcll(__ballot(i < nq && as_accept(best[i], max_dist)));
  }
  __shared__ int wc[AS_THREADS / 64];
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// accepted query i -> slot (accepted before i); rejected query i -> slot i - (accepted before i)
__global__ __launch_bounds__(AS_THREADS) void as_scatter_kernel(const nn_key_t* __restrict__ best, int nq,
                                                                float max_dist, const int* __restrict__ block_offsets,
                                                                int32_t* __restrict__ assoc_q,
                                                                int32_t* __restrict__ assoc_t,
                                                                float* __restrict__ assoc_d,
                                                                int32_t* __restrict__ rej_q) {
  __shared__ int wcount[AS_PER_THREAD][AS_THREADS / 64];
  const int base = blockIdx.x * AS_BLOCK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long mask[AS_PER_THREAD];
  nn_key_t key[AS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < AS_PER_THREAD; ++k) {
    const int i = base + k * AS_THREADS + threadIdx.x;
    key[k] = i < nq ? best[i] : NN_KEY_INIT;
    mask[k] = __ballot(i < nq && as_accept(key[k], max_dist));
    if (lane == 0) wcount[k][wave] = __popcll(mask[k]);
  }
  __syncthreads();
  int off = block_offsets[blockIdx.x];
#pragma unroll
  for (int k = 0; k < AS_PER_THREAD; ++k) {
    const int i = base + k * AS_THREADS + threadIdx.x;
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wcount[k][w];
    const int rank = off + before + __popcll(mask[k] & ((1ull << lane) - 1ull));
    if (i < nq) {
      if ((mask[k] >> lane) & 1ull) {
        assoc_q[rank] = i;
        assoc_t[rank] = (int32_t)(unsigned)(key[k] & 0xffffffffu);
        assoc_d[rank] = __uint_as_float((unsigned)(key[k] >> 32));
      } else {
        rej_q[i - rank] = i;
      }
    }
    off += wcount[k][0] + wcount[k][1] + wcount[k][2] + wcount[k][3];
  }
}

// exclusive scan in place of <= a few thousand block counts; counts[nblocks] and *total = sum
__global__ void as_scan_kernel(int* __restrict__ counts, int nblocks, int* __restrict__ total) {
  __shared__ int carry;
  __shared__ int wsum[4];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? counts[i] : 0;
    int incl = v;
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
    if (i < nblocks) counts[i] = c + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = c + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counts[nblocks] = carry;
    *total = carry;
  }
}

void launch_assoc_split(const nn_key_t* best, int nq, float max_dist, int* block_counts, int* n_accepted,
                        int32_t* assoc_q, int32_t* assoc_t, float* assoc_d, int32_t* rej_q, hipStream_t s) {
  if (nq <= 0) return;
  const int nblocks = (nq + AS_BLOCK - 1) / AS_BLOCK;
  hipLaunchKernelGGL(as_count_kernel, dim3(nblocks), dim3(AS_THREADS), 0, s, best, nq, max_dist, block_counts);
  hipLaunchKernelGGL(as_scan_kernel, dim3(1), dim3(256), 0, s, block_counts, nblocks, n_accepted);
  hipLaunchKernelGGL(as_scatter_kernel, dim3(nblocks), dim3(AS_THREADS), 0, s, best, nq, max_dist, block_counts, assoc_q,
                     assoc_t, assoc_d, rej_q);
}

// ---- depth filter ---------------------------------------------------------------------------
// One workgroup = a 64 x 16 tile of output pixels.  Output pixel (y, x) is the minimum over rows
// y - ay .. y - ay + 4, columns x - ax .. x - ax + 4 of the DILATED image, whose pixel (v, u) is the
// maximum over rows v - ay .. v - ay + 4, columns u - ax .. u - ax + 4 of the range-clamped input:
// the tile needs the input from (y0 - 2 ay, x0 - 2 ax), 24 rows x 72 columns.  Out-of-image pixels
// never win (OpenCV's default BORDER_CONSTANT with morphologyDefaultBorderValue()): 0 for the
// maximum, 65535 for the minimum -- the dilated value of a position outside the image is
// therefore forced to 65535 before the erode stage.  morph == 0: range clamp only.
constexpr int DF_TW = 64, DF_TH = 16, DF_K = 5;
constexpr int DF_IW = DF_TW + 2 * (DF_K - 1), DF_IH = DF_TH + 2 * (DF_K - 1);  // 72 x 24 input
constexpr int DF_DW = DF_TW + (DF_K - 1), DF_DH = DF_TH + (DF_K - 1);          // 68 x 20 dilated

__device__ __forceinline__ unsigned df_clamp(unsigned d, unsigned min_d, unsigned max_d) {
  return (d > max_d || d < min_d) ? 0u : d;  // SLAM.cpp:558-566
}

__global__ __launch_bounds__(256) void depth_filter_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                           int rows, int cols, unsigned min_d, unsigned max_d, int ax,
                                                           int ay, int morph) {
  __shared__ uint16_t A[DF_IH][DF_IW];   // clamped input
  __shared__ uint16_t B[DF_IH][DF_DW];   // horizontal max
  __shared__ uint16_t C[DF_DH][DF_DW];   // dilated
  __shared__ uint16_t D[DF_DH][DF_TW];   // horizontal min
  const int x0 = blockIdx.x * DF_TW, y0 = blockIdx.y * DF_TH;
  const int t = threadIdx.x;
  if (!morph) {
    for (int p = t; p < DF_TW * DF_TH; p += 256) {
      const int y = y0 + p / DF_TW, x = x0 + p % DF_TW;
      if (y < rows && x < cols) out[(size_t)y * cols + x] = (uint16_t)df_clamp(in[(size_t)y * cols + x], min_d, max_d);
    }
    return;
  }
  const int ix0 = x0 - 2 * ax, iy0 = y0 - 2 * ay;  // input origin of the tile
  for (int p = t; p < DF_IW * DF_IH; p += 256) {
    const int r = p / DF_IW, c = p % DF_IW;
    const int y = iy0 + r, x = ix0 + c;
    unsigned v = 0u;
    if (y >= 0 && y < rows && x >= 0 && x < cols) v = df_clamp(in[(size_t)y * cols + x], min_d, max_d);
    A[r][c] = (uint16_t)v;
  }
  __syncthreads();
  for (int p = t; p < DF_DW * DF_IH; p += 256) {
    const int r = p / DF_DW, c = p % DF_DW;
    unsigned m = A[r][c];
#pragma unroll
    for (int k = 1; k < DF_K; ++k) m = max(m, (unsigned)A[r][c + k]);
    B[r][c] = (uint16_t)m;
  }
  __syncthreads();
  const int dx0 = x0 - ax, dy0 = y0 - ay;  // origin of the dilated region
  for (int p = t; p < DF_DW * DF_DH; p += 256) {
    const int r = p / DF_DW, c = p % DF_DW;
    unsigned m = B[r][c];
#pragma unroll
    for (int k = 1; k < DF_K; ++k) m = max(m, (unsigned)B[r + k][c]);
    const int y = dy0 + r, x = dx0 + c;
    if (!(y >= 0 && y < rows && x >= 0 && x < cols)) m = 65535u;  // outside the image: never the minimum
    C[r][c] = (uint16_t)m;
  }
  __syncthreads();
  for (int p = t; p < DF_TW * DF_DH; p += 256) {
    const int r = p / DF_TW, c = p % DF_TW;
    unsigned m = C[r][c];
#pragma unroll
    for (int k = 1; k < DF_K; ++k) m = min(m, (unsigned)C[r][c + k]);
    D[r][c] = (uint16_t)m;
  }
  __syncthreads();
  for (int p = t; p < DF_TW * DF_TH; p += 256) {
    const int r = p / DF_TW, c = p % DF_TW;
    unsigned m = D[r][c];
#pragma unroll
    for (int k = 1; k < DF_K; ++k) m = min(m, (unsigned)D[r + k][c]);
    const int y = y0 + r, x = x0 + c;
    if (y < rows && x < cols) out[(size_t)y * cols + x] = (uint16_t)m;
  }
}

void launch_depth_filter(const uint16_t* in, uint16_t* out, int rows, int cols, int min_d, int max_d, int ax, int ay,
                         int morph, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return;
  const unsigned lo = min_d < 0 ? 0u : (unsigned)min_d;
  const unsigned hi = max_d > 65535 ? 65535u : (max_d < 0 ? 0u : (unsigned)max_d);
  hipLaunchKernelGGL(depth_filter_kernel, dim3((cols + DF_TW - 1) / DF_TW, (rows + DF_TH - 1) / DF_TH), dim3(256), 0, s,
                     in, out, rows, cols, lo, hi, ax, ay, morph);
}

}  // namespace icpk
