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
#include "nn_device.h"
#include "solve_impl.h"

namespace icpk {

__global__ void fill_u64_kernel(nn_key_t* __restrict__ p, int n, nn_key_t v, const int* __restrict__ stop) {
  if (loop_stopped(stop)) return;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

void launch_fill_u64(nn_key_t* p, int n, nn_key_t v, const int* stop, hipStream_t s) {
  if (n <= 0) return;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_u64_kernel, dim3(blocks), dim3(256), 0, s, p, n, v, stop);
}

// ---- K1 exact -----------------------------------------------------------------
__global__ __launch_bounds__(NN_THREADS) void nn_exact_kernel(NnArgs a) {
  if (loop_stopped(a.stop)) return;
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

// ---- K1 filtered ------------------------------------------------------------------
// Same result as nn_exact_kernel, ~7 fp32 VALU per pair instead of ~32 mixed
// fp64/fp32: every query starts from a SEED candidate (its match of the previous
// ICP iteration, or of a coarse pre-pass over a decimated target) whose exact
// distance d_b bounds the search.  A target can only change the answer if its
// float distance is <= d_b, which implies ("filter bound")
//     e = fmaf(dz,dz, fmaf(dy,dy, dx*dx))  <=  T(d_b) = d_b^2 (1 + 2^-20) + 2^-120
// so the inner loop evaluates only e (fp32) and compares it with the per-lane
// threshold; the rare groups in which some lane passes are re-evaluated with the
// exact pair_dist() and merged lexicographically on (distance, index), which is
// order independent and equals the reference scan's "strict <, lowest index".
//
// Targets are wave-uniform, so they are fetched with scalar loads (s_load_dwordx8
// through the scalar cache) and consumed as SGPR operands: no LDS staging, no
// barriers, no VGPRs for target data.
// scan targets [j0, j1) (multiples of 2*NNF_G) for the Q queries of this lane.
// Two SGPR buffers (A, B) of NNF_G targets: the scalar loads of one group are in
// flight while the other is consumed.  The last prefetch reads NNF_G floats past
// j1; every cloud allocation carries that much slack.
template <int Q, bool PERM>
__device__ __forceinline__ void scan_range(const float* __restrict__ txp, const float* __restrict__ typ,
                                           const float* __restrict__ tzp, const int* __restrict__ tperm, int j0,
                                           int j1, const float (&qx)[Q],
                                           const float (&qy)[Q], const float (&qz)[Q], float (&bd)[Q], int (&bj)[Q],
                                           float (&T)[Q]) {
  float XA[NNF_G], YA[NNF_G], ZA[NNF_G], XB[NNF_G], YB[NNF_G], ZB[NNF_G];
#pragma unroll
  for (int k = 0; k < NNF_G; ++k) {
    XA[k] = txp[j0 + k];
    YA[k] = typ[j0 + k];
    ZA[k] = tzp[j0 + k];
  }
  for (int j = j0; j < j1; j += 2 * NNF_G) {
#pragma unroll
    for (int k = 0; k < NNF_G; ++k) {
      XB[k] = txp[j + NNF_G + k];
      YB[k] = typ[j + NNF_G + k];
      ZB[k] = tzp[j + NNF_G + k];
    }
    float e[NNF_G][Q], m[Q];
    bool hit;
    group_estimates<Q>(qx, qy, qz, XA, YA, ZA, e, m);
    hit = false;
#pragma unroll
    for (int u = 0; u < Q; ++u) hit |= (m[u] <= T[u]);
    if (__builtin_amdgcn_ballot_w64(hit) != 0) group_exact<Q, PERM>(qx, qy, qz, XA, YA, ZA, e, j, tperm, bd, bj, T);
#pragma unroll
    for (int k = 0; k < NNF_G; ++k) {
      XA[k] = txp[j + 2 * NNF_G + k];
      YA[k] = typ[j + 2 * NNF_G + k];
      ZA[k] = tzp[j + 2 * NNF_G + k];
    }
    group_estimates<Q>(qx, qy, qz, XB, YB, ZB, e, m);
    hit = false;
#pragma unroll
    for (int u = 0; u < Q; ++u) hit |= (m[u] <= T[u]);
    if (__builtin_amdgcn_ballot_w64(hit) != 0)
      group_exact<Q, PERM>(qx, qy, qz, XB, YB, ZB, e, j + NNF_G, tperm, bd, bj, T);
  }
}

// every target of the chunk is scanned (brute force over all Nq x Nt pairs); the
// box-pruned variant lives in kernels_nn_pruned.hip
template <int Q>
__global__ __launch_bounds__(NN_THREADS) void nn_filtered_kernel(
    const float* __restrict__ qxp, const float* __restrict__ qyp, const float* __restrict__ qzp, int nq,
    const float* __restrict__ txp, const float* __restrict__ typ, const float* __restrict__ tzp, int nt_pad,
    int tiles_per_chunk, const nn_key_t* __restrict__ seed, int seed_scale, nn_key_t* __restrict__ best,
    const int* __restrict__ stop) {
  if (loop_stopped(stop)) return;
  const int tid = threadIdx.x;
  const int ibase = blockIdx.x * (NN_THREADS * Q) + tid;
  float qx[Q], qy[Q], qz[Q], bd[Q], T[Q];
  int bj[Q], qi[Q];
  nn_key_t key0[Q];
#pragma unroll
  for (int u = 0; u < Q; ++u) {
    const int ip = ibase + u * NN_THREADS;
    const bool live = ip < nq;
    const int i = live ? ip : 0;
    qi[u] = live ? i : -1;
    qx[u] = live ? qxp[i] : 0.f;
    qy[u] = live ? qyp[i] : 0.f;
    qz[u] = live ? qzp[i] : 0.f;
    int js = live ? (int)(unsigned)(seed[i] & 0xffffffffu) * seed_scale : 0;
    float ds = pair_dist(qx[u], qy[u], qz[u], txp[js], typ[js], tzp[js]);
    if (!(ds <= 3.402823466e38f)) {  // inf/NaN: fall back to the reference's literal seed, element 0
      js = 0;
      ds = pair_dist(qx[u], qy[u], qz[u], txp[0], typ[0], tzp[0]);
    }
    bd[u] = ds;
    bj[u] = js;
    T[u] = filt_threshold(ds);
    key0[u] = ((nn_key_t)__float_as_uint(ds) << 32) | (nn_key_t)(unsigned)js;
  }

  const int ntiles = nt_pad / NN_TILE;
  const int tile0 = blockIdx.y * tiles_per_chunk;
  int tile1 = tile0 + tiles_per_chunk;
  if (tile1 > ntiles) tile1 = ntiles;

  scan_range<Q, false>(txp, typ, tzp, nullptr, tile0 * NN_TILE, tile1 * NN_TILE, qx, qy, qz, bd, bj, T);

#pragma unroll
  for (int u = 0; u < Q; ++u) {
    const nn_key_t key = ((nn_key_t)__float_as_uint(bd[u]) << 32) | (nn_key_t)(unsigned)bj[u];
    // chunk 0 always publishes (so the seed candidate itself is in the result);
    // the others only if they improved on it
    if (qi[u] >= 0 && (blockIdx.y == 0 || key < key0[u])) atomicMin(&best[qi[u]], key);
  }
}

void launch_nn_filtered(const NnArgs& a, const nn_key_t* seed, int seed_scale, int q_per_lane, hipStream_t s) {
  const int ntiles = a.nt_pad / NN_TILE;
  const int nchunks = (ntiles + a.tiles_per_chunk - 1) / a.tiles_per_chunk;
  const int q = q_per_lane == 2 ? 2 : 1;
  dim3 grid((a.nq + q * NN_THREADS - 1) / (q * NN_THREADS), nchunks);
  if (q == 2)
    hipLaunchKernelGGL(nn_filtered_kernel<2>, grid, dim3(NN_THREADS), 0, s, a.qx, a.qy, a.qz, a.nq, a.tx, a.ty, a.tz,
                       a.nt_pad, a.tiles_per_chunk, seed, seed_scale, a.best, a.stop);
  else
    hipLaunchKernelGGL(nn_filtered_kernel<1>, grid, dim3(NN_THREADS), 0, s, a.qx, a.qy, a.qz, a.nq, a.tx, a.ty, a.tz,
                       a.nt_pad, a.tiles_per_chunk, seed, seed_scale, a.best, a.stop);
}

// Bounding boxes of the target cloud: one 512-lane workgroup per 1024-point tile, one
// wave per 128-point sub-tile (2 points per lane, wave64 min/max butterfly).  Padded
// (+inf) slots are ignored; an all-padding sub-tile gets the empty box (+inf, -inf).
__global__ __launch_bounds__(512) void tile_boxes_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, int n, float* __restrict__ tbox,
                                                         int tbox_stride, float* __restrict__ sbox, int sbox_stride) {
  __shared__ float wlo[3][NN_SUBS], whi[3][NN_SUBS];
  const int t = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* p[3] = {x, y, z};
  float lo[3], hi[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    lo[c] = __builtin_inff();
    hi[c] = -__builtin_inff();
#pragma unroll
    for (int r = 0; r < NN_SUB / 64; ++r) {
      const int j = t * NN_TILE + wave * NN_SUB + r * 64 + lane;
      if (j < n) {
        const float v = p[c][j];
        lo[c] = __builtin_fminf(lo[c], v);
        hi[c] = __builtin_fmaxf(hi[c], v);
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      lo[c] = __builtin_fminf(lo[c], __shfl_xor(lo[c], m, 64));
      hi[c] = __builtin_fmaxf(hi[c], __shfl_xor(hi[c], m, 64));
    }
    if (lane == 0) {
      sbox[c * sbox_stride + t * NN_SUBS + wave] = lo[c];
      sbox[(3 + c) * sbox_stride + t * NN_SUBS + wave] = hi[c];
      wlo[c][wave] = lo[c];
      whi[c][wave] = hi[c];
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    float l = wlo[c][0], h = whi[c][0];
    for (int k = 1; k < NN_SUBS; ++k) {
      l = __builtin_fminf(l, wlo[c][k]);
      h = __builtin_fmaxf(h, whi[c][k]);
    }
    tbox[c * tbox_stride + t] = l;
    tbox[(3 + c) * tbox_stride + t] = h;
  }
}

void launch_tile_boxes(const float* x, const float* y, const float* z, int n, int ntiles, const NnBoxes& b,
                       hipStream_t s) {
  hipLaunchKernelGGL(tile_boxes_kernel, dim3(ntiles), dim3(512), 0, s, x, y, z, n, b.tbox, b.tbox_stride, b.sbox,
                     b.sbox_stride);
}

// every `stride`-th target -> coarse cloud for the seeding pre-pass
__global__ void decimate_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                int n, int stride, float* __restrict__ ox, float* __restrict__ oy,
                                float* __restrict__ oz, int n_out_pad) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_out_pad) return;
  const long long j = (long long)k * stride;
  const bool ok = j < n;
  ox[k] = ok ? x[j] : __builtin_inff();
  oy[k] = ok ? y[j] : __builtin_inff();
  oz[k] = ok ? z[j] : __builtin_inff();
}

void launch_decimate(const float* x, const float* y, const float* z, int n, int stride, float* ox, float* oy, float* oz,
                     int n_out_pad, hipStream_t s) {
  hipLaunchKernelGGL(decimate_kernel, dim3((n_out_pad + 255) / 256), dim3(256), 0, s, x, y, z, n, stride, ox, oy, oz,
                     n_out_pad);
}

// ---- test hook: the pair distance on its own --------------------------------
// point3 == 0: icp.cpp:606-620 (the loop's distance); 1: icp.cpp:595-602 (cv::Point3f overload)
__global__ void pair_distance_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                     int n, int point3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = point3 ? distance3(a[i], a[n + i], a[2 * n + i], b[i], b[n + i], b[2 * n + i])
                  : pair_dist(a[i], a[n + i], a[2 * n + i], b[i], b[n + i], b[2 * n + i]);
}

void launch_pair_distance(const float* a, const float* b, float* out, int n, int point3, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(pair_distance_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, out, n, point3);
}

}  // namespace icpk
