// kernels_reduce.hip -- K2: masked reduction over the associations of one sweep.
//
// Folds into one pass what the reference does with four O(N) loops:
//   icp.cpp:186-200  association split + N x 3 matrix export
//   icp.cpp:212      M = previousMat.t() * dataMat      (9 sums, M[r][c] = sum b_r a_c)
//   icp.cpp:314-344  calculateOffset                    (3 sums of (float)(a - b))
//   icp.cpp:622-638  meanSquareError                    (1 sum of distances, count)
// plus sum a / sum b (6) for the centred Kabsch flavour (rigid_transform_3D.py:14-20).
//
// Summation order is the canonical tree of include/icpk.h (ICPK_RED_*): double
// accumulators, wave64 xor butterfly via __shfl_xor, fixed wave and block
// order -- no atomics, so results are bit-reproducible and equal to the
// oracle's orc_sums_canonical.
#include "icpk_internal.h"
#include "wave_sum.h"

namespace icpk {

// Diagnostic build only (-DICPK_RED_STAMPS, tools/stamp_reduce.py): wall_clock64 stamps per block
#ifdef ICPK_RED_STAMPS
__device__ unsigned long long red_dbg[8 * 256];
#define RED_STAMP(k)                                                               \
  do {                                                                             \
    if (threadIdx.x == 0) red_dbg[blockIdx.x * 8 + (k)] = wall_clock64();          \
  } while (0)
extern "C" int icpk_debug_read_red_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(red_dbg), sizeof(red_dbg));
}
#else
#define RED_STAMP(k)
#endif

// NACT: how many of the NSUM sums the consumer needs.  icpk_reduce and the Kabsch flavour take
// all 19; the reference flavour's loop step only reads [0..12] (M, mean difference, distance
// sum), so the device loop skips the sums of a and b: a third less butterfly.
// `block` of `nblocks` (the canonical geometry of the pair: nblocks = red_blocks(nq)); shared
// by the single-pair kernel and the frame-batch kernel (blockIdx.y = pair).
template <int NACT>
__device__ __forceinline__ void assoc_reduce_body(
    const nn_key_t* __restrict__ best, const float* __restrict__ ax, const float* __restrict__ ay,
    const float* __restrict__ az, int nq, const float* __restrict__ tx, const float* __restrict__ ty,
    const float* __restrict__ tz, const float4* __restrict__ o4, const float4* __restrict__ rec, float max_dist,
    int32_t* __restrict__ idx_out, float* __restrict__ dist_out, double* __restrict__ partial, int* __restrict__ pcount,
    LoopState* __restrict__ st, const int block, const int nblocks) {
  const int tid = threadIdx.x;
  const int P = nblocks * RED_THREADS;
  // (records path: a lane's first record is asked for BEFORE the loop state is looked at -- two cold round trips side by
  // side instead of one after the other, ~0.7 us of a 5 us kernel; if the loop has ended the loads were for nothing)
  const int i_first = block * RED_THREADS + tid;
  float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
  if (rec && i_first < nq) {
    f0 = rec[2 * (size_t)i_first];
    f1 = rec[2 * (size_t)i_first + 1];
  }
  if (st) {
    if (st->done | st->stop_after_transform) return;
    if (block == 0 && threadIdx.x == 0) st->sweeps += 1;  // this sweep's associations are consumed
  }
  RED_STAMP(0);
  double v[NACT];
#pragma unroll
  for (int s = 0; s < NACT; ++s) v[s] = 0.0;
  int cnt = 0;

  // the accumulation of one accepted pair (a = moved query, b = its match, d = their distance)
  auto add_pair = [&](float a0, float a1, float a2, float b0, float b1, float b2, float d) {
      const double da0 = a0, da1 = a1, da2 = a2, db0 = b0, db1 = b1, db2 = b2;
      v[0] += db0 * da0; v[1] += db0 * da1; v[2] += db0 * da2;
      v[3] += db1 * da0; v[4] += db1 * da1; v[5] += db1 * da2;
      v[6] += db2 * da0; v[7] += db2 * da1; v[8] += db2 * da2;
      v[9] += (double)(a0 - b0);
      v[10] += (double)(a1 - b1);
      v[11] += (double)(a2 - b2);
      v[12] += (double)d;
      if constexpr (NACT > 13) {
        v[13] += da0; v[14] += da1; v[15] += da2;
        v[16] += db0; v[17] += db1; v[18] += db2;
      }
      ++cnt;
  };
  if (rec) {  // (uniform) behind a grid sweep of the device loop: one coalesced 32-byte record per query, no gather
    for (int i = i_first; i < nq; i += P) {  // (the next record is on its way while this one is added)
      const float4 r0 = f0, r1 = f1;
      if (i + P < nq) {
        f0 = rec[2 * (size_t)(i + P)];
        f1 = rec[2 * (size_t)(i + P) + 1];
      }
      if (r0.w < max_dist) add_pair(r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, r0.w);  // icp.cpp:553 (false for NaN)
    }
  } else {
    for (int i = block * RED_THREADS + tid; i < nq; i += P) {
      const nn_key_t key = best[i];
      const float d = __uint_as_float((unsigned)(key >> 32));
      const int j = (int)(unsigned)(key & 0xffffffffu);
      if (idx_out) {  // (null in the device loop: icpk_get_associations unpacks on demand)
        idx_out[i] = j;
        dist_out[i] = d;
      }
      if (d < max_dist) {  // icp.cpp:553 (false for NaN)
        const float a0 = ax[i], a1 = ay[i], a2 = az[i];
        float b0, b1, b2;
        if (o4) {  // (uniform) one 16-byte gather instead of three 4-byte ones
          const float4 b = o4[j];
          b0 = b.x;
          b1 = b.y;
          b2 = b.z;
        } else {
          b0 = tx[j];
          b1 = ty[j];
          b2 = tz[j];
        }
        add_pair(a0, a1, a2, b0, b1, b2, d);
      }
    }
  }

  if (v[12] > -1.0) RED_STAMP(1);  // loads and accumulation done
  // the canonical wave64 tree (order 32, 16, ..., 1) as a reduce-scatter: wave_sum.h
  double u[WaveScatter<NACT>::H2];
  wave_reduce_scatter<NACT>(v, u, cnt);
  if (u[0] > -1.0) RED_STAMP(2);  // butterfly done

  __shared__ double ws[RED_THREADS / 64][NACT];
  __shared__ int wc[RED_THREADS / 64];
  const int wave = tid >> 6, lane = tid & 63;
  wave_scatter_store<NACT>(u, lane, ws[wave]);
  if (lane == 0) wc[wave] = cnt;
  __syncthreads();
  if (tid < NACT) partial[tid * RED_MAX_BLOCKS + block] = ((ws[0][tid] + ws[1][tid]) + ws[2][tid]) + ws[3][tid];
  if (tid == NACT) pcount[block] = wc[0] + wc[1] + wc[2] + wc[3];
  RED_STAMP(3);
}

template <int NACT>
__global__ __launch_bounds__(RED_THREADS) void assoc_reduce_kernel(
    const nn_key_t* __restrict__ best, const float* __restrict__ ax, const float* __restrict__ ay,
    const float* __restrict__ az, int nq, const float* __restrict__ tx, const float* __restrict__ ty,
    const float* __restrict__ tz, const float4* __restrict__ o4, const float4* __restrict__ rec, float max_dist,
    int32_t* __restrict__ idx_out, float* __restrict__ dist_out, double* __restrict__ partial, int* __restrict__ pcount,
    LoopState* __restrict__ st) {
  assoc_reduce_body<NACT>(best, ax, ay, az, nq, tx, ty, tz, o4, rec, max_dist, idx_out, dist_out, partial, pcount, st,
                          blockIdx.x, gridDim.x);
}

// frame-batch mode: blockIdx.y = pair; every pair keeps ITS canonical geometry (its own
// nblocks), so its sums equal those of a single-pair launch bit for bit
template <int NACT>
__global__ __launch_bounds__(RED_THREADS) void assoc_reduce_batch_kernel(const ReduceBatch b, float max_dist) {
  const ReduceArgs& a = b.p[blockIdx.y];
  if ((int)blockIdx.x >= a.nblocks) return;
  assoc_reduce_body<NACT>(a.best, a.ax, a.ay, a.az, a.nq, a.tx, a.ty, a.tz, a.o4, a.rec, max_dist, nullptr, nullptr,
                          a.partial, a.pcount, a.st, blockIdx.x, a.nblocks);
}

// K5: normal equations of the linearised point-to-plane step (extension; the
// reference only plans it, TODO:9).  Per accepted pair (dist < max_dist and a
// non-zero target normal n): J = [p x n ; n], r = (p - q).n; 21 upper-triangle
// entries of J J^T, 6 of J r, 1 distance sum -- same canonical tree as K2.
__global__ __launch_bounds__(RED_THREADS) void p2l_reduce_kernel(
    const nn_key_t* __restrict__ best, const float* __restrict__ ax, const float* __restrict__ ay,
    const float* __restrict__ az, int nq, const float* __restrict__ tx, const float* __restrict__ ty,
    const float* __restrict__ tz, const float* __restrict__ nxp, const float* __restrict__ nyp,
    const float* __restrict__ nzp, const float4* __restrict__ rec, float max_dist, int32_t* __restrict__ idx_out,
    float* __restrict__ dist_out, double* __restrict__ partial, int* __restrict__ pcount, LoopState* __restrict__ st) {
  const int tid = threadIdx.x;
  const int P = gridDim.x * RED_THREADS;
  // (records path: a lane's first record is asked for before the loop state is looked at, and inside the loop the NEXT
  // record while the normals of the current one are gathered -- a lane has 3-4 of them at Kinect-v2 size, each a chain of
  // record -> normal -> arithmetic otherwise)
  const int i_first = blockIdx.x * RED_THREADS + tid;
  float4 nx0 = make_float4(0.f, 0.f, 0.f, 0.f), nx1 = nx0;
  if (rec && i_first < nq) {
    nx0 = rec[2 * (size_t)i_first];
    nx1 = rec[2 * (size_t)i_first + 1];
  }
  if (st) {
    if (st->done | st->stop_after_transform) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->sweeps += 1;
  }
  double v[NP2L];
#pragma unroll
  for (int s = 0; s < NP2L; ++s) v[s] = 0.0;
  int cnt = 0;
  for (int i = i_first; i < nq; i += P) {
    float d, a0, a1, a2, b0 = 0.f, b1 = 0.f, b2 = 0.f;
    int j;
    if (rec) {  // (uniform) behind a grid sweep of the device loop: query, match and distance in one 32-byte record
      const float4 r0 = nx0, r1 = nx1;
      if (i + P < nq) {
        nx0 = rec[2 * (size_t)(i + P)];
        nx1 = rec[2 * (size_t)(i + P) + 1];
      }
      a0 = r0.x, a1 = r0.y, a2 = r0.z, d = r0.w;
      b0 = r1.x, b1 = r1.y, b2 = r1.z, j = __float_as_int(r1.w);
    } else {
      const nn_key_t key = best[i];
      d = __uint_as_float((unsigned)(key >> 32));
      j = (int)(unsigned)(key & 0xffffffffu);
      if (idx_out) {  // (null in the device loop: icpk_get_associations unpacks on demand)
        idx_out[i] = j;
        dist_out[i] = d;
      }
      a0 = ax[i], a1 = ay[i], a2 = az[i];
    }
    if (d < max_dist) {
      const double n0 = nxp[j], n1 = nyp[j], n2 = nzp[j];
      if (!(n0 == 0.0 && n1 == 0.0 && n2 == 0.0)) {
        if (!rec) b0 = tx[j], b1 = ty[j], b2 = tz[j];
        const double p0 = a0, p1 = a1, p2 = a2;
        const double q0 = b0, q1 = b1, q2 = b2;
        double J[6];
        J[0] = p1 * n2 - p2 * n1;
        J[1] = p2 * n0 - p0 * n2;
        J[2] = p0 * n1 - p1 * n0;
        J[3] = n0;
        J[4] = n1;
        J[5] = n2;
        const double r = ((p0 - q0) * n0 + (p1 - q1) * n1) + (p2 - q2) * n2;
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = a; b < 6; ++b) v[k++] += J[a] * J[b];
#pragma unroll
        for (int a = 0; a < 6; ++a) v[21 + a] += J[a] * r;
        v[27] += (double)d;
        ++cnt;
      }
    }
  }
  double u[WaveScatter<NP2L>::H2];
  wave_reduce_scatter<NP2L>(v, u, cnt);
  __shared__ double ws[RED_THREADS / 64][NP2L];
  __shared__ int wc[RED_THREADS / 64];
  const int wave = tid >> 6, lane = tid & 63;
  wave_scatter_store<NP2L>(u, lane, ws[wave]);
  if (lane == 0) wc[wave] = cnt;
  __syncthreads();
  if (tid < NP2L) partial[tid * RED_MAX_BLOCKS + blockIdx.x] = ((ws[0][tid] + ws[1][tid]) + ws[2][tid]) + ws[3][tid];
  if (tid == NP2L) pcount[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

void launch_p2l_reduce(const nn_key_t* best, const float* ax, const float* ay, const float* az, int nq, const float* tx,
                       const float* ty, const float* tz, const float* nx, const float* ny, const float* nz,
                       const float4* rec, float max_dist, int32_t* idx_out, float* dist_out, double* partial, int* pcount,
                       double* out, LoopState* st, hipStream_t s) {
  const int B = red_blocks(nq);
  hipLaunchKernelGGL(p2l_reduce_kernel, dim3(B), dim3(RED_THREADS), 0, s, best, ax, ay, az, nq, tx, ty, tz, nx, ny, nz,
                     rec, max_dist, idx_out, dist_out, partial, pcount, st);
  if (out) launch_reduce_final(partial, pcount, B, NP2L, out, s);
}

void launch_assoc_reduce(const nn_key_t* best, const float* ax, const float* ay, const float* az, int nq,
                         const float* tx, const float* ty, const float* tz, const float4* o4, const float4* rec,
                         float max_dist, int32_t* idx_out, float* dist_out, double* partial, int* pcount, double* out,
                         LoopState* st, int nact, hipStream_t s) {
  const int B = red_blocks(nq);
  if (nact == NSUM_REF && !out)
    hipLaunchKernelGGL(assoc_reduce_kernel<NSUM_REF>, dim3(B), dim3(RED_THREADS), 0, s, best, ax, ay, az, nq, tx, ty, tz,
                       o4, rec, max_dist, idx_out, dist_out, partial, pcount, st);
  else
    hipLaunchKernelGGL(assoc_reduce_kernel<NSUM>, dim3(B), dim3(RED_THREADS), 0, s, best, ax, ay, az, nq, tx, ty, tz, o4,
                       rec, max_dist, idx_out, dist_out, partial, pcount, st);
  if (out) launch_reduce_final(partial, pcount, B, NSUM, out, s);
}

void launch_assoc_reduce_batch(const ReduceBatch& b, int count, float max_dist, int nact, hipStream_t s) {
  int bmax = 0;
  for (int k = 0; k < count; ++k) bmax = b.p[k].nblocks > bmax ? b.p[k].nblocks : bmax;
  if (count <= 0 || bmax <= 0) return;
  if (nact == NSUM_REF)
    hipLaunchKernelGGL(assoc_reduce_batch_kernel<NSUM_REF>, dim3(bmax, count), dim3(RED_THREADS), 0, s, b, max_dist);
  else
    hipLaunchKernelGGL(assoc_reduce_batch_kernel<NSUM>, dim3(bmax, count), dim3(RED_THREADS), 0, s, b, max_dist);
}

}  // namespace icpk
