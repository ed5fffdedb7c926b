// kernels_loop.hip -- the part of an ICP iteration the reference runs on the host
// between two NN sweeps, moved onto the device so that a whole alignment is enqueued
// without a host round trip per iteration:
//   * stage 2 of the canonical reduction (the <= 256 per-block partial sums go through
//     the same 4-wave xor-butterfly tree as stage 1);
//   * loop control of icp.cpp:155 (mse > threshold && i < maxIterations) and the
//     < 3 pairs fallback of icp.cpp:163-182;
//   * the solve (icp.cpp:212-246 / rigid_transform_3D.py:9-40 / point-to-plane), by the
//     very same source as the host loop (solve_impl.h), on one lane in float64;
//   * pose accumulation (icp.cpp:227-233, 266-268) and the per-iteration trace.
// K3 then reads its transform from the state instead of kernel arguments.
#include "icpk_internal.h"
#include "loop_init.h"
#include "solve_impl.h"
#include "wave_sum.h"

namespace icpk {

// Diagnostic build only (tools/stamp_step.py, -DICPK_STEP_STAMPS): wall_clock64 (100 MHz)
// stamps of the phases of loop_step_kernel, one row per iteration.
#ifdef ICPK_STEP_STAMPS
__device__ unsigned long long step_dbg[8 * 64];
#define STEP_STAMP(i, k)                                                        \
  do {                                                                          \
    if (threadIdx.x == 0 && (i) < 64) step_dbg[(i) * 8 + (k)] = wall_clock64(); \
  } while (0)
extern "C" int icpk_debug_read_step_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(step_dbg), sizeof(step_dbg));
}
#else
#define STEP_STAMP(i, k)
#endif

// canonical stage 2: 256 slots (slot b = sums of block b, +0.0 beyond nblocks), one slot
// per lane; wave butterfly, ((w0+w1)+w2)+w3.  Result in sums[] of thread 0.
template <int NS, int NACT = NS>
__device__ __forceinline__ void tree_stage2(const double* __restrict__ partial, const int* __restrict__ pcount,
                                            int nblocks, double (&sums)[NS], long long& count) {
  // only the first NACT sums were produced (and are consumed); the others read as 0
  __shared__ double ws[4][NACT];
  __shared__ int wc[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double v[NACT];
  int c = 0;
#pragma unroll
  for (int s = 0; s < NACT; ++s) v[s] = tid < nblocks ? partial[s * RED_MAX_BLOCKS + tid] : 0.0;  // [sum][block]: coalesced
  if (tid < nblocks) c = pcount[tid];
  double u[WaveScatter<NACT>::H2];
  wave_reduce_scatter<NACT>(v, u, c);
  wave_scatter_store<NACT>(u, lane, ws[wave]);
  if (lane == 0) wc[wave] = c;
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) sums[s] = s < NACT ? ((ws[0][s] + ws[1][s]) + ws[2][s]) + ws[3][s] : 0.0;
    count = (long long)wc[0] + wc[1] + wc[2] + wc[3];
  }
}

// count_as_double: the pair count as a float64 in out[NS] (exact below 2^53) instead of an int64 -- the query-sharded
// loop all-reduces sums and count in ONE float64 message; st != nullptr: a launch inside a device loop (no-op once
// the loop has exited)
template <int NS>
__global__ __launch_bounds__(256) void reduce_final_kernel(const double* __restrict__ partial,
                                                           const int* __restrict__ pcount, int nblocks,
                                                           double* __restrict__ out, int count_as_double,
                                                           const LoopState* __restrict__ st) {
  if (st && st->done) return;
  double sums[NS];
  long long count = 0;
  tree_stage2<NS>(partial, pcount, nblocks, sums, count);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) out[s] = sums[s];
    if (count_as_double)
      out[NS] = (double)count;
    else
      reinterpret_cast<long long*>(out)[NS] = count;
  }
}

void launch_reduce_final(const double* partial, const int* pcount, int nblocks, int nsum, double* out, hipStream_t s) {
  if (nsum == NP2L)
    hipLaunchKernelGGL(reduce_final_kernel<NP2L>, dim3(1), dim3(256), 0, s, partial, pcount, nblocks, out, 0, nullptr);
  else
    hipLaunchKernelGGL(reduce_final_kernel<NSUM>, dim3(1), dim3(256), 0, s, partial, pcount, nblocks, out, 0, nullptr);
}
void launch_reduce_final_shard(const double* partial, const int* pcount, int nblocks, double* out, const LoopState* st,
                               hipStream_t s) {
  hipLaunchKernelGGL(reduce_final_kernel<NSUM>, dim3(1), dim3(256), 0, s, partial, pcount, nblocks, out, 1, st);
}

__device__ void compose_rt(const float Rf[9], const float tf[3], double Tk[12]) {
  double Tn[12];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += (double)Rf[3 * r + k] * Tk[4 * k + c];
      Tn[4 * r + c] = s + (c == 3 ? (double)tf[r] : 0.0);
    }
  for (int k = 0; k < 12; ++k) Tk[k] = Tn[k];
}

// thread 0, on every way out of a loop step: tell the host (see LoopState::progress).  The words are hints
// that only decide how much the host enqueues -- a stale read costs a no-op launch, never a result -- so
// the stores are relaxed: no release, which at system scope would write the whole L2 back.
__device__ __forceinline__ void publish_progress(LoopState* __restrict__ st) {
  const int k = st->steps + 1;
  st->steps = k;
  int* pr = st->progress;
  if (pr) {
    const int e = st->epoch;
    __hip_atomic_store(pr + 1, (e << 2) | ((st->done != 0) << 1) | ((st->done | st->stop_after_transform) != 0), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(pr, (e << 10) | k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// thread 0, when the outputs are final (the step that sets `done`, or the statistics-only step behind the last sweep):
// the host-visible copy (LoopState::mirror), then its ready word with a system-scope release -- once per alignment
__device__ __forceinline__ void publish_result(LoopState* __restrict__ st) {
  LoopState* __restrict__ m = st->mirror;
  if (!m) return;
  m->done = st->done;
  m->iterations = st->iterations;
  m->status = st->status;
  m->sweeps = st->sweeps;
  m->pairs = st->pairs;
  m->mse = st->mse;
  for (int k = 0; k < 9; ++k) m->Trot[k] = st->Trot[k];
  for (int k = 0; k < 3; ++k) m->offset[k] = st->offset[k];
  for (int k = 0; k < 12; ++k) m->Tk[k] = st->Tk[k];
  __hip_atomic_store(st->progress + 2, (st->epoch << 1) | 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int NS, int NACT = NS>
__device__ __forceinline__ void loop_step_body(const double* __restrict__ partial, const int* __restrict__ pcount,
                                               int nblocks, LoopState* __restrict__ st, int stats_only) {
  // the control words are fetched together with the partial sums (independent loads in
  // flight at once) and only then acted upon
  const int done = st->done, stop_after = st->stop_after_transform, i = st->iterations;
  const int max_iterations = st->max_iterations, min_pairs = st->min_pairs, fixed = st->fixed_iterations;
  const float threshold = st->threshold;
  STEP_STAMP(i, 0);
  double sums[NS];
  long long npairs = 0;
  if (nblocks < 0) {
    // query-sharded loop: `partial` holds the sums of ALL ranks, already reduced (this rank's canonical tree, then one
    // float64 all-reduce over the ranks): NS doubles + the pair count as a double
    if (threadIdx.x == 0) {
#pragma unroll
      for (int s = 0; s < NS; ++s) sums[s] = partial[s];
      npairs = (long long)(partial[NS] + 0.5);
    }
  } else {
    tree_stage2<NS, NACT>(partial, pcount, nblocks, sums, npairs);  // sums [NACT..NS) read as 0
  }
  STEP_STAMP(i, 1);
  if (done) {
    if (threadIdx.x == 0) publish_progress(st);
    return;
  }
  if (stop_after) {  // the fallback motion has been applied by the previous transform
    if (threadIdx.x == 0) {
      st->done = 1;
      publish_result(st);
      publish_progress(st);
    }
    return;
  }
  if (threadIdx.x != 0) return;

  // statistics of the sweep just reduced (icp.cpp:622-638 from the double sum)
  float mse = 0.f;
  if (npairs > 0) {
    const float m = (float)(sums[NS == NP2L ? 27 : 12] / (double)npairs);
    mse = (float)((double)m * (double)m);
  }
  st->pairs = npairs;
  st->mse = mse;
  if (stats_only) {
    publish_result(st);
    return;
  }

  if (!((fixed || mse > threshold) && i < max_iterations)) {  // icp.cpp:155
    st->done = 1;
    publish_result(st);
    publish_progress(st);
    return;
  }
  if (npairs < min_pairs) {  // icp.cpp:163-182
    for (int k = 0; k < 9; ++k) st->rt.R[k] = st->last_rotation[k];
    for (int k = 0; k < 3; ++k) {
      st->rt.t[k] = st->last_translation[k];
      st->offset[k] = -st->last_translation[k];
    }
    for (int k = 0; k < 9; ++k) st->Rd[k] = (double)st->rt.R[k];
    st->status = 1;  // ICPK_W_TOO_FEW_PAIRS
    st->stop_after_transform = 1;
    publish_progress(st);
    return;
  }
  st->trace_pairs[i] = (int)npairs;
  st->trace_mse[i] = mse;
  LoopState* __restrict__ mir = st->mirror;
  if (mir) {
    mir->trace_pairs[i] = (int)npairs;
    mir->trace_mse[i] = mse;
  }
  float Rrec[9], trec[3];
  STEP_STAMP(i, 2);
  if (NS == NP2L) {
    double Rd[9], td[3];
    if (!solve_p2l(sums, Rd, td)) {
      st->status = 2;  // ICPK_W_DEGENERATE
      st->done = 1;
      publish_result(st);
      publish_progress(st);
      return;
    }
    for (int k = 0; k < 9; ++k) Rrec[k] = (float)Rd[k];
    for (int k = 0; k < 3; ++k) trec[k] = (float)td[k];
    for (int k = 0; k < 9; ++k) st->rt.R[k] = Rrec[k];
    for (int k = 0; k < 3; ++k) st->rt.t[k] = trec[k];
    compose_rt(Rrec, trec, st->Tk);
  } else if (st->solve == 0) {  // ICPK_SOLVE_REFERENCE
    float M[9], Rinv[9];
    for (int k = 0; k < 9; ++k) M[k] = (float)sums[k];  // icp.cpp:212
    solve_reference(M, Rrec);                           // icp.cpp:215-223
    STEP_STAMP(i, 5);
    if (i == 0) {
      for (int k = 0; k < 9; ++k) st->Trot[k] = Rrec[k];  // icp.cpp:227-229
    } else {
      float tmp[9], cur[9];
      for (int k = 0; k < 9; ++k) cur[k] = st->Trot[k];
      mul3f(Rrec, cur, tmp);  // icp.cpp:231-232
      for (int k = 0; k < 9; ++k) st->Trot[k] = tmp[k];
    }
    invert3f(Rrec, Rinv);  // icp.cpp:235
    for (int k = 0; k < 3; ++k) {
      trec[k] = (float)(sums[9 + k] / (double)npairs);  // icp.cpp:240
      st->offset[k] = trec[k];
      st->rt.t[k] = -trec[k];  // icp.cpp:245
    }
    for (int k = 0; k < 9; ++k) st->rt.R[k] = Rinv[k];
  } else {  // ICPK_SOLVE_KABSCH
    double sa[3], sb[3], sab[9], Rd[9], td[3];
    for (int k = 0; k < 3; ++k) {
      sa[k] = sums[13 + k];
      sb[k] = sums[16 + k];
    }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) sab[3 * r + c] = sums[3 * c + r];
    solve_kabsch(npairs, sa, sb, sab, Rd, td);
    for (int k = 0; k < 9; ++k) Rrec[k] = (float)Rd[k];
    for (int k = 0; k < 3; ++k) trec[k] = (float)td[k];
    for (int k = 0; k < 9; ++k) st->rt.R[k] = Rrec[k];
    for (int k = 0; k < 3; ++k) st->rt.t[k] = trec[k];
    compose_rt(Rrec, trec, st->Tk);
  }
  STEP_STAMP(i, 3);
  for (int k = 0; k < 9; ++k) st->Rd[k] = (double)st->rt.R[k];
  for (int k = 0; k < 9; ++k) st->trace_R[9 * i + k] = Rrec[k];
  for (int k = 0; k < 3; ++k) st->trace_t[3 * i + k] = trec[k];
  if (mir) {
    for (int k = 0; k < 9; ++k) mir->trace_R[9 * i + k] = Rrec[k];
    for (int k = 0; k < 3; ++k) mir->trace_t[3 * i + k] = trec[k];
  }
  st->iterations = i + 1;  // icp.cpp:257 (the sweep that follows is already enqueued)
  publish_progress(st);
  STEP_STAMP(i, 4);
}

template <int NS, int NACT = NS>
__global__ __launch_bounds__(256) void loop_step_kernel(const double* __restrict__ partial,
                                                        const int* __restrict__ pcount, int nblocks,
                                                        LoopState* __restrict__ st, int stats_only) {
  loop_step_body<NS, NACT>(partial, pcount, nblocks, st, stats_only);
}

// frame-batch mode: one workgroup per pair
template <int NS, int NACT = NS>
__global__ __launch_bounds__(256) void loop_step_batch_kernel(const StepBatch b, int stats_only) {
  const StepArgs& a = b.p[blockIdx.x];
  loop_step_body<NS, NACT>(a.partial, a.pcount, a.nblocks, a.st, stats_only);
}

__global__ __launch_bounds__(64) void loop_init_kernel(const LoopInitArgs a) { loop_init_body(a); }
__global__ __launch_bounds__(64) void loop_init_batch_kernel(const SetupBatchOf<LoopInitArgs> b) { loop_init_body(b.p[blockIdx.x]); }

void launch_loop_init(const LoopInitArgs& a, hipStream_t s) {
  if (SetupRecorder* r = setup_recorder()) {
    if (r->n < SETUP_MAX_CALLS) {
      r->calls[r->n].kind = SK_LOOP_INIT;
      r->calls[r->n++].loop_init = a;
    } else {
      r->overflow = true;
    }
    return;
  }
  hipLaunchKernelGGL(loop_init_kernel, dim3(1), dim3(64), 0, s, a);
}
void launch_loop_init_batch(const SetupBatchOf<LoopInitArgs>& b, int count, hipStream_t s) {
  if (count > 0) hipLaunchKernelGGL(loop_init_batch_kernel, dim3(count), dim3(64), 0, s, b);
}

void launch_loop_step(const double* partial, const int* pcount, int nblocks, int nsum, LoopState* st, int stats_only,
                      hipStream_t s) {
  if (nsum == NP2L)
    hipLaunchKernelGGL(loop_step_kernel<NP2L>, dim3(1), dim3(256), 0, s, partial, pcount, nblocks, st, stats_only);
  else if (nsum == NSUM_REF)  // reference flavour: the reduction produced sums [0..12] only
    hipLaunchKernelGGL((loop_step_kernel<NSUM, NSUM_REF>), dim3(1), dim3(256), 0, s, partial, pcount, nblocks, st,
                       stats_only);
  else
    hipLaunchKernelGGL(loop_step_kernel<NSUM>, dim3(1), dim3(256), 0, s, partial, pcount, nblocks, st, stats_only);
}

void launch_loop_step_batch(const StepBatch& b, int count, int nsum, int stats_only, hipStream_t s) {
  if (count <= 0) return;
  if (nsum == NSUM_REF)
    hipLaunchKernelGGL((loop_step_batch_kernel<NSUM, NSUM_REF>), dim3(count), dim3(256), 0, s, b, stats_only);
  else
    hipLaunchKernelGGL(loop_step_batch_kernel<NSUM>, dim3(count), dim3(256), 0, s, b, stats_only);
}

// K3 with the transform taken from the loop state (same arithmetic as transform_kernel)
__device__ __forceinline__ float rot_row_s(float r0, float r1, float r2, float x, float y, float z) {
  return (float)__builtin_fma((double)r2, (double)z, __builtin_fma((double)r1, (double)y, (double)r0 * (double)x));
}

__global__ __launch_bounds__(256) void transform_state_kernel(float* __restrict__ x, float* __restrict__ y,
                                                              float* __restrict__ z, int n4,
                                                              const LoopState* __restrict__ st) {
  if (st->done) return;  // stop_after_transform: this launch still applies the fallback motion
  const Rt rt = st->rt;
  float4* x4 = reinterpret_cast<float4*>(x);
  float4* y4 = reinterpret_cast<float4*>(y);
  float4* z4 = reinterpret_cast<float4*>(z);
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 X = x4[i], Y = y4[i], Z = z4[i];
    float4 ox, oy, oz;
#define ICPK_ROW(c)                                                               \
  ox.c = rot_row_s(rt.R[0], rt.R[1], rt.R[2], X.c, Y.c, Z.c) + rt.t[0];           \
  oy.c = rot_row_s(rt.R[3], rt.R[4], rt.R[5], X.c, Y.c, Z.c) + rt.t[1];           \
  oz.c = rot_row_s(rt.R[6], rt.R[7], rt.R[8], X.c, Y.c, Z.c) + rt.t[2];
    ICPK_ROW(x) ICPK_ROW(y) ICPK_ROW(z) ICPK_ROW(w)
#undef ICPK_ROW
    x4[i] = ox;
    y4[i] = oy;
    z4[i] = oz;
  }
}

void launch_transform_state(float* x, float* y, float* z, int n, const LoopState* st, hipStream_t s) {
  if (n <= 0) return;
  const int n4 = (n + 3) / 4;
  int blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(transform_state_kernel, dim3(blocks), dim3(256), 0, s, x, y, z, n4, st);
}

}  // namespace icpk
