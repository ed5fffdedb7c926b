// loop_init.h -- the initial LoopState, shared by kernels_loop.hip (loop_init_kernel) and kernels_grid.hip
// (grid_begin_kernel: the first launch of a fresh frame pair's set-up does it on the side).
#pragma once
#include <cstddef>

#include "icpk_internal.h"

namespace icpk {

// initial LoopState (everything in front of the trace arrays) from values that travel in the kernel
// arguments: no staging copy, and one launch for a whole lock-step group
__device__ __forceinline__ void loop_init_body(const LoopInitArgs& a) {
  constexpr int WORDS = (int)(offsetof(LoopState, trace_R) / 4);
  int* w = reinterpret_cast<int*>(a.st);
  for (int i = threadIdx.x; i < WORDS; i += blockDim.x) w[i] = 0;
  __syncthreads();
  if (threadIdx.x != 0) return;
  LoopState* st = a.st;
  st->Trot[0] = st->Trot[4] = st->Trot[8] = 1.f;
  st->Tk[0] = st->Tk[5] = st->Tk[10] = 1.0;
  st->max_iterations = a.max_iterations;
  st->min_pairs = a.min_pairs;
  st->solve = a.solve;
  st->fixed_iterations = a.fixed_iterations;
  st->threshold = a.threshold;
  st->epoch = a.epoch;
  st->progress = a.progress;
  st->mirror = a.mirror;
  for (int k = 0; k < 9; ++k) st->last_rotation[k] = a.last_rotation[k];
  for (int k = 0; k < 3; ++k) st->last_translation[k] = a.last_translation[k];
}

}  // namespace icpk
