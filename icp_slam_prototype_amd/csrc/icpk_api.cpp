// icpk_api.cpp -- host side of libicpk.so: context, device buffers, the ICP loop
// (icp.cpp:98-268 in its frame-pair formulation) and the C ABI of include/icpk.h.
//
// One context = one GPU + one HIP stream.  Per iteration the host sees exactly
// one small device->host copy (19 sums + count, 160 bytes, pinned) and sends the
// next 3x4 transform as kernel arguments.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "icpk.h"
#include "icpk_ctx.h"
#include "icpk_internal.h"
#include "solve_impl.h"

using namespace icpk;

namespace {

int fail(icpk_ctx* ctx, int code, const char* msg) {
  if (ctx) ctx->err = msg;
  return code;
}

int ensure_cloud(icpk_ctx* ctx, Cloud& c, int n) {
  if (ctx && (&c == &ctx->src0 || &c == &ctx->src)) ctx->src_pristine = false;  // (about to be resized or rewritten)
  if (ctx && &c == &ctx->src) ctx->rec_pending = false;  // (whatever was to be unpacked into it is superseded)
  if (ctx && (&c == &ctx->src0 || &c == &ctx->tgt)) ctx->have_pix_seed = false;  // (other points than the pixel maps describe)
  const int cap = round_up(n < 1 ? 1 : n, NN_TILE);
  if (cap > c.cap) {
    if (c.base) ICPK_HIP(ctx, hipFree(c.base));
    c.base = nullptr;
    c.cap = 0;
    // +64 floats: the filtered NN kernel prefetches one group past its chunk
    ICPK_HIP(ctx, hipMalloc((void**)&c.base, ((size_t)3 * cap + 64) * sizeof(float)));
    c.cap = cap;
  }
  c.n = n;
  return ICPK_OK;
}

int ensure_assoc(icpk_ctx* ctx, int nq) {
  const int cap = round_up(nq < 1 ? 1 : nq, NN_TILE);
  if (cap > ctx->assoc_cap) {
    if (ctx->best) ICPK_HIP(ctx, hipFree(ctx->best));
    if (ctx->seed) ICPK_HIP(ctx, hipFree(ctx->seed));
    if (ctx->best_m) ICPK_HIP(ctx, hipFree(ctx->best_m));
    if (ctx->seed_m) ICPK_HIP(ctx, hipFree(ctx->seed_m));
    ctx->best_m = ctx->seed_m = nullptr;
    if (ctx->idx) ICPK_HIP(ctx, hipFree(ctx->idx));
    if (ctx->dist) ICPK_HIP(ctx, hipFree(ctx->dist));
    ctx->best = ctx->seed = nullptr;
    ctx->idx = nullptr;
    ctx->dist = nullptr;
    ctx->assoc_cap = 0;
    ctx->have_seed = false;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->best, (size_t)cap * sizeof(nn_key_t)));
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->seed, (size_t)cap * sizeof(nn_key_t)));
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->best_m, (size_t)cap * sizeof(nn_key_t)));
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->seed_m, (size_t)cap * sizeof(nn_key_t)));
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->idx, (size_t)cap * sizeof(int32_t)));
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->dist, (size_t)cap * sizeof(float)));
    ctx->assoc_cap = cap;
  }
  return ICPK_OK;
}

// pad value: +inf for targets (a padded target is never nearer than a real
// one), 0 for sources (padded queries are computed and discarded)
int upload_cloud(icpk_ctx* ctx, Cloud& c, const float* x, const float* y, const float* z, int n, float pad,
                 hipMemcpyKind kind, bool sync = true) {
  if (n < 0 || (n > 0 && (!x || !y || !z))) return fail(ctx, ICPK_E_ARG, "bad cloud pointers/size");
  int rc = ensure_cloud(ctx, c, n);
  if (rc) return rc;
  if (n > 0 && kind == hipMemcpyHostToDevice && !sync) {
    // frame-batch slots: pageable host memory crosses PCIe through a staging copy inside the
    // runtime, one plane after the other and synchronously; copying into the slot's own pinned
    // buffer here (the set-up threads do it in parallel) leaves ONE truly asynchronous DMA per cloud
    float*& stage = (&c == &ctx->tgt) ? ctx->stage_t : ctx->stage_s;
    int& cap = (&c == &ctx->tgt) ? ctx->stage_t_cap : ctx->stage_s_cap;
    if (c.cap > cap) {
      if (stage) ICPK_HIP(ctx, hipHostFree(stage));
      stage = nullptr;
      cap = 0;
      ICPK_HIP(ctx, hipHostMalloc((void**)&stage, (size_t)3 * c.cap * sizeof(float), hipHostMallocDefault));
      cap = c.cap;
    }
    std::memcpy(stage, x, (size_t)n * sizeof(float));
    std::memcpy(stage + c.cap, y, (size_t)n * sizeof(float));
    std::memcpy(stage + 2 * (size_t)c.cap, z, (size_t)n * sizeof(float));
    // (the planes sit at the device cloud's own stride: one contiguous copy; the tails are padded below)
    ICPK_HIP(ctx, hipMemcpyAsync(c.base, stage, ((size_t)2 * c.cap + n) * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  } else if (n > 0) {
    ICPK_HIP(ctx, hipMemcpyAsync(c.x(), x, (size_t)n * sizeof(float), kind, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(c.y(), y, (size_t)n * sizeof(float), kind, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(c.z(), z, (size_t)n * sizeof(float), kind, ctx->stream));
  }
  launch_fill_f32(c.x() + n, c.cap - n, pad, ctx->stream);
  launch_fill_f32(c.y() + n, c.cap - n, pad, ctx->stream);
  launch_fill_f32(c.z() + n, c.cap - n, pad, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  // host buffers are only valid for the duration of the call (the frame-batch entry points
  // keep theirs alive until they return and synchronise once at the end)
  if (sync) ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ICPK_OK;
}

int copy_src0_to_src(icpk_ctx* ctx) {
  // src_pristine: the working copy is known to hold the committed source already (icpk_backproject_pair writes both
  // at once; nothing has touched either since) -- the frame path's icpk_align starts without this copy
  if (ctx->src_pristine && ctx->pristine_skip && ctx->src.n == ctx->src0.n) return ICPK_OK;
  ctx->rec_pending = false;  // (the working source is overwritten: nothing of the last loop's is wanted any more)
  int rc = ensure_cloud(ctx, ctx->src, ctx->src0.n);
  if (rc) return rc;
  const Cloud &a = ctx->src0, &b = ctx->src;
  if (a.cap == b.cap) {
    ICPK_HIP(ctx, hipMemcpyAsync(b.base, a.base, (size_t)3 * a.cap * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  } else {  // capacities can differ after a shrink: copy plane by plane, padded part included
    const int m = round_up(a.n < 1 ? 1 : a.n, NN_TILE);
    ICPK_HIP(ctx, hipMemcpyAsync(b.x(), a.x(), (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(b.y(), a.y(), (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(b.z(), a.z(), (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  }
  ctx->src_pristine = true;
  return ICPK_OK;
}

hipEvent_t get_event(icpk_ctx* ctx, size_t k) {
  while (ctx->events.size() <= k) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    ctx->events.push_back(e);
  }
  return ctx->events[k];
}

void log_delta(icpk_ctx* ctx, int key, int quantity) {
  if (!ctx->log_fn) return;
  const auto now = std::chrono::steady_clock::now();
  const double us = std::chrono::duration<double, std::micro>(now - ctx->log_last).count();
  ctx->log_last = now;
  ctx->log_fn(key, quantity, us, ctx->log_user);
}

int check_ready(icpk_ctx* ctx) {
  if (!ctx) return ICPK_E_ARG;
  if (!ctx->have_tgt || !ctx->have_src) return fail(ctx, ICPK_E_NOT_SET, "source or target cloud not set");
  if (ctx->tgt.n <= 0) return fail(ctx, ICPK_E_EMPTY_TARGET, "target cloud is empty");
  return ICPK_OK;
}

int ensure_sort_buffers(icpk_ctx* ctx, int n) {
  const int cap = round_up(n < 1 ? 1 : n, NN_TILE);
  if (cap > ctx->sort_cap) {
    if (ctx->sort_keys) ICPK_HIP(ctx, hipFree(ctx->sort_keys));
    if (ctx->sort_vals) ICPK_HIP(ctx, hipFree(ctx->sort_vals));
    ctx->sort_keys = nullptr;
    ctx->sort_vals = nullptr;
    ctx->sort_cap = 0;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->sort_keys, (size_t)2 * cap * sizeof(unsigned)));
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->sort_vals, (size_t)cap * sizeof(int)));
    ctx->sort_cap = cap;
  }
  if (!ctx->bounds) ICPK_HIP(ctx, hipMalloc((void**)&ctx->bounds, 6 * sizeof(float)));
  return ICPK_OK;
}

int ensure_scan_buffers(icpk_ctx* ctx);

// Morton order of `c` (cells of the target's bounding box) -> perm_out[k] = index of the k-th point; the sorted keys
// land in sort_keys[sort_cap ...), the unsorted ones stay in sort_keys[0 ... n)
int enqueue_morton_order(icpk_ctx* ctx, const Cloud& c, int* perm_out) {
  int rc = ensure_sort_buffers(ctx, c.n);
  if (rc) return rc;
  rc = ensure_scan_buffers(ctx);
  if (rc) return rc;
  if (!ctx->morton_table) ICPK_HIP(ctx, hipMalloc((void**)&ctx->morton_table, sizeof(GridInfo)));
  unsigned* ka = ctx->sort_keys;
  unsigned* kb = ctx->sort_keys + ctx->sort_cap;
  const int bits = ctx->grid_max_cells >= (1 << 21) + 1 ? 7 : 6;  // 8^bits cells + 1 bin + 1 must fit the count table
  ctx->qcount_dirty = true;
  launch_morton_order(c.x(), c.y(), c.z(), c.n, ctx->bounds, bits, ka, ctx->sort_vals, ctx->qcount, ctx->qstart, ctx->scan_bsum,
                      ctx->morton_table, kb, perm_out, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ctx->qcount_dirty = false;
  return ICPK_OK;
}

// boxes + Morton-ordered target for the pruned scan (once per target cloud)
int prepare_pruned_target(icpk_ctx* ctx, NnBoxes& bx) {
  const int nt = ctx->tgt.n;
  const int nt_pad = round_up(nt, NN_TILE);
  const int ntiles = nt_pad / NN_TILE;
  if (ntiles > ctx->boxes_tiles_cap) {
    if (ctx->boxes) ICPK_HIP(ctx, hipFree(ctx->boxes));
    ctx->boxes = nullptr;
    ctx->boxes_tiles_cap = 0;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->boxes, (size_t)6 * (ntiles + 16) * (1 + NN_SUBS) * sizeof(float)));
    ctx->boxes_tiles_cap = ntiles;
    ctx->have_boxes = false;
  }
  if (nt_pad > ctx->tperm_cap) {
    if (ctx->tperm) ICPK_HIP(ctx, hipFree(ctx->tperm));
    ctx->tperm = nullptr;
    ctx->tperm_cap = 0;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->tperm, ((size_t)nt_pad + 64) * sizeof(int)));
    if (ctx->tkeys) ICPK_HIP(ctx, hipFree(ctx->tkeys));
    ctx->tkeys = nullptr;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->tkeys, ((size_t)nt_pad + 64) * sizeof(unsigned)));
    ctx->tperm_cap = nt_pad;
    ctx->have_boxes = false;
  }
  bx.tbox_stride = ctx->boxes_tiles_cap + 16;
  bx.sbox_stride = (ctx->boxes_tiles_cap + 16) * NN_SUBS;
  bx.tbox = ctx->boxes;
  bx.sbox = ctx->boxes + (size_t)6 * bx.tbox_stride;
  bx.ox = ctx->tgt.x();
  bx.oy = ctx->tgt.y();
  bx.oz = ctx->tgt.z();
  bx.tperm = ctx->tperm;
  bx.qperm = ctx->qperm;
  if (ctx->have_boxes) return ICPK_OK;
  int rc = ensure_cloud(ctx, ctx->sorted, nt);
  if (rc) return rc;
  rc = ensure_sort_buffers(ctx, nt);
  if (rc) return rc;
  // bounds of the cloud from boxes of the caller's order, then sort, gather, final boxes
  launch_tile_boxes(ctx->tgt.x(), ctx->tgt.y(), ctx->tgt.z(), nt, ntiles, bx, ctx->stream);
  launch_bounds(bx.tbox, bx.tbox_stride, ntiles, ctx->bounds, ctx->stream);
  rc = enqueue_morton_order(ctx, ctx->tgt, ctx->tperm);
  if (rc) return rc;
  ICPK_HIP(ctx, hipMemcpyAsync(ctx->tkeys, ctx->sort_keys + ctx->sort_cap, (size_t)nt * sizeof(unsigned),
                               hipMemcpyDeviceToDevice, ctx->stream));
  launch_gather_planes(ctx->tgt.x(), ctx->tgt.y(), ctx->tgt.z(), ctx->tperm, nt, nt_pad, __builtin_inff(),
                       ctx->sorted.x(), ctx->sorted.y(), ctx->sorted.z(), ctx->tperm, ctx->stream);
  launch_tile_boxes(ctx->sorted.x(), ctx->sorted.y(), ctx->sorted.z(), nt, ntiles, bx, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ctx->have_boxes = true;
  return ICPK_OK;
}

// counts / starts of the counting sorts by cell (targets: cell_start; queries: qstart).  The count table is
// all zero between two sorts (the scan hands it back zeroed); a sort that did not get as far as its scan --
// a failed launch -- leaves it marked dirty, and the next one clears all of it first.
int ensure_scan_buffers(icpk_ctx* ctx) {
  const size_t bytes = ((size_t)ctx->grid_max_cells + 1) * sizeof(int);
  if (!ctx->qcount) {
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->qcount, bytes));
    ctx->qcount_dirty = true;
  }
  if (ctx->qcount_dirty) {
    ICPK_HIP(ctx, hipMemsetAsync(ctx->qcount, 0, bytes, ctx->stream));
    ctx->qcount_dirty = false;
  }
  if (!ctx->qstart) ICPK_HIP(ctx, hipMalloc((void**)&ctx->qstart, bytes));
  if (!ctx->scan_bsum) ICPK_HIP(ctx, hipMalloc((void**)&ctx->scan_bsum, (size_t)GRID_SCAN_BLOCKS * sizeof(int)));
  return ICPK_OK;
}

// cell table + cell-sorted AoS copy of the target for the grid scan (once per target cloud).
// Everything is enqueued: the grid's size stays on the device (GridInfo), the counting sort's
// zero fill and scan read it there -- no host round trip, so the frame-batch mode can build the
// next group's grids in the shadow of the running loop.
int prepare_grid_target(icpk_ctx* ctx) {
  const int nt = ctx->tgt.n;
  if (nt > (1 << 28)) return fail(ctx, ICPK_E_ARG, "the grid search addresses its cell-sorted targets with 32-bit byte offsets: at most 2^28 target points");
  if (!ctx->grid_info) ICPK_HIP(ctx, hipMalloc((void**)&ctx->grid_info, sizeof(GridInfo)));
  if (!ctx->grid_bounds)
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->grid_bounds, (size_t)GRID_BOUNDS_PARTS * 6 * sizeof(float)));
  if (!ctx->cell_start) ICPK_HIP(ctx, hipMalloc((void**)&ctx->cell_start, ((size_t)ctx->grid_max_cells + 1) * sizeof(int)));
  if (nt > ctx->t4_cap) {
    if (ctx->t4) ICPK_HIP(ctx, hipFree(ctx->t4));
    if (ctx->o4) ICPK_HIP(ctx, hipFree(ctx->o4));
    ctx->t4 = ctx->o4 = nullptr;
    ctx->t4_cap = 0;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->t4, ((size_t)round_up(nt, NN_TILE) + 64) * sizeof(float4)));
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->o4, ((size_t)round_up(nt, NN_TILE) + 64) * sizeof(float4)));
    ctx->t4_cap = round_up(nt, NN_TILE);
    ctx->have_grid = false;
  }
  if (ctx->have_grid) return ICPK_OK;
  int rc = ensure_sort_buffers(ctx, nt);
  if (rc) return rc;
  rc = ensure_scan_buffers(ctx);
  if (rc) return rc;
  launch_grid_bounds(ctx->tgt.x(), ctx->tgt.y(), ctx->tgt.z(), nt, ctx->grid_bounds, ctx->stream);
  launch_grid_info(ctx->grid_bounds, nt, ctx->grid_ppc, ctx->grid_xdiv, ctx->grid_max_cells, ctx->grid_info, ctx->stream);
  // counting sort of the targets by cell: slot within the cell by atomics (the order inside a
  // cell is irrelevant: candidates are merged lexicographically), cell starts by an exclusive
  // scan of the counts (entry ncells = Nt), scatter into the AoS copy
  int* tcell = reinterpret_cast<int*>(ctx->sort_keys + ctx->sort_cap);
  int* tslot = ctx->sort_vals;
  ctx->qcount_dirty = true;
  launch_grid_qslot(ctx->tgt.x(), ctx->tgt.y(), ctx->tgt.z(), nt, ctx->grid_info, ctx->qcount, tcell, tslot, 0,
                    ctx->stream);
  launch_grid_scan(ctx->qcount, ctx->cell_start, ctx->scan_bsum, ctx->grid_info, 0, ctx->stream);
  launch_grid_tscatter(ctx->tgt.x(), ctx->tgt.y(), ctx->tgt.z(), tcell, tslot, ctx->cell_start, nt, ctx->t4, ctx->o4,
                       ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ctx->qcount_dirty = false;
  ctx->have_grid = true;
  return ICPK_OK;
}

// query order for the grid scan: counting sort of the source by cell of the target's grid.
// with_points: the scatter also writes the scan-order queries and element 0 as everybody's seed
// (the first sweep of an alignment that has no seeds)
int enqueue_cell_order(icpk_ctx* ctx, bool with_points) {
  const int nq = ctx->src.n;
  int rc = ensure_sort_buffers(ctx, nq);
  if (rc) return rc;
  rc = ensure_scan_buffers(ctx);
  if (rc) return rc;
  int* qcell = reinterpret_cast<int*>(ctx->sort_keys + ctx->sort_cap);
  int* qslot = ctx->sort_vals;
  // (locality only: the coarser table, xdiv times fewer counts to scan)
  ctx->qcount_dirty = true;
  launch_grid_qslot(ctx->src.x(), ctx->src.y(), ctx->src.z(), nq, ctx->grid_info, ctx->qcount, qcell, qslot, 1,
                    ctx->stream);
  launch_grid_scan(ctx->qcount, ctx->qstart, ctx->scan_bsum, ctx->grid_info, 1, ctx->stream);
  launch_grid_qscatter(qcell, qslot, ctx->qstart, nq, ctx->qperm, ctx->src.x(), ctx->src.y(), ctx->src.z(), ctx->tgt.x(),
                       ctx->tgt.y(), ctx->tgt.z(), with_points ? ctx->qm4 : nullptr, ctx->sp_in, ctx->seed_m, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ctx->qcount_dirty = false;
  return ICPK_OK;
}

int ensure_query_points(icpk_ctx* ctx, int nq);

// After a device loop of grid sweeps the caller-order views -- the moved source planes, the association keys -- exist
// only as records (ctx->rec, kernels_grid.hip); they are unpacked when something asks for them (icpk_get_source,
// icpk_get_associations, icpk_commit_source, icpk_transform_source, icpk_nn, icpk_reduce ...) and dropped when the
// working source is overwritten first (the next alignment, a new source): a tracker that only wants the pose never
// pays the launch.  ICPK_LAZY_UNPACK=0: unpack at the end of every loop.
int ensure_unpacked(icpk_ctx* ctx) {
  if (!ctx->rec_pending) return ICPK_OK;
  ctx->rec_pending = false;
  launch_grid_unpack(ctx->qm4, ctx->rec, ctx->src.n, ctx->src.x(), ctx->src.y(), ctx->src.z(), ctx->best, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  return ICPK_OK;
}

// the deferred initial LoopState of a device loop (device_loop_begin), if no set-up launch has carried it
int flush_loop_init(icpk_ctx* ctx) {
  if (!ctx->init_pending) return ICPK_OK;
  ctx->init_pending = false;
  launch_loop_init(ctx->pending_init, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  return ICPK_OK;
}

// A FRESH pair (new target, new source, no seeds: every frame of the drop-in path): the target's grid and the
// query order in 6 launches instead of 10 -- the two counting sorts run side by side (cell slots of both clouds in one
// launch, both scans in two, both scatters in one), each with its own count table.  Same kernels' bodies as
// prepare_grid_target + enqueue_cell_order: same tables, same copies.  ~5 us of launch latency per launch saved on a
// path that is a chain of tiny dependent kernels.
int build_grid_and_order(icpk_ctx* ctx) {
  const int nt = ctx->tgt.n, nq = ctx->src.n;
  if (nt > (1 << 28)) return fail(ctx, ICPK_E_ARG, "the grid search addresses its cell-sorted targets with 32-bit byte offsets: at most 2^28 target points");
  if (!ctx->grid_info) ICPK_HIP(ctx, hipMalloc((void**)&ctx->grid_info, sizeof(GridInfo)));
  if (!ctx->grid_bounds)
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->grid_bounds, (size_t)GRID_BOUNDS_PARTS * 6 * sizeof(float)));
  if (!ctx->cell_start) ICPK_HIP(ctx, hipMalloc((void**)&ctx->cell_start, ((size_t)ctx->grid_max_cells + 1) * sizeof(int)));
  if (nt > ctx->t4_cap) {
    if (ctx->t4) ICPK_HIP(ctx, hipFree(ctx->t4));
    if (ctx->o4) ICPK_HIP(ctx, hipFree(ctx->o4));
    ctx->t4 = ctx->o4 = nullptr;
    ctx->t4_cap = 0;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->t4, ((size_t)round_up(nt, NN_TILE) + 64) * sizeof(float4)));
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->o4, ((size_t)round_up(nt, NN_TILE) + 64) * sizeof(float4)));
    ctx->t4_cap = round_up(nt, NN_TILE);
  }
  int rc = ensure_sort_buffers(ctx, nq > nt ? nq : nt);
  if (rc) return rc;
  rc = ensure_scan_buffers(ctx);
  if (rc) return rc;
  const size_t table = ((size_t)ctx->grid_max_cells + 1) * sizeof(int);
  if (!ctx->qcount2) {
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->qcount2, table));
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->scan_bsum2, (size_t)GRID_SCAN_BLOCKS * sizeof(int)));
    ctx->qcount2_dirty = true;
  }
  if (ctx->qcount2_dirty) {  // (first use, or a sort that was cut short: the scans hand the table back zeroed otherwise)
    ICPK_HIP(ctx, hipMemsetAsync(ctx->qcount2, 0, table, ctx->stream));
    ctx->qcount2_dirty = false;
  }
  if (ctx->sort_cap > ctx->sort_vals2_cap) {
    if (ctx->sort_vals2) ICPK_HIP(ctx, hipFree(ctx->sort_vals2));
    ctx->sort_vals2 = nullptr;
    ctx->sort_vals2_cap = 0;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->sort_vals2, (size_t)ctx->sort_cap * sizeof(int)));
    ctx->sort_vals2_cap = ctx->sort_cap;
  }
  int* tcell = reinterpret_cast<int*>(ctx->sort_keys + ctx->sort_cap);
  int* tslot = ctx->sort_vals;
  int* qcell = reinterpret_cast<int*>(ctx->sort_keys);
  int* qslot = ctx->sort_vals2;
  if (!ctx->grid_ticket) {
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->grid_ticket, sizeof(int)));
    ICPK_HIP(ctx, hipMemsetAsync(ctx->grid_ticket, 0, sizeof(int), ctx->stream));
  }
  // bounds + geometry (+ the pending initial LoopState of the alignment being enqueued) in ONE launch
  launch_grid_begin(ctx->tgt.x(), ctx->tgt.y(), ctx->tgt.z(), nt, ctx->grid_bounds, ctx->grid_ppc, ctx->grid_xdiv,
                    ctx->grid_max_cells, ctx->grid_info, ctx->init_pending ? &ctx->pending_init : nullptr, ctx->grid_ticket,
                    ctx->stream);
  ctx->init_pending = false;
  ctx->qcount_dirty = ctx->qcount2_dirty = true;
  SetupBatchOf<QslotArgs> qb{};
  qb.p[0] = QslotArgs{ctx->tgt.x(), ctx->tgt.y(), ctx->tgt.z(), ctx->grid_info, ctx->qcount, tcell, tslot, nt, 0};
  qb.p[1] = QslotArgs{ctx->src.x(), ctx->src.y(), ctx->src.z(), ctx->grid_info, ctx->qcount2, qcell, qslot, nq, 1};
  // A source that icpk_backproject_pair has just made is in row-major IMAGE order: eight consecutive points are a short
  // run of one image row, as close together as a grid cell's -- the queries are swept in the caller's order and their
  // counting sort is left out (-3.5 us per 92k-point pair, -14 us at 306k; ICPK_IMAGE_ORDER=0: sort them all the same).
  const bool ident = ctx->have_pix_seed && ctx->image_order;
  launch_grid_qslot_batch(qb, ident ? 1 : 2, ctx->stream);
  SetupBatchOf<ScanArgs> sb{};
  sb.p[0] = ScanArgs{ctx->qcount, ctx->cell_start, ctx->scan_bsum, ctx->grid_info, 0, 0};
  sb.p[1] = ScanArgs{ctx->qcount2, ctx->qstart, ctx->scan_bsum2, ctx->grid_info, 1, 0};
  launch_grid_scan_batch(sb, ident ? 1 : 2, ctx->stream);
  const TscatterArgs ta{ctx->tgt.x(), ctx->tgt.y(), ctx->tgt.z(), tcell, tslot, ctx->cell_start, ctx->t4, ctx->o4, nt, 0};
  const bool pix = ctx->have_pix_seed && ctx->pixel_seeds;
  const QscatterArgs qa{ident ? nullptr : qcell,        qslot,        ctx->qstart,  ctx->qperm,   ctx->src.x(), ctx->src.y(), ctx->src.z(), ctx->tgt.x(),
                        ctx->tgt.y(), ctx->tgt.z(), ctx->qm4,     ctx->sp_in,   ctx->seed_m,  nq,           0,
                        pix ? ctx->pix_src : nullptr, pix ? ctx->pix_tidx : nullptr, ctx->pix_rows, ctx->pix_cols};
  launch_grid_tqscatter(ta, qa, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ctx->qcount_dirty = ctx->qcount2_dirty = false;
  ctx->have_grid = true;
  return ICPK_OK;
}

// scan-order copies of the queries and of their seed points (grid scan)
int ensure_query_points(icpk_ctx* ctx, int nq) {
  if (nq <= ctx->qm4_cap) return ICPK_OK;
  for (float4** pp : {&ctx->qm4, &ctx->sp_in, &ctx->sp_out, &ctx->rec}) {
    if (*pp) ICPK_HIP(ctx, hipFree(*pp));
    *pp = nullptr;
  }
  ctx->qm4_cap = 0;
  const size_t bytes = ((size_t)round_up(nq, NN_TILE) + 64) * sizeof(float4);
  ICPK_HIP(ctx, hipMalloc((void**)&ctx->qm4, bytes));
  ICPK_HIP(ctx, hipMalloc((void**)&ctx->sp_in, bytes));
  ICPK_HIP(ctx, hipMalloc((void**)&ctx->sp_out, bytes));
  ICPK_HIP(ctx, hipMalloc((void**)&ctx->rec, 2 * bytes));
  ctx->qm4_cap = round_up(nq, NN_TILE);
  ctx->grid_chain = false;
  return ICPK_OK;
}

// Everything a pruned / grid sweep needs before its K1 launch: query order (once per alignment),
// seeds in scan order, buffer rotation.  Enqueues on ctx->stream only for the FIRST sweep of a
// chain; for a sweep that continues a chain of grid sweeps inside a device loop it merely
// rotates pointers.  recheck = 1: the seeds are loose (first sweep).
int prepare_sorted_sweep(icpk_ctx* ctx, int nn_mode, NnArgs& a, NnBoxes& bx, int& recheck) {
  const int nq = ctx->src.n;
  int rc = ICPK_OK;
  if (round_up(nq, NN_TILE) > ctx->qperm_cap) {
    if (ctx->qperm) ICPK_HIP(ctx, hipFree(ctx->qperm));
    ctx->qperm = nullptr;
    ctx->qperm_cap = 0;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->qperm, (size_t)round_up(nq, NN_TILE) * sizeof(int)));
    ctx->qperm_cap = round_up(nq, NN_TILE);
    ctx->have_qperm = false;
  }
  bx = NnBoxes{};
  bool fresh = false;
  if (nn_mode == ICPK_NN_GRID) {
    bx.ox = ctx->tgt.x();
    bx.oy = ctx->tgt.y();
    bx.oz = ctx->tgt.z();
    // (one size for both counting sorts up front: the target's sort must not see its scratch re-allocated by
    // the queries' -- its launches may only have been recorded so far, see SetupRecorder)
    rc = ensure_sort_buffers(ctx, nq > ctx->tgt.n ? nq : ctx->tgt.n);
    // a fresh pair (no grid yet, no seeds, launches not being recorded for a lock-step group): both sorts side by side
    fresh = !rc && !ctx->have_grid && !ctx->have_seed && !setup_recorder() && ctx->merged_setup && ctx->tgt.n > 0;
    if (fresh) {
      rc = ensure_query_points(ctx, nq);
      if (!rc) rc = build_grid_and_order(ctx);
    } else {
      if (!rc) rc = prepare_grid_target(ctx);
      if (!rc) rc = ensure_query_points(ctx, nq);
    }
  } else {
    rc = prepare_pruned_target(ctx, bx);
  }
  if (rc) return rc;
  bool new_order = false;
  bool points_written = false;  // qm4 / sp_in / seed_m already hold this sweep's queries and seeds
  const int want_kind = nn_mode == ICPK_NN_GRID ? 2 : 1;
  if (fresh) {  // order, scan-order queries and literal seeds were written by build_grid_and_order
    ctx->have_qperm = true;
    ctx->qperm_kind = want_kind;
    new_order = true;
    points_written = true;
  } else if (!ctx->have_qperm || !ctx->have_seed || ctx->qperm_kind != want_kind) {
    // query order (once per alignment), from the source at its current pose: Morton order
    // for the pruned scan (the unsorted Morton keys of the queries stay in sort_keys[0..nq)
    // for its first-sweep seeds), order by grid cell (a cheaper counting sort) for the grid scan
    if (nn_mode == ICPK_NN_GRID) {
      points_written = !ctx->have_seed;
      rc = enqueue_cell_order(ctx, points_written);
    } else {
      rc = enqueue_morton_order(ctx, ctx->src, ctx->qperm);
    }
    if (rc) return rc;
    ctx->have_qperm = true;
    ctx->qperm_kind = want_kind;
    new_order = true;
  }
  recheck = 0;
  if (ctx->have_seed && ctx->have_seed_m && !new_order) {
    // matches of the previous pruned sweep, already in query Morton order
    std::swap(ctx->seed_m, ctx->best_m);
    std::swap(ctx->seed, ctx->best);
  } else if (ctx->have_seed) {  // matches of a sweep by another kernel: bring them into Morton order
    launch_seed_gather(ctx->best, ctx->qperm, nq, ctx->seed_m, ctx->stream);
    std::swap(ctx->seed, ctx->best);
  } else if (nn_mode == ICPK_NN_GRID) {
    // first sweep of the grid scan: the reference's own literal seed, element 0 (icp.cpp:572);
    // the expanding search does not depend on the seed's quality
    if (!points_written) launch_fill_u64(ctx->seed_m, nq, 0ull, nullptr, ctx->stream);
    recheck = 1;
  } else {  // first sweep: the target with the nearest Morton code; loose, so re-check lazily
    launch_seed_morton(ctx->sort_keys, ctx->qperm, nq, ctx->src.x(), ctx->src.y(), ctx->src.z(), ctx->tkeys,
                       ctx->sorted.x(), ctx->sorted.y(), ctx->sorted.z(), ctx->tperm, ctx->tgt.n, ctx->seed_m,
                       ctx->stream);
    recheck = 1;
  }
  a.tx = ctx->sorted.x();
  a.ty = ctx->sorted.y();
  a.tz = ctx->sorted.z();
  a.tiles_per_chunk = a.nt_pad / NN_TILE;
  a.best = ctx->best;
  if (nn_mode == ICPK_NN_GRID) {
    // inside a device loop the grid sweeps keep qm4 / the seed points current themselves;
    // anywhere else the source may have been moved by other kernels: gather afresh
    if (!(ctx->st_active && ctx->grid_chain) && !points_written)
      launch_grid_query_points(ctx->src.x(), ctx->src.y(), ctx->src.z(), ctx->qperm, nq, ctx->seed_m, bx.ox, bx.oy,
                               bx.oz, ctx->qm4, ctx->sp_in, ctx->stream);
  }
  ICPK_HIP(ctx, hipGetLastError());
  return ICPK_OK;
}

// lanes per query of the grid scan (measured with cells 4x finer along x: 8 is best up to 217k queries, 4 from
// 307k on; a launch that fills the GPU several times over is issue-bound and prefers fewer, longer lanes)
int grid_slices_for(const icpk_ctx* ctx, int nq) { return ctx->grid_slices ? ctx->grid_slices : (nq > 524288 ? 4 : 8); }

// the K1d arguments of the sweep prepare_sorted_sweep has just set up, and the bookkeeping
// that follows its launch
GridSweepArgs grid_sweep_args(icpk_ctx* ctx, const NnArgs& a, const NnBoxes& bx) {
  GridSweepArgs g{};
  g.qx = const_cast<float*>(a.qx);
  g.qy = const_cast<float*>(a.qy);
  g.qz = const_cast<float*>(a.qz);
  g.nq = a.nq;
  g.qm4 = ctx->qm4;
  g.t4 = ctx->t4;
  g.cell_start = ctx->cell_start;
  g.gi = ctx->grid_info;
  g.ox = bx.ox;
  g.oy = bx.oy;
  g.oz = bx.oz;
  g.sp_in = ctx->sp_in;
  g.sp_out = ctx->sp_out;
  g.best = a.best;
  g.best_m = ctx->best_m;
  g.st = ctx->st_active;
  g.rec = ctx->st_active ? ctx->rec : nullptr;  // inside a device loop K2 reads the records; planes / keys once at the end
  return g;
}
void after_grid_sweep(icpk_ctx* ctx) {
  std::swap(ctx->sp_in, ctx->sp_out);
  ctx->grid_chain = ctx->st_active != nullptr;
  ctx->have_assoc = true;
  ctx->have_seed = true;
  ctx->have_seed_m = true;
}

NnArgs base_nn_args(const icpk_ctx* ctx) {
  NnArgs a;
  a.qx = ctx->src.x();
  a.qy = ctx->src.y();
  a.qz = ctx->src.z();
  a.nq = ctx->src.n;
  a.tx = ctx->tgt.x();
  a.ty = ctx->tgt.y();
  a.tz = ctx->tgt.z();
  a.nt_pad = round_up(ctx->tgt.n, NN_TILE);
  a.tiles_per_chunk = a.nt_pad / NN_TILE;
  a.best = ctx->best;
  a.stop = ctx->stop;
  return a;
}

// enqueue one NN sweep (K1) over the working source; ev0/ev1 (optional) are recorded
// immediately before/after the K1 launch itself, so that set-up kernels of a first sweep
// (sort, seeding, fills) do not count as kernel time
int enqueue_nn(icpk_ctx* ctx, int nn_mode, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
  auto mark = [&](hipEvent_t e) -> int {
    if (e) ICPK_HIP(ctx, hipEventRecord(e, ctx->stream));
    return ICPK_OK;
  };
  if (nn_mode != ICPK_NN_EXACT && nn_mode != ICPK_NN_FILTERED && nn_mode != ICPK_NN_PRUNED && nn_mode != ICPK_NN_GRID)
    return fail(ctx, ICPK_E_ARG, "unknown nn_mode");
  const int nq = ctx->src.n;
  int rc = ensure_assoc(ctx, nq);
  if (rc) return rc;
  if (nq == 0) {
    ctx->have_assoc = true;
    return ICPK_OK;
  }
  auto chunking = [&](int nqb, int ntiles) {
    int nchunks = (ctx->target_blocks + nqb - 1) / nqb;
    if (nchunks < 1) nchunks = 1;
    if (nchunks > ntiles) nchunks = ntiles;
    return (ntiles + nchunks - 1) / nchunks;
  };
  NnArgs a = base_nn_args(ctx);
  const int ntiles = a.nt_pad / NN_TILE;
  if (nn_mode != ICPK_NN_GRID && (rc = flush_loop_init(ctx))) return rc;  // (their fills and kernels look at the state)
  if (nn_mode == ICPK_NN_EXACT) {
    a.tiles_per_chunk = chunking((nq + NN_THREADS - 1) / NN_THREADS, ntiles);
    a.best = ctx->best;
    launch_fill_u64(ctx->best, nq, NN_KEY_INIT, ctx->stop, ctx->stream);
    if ((rc = mark(ev0))) return rc;
    launch_nn_exact(a, ctx->stream);
    if ((rc = mark(ev1))) return rc;
  } else if (nn_mode == ICPK_NN_PRUNED || nn_mode == ICPK_NN_GRID) {
    NnBoxes bx{};
    int recheck = 0;
    rc = prepare_sorted_sweep(ctx, nn_mode, a, bx, recheck);
    if (rc) return rc;
    if ((rc = flush_loop_init(ctx))) return rc;  // (unless the set-up's first launch has carried it)
    if ((rc = mark(ev0))) return rc;
    if (nn_mode == ICPK_NN_GRID) {
      launch_nn_grid(grid_sweep_args(ctx, a, bx), grid_slices_for(ctx, nq), recheck, ctx->stream);
      after_grid_sweep(ctx);
    } else {
      // lanes per query: as many as keep the launch at <= ~10k waves (measured best: 16 at 10k
      // queries, 4 at 92k, 2 at 217k-307k, 1 at 10^6)
      int slices = ctx->slices;
      if (slices == 0) {
        slices = 16;
        while (slices > 1 && (long long)nq * slices / 64 > 10000) slices >>= 1;
      }
      launch_nn_pruned(a, ctx->seed_m, ctx->best_m, bx, slices, recheck, ctx->st_active, ctx->stream);
      ctx->grid_chain = false;
      ctx->have_assoc = true;
      ctx->have_seed = true;
      ctx->have_seed_m = true;
    }
    if ((rc = mark(ev1))) return rc;
    ICPK_HIP(ctx, hipGetLastError());
    return ICPK_OK;
  } else {
    int seed_scale = 1;
    if (ctx->have_seed) {
      // matches of the previous sweep (same clouds, source possibly moved) seed this one
      std::swap(ctx->seed, ctx->best);
    } else {
      // coarse pre-pass: exact NN against every NN_SEED_STRIDE-th target
      if (!ctx->have_dec) {
        const int nd = (ctx->tgt.n + NN_SEED_STRIDE - 1) / NN_SEED_STRIDE;
        rc = ensure_cloud(ctx, ctx->dec, nd);
        if (rc) return rc;
        launch_decimate(ctx->tgt.x(), ctx->tgt.y(), ctx->tgt.z(), ctx->tgt.n, NN_SEED_STRIDE, ctx->dec.x(),
                        ctx->dec.y(), ctx->dec.z(), round_up(nd, NN_TILE), ctx->stream);
        ctx->have_dec = true;
      }
      NnArgs c = a;
      c.tx = ctx->dec.x();
      c.ty = ctx->dec.y();
      c.tz = ctx->dec.z();
      c.nt_pad = round_up(ctx->dec.n, NN_TILE);
      c.tiles_per_chunk = chunking((nq + NN_THREADS - 1) / NN_THREADS, c.nt_pad / NN_TILE);
      c.best = ctx->seed;
      launch_fill_u64(ctx->seed, nq, NN_KEY_INIT, ctx->stop, ctx->stream);
      launch_nn_exact(c, ctx->stream);
      seed_scale = NN_SEED_STRIDE;
    }
    const int q = ctx->q_per_lane > 0 ? ctx->q_per_lane : (nq >= 65536 ? 2 : 1);
    a.tiles_per_chunk = chunking((nq + NN_THREADS * q - 1) / (NN_THREADS * q), ntiles);
    a.best = ctx->best;
    launch_fill_u64(ctx->best, nq, NN_KEY_INIT, ctx->stop, ctx->stream);
    if ((rc = mark(ev0))) return rc;
    launch_nn_filtered(a, ctx->seed, seed_scale, q, ctx->stream);
    if ((rc = mark(ev1))) return rc;
  }
  ICPK_HIP(ctx, hipGetLastError());
  ctx->have_assoc = true;
  ctx->have_seed = true;
  ctx->have_seed_m = false;  // exact / filtered sweeps leave their matches in the caller's order only
  return ICPK_OK;
}

// the records of the grid sweep just enqueued, if it was one of a device loop (then planes and keys are stale until
// the loop's final unpack)
const float4* loop_rec(const icpk_ctx* ctx) { return ctx->st_active && ctx->grid_chain ? ctx->rec : nullptr; }

// enqueue K2 and the 160-byte read-back; caller synchronises
int enqueue_reduce(icpk_ctx* ctx, float max_dist) {
  const int nq = ctx->src.n;
  launch_assoc_reduce(ctx->best, ctx->src.x(), ctx->src.y(), ctx->src.z(), nq, ctx->tgt.x(), ctx->tgt.y(), ctx->tgt.z(),
                      ctx->have_grid ? ctx->o4 : nullptr, loop_rec(ctx), max_dist, ctx->st_active ? nullptr : ctx->idx, ctx->st_active ? nullptr : ctx->dist, ctx->partial,
                      ctx->pcount, ctx->st_active ? nullptr : ctx->red_out,
                      ctx->st_active, ctx->st_active ? ctx->loop_nact : NSUM, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  if (!ctx->st_active)
    ICPK_HIP(ctx, hipMemcpyAsync(ctx->red_host, ctx->red_out, (NSUM + 1) * sizeof(double), hipMemcpyDeviceToHost,
                                 ctx->stream));
  return ICPK_OK;
}

// point-to-plane flavour of enqueue_reduce (K5)
int enqueue_reduce_p2l(icpk_ctx* ctx, float max_dist) {
  launch_p2l_reduce(ctx->best, ctx->src.x(), ctx->src.y(), ctx->src.z(), ctx->src.n, ctx->tgt.x(), ctx->tgt.y(),
                    ctx->tgt.z(), ctx->nrm.x(), ctx->nrm.y(), ctx->nrm.z(), loop_rec(ctx), max_dist, ctx->st_active ? nullptr : ctx->idx,
                    ctx->st_active ? nullptr : ctx->dist, ctx->partial,
                    ctx->pcount, ctx->st_active ? nullptr : ctx->red_out, ctx->st_active, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  if (!ctx->st_active)
    ICPK_HIP(ctx, hipMemcpyAsync(ctx->red_host, ctx->red_out, (NP2L + 1) * sizeof(double), hipMemcpyDeviceToHost,
                                 ctx->stream));
  return ICPK_OK;
}

float mse_from(const double* sums, int64_t n) {
  // icp.cpp:622-638: (mean distance)^2, evaluated from the double sum
  if (n <= 0) return 0.f;
  const float m = (float)(sums[12] / (double)n);
  return (float)((double)m * (double)m);
}

}  // namespace

int icpk_host_fail(icpk_ctx* ctx, int code, const char* msg) { return fail(ctx, code, msg); }
int icpk_host_ensure_cloud(icpk_ctx* ctx, icpk::Cloud& c, int n) { return ensure_cloud(ctx, c, n); }
int icpk_host_target_replaced(icpk_ctx* ctx) {
  Cloud& c = ctx->tgt;
  const int padded = round_up(c.n < 1 ? 1 : c.n, NN_TILE);
  launch_fill_f32(c.x() + c.n, padded - c.n, __builtin_inff(), ctx->stream);
  launch_fill_f32(c.y() + c.n, padded - c.n, __builtin_inff(), ctx->stream);
  launch_fill_f32(c.z() + c.n, padded - c.n, __builtin_inff(), ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ctx->have_tgt = true;
  ctx->have_assoc = false;
  ctx->have_dec = false;
  ctx->have_boxes = false;
  ctx->have_grid = false;
  ctx->have_seed = false;
  ctx->have_normals = false;
  return ICPK_OK;
}

// ---- deferred set-up launches (icpk_internal.h) ----
namespace icpk {
SetupRecorder*& setup_recorder() {
  static thread_local SetupRecorder* rec = nullptr;
  return rec;
}

// every pair of the group recorded the same steps in the same order?
static bool same_sequences(const SetupRecorder* recs, int count) {
  for (int k = 0; k < count; ++k) {
    if (recs[k].overflow || recs[k].n != recs[0].n) return false;
    for (int j = 0; j < recs[0].n; ++j)
      if (recs[k].calls[j].kind != recs[0].calls[j].kind) return false;
  }
  return true;
}

bool flush_setup_batches(const SetupRecorder* recs, int count, hipStream_t s) {
  if (count <= 0) return true;
  if (count > BATCH_MAX || !same_sequences(recs, count)) return false;
  for (int j = 0; j < recs[0].n; ++j) {
#define ICPK_STEP(KIND, FIELD, TYPE, LAUNCH)                        \
  case KIND: {                                                      \
    SetupBatchOf<TYPE> b;                                           \
    for (int k = 0; k < count; ++k) b.p[k] = recs[k].calls[j].FIELD; \
    LAUNCH(b, count, s);                                            \
    break;                                                          \
  }
    switch (recs[0].calls[j].kind) {
      ICPK_STEP(SK_INGEST, ingest, IngestArgs, launch_ingest_batch)
      ICPK_STEP(SK_LOOP_INIT, loop_init, LoopInitArgs, launch_loop_init_batch)
      ICPK_STEP(SK_BOUNDS, bounds, BoundsArgs, launch_grid_bounds_batch)
      ICPK_STEP(SK_INFO, info, InfoArgs, launch_grid_info_batch)
      ICPK_STEP(SK_QSLOT, qslot, QslotArgs, launch_grid_qslot_batch)
      ICPK_STEP(SK_SCAN, scan, ScanArgs, launch_grid_scan_batch)
      ICPK_STEP(SK_TSCATTER, tscatter, TscatterArgs, launch_grid_tscatter_batch)
      ICPK_STEP(SK_QSCATTER, qscatter, QscatterArgs, launch_grid_qscatter_batch)
      default: return false;
    }
#undef ICPK_STEP
  }
  return true;
}

// one pair's recorded steps, launched one by one (the sequences of a group differed)
void replay_setup(const SetupRecorder& rec, hipStream_t s) {
  SetupRecorder* const saved = setup_recorder();
  setup_recorder() = nullptr;
  for (int j = 0; j < rec.n; ++j) {
    const SetupCall& c = rec.calls[j];
    switch (c.kind) {
      case SK_INGEST: {
        const IngestArgs& a = c.ingest;
        launch_ingest_cloud(a.x, a.y, a.z, a.n, a.n_pad, a.pad, a.d1, a.cap1, a.d2, a.cap2, s);
        break;
      }
      case SK_LOOP_INIT: launch_loop_init(c.loop_init, s); break;
      case SK_BOUNDS: launch_grid_bounds(c.bounds.x, c.bounds.y, c.bounds.z, c.bounds.n, c.bounds.fb, s); break;
      case SK_INFO: launch_grid_info(c.info.fb, c.info.n, c.info.ppc, c.info.xdiv, c.info.max_cells, c.info.g, s); break;
      case SK_QSLOT: {
        const QslotArgs& a = c.qslot;
        launch_grid_qslot(a.x, a.y, a.z, a.n, a.gi, a.count, a.cell, a.slot, a.coarse, s);
        break;
      }
      case SK_SCAN: launch_grid_scan(c.scan.count, c.scan.out, c.scan.bsum, c.scan.g, c.scan.coarse, s); break;
      case SK_TSCATTER: {
        const TscatterArgs& a = c.tscatter;
        launch_grid_tscatter(a.x, a.y, a.z, a.tcell, a.tslot, a.cell_start, a.n, a.t4, a.o4, s);
        break;
      }
      case SK_QSCATTER: {
        const QscatterArgs& a = c.qscatter;
        launch_grid_qscatter(a.qcell, a.qslot, a.qstart, a.n, a.qperm, a.qx, a.qy, a.qz, a.ox, a.oy, a.oz, a.qm4, a.sp,
                             a.seed_m, s);
        break;
      }
      default: break;
    }
  }
  setup_recorder() = saved;
}
}  // namespace icpk

extern "C" {

const char* icpk_version(void) { return ICPK_VERSION_STRING; }

void icpk_default_params(icpk_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->max_iterations = ICPK_DEFAULT_MAX_ITERATIONS;
  p->threshold = ICPK_DEFAULT_THRESHOLD;
  p->max_nn_dist = ICPK_MAX_NN_DISTANCE;
  p->min_pairs = ICPK_MIN_PAIRS;
  p->solve = ICPK_SOLVE_REFERENCE;
  p->nn_mode = ICPK_NN_GRID;  // same results as ICPK_NN_EXACT (tests), fastest
  p->last_rotation[0] = p->last_rotation[4] = p->last_rotation[8] = 1.f;
}

// stream + the fixed-size buffers every context owns; `parent` != nullptr: a frame-batch slot
// (inherits the tuning knobs)
static icpk_ctx* make_context(int device_id, const icpk_ctx* parent) {
  icpk_ctx* ctx = new icpk_ctx();
  ctx->device = device_id;
  bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipMalloc((void**)&ctx->partial, (size_t)RED_MAX_BLOCKS * NSUM_MAX * sizeof(double)) == hipSuccess;
  ok = ok && hipMalloc((void**)&ctx->pcount, (size_t)RED_MAX_BLOCKS * sizeof(int)) == hipSuccess;
  ok = ok && hipMalloc((void**)&ctx->red_out, (NSUM_MAX + 1) * sizeof(double)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&ctx->red_host, (NSUM_MAX + 1) * sizeof(double), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&ctx->bp_n_host, 2 * sizeof(int), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipMalloc((void**)&ctx->st_dev, sizeof(LoopState)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&ctx->st_host, sizeof(LoopState), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&ctx->progress, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
  ok = ok && hipHostGetDevicePointer((void**)&ctx->progress_dev, ctx->progress, 0) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&ctx->st_mirror, sizeof(LoopState), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
  ok = ok && hipHostGetDevicePointer((void**)&ctx->st_mirror_dev, ctx->st_mirror, 0) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->ready_ev, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->group_ev[0], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ctx->group_ev[1], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    icpk_destroy(ctx);
    return nullptr;
  }
  if (parent) {
    ctx->target_blocks = parent->target_blocks;
    ctx->slices = parent->slices;
    ctx->grid_ppc = parent->grid_ppc;
    ctx->grid_xdiv = parent->grid_xdiv;
    ctx->grid_slices = parent->grid_slices;
    ctx->grid_max_cells = GRID_MAX_CELLS_SLOT;
    ctx->q_per_lane = parent->q_per_lane;
  }
  ctx->log_last = std::chrono::steady_clock::now();
  return ctx;
}

int icpk_create(icpk_ctx** out, int device_id) {
  if (!out) return ICPK_E_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ICPK_E_NO_DEVICE;
  if (device_id < 0 || device_id >= ndev) return ICPK_E_NO_DEVICE;
  if (hipSetDevice(device_id) != hipSuccess) return ICPK_E_NO_DEVICE;
  icpk_ctx* ctx = make_context(device_id, nullptr);
  if (!ctx) return ICPK_E_HIP;
  if (const char* e = std::getenv("ICPK_NN_TARGET_BLOCKS")) {
    const int v = std::atoi(e);
    if (v > 0) ctx->target_blocks = v;
  }
  if (const char* e = std::getenv("ICPK_NN_SLICES")) {
    const int v = std::atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) ctx->slices = v;
  }
  if (const char* e = std::getenv("ICPK_GRID_PPC")) {
    const float v = (float)std::atof(e);
    if (v > 0.f) ctx->grid_ppc = v;
  }
  if (const char* e = std::getenv("ICPK_GRID_XDIV")) {
    const int v = std::atoi(e);
    if (v >= 1 && v <= 64) ctx->grid_xdiv = v;
  }
  if (const char* e = std::getenv("ICPK_GRID_SLICES")) {
    const int v = std::atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8) ctx->grid_slices = v;
  }
  if (const char* e = std::getenv("ICPK_MERGED_SETUP")) ctx->merged_setup = std::atoi(e);  // 0: the two sorts of a fresh pair one after the other
  if (const char* e = std::getenv("ICPK_NN_Q")) {
    const int v = std::atoi(e);
    if (v == 1 || v == 2) ctx->q_per_lane = v;
  }
  if (const char* e = std::getenv("ICPK_BATCH_GROUP")) {
    const int v = std::atoi(e);
    if (v >= 1 && v <= BATCH_MAX) ctx->batch_group = v;
  }
  if (const char* e = std::getenv("ICPK_BATCH_THREADS")) {
    const int v = std::atoi(e);
    if (v >= 1 && v <= 16) ctx->batch_threads = v;
  }
  if (const char* e = std::getenv("ICPK_BATCH_SETUP")) ctx->batch_setup = std::atoi(e);  // 0: per-pair launches; 2: batched launches for single-group host-pointer batches too; 3: as 2, replayed pair by pair
  if (const char* e = std::getenv("ICPK_PRISTINE_SKIP")) ctx->pristine_skip = std::atoi(e) != 0;
  if (const char* e = std::getenv("ICPK_PIXEL_SEEDS")) ctx->pixel_seeds = std::atoi(e) != 0;
  if (const char* e = std::getenv("ICPK_LAZY_UNPACK")) ctx->lazy_unpack = std::atoi(e) != 0;
  if (const char* e = std::getenv("ICPK_IMAGE_ORDER")) ctx->image_order = std::atoi(e) != 0;
  if (const char* e = std::getenv("ICPK_ZERO_COPY_UPLOAD")) ctx->zero_copy_upload = std::atoi(e) != 0;
  if (const char* e = std::getenv("ICPK_RESULT_MIRROR")) ctx->result_mirror = std::atoi(e) != 0;
  if (const char* e = std::getenv("ICPK_LOOP_AHEAD")) {  // 0: enqueue every iteration up front
    const int v = std::atoi(e);
    if (v >= 0 && v <= LOOP_MAX_ITER) ctx->loop_ahead = v;
  }
  *out = ctx;
  return ICPK_OK;
}

void icpk_destroy(icpk_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  icpk_comm_release(ctx);
  for (icpk_ctx* sl : ctx->slots) icpk_destroy(sl);
  ctx->slots.clear();
  for (hipStream_t st : {ctx->setup_stream[0], ctx->setup_stream[1]})
    if (st) (void)hipStreamDestroy(st);
  for (hipEvent_t e : {ctx->ready_ev, ctx->group_ev[0], ctx->group_ev[1], ctx->setup_ev[0], ctx->setup_ev[1], ctx->batch_t0[0],
                       ctx->batch_t0[1], ctx->batch_t1[0], ctx->batch_t1[1]})
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
  void* dev[] = {ctx->qcount, ctx->qstart, ctx->scan_bsum, ctx->qcount2, ctx->scan_bsum2, ctx->sort_vals2, ctx->qm4, ctx->sp_in, ctx->sp_out, ctx->rec, ctx->grid_info, ctx->grid_bounds, ctx->cell_start, ctx->t4, ctx->o4, ctx->best_m, ctx->seed_m, ctx->st_pooled ? nullptr : (void*)ctx->st_dev, ctx->sorted.base, ctx->tkeys, ctx->tperm, ctx->qperm, ctx->bounds, ctx->sort_keys, ctx->sort_vals, ctx->morton_table,
                 ctx->nrm.base, ctx->boxes, ctx->dec.base, ctx->tgt.base, ctx->src0.base, ctx->src.base, ctx->best,      ctx->seed,     ctx->idx,
                 ctx->dist,     ctx->partial,   ctx->pcount,   ctx->red_out,   ctx->depth_dev, ctx->depth_flt, ctx->ks_buf, ctx->bp_counts};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (ctx->red_host) (void)hipHostFree(ctx->red_host);
  if (ctx->stage_t) (void)hipHostFree(ctx->stage_t);
  if (ctx->stage_s) (void)hipHostFree(ctx->stage_s);
  if (ctx->bp_n_host) (void)hipHostFree(ctx->bp_n_host);
  if (ctx->st_host && !ctx->st_pooled) (void)hipHostFree(ctx->st_host);
  if (ctx->slot_states) (void)hipFree(ctx->slot_states);
  if (ctx->slot_states_host) (void)hipHostFree(ctx->slot_states_host);
  if (ctx->progress) (void)hipHostFree(ctx->progress);
  if (ctx->st_mirror) (void)hipHostFree(ctx->st_mirror);
  if (ctx->grid_ticket) (void)hipFree(ctx->grid_ticket);
  if (ctx->stage_depth) (void)hipHostFree(ctx->stage_depth);
  for (const icpk_ctx::HostRange& r : ctx->registered) (void)hipHostUnregister(const_cast<char*>(r.host));
  ctx->registered.clear();
  if (ctx->pix_tidx) (void)hipFree(ctx->pix_tidx);
  if (ctx->pix_src) (void)hipFree(ctx->pix_src);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* icpk_last_error(const icpk_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int icpk_set_log_callback(icpk_ctx* ctx, icpk_log_fn fn, void* user) {
  if (!ctx) return ICPK_E_ARG;
  ctx->log_fn = fn;
  ctx->log_user = user;
  ctx->log_last = std::chrono::steady_clock::now();
  return ICPK_OK;
}

void* icpk_stream(icpk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

static int set_target_impl(icpk_ctx* ctx, const float* x, const float* y, const float* z, int32_t n, hipMemcpyKind k,
                           bool sync = true) {
  if (!ctx) return ICPK_E_ARG;
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  int rc = upload_cloud(ctx, ctx->tgt, x, y, z, n, __builtin_inff(), k, sync);
  if (rc) return rc;
  ctx->have_tgt = true;
  ctx->have_assoc = false;
  ctx->have_dec = false;
  ctx->have_boxes = false;
  ctx->have_grid = false;
  ctx->have_seed = false;
  ctx->have_normals = false;
  return ICPK_OK;
}

static int set_source_impl(icpk_ctx* ctx, const float* x, const float* y, const float* z, int32_t n, hipMemcpyKind k,
                           bool sync = true) {
  if (!ctx) return ICPK_E_ARG;
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  int rc = upload_cloud(ctx, ctx->src0, x, y, z, n, 0.f, k, sync);
  if (rc) return rc;
  ctx->have_src = true;
  ctx->have_assoc = false;
  ctx->have_seed = false;
  ctx->have_qperm = false;
  // (no second wait: the host buffers have been consumed by upload_cloud; the device-side copy of the
  // working source is stream-ordered before anything that uses it)
  return copy_src0_to_src(ctx);
}

int icpk_set_target(icpk_ctx* ctx, const float* x, const float* y, const float* z, int32_t n) {
  return set_target_impl(ctx, x, y, z, n, hipMemcpyHostToDevice);
}
int icpk_set_source(icpk_ctx* ctx, const float* x, const float* y, const float* z, int32_t n) {
  return set_source_impl(ctx, x, y, z, n, hipMemcpyHostToDevice);
}
int icpk_set_target_device(icpk_ctx* ctx, const float* x, const float* y, const float* z, int32_t n) {
  return set_target_impl(ctx, x, y, z, n, hipMemcpyDeviceToDevice);
}
int icpk_set_source_device(icpk_ctx* ctx, const float* x, const float* y, const float* z, int32_t n) {
  return set_source_impl(ctx, x, y, z, n, hipMemcpyDeviceToDevice);
}

int icpk_reset_source(icpk_ctx* ctx) {
  if (!ctx) return ICPK_E_ARG;
  if (!ctx->have_src) return fail(ctx, ICPK_E_NOT_SET, "source cloud not set");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  int rc = copy_src0_to_src(ctx);
  if (rc) return rc;
  ctx->have_assoc = false;
  ctx->have_seed = false;
  ctx->have_qperm = false;
  return ICPK_OK;  // device-side copy, stream-ordered: no host wait
}

int icpk_commit_source(icpk_ctx* ctx) {
  if (!ctx) return ICPK_E_ARG;
  if (!ctx->have_src) return fail(ctx, ICPK_E_NOT_SET, "source cloud not set");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  if (int ru = ensure_unpacked(ctx)) return ru;
  const Cloud &a = ctx->src, &b = ctx->src0;  // src0.cap >= src.cap by construction
  const int m = round_up(a.n < 1 ? 1 : a.n, NN_TILE);
  ICPK_HIP(ctx, hipMemcpyAsync(b.x(), a.x(), (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  ICPK_HIP(ctx, hipMemcpyAsync(b.y(), a.y(), (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  ICPK_HIP(ctx, hipMemcpyAsync(b.z(), a.z(), (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  ctx->src_pristine = a.n == b.n;  // (the two copies are equal again)
  return ICPK_OK;  // stream-ordered: no host wait
}

int icpk_get_source(icpk_ctx* ctx, float* x, float* y, float* z) {
  if (!ctx || !x || !y || !z) return ICPK_E_ARG;
  if (!ctx->have_src) return fail(ctx, ICPK_E_NOT_SET, "source cloud not set");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  if (int ru = ensure_unpacked(ctx)) return ru;
  const size_t b = (size_t)ctx->src.n * sizeof(float);
  if (b) {
    ICPK_HIP(ctx, hipMemcpyAsync(x, ctx->src.x(), b, hipMemcpyDeviceToHost, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(y, ctx->src.y(), b, hipMemcpyDeviceToHost, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(z, ctx->src.z(), b, hipMemcpyDeviceToHost, ctx->stream));
  }
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ICPK_OK;
}

int icpk_get_target(icpk_ctx* ctx, float* x, float* y, float* z) {
  if (!ctx || !x || !y || !z) return ICPK_E_ARG;
  if (!ctx->have_tgt) return fail(ctx, ICPK_E_NOT_SET, "target cloud not set");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  const size_t b = (size_t)ctx->tgt.n * sizeof(float);
  if (b) {
    ICPK_HIP(ctx, hipMemcpyAsync(x, ctx->tgt.x(), b, hipMemcpyDeviceToHost, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(y, ctx->tgt.y(), b, hipMemcpyDeviceToHost, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(z, ctx->tgt.z(), b, hipMemcpyDeviceToHost, ctx->stream));
  }
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ICPK_OK;
}

int32_t icpk_source_size(const icpk_ctx* ctx) { return ctx && ctx->have_src ? ctx->src0.n : 0; }
int32_t icpk_target_size(const icpk_ctx* ctx) { return ctx && ctx->have_tgt ? ctx->tgt.n : 0; }

int icpk_get_associations(icpk_ctx* ctx, int32_t* idx_out, float* dist_out) {
  if (!ctx) return ICPK_E_ARG;
  if (!ctx->have_assoc) return fail(ctx, ICPK_E_NOT_SET, "no nearest-neighbour sweep has run");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  const int nq = ctx->src.n;
  {
    const int ru = ensure_unpacked(ctx);  // (after a device loop of grid sweeps; a frame-batch slot after its lock-step loop)
    if (ru) return ru;
  }
  if (nq > 0) {
    // K2 unpacks (distance, index) keys into the idx/dist planes
    int rc = enqueue_reduce(ctx, __builtin_inff());
    if (rc) return rc;
    if (idx_out)
      ICPK_HIP(ctx, hipMemcpyAsync(idx_out, ctx->idx, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (dist_out)
      ICPK_HIP(ctx, hipMemcpyAsync(dist_out, ctx->dist, (size_t)nq * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  }
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ICPK_OK;
}

int icpk_nn(icpk_ctx* ctx, int32_t nn_mode, int32_t* idx_out, float* dist_out) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  if (int ru = ensure_unpacked(ctx)) return ru;
  rc = enqueue_nn(ctx, nn_mode);
  if (rc) return rc;
  if (idx_out || dist_out) return icpk_get_associations(ctx, idx_out, dist_out);
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ICPK_OK;
}

int icpk_reduce(icpk_ctx* ctx, float max_dist, double* sums, int64_t* count) {
  if (!ctx || !sums) return ICPK_E_ARG;
  if (!ctx->have_assoc) return fail(ctx, ICPK_E_NOT_SET, "no nearest-neighbour sweep has run");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  if (int ru = ensure_unpacked(ctx)) return ru;
  if (ctx->src.n == 0) {
    std::memset(sums, 0, NSUM * sizeof(double));
    if (count) *count = 0;
    return ICPK_OK;
  }
  int rc = enqueue_reduce(ctx, max_dist);
  if (rc) return rc;
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(sums, ctx->red_host, NSUM * sizeof(double));
  if (count) std::memcpy(count, ctx->red_host + NSUM, sizeof(int64_t));
  return ICPK_OK;
}

int icpk_transform_source(icpk_ctx* ctx, const float R[9], const float t[3]) {
  if (!ctx || !R || !t) return ICPK_E_ARG;
  if (!ctx->have_src) return fail(ctx, ICPK_E_NOT_SET, "source cloud not set");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  if (int ru = ensure_unpacked(ctx)) return ru;
  Rt rt;
  std::memcpy(rt.R, R, sizeof(rt.R));
  std::memcpy(rt.t, t, sizeof(rt.t));
  ctx->src_pristine = false;
  launch_transform(ctx->src.x(), ctx->src.y(), ctx->src.z(), ctx->src.n, rt, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ctx->have_assoc = false;
  return ICPK_OK;  // (R and t travel by value in the kernel arguments: no host wait)
}

// ---- device-side loop: begin / finish, shared by the single-pair and the frame-batch path ----
// sums the loop step of this flavour consumes: the reference flavour reads [0..12] only
static int loop_nsum(const icpk_params* p) {
  return p->solve == ICPK_SOLVE_POINT_TO_PLANE ? NP2L : (p->solve == ICPK_SOLVE_REFERENCE ? NSUM_REF : NSUM);
}

// initial LoopState -> device (on ctx->stream), stop flags armed
// defer: the launch is left to the first set-up launch that can carry it (build_grid_and_order) or, failing that, to
// flush_loop_init right before the first kernel that reads the state
static int device_loop_begin(icpk_ctx* ctx, const icpk_params* p, bool throttled = false, bool mirror = false,
                             bool defer = false) {
  LoopInitArgs a{};
  a.st = ctx->st_dev;
  if (throttled || mirror) {
    ctx->loop_epoch = (ctx->loop_epoch % 1000000) + 1;  // (<< 10 must fit an int)
    a.epoch = ctx->loop_epoch;
    a.progress = ctx->progress_dev;
    a.mirror = mirror ? ctx->st_mirror_dev : nullptr;
  }
  a.max_iterations = p->max_iterations;
  a.min_pairs = p->min_pairs;
  a.solve = p->solve;
  a.fixed_iterations = p->fixed_iterations;
  a.threshold = p->threshold;
  std::memcpy(a.last_rotation, p->last_rotation, sizeof(a.last_rotation));
  std::memcpy(a.last_translation, p->last_translation, sizeof(a.last_translation));
  if (defer) {
    ctx->pending_init = a;
    ctx->init_pending = true;
  } else {
    launch_loop_init(a, ctx->stream);  // (values travel in the kernel arguments: no staging copy)
    ICPK_HIP(ctx, hipGetLastError());
  }
  const int nsum = loop_nsum(p);
  ctx->loop_nact = nsum == NSUM_REF ? NSUM_REF : NSUM;
  // the stop flags are only meaningful while this alignment is being enqueued
  ctx->stop = &ctx->st_dev->done;
  ctx->st_active = ctx->st_dev;
  ctx->grid_chain = false;
  ctx->best_of_sweep.clear();
  return ICPK_OK;
}

static void device_loop_disarm(icpk_ctx* ctx) {
  ctx->init_pending = false;
  ctx->stop = nullptr;
  ctx->st_active = nullptr;
  ctx->grid_chain = false;
}

// after the LoopState has landed in ctx->st_host: outputs of the alignment
static int device_loop_finish(icpk_ctx* ctx, const icpk_params* p, float T_out[16], icpk_stats* stats,
                              const LoopState* h = nullptr) {
  device_loop_disarm(ctx);
  if (!h) h = ctx->st_host;
  // the associations of the last EXECUTED sweep are the result
  const int k = h->sweeps;
  if (k >= 1 && k <= (int)ctx->best_of_sweep.size()) {
    nn_key_t* fin = ctx->best_of_sweep[k - 1];
    if (fin != ctx->best) {
      ctx->seed = ctx->best;
      ctx->best = fin;
    }
  }
  ctx->have_seed_m = false;  // the Morton-ordered copy may belong to a skipped sweep: re-gather on demand
  const int it = h->iterations;
  if (p->solve == ICPK_SOLVE_REFERENCE) {
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T_out[4 * r + c] = h->Trot[3 * r + c];
      T_out[4 * r + 3] = h->offset[r];  // icp.cpp:266-268
    }
  } else {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) T_out[4 * r + c] = (float)h->Tk[4 * r + c];
  }
  T_out[12] = T_out[13] = T_out[14] = 0.f;
  T_out[15] = 1.f;
  ctx->trace_R.assign(h->trace_R, h->trace_R + 9 * it);
  ctx->trace_t.assign(h->trace_t, h->trace_t + 3 * it);
  ctx->trace_mse.assign(h->trace_mse, h->trace_mse + it);
  ctx->trace_pairs.assign(h->trace_pairs, h->trace_pairs + it);
  if (stats) {
    stats->iterations = it;
    stats->status = h->status;
    stats->final_pairs = (int32_t)h->pairs;
    stats->final_mse = h->mse;
    stats->nn_launches = k;
  }
  return h->status;
}

// host side of LoopState::progress: returns once `steps` loop steps have run on the device or the loop has
// exited.  Spins (the wait is a fraction of one iteration), yields when a sweep is long, and looks at the
// stream now and then so that a faulted kernel ends the wait with an error instead of hanging the caller.
// steps < 0: returns once the loop's outputs have landed in ctx->st_mirror (progress word 2).
static int wait_loop_progress(icpk_ctx* ctx, int steps, bool* exited) {
  volatile int* pr = ctx->progress;
  const int e = ctx->loop_epoch;
  auto look = [&]() -> int {  // 1: the loop has exited, 2: `steps` steps have run, 0: neither yet
    if (steps < 0) return __atomic_load_n(&pr[2], __ATOMIC_ACQUIRE) == ((e << 1) | 1) ? 1 : 0;
    const int w0 = __atomic_load_n(&pr[0], __ATOMIC_ACQUIRE), w1 = pr[1];
    if ((w1 >> 2) == e && (w1 & 1)) return 1;
    return ((w0 >> 10) == e && (w0 & 1023) >= steps) ? 2 : 0;
  };
  auto t_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
  for (unsigned spin = 1;; ++spin) {
    const int got = look();
    if (got) {
      *exited = got == 1;
      return ICPK_OK;
    }
    __builtin_ia32_pause();
    if ((spin & 0x3ff) != 0) continue;
    std::this_thread::yield();
    const auto now = std::chrono::steady_clock::now();
    if (now < t_query) continue;
    // (not more often: a stream query may itself put a marker into the queue)
    t_query = now + std::chrono::milliseconds(20);
    const hipError_t q = hipStreamQuery(ctx->stream);
    if (q == hipSuccess) {  // drained: the words are final
      const int fin = look();
      if (!fin) return fail(ctx, ICPK_E_HIP, "device loop made no progress");
      *exited = fin == 1;
      return ICPK_OK;
    }
    if (q != hipErrorNotReady) return fail(ctx, ICPK_E_HIP, hipGetErrorString(q));
  }
}

// Whole alignment enqueued up front (or, when the loop may leave early, a few iterations ahead of the
// device); loop test, solve and pose accumulation run on the device (kernels_loop.hip).  Same results
// as the host loop below.
static int align_device_loop(icpk_ctx* ctx, const icpk_params* p, float T_out[16], icpk_stats* stats) {
  const bool prof = p->profile != 0;
  const bool prof_all = p->profile >= 2;  // 1: NN kernels only (2 events per sweep); 2: every stage
  const bool p2l = p->solve == ICPK_SOLVE_POINT_TO_PLANE;
  const bool fused = p->nn_mode == ICPK_NN_PRUNED || p->nn_mode == ICPK_NN_GRID;  // K3 runs inside the sweep
  const int nsum = loop_nsum(p);
  const int B = red_blocks(ctx->src.n);
  size_t nev = 0;
  std::vector<size_t> ev_nn, ev_red, ev_tr;
  auto stamp = [&](std::vector<size_t>* list) -> int {
    if (!prof) return ICPK_OK;
    hipEvent_t e = get_event(ctx, nev);
    if (!e) return fail(ctx, ICPK_E_HIP, "hipEventCreate failed");
    ICPK_HIP(ctx, hipEventRecord(e, ctx->stream));
    if (list) list->push_back(nev);
    ++nev;
    return ICPK_OK;
  };
  struct Guard {
    icpk_ctx* c;
    ~Guard() { device_loop_disarm(c); }
  } guard{ctx};
  // a loop that may leave early is enqueued loop_ahead iterations ahead of the device, not all at once
  const int ahead = ctx->loop_ahead;
  const bool throttled = !p->fixed_iterations && !prof && ahead > 0 && p->max_iterations > ahead;
  // the outputs come back through the host-visible mirror the last step writes (LoopState::mirror): no copy kernel,
  // and in a throttled loop no stream wait either -- the call returns when the deciding step has run
  const bool mirror = ctx->result_mirror && !prof;
  int rc = device_loop_begin(ctx, p, throttled, mirror, /*defer=*/p->nn_mode == ICPK_NN_GRID && !prof);
  if (rc) return rc;

  int nsweep = 0;
  const int phase = ctx->profile_phase++;  // successive alignments bracket different sweeps: unbiased sample
  // throttled loop: has the device loop exited (LoopState::progress word 1)?  A glance at pinned memory before every
  // launch: whatever would be enqueued after the exit is a no-op that still costs its dispatch (4-5 us each).
  auto gone = [&]() -> bool {
    if (!throttled) return false;
    const volatile int* pr = ctx->progress;
    const int w1 = pr[1];
    return (w1 >> 2) == ctx->loop_epoch && (w1 & 1);
  };
  auto sweep = [&]() -> int {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const int nth = nsweep++;
    if (prof && (prof_all || p->profile_stride <= 1 || (nth + phase) % p->profile_stride == 0)) {
      // two events tightly around the K1 launch
      e0 = get_event(ctx, nev);
      e1 = get_event(ctx, nev + 1);
      if (!e0 || !e1) return fail(ctx, ICPK_E_HIP, "hipEventCreate failed");
      ev_nn.push_back(nev);
      nev += 2;
    }
    int r = enqueue_nn(ctx, p->nn_mode, e0, e1);
    if (r) return r;
    if (!loop_rec(ctx)) ctx->best_of_sweep.push_back(ctx->best);  // (grid sweeps keep ONE set of records: a sweep that runs at all supersedes the previous one)
    if (gone()) return ICPK_OK;  // (the loop has exited meanwhile: K2 would be a no-op launch)
    if (prof_all) {
      r = stamp(&ev_red);
      if (r) return r;
    }
    r = p2l ? enqueue_reduce_p2l(ctx, p->max_nn_dist) : enqueue_reduce(ctx, p->max_nn_dist);
    if (r) return r;
    return prof_all ? stamp(nullptr) : ICPK_OK;
  };
  rc = sweep();  // icp.cpp:98
  if (rc) return rc;
  if ((rc = flush_loop_init(ctx))) return rc;  // (normally carried by the set-up or flushed before the sweep already)
  for (int i = 0; i < p->max_iterations; ++i) {
    if (throttled && i >= ahead) {
      bool exited = false;
      rc = wait_loop_progress(ctx, i - ahead + 1, &exited);
      if (rc) return rc;
      if (exited) break;  // everything from here on would find `done` set and do nothing
    }
    launch_loop_step(ctx->partial, ctx->pcount, B, nsum, ctx->st_dev, 0, ctx->stream);
    if (gone()) break;
    if (!fused) {  // the pruned sweep applies the transform itself (K3 fused into K1c)
      if (prof_all) {
        rc = stamp(&ev_tr);
        if (rc) return rc;
      }
      launch_transform_state(ctx->src.x(), ctx->src.y(), ctx->src.z(), ctx->src.n, ctx->st_dev, ctx->stream);
      if (prof_all) {
        rc = stamp(nullptr);
        if (rc) return rc;
      }
    }
    rc = sweep();  // icp.cpp:255
    if (rc) return rc;
  }
  // a step that set `done` has published the outputs already: the statistics-only step would be a no-op launch
  bool done_seen = false;
  if (throttled && mirror) {
    const int w1 = ((const volatile int*)ctx->progress)[1];
    done_seen = (w1 >> 2) == ctx->loop_epoch && (w1 & 2);
  }
  if (!done_seen) launch_loop_step(ctx->partial, ctx->pcount, B, nsum, ctx->st_dev, 1, ctx->stream);
  if (loop_rec(ctx)) {  // the caller-order planes and keys the grid sweeps did not keep current: once, and only if asked for
    ctx->rec_pending = true;
    if (!ctx->lazy_unpack && (rc = ensure_unpacked(ctx))) return rc;
  }
  ICPK_HIP(ctx, hipGetLastError());
  const LoopState* result = nullptr;
  if (mirror) {
    // a loop enqueued whole is waited for the same way when it is short (well under a millisecond of device time: the
    // host would otherwise sleep through the unpack and its own wake-up); long ones leave the core alone
    const bool brief = (long long)ctx->src.n * p->max_iterations <= 8000000ll;
    if (throttled || brief) {  // (what is still enqueued -- a no-op sweep, the unpack -- is stream-ordered before whatever comes next)
      bool ready = false;
      rc = wait_loop_progress(ctx, -1, &ready);
      if (rc) return rc;
    } else {
      ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (__atomic_load_n(&ctx->progress[2], __ATOMIC_ACQUIRE) != ((ctx->loop_epoch << 1) | 1))
        return fail(ctx, ICPK_E_HIP, "device loop ended without publishing its result");
    }
    result = ctx->st_mirror;
  } else {
    ICPK_HIP(ctx, hipMemcpyAsync(ctx->st_host, ctx->st_dev, sizeof(LoopState), hipMemcpyDeviceToHost, ctx->stream));
    ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }

  rc = device_loop_finish(ctx, p, T_out, stats, result);
  if (stats) {
    stats->nn_timed_launches = (int32_t)ev_nn.size();
    if (prof && nev >= 2) {
      auto span = [&](size_t a, size_t b) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ctx->events[a], ctx->events[b]);
        return ms;
      };
      for (size_t e : ev_nn) stats->nn_ms_total += span(e, e + 1);
      for (size_t e : ev_red) stats->reduce_ms_total += span(e, e + 1);
      for (size_t e : ev_tr) stats->transform_ms_total += span(e, e + 1);
      stats->total_ms = span(0, nev - 1);
    }
  }
  return rc;
}

// Query-sharded alignment of ONE pair over the ranks of the context's communicator ("single huge
// pair"; the frame-pair formulation icp.cpp:541-563 with the queries split): the target is the same on every rank
// (icpk_comm_broadcast_target), the source is this rank's slice of the queries.  The whole loop is enqueued: per
// iteration the grid sweep (K3 fused) and K2 on the slice, the canonical second tree stage, ONE in-stream float64
// all-reduce of the 19 sums + the pair count (160 bytes), and the loop step on the reduced sums -- replicated, so
// every rank applies the same transform, takes the same exit and returns the same T.  No host round trip and no
// host copy per iteration (the host-driven loop of round 2 paid a stream sync + two staging copies each).
// Results agree with icpk_align on the whole pair to ~1e-6 on T (the sums of the ranks are added by the collective:
// another order than the single-GPU canonical tree); with one rank they are bit-identical.
int icpk_align_query_sharded(icpk_ctx* ctx, const icpk_params* p, float T_out[16], icpk_stats* stats) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (!p || !T_out) return ICPK_E_ARG;
  if (!ctx->comm) return fail(ctx, ICPK_E_NOT_SET, "icpk_comm_init_rccl has not been called");
  if (p->solve != ICPK_SOLVE_REFERENCE && p->solve != ICPK_SOLVE_KABSCH)
    return fail(ctx, ICPK_E_ARG, "the query-sharded loop supports the reference and Kabsch flavours");
  if (p->max_iterations < 0 || p->max_iterations > LOOP_MAX_ITER) return fail(ctx, ICPK_E_ARG, "max_iterations out of range");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  for (int k = 0; k < 16; ++k) T_out[k] = (k % 5 == 0) ? 1.f : 0.f;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  icpk_params q = *p;
  q.nn_mode = ICPK_NN_GRID;
  rc = copy_src0_to_src(ctx);  // like icpk_align: the alignment starts from the source as set / committed
  ctx->src_pristine = false;   // (the loop moves the working copy)
  if (rc) return rc;
  ctx->have_seed = false;
  ctx->have_qperm = false;
  ctx->trace_R.clear();
  ctx->trace_t.clear();
  ctx->trace_mse.clear();
  ctx->trace_pairs.clear();
  struct Guard {
    icpk_ctx* c;
    ~Guard() { device_loop_disarm(c); }
  } guard{ctx};
  rc = device_loop_begin(ctx, &q, false);
  if (rc) return rc;
  ctx->loop_nact = NSUM;  // both flavours through the full 19 sums: one message shape
  const int B = red_blocks(ctx->src.n);
  auto sweep = [&]() -> int {
    // (an empty slice still takes part: its sums are zero)
    int r = enqueue_nn(ctx, ICPK_NN_GRID);
    if (r) return r;
    r = enqueue_reduce(ctx, q.max_nn_dist);
    if (r) return r;
    launch_reduce_final_shard(ctx->partial, ctx->pcount, B, ctx->red_out, ctx->st_dev, ctx->stream);
    return icpk_comm_allreduce_device(ctx, ctx->red_out, NSUM + 1);
  };
  rc = sweep();
  for (int i = 0; rc == ICPK_OK && i < q.max_iterations; ++i) {
    launch_loop_step(ctx->red_out, nullptr, -1, NSUM, ctx->st_dev, 0, ctx->stream);
    rc = sweep();
  }
  if (rc) {
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
  }
  launch_loop_step(ctx->red_out, nullptr, -1, NSUM, ctx->st_dev, 1, ctx->stream);
  if (loop_rec(ctx))
    launch_grid_unpack(ctx->qm4, ctx->rec, ctx->src.n, ctx->src.x(), ctx->src.y(), ctx->src.z(), ctx->best, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ICPK_HIP(ctx, hipMemcpyAsync(ctx->st_host, ctx->st_dev, sizeof(LoopState), hipMemcpyDeviceToHost, ctx->stream));
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return device_loop_finish(ctx, &q, T_out, stats);
}

int icpk_transform_target(icpk_ctx* ctx, const float R[9], const float t[3]) {
  if (!ctx || !R || !t) return ICPK_E_ARG;
  if (!ctx->have_tgt) return fail(ctx, ICPK_E_NOT_SET, "target cloud not set");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  Rt rt;
  std::memcpy(rt.R, R, sizeof(rt.R));
  std::memcpy(rt.t, t, sizeof(rt.t));
  Cloud& c = ctx->tgt;
  launch_transform(c.x(), c.y(), c.z(), c.n, rt, ctx->stream);
  // the kernel works on whole float4s: restore the +inf padding it touched
  const int padded = round_up(c.n < 1 ? 1 : c.n, NN_TILE);
  launch_fill_f32(c.x() + c.n, padded - c.n, __builtin_inff(), ctx->stream);
  launch_fill_f32(c.y() + c.n, padded - c.n, __builtin_inff(), ctx->stream);
  launch_fill_f32(c.z() + c.n, padded - c.n, __builtin_inff(), ctx->stream);
  if (ctx->have_normals) {  // normals rotate with the cloud (no translation)
    Rt rn = rt;
    rn.t[0] = rn.t[1] = rn.t[2] = 0.f;
    launch_transform(ctx->nrm.x(), ctx->nrm.y(), ctx->nrm.z(), c.n, rn, ctx->stream);
  }
  ICPK_HIP(ctx, hipGetLastError());
  ctx->have_assoc = false;
  ctx->have_dec = false;
  ctx->have_boxes = false;
  ctx->have_grid = false;
  ctx->have_seed = false;
  return ICPK_OK;  // stream-ordered: no host wait
}

int icpk_get_trace(icpk_ctx* ctx, int32_t* n_iter, float* R_out, float* t_out, int32_t* pairs_out, float* mse_out) {
  if (!ctx || !n_iter) return ICPK_E_ARG;
  const size_t n = ctx->trace_R.size() / 9;  // completed solves (a fallback iteration records none)
  *n_iter = (int32_t)n;
  if (R_out && n) std::memcpy(R_out, ctx->trace_R.data(), n * 9 * sizeof(float));
  if (t_out && n) std::memcpy(t_out, ctx->trace_t.data(), n * 3 * sizeof(float));
  if (pairs_out && n) std::memcpy(pairs_out, ctx->trace_pairs.data(), n * sizeof(int32_t));
  if (mse_out && n) std::memcpy(mse_out, ctx->trace_mse.data(), n * sizeof(float));
  return ICPK_OK;
}

int icpk_align(icpk_ctx* ctx, const icpk_params* p, float T_out[16], icpk_stats* stats) {
  if (T_out)
    for (int k = 0; k < 16; ++k) T_out[k] = (k % 5 == 0) ? 1.f : 0.f;  // identity on failure
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (!ctx || !p || !T_out) return ICPK_E_ARG;
  if (p->max_iterations < 0 || p->solve < ICPK_SOLVE_REFERENCE || p->solve > ICPK_SOLVE_POINT_TO_PLANE)
    return fail(ctx, ICPK_E_ARG, "bad params");
  if (p->solve == ICPK_SOLVE_POINT_TO_PLANE && ctx && !ctx->have_normals)
    return fail(ctx, ICPK_E_NOT_SET, "point-to-plane needs target normals");
  int rc = check_ready(ctx);
  if (rc) {
    if (stats) stats->status = rc;
    return rc;
  }
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  rc = copy_src0_to_src(ctx);
  if (rc) return rc;
  ctx->src_pristine = false;  // (the loop moves the working copy)
  ctx->have_seed = false;  // matches of an earlier alignment belong to a different source pose
  ctx->have_qperm = false;
  ctx->trace_R.clear();
  ctx->trace_t.clear();
  ctx->trace_mse.clear();
  ctx->trace_pairs.clear();
  if (!p->host_loop && !ctx->log_fn && ctx->src.n > 0 && p->max_iterations <= LOOP_MAX_ITER)
    return align_device_loop(ctx, p, T_out, stats);

  const bool prof = p->profile != 0;
  const bool prof_all = p->profile >= 2;
  size_t nev = 0;
  std::vector<size_t> ev_nn, ev_red, ev_tr;  // indices of (start, stop) pairs
  auto stamp = [&](std::vector<size_t>* list) -> int {
    if (!prof) return ICPK_OK;
    hipEvent_t e = get_event(ctx, nev);
    if (!e) return fail(ctx, ICPK_E_HIP, "hipEventCreate failed");
    ICPK_HIP(ctx, hipEventRecord(e, ctx->stream));
    if (list) list->push_back(nev);
    ++nev;
    return ICPK_OK;
  };

  const bool p2l = p->solve == ICPK_SOLVE_POINT_TO_PLANE;
  const int nsum = p2l ? NP2L : NSUM;
  double sums[NSUM_MAX];
  int64_t npairs = 0;
  float mse = 0.f;
  int sweeps = 0;
  int nsweep = 0;
  const int phase = ctx->profile_phase++;  // successive alignments bracket different sweeps: unbiased sample
  auto sweep = [&]() -> int {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const int nth = nsweep++;
    if (prof && (prof_all || p->profile_stride <= 1 || (nth + phase) % p->profile_stride == 0)) {
      // two events tightly around the K1 launch
      e0 = get_event(ctx, nev);
      e1 = get_event(ctx, nev + 1);
      if (!e0 || !e1) return fail(ctx, ICPK_E_HIP, "hipEventCreate failed");
      ev_nn.push_back(nev);
      nev += 2;
    }
    int r = enqueue_nn(ctx, p->nn_mode, e0, e1);
    if (r) return r;
    r = stamp(&ev_red);  // start of reduce
    if (r) return r;
    if (ctx->src.n > 0) {
      r = p2l ? enqueue_reduce_p2l(ctx, p->max_nn_dist) : enqueue_reduce(ctx, p->max_nn_dist);
      if (r) return r;
    }
    r = stamp(nullptr);  // end of reduce
    if (r) return r;
    ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->src.n > 0) {
      std::memcpy(sums, ctx->red_host, nsum * sizeof(double));
      std::memcpy(&npairs, ctx->red_host + nsum, sizeof(int64_t));
    } else {
      std::memset(sums, 0, sizeof(sums));
      npairs = 0;
    }
    if (p2l) {  // distance sum sits in the last slot
      const float m = npairs > 0 ? (float)(sums[27] / (double)npairs) : 0.f;
      mse = (float)((double)m * (double)m);
    } else {
      mse = mse_from(sums, npairs);
    }
    ++sweeps;
    log_delta(ctx, ICPK_LOG_NEAREST_NEIGHBOR, (int)npairs);  // icp.cpp:561
    log_delta(ctx, ICPK_LOG_MSE, (int)npairs);               // icp.cpp:635
    return ICPK_OK;
  };
  auto apply = [&](const float R[9], const float t[3]) -> int {
    Rt rt;
    std::memcpy(rt.R, R, sizeof(rt.R));
    std::memcpy(rt.t, t, sizeof(rt.t));
    int r = stamp(&ev_tr);
    if (r) return r;
    launch_transform(ctx->src.x(), ctx->src.y(), ctx->src.z(), ctx->src.n, rt, ctx->stream);
    ICPK_HIP(ctx, hipGetLastError());
    return stamp(nullptr);
  };

  ctx->trace_R.clear();
  ctx->trace_t.clear();
  ctx->trace_mse.clear();
  ctx->trace_pairs.clear();
  ctx->log_last = std::chrono::steady_clock::now();
  rc = sweep();  // icp.cpp:98
  if (rc) return rc;

  float Trot[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  float offset[3] = {0, 0, 0};
  double Tk[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  int status = ICPK_OK;
  int i = 0;
  while ((p->fixed_iterations || mse > p->threshold) && i < p->max_iterations) {  // icp.cpp:155
    if (npairs < p->min_pairs) {  // icp.cpp:163-182: reuse the caller's last motion
      rc = apply(p->last_rotation, p->last_translation);
      if (rc) return rc;
      for (int k = 0; k < 3; ++k) offset[k] = -p->last_translation[k];
      status = ICPK_W_TOO_FEW_PAIRS;
      break;
    }
    ctx->trace_pairs.push_back((int32_t)npairs);
    ctx->trace_mse.push_back(mse);
    if (p->solve == ICPK_SOLVE_REFERENCE) {
      float M[9], R[9], Rinv[9], neg[3];
      for (int k = 0; k < 9; ++k) M[k] = (float)sums[k];  // icp.cpp:212 (CV_32F result)
      log_delta(ctx, ICPK_LOG_RECONSTRUCT_POINT_CLOUDS, 0);  // icp.cpp:210
      solve_reference(M, R);                                 // icp.cpp:215-223
      log_delta(ctx, ICPK_LOG_SVD, 0);                       // icp.cpp:225
      if (i == 0)
        std::memcpy(Trot, R, sizeof(Trot));  // icp.cpp:227-229
      else
        mul3f(R, Trot, Trot);  // icp.cpp:231-232
      invert3f(R, Rinv);       // icp.cpp:235
      for (int k = 0; k < 3; ++k) {
        offset[k] = (float)(sums[9 + k] / (double)npairs);  // icp.cpp:240 (pre-rotation pairs)
        neg[k] = -offset[k];
      }
      rc = apply(Rinv, neg);  // icp.cpp:236,245
      if (rc) return rc;
      ctx->trace_R.insert(ctx->trace_R.end(), R, R + 9);
      ctx->trace_t.insert(ctx->trace_t.end(), offset, offset + 3);
    } else if (p2l) {
      double Rd[9], td[3];
      if (!solve_p2l(sums, Rd, td)) {
        status = ICPK_W_DEGENERATE;
        break;
      }
      log_delta(ctx, ICPK_LOG_SVD, 0);
      float Rf[9], tf[3];
      for (int k = 0; k < 9; ++k) Rf[k] = (float)Rd[k];
      for (int k = 0; k < 3; ++k) tf[k] = (float)td[k];
      rc = apply(Rf, tf);
      if (rc) return rc;
      ctx->trace_R.insert(ctx->trace_R.end(), Rf, Rf + 9);
      ctx->trace_t.insert(ctx->trace_t.end(), tf, tf + 3);
      double Tn[12];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
          double s = 0;
          for (int k = 0; k < 3; ++k) s += (double)Rf[3 * r + k] * Tk[4 * k + c];
          Tn[4 * r + c] = s + (c == 3 ? (double)tf[r] : 0.0);
        }
      std::memcpy(Tk, Tn, sizeof(Tk));
    } else {
      double sa[3], sb[3], sab[9], Rd[9], td[3];
      for (int k = 0; k < 3; ++k) {
        sa[k] = sums[13 + k];
        sb[k] = sums[16 + k];
      }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) sab[3 * r + c] = sums[3 * c + r];  // sum a_r b_c = M^T
      log_delta(ctx, ICPK_LOG_RECONSTRUCT_POINT_CLOUDS, 0);
      solve_kabsch(npairs, sa, sb, sab, Rd, td);
      log_delta(ctx, ICPK_LOG_SVD, 0);
      float Rf[9], tf[3];
      for (int k = 0; k < 9; ++k) Rf[k] = (float)Rd[k];
      for (int k = 0; k < 3; ++k) tf[k] = (float)td[k];
      rc = apply(Rf, tf);
      if (rc) return rc;
      ctx->trace_R.insert(ctx->trace_R.end(), Rf, Rf + 9);
      ctx->trace_t.insert(ctx->trace_t.end(), tf, tf + 3);
      double Tn[12];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
          double s = 0;
          for (int k = 0; k < 3; ++k) s += (double)Rf[3 * r + k] * Tk[4 * k + c];
          Tn[4 * r + c] = s + (c == 3 ? (double)tf[r] : 0.0);
        }
      std::memcpy(Tk, Tn, sizeof(Tk));
    }
    log_delta(ctx, ICPK_LOG_ROTATE, 0);  // icp.cpp:250
    rc = sweep();                        // icp.cpp:255
    if (rc) return rc;
    ++i;  // icp.cpp:257
  }
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));

  if (p->solve == ICPK_SOLVE_REFERENCE) {
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T_out[4 * r + c] = Trot[3 * r + c];
      T_out[4 * r + 3] = offset[r];  // icp.cpp:266-268: the LAST offset only
    }
  } else {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) T_out[4 * r + c] = (float)Tk[4 * r + c];
  }
  if (stats) {
    stats->iterations = i;
    stats->status = status;
    stats->final_pairs = (int32_t)npairs;
    stats->final_mse = mse;
    stats->nn_launches = sweeps;
    stats->nn_timed_launches = (int32_t)ev_nn.size();
    if (prof && nev >= 2) {
      auto span = [&](size_t a, size_t b) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ctx->events[a], ctx->events[b]);
        return ms;
      };
      for (size_t k : ev_nn) stats->nn_ms_total += span(k, k + 1);
      for (size_t k : ev_red) stats->reduce_ms_total += span(k, k + 1);
      for (size_t k : ev_tr) stats->transform_ms_total += span(k, k + 1);
      stats->total_ms = span(0, nev - 1);
    }
  }
  return status;
}

// ---- frame-batch mode (the frame-pair formulation of icp.cpp:541-563) ----------
// Independent pairs, `batch_group` of them advancing in LOCK STEP: every stage of an iteration is
// ONE launch for the whole group (K1d and K2 with blockIdx.y = pair, the loop step with one
// workgroup per pair), so the two small kernels and the launch gaps of the dependent chain
// sweep -> reduce -> step, which leave most of the GPU idle for a single pair, are shared by the
// group.  Each pair lives in a slot (a child context: its own clouds, grid, loop state and a
// stream for its set-up work); two sets of slots alternate so that the set-up of the next group
// (uploads, grid build, query order) overlaps the loop of the current one.  Results are those
// of icpk_align on the same pair, bit for bit (same kernels' bodies, same canonical reduction
// geometry per pair).
namespace {

struct GroupRun {
  int first = 0, count = 0, set = 0;  // pairs [first, first + count) live in slots [set * G, ...)
  std::vector<int> rc;                // per pair: set-up status (< 0: failed, not in the loop)
};

bool batch_eligible(const icpk_ctx* ctx, const icpk_params* p) {
  return p->nn_mode == ICPK_NN_GRID && !p->host_loop && !ctx->log_fn && p->profile <= 1 &&
         (p->solve == ICPK_SOLVE_REFERENCE || p->solve == ICPK_SOLVE_KABSCH) && p->max_iterations >= 0 &&
         p->max_iterations <= LOOP_MAX_ITER;
}

int ensure_slots(icpk_ctx* ctx, int n) {
  if (n > 2 * BATCH_MAX) return fail(ctx, ICPK_E_ARG, "too many frame-batch slots");
  if (!ctx->slot_states) {
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->slot_states, (size_t)2 * BATCH_MAX * sizeof(LoopState)));
    ICPK_HIP(ctx, hipHostMalloc((void**)&ctx->slot_states_host, (size_t)2 * BATCH_MAX * sizeof(LoopState), hipHostMallocDefault));
  }
  while ((int)ctx->slots.size() < n) {
    icpk_ctx* sl = make_context(ctx->device, ctx);
    if (!sl) return fail(ctx, ICPK_E_HIP, "frame-batch slot allocation failed");
    // the slot's loop state lives in the parent's pools (one copy brings a whole group's states back)
    const size_t k = ctx->slots.size();
    (void)hipFree(sl->st_dev);
    (void)hipHostFree(sl->st_host);
    sl->st_dev = ctx->slot_states + k;
    sl->st_host = ctx->slot_states_host + k;
    sl->st_pooled = true;
    ctx->slots.push_back(sl);
  }
  return ICPK_OK;
}

// uploads + everything up to (not including) the first host wait of the pair's set-up
// device-resident pair into a slot: one launch per cloud (planes, padding and, for the source,
// the working copy) instead of the seven copies and six fills of the general upload path
int slot_ingest_device(icpk_ctx* sl, const icpk_pair& pr) {
  if (pr.nt < 0 || pr.ns < 0 || (pr.nt > 0 && (!pr.tx || !pr.ty || !pr.tz)) || (pr.ns > 0 && (!pr.sx || !pr.sy || !pr.sz)))
    return fail(sl, ICPK_E_ARG, "bad cloud pointers/size");
  ICPK_HIP(sl, hipSetDevice(sl->device));
  int rc = ensure_cloud(sl, sl->tgt, pr.nt);
  if (rc == ICPK_OK) rc = ensure_cloud(sl, sl->src0, pr.ns);
  if (rc == ICPK_OK) rc = ensure_cloud(sl, sl->src, pr.ns);
  if (rc) return rc;
  launch_ingest_cloud(pr.tx, pr.ty, pr.tz, pr.nt, sl->tgt.cap, __builtin_inff(), sl->tgt.base, sl->tgt.cap, nullptr, 0,
                      sl->stream);
  // (src.cap <= src0.cap always; both paddings reach their own capacity's first NN_TILE multiple above n)
  const int spad = round_up(pr.ns < 1 ? 1 : pr.ns, NN_TILE);
  launch_ingest_cloud(pr.sx, pr.sy, pr.sz, pr.ns, spad, 0.f, sl->src0.base, sl->src0.cap, sl->src.base, sl->src.cap,
                      sl->stream);
  ICPK_HIP(sl, hipGetLastError());
  sl->have_tgt = sl->have_src = true;
  sl->have_assoc = sl->have_dec = sl->have_boxes = sl->have_grid = sl->have_seed = sl->have_normals = false;
  sl->have_qperm = false;
  return ICPK_OK;
}

int slot_setup_phase1(icpk_ctx* sl, const icpk_pair& pr, hipMemcpyKind kind) {
  int rc;
  if (kind == hipMemcpyDeviceToDevice) {
    rc = slot_ingest_device(sl, pr);
  } else {
    rc = set_target_impl(sl, pr.tx, pr.ty, pr.tz, pr.nt, kind, false);
    if (rc == ICPK_OK) rc = set_source_impl(sl, pr.sx, pr.sy, pr.sz, pr.ns, kind, false);
  }
  if (rc) return rc;
  rc = check_ready(sl);
  if (rc) return rc;
  if (sl->src.n <= 0) return ICPK_OK;  // an empty source takes the single-pair path
  sl->have_seed = false;
  sl->have_qperm = false;
  sl->rec_pending = false;
  return ensure_assoc(sl, sl->src.n);
}

// grid of the target (waits for its 36-byte info), query order, scan-order queries and seeds,
// initial loop state: the slot is then ready for the group's first sweep
int slot_setup_phase2(icpk_ctx* sl, const icpk_params* p, GridSweepArgs& first) {
  int rc = device_loop_begin(sl, p);
  if (rc) return rc;
  NnArgs a = base_nn_args(sl);
  NnBoxes bx{};
  int recheck = 0;
  rc = prepare_sorted_sweep(sl, ICPK_NN_GRID, a, bx, recheck);
  if (rc) return rc;
  first = grid_sweep_args(sl, a, bx);
  after_grid_sweep(sl);
  sl->rec_pending = true;  // planes / keys come from qm4 / rec on demand (icpk_get_associations)
  // (while the launches are being recorded the group's set-up event, recorded after the flush, takes its place)
  if (!setup_recorder()) ICPK_HIP(sl, hipEventRecord(sl->ready_ev, sl->stream));
  return ICPK_OK;
}

ReduceArgs slot_reduce_args(const icpk_ctx* sl) {
  ReduceArgs r{};
  r.best = sl->best;
  r.ax = sl->src.x();
  r.ay = sl->src.y();
  r.az = sl->src.z();
  r.tx = sl->tgt.x();
  r.ty = sl->tgt.y();
  r.tz = sl->tgt.z();
  r.o4 = sl->have_grid ? sl->o4 : nullptr;
  r.rec = sl->rec;
  r.partial = sl->partial;
  r.pcount = sl->pcount;
  r.st = sl->st_dev;
  r.nq = sl->src.n;
  r.nblocks = red_blocks(sl->src.n);
  return r;
}

// the whole loop of a group on the parent's stream, then the read-back of every loop state
int enqueue_group_loop(icpk_ctx* ctx, const icpk_params* p, const std::vector<icpk_ctx*>& act,
                       const std::vector<GridSweepArgs>& first, int set, const std::vector<bool>& own_event) {
  const int n = (int)act.size();
  if (n == 0) return ICPK_OK;
  const int nsum = loop_nsum(p);
  const int nact = nsum == NSUM_REF ? NSUM_REF : NSUM;
  long long nq_total = 0;
  bool group_event = false;
  for (int k = 0; k < n; ++k) {
    icpk_ctx* sl = act[k];
    if (own_event[k])
      ICPK_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl->ready_ev, 0));
    else
      group_event = true;  // set up by the group's batched launches
    nq_total += sl->src.n;
  }
  if (group_event) ICPK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->setup_ev[set], 0));
  // lanes per query: a single pair is latency-bound and wants 8; a group that fills the GPU
  // several times over is issue-bound and does better with fewer, longer lanes (measured on
  // 8 config-2 pairs: 43.4k iter/s with 8, 50.6k with 4, 48.2k with 2)
  const int slices = ctx->grid_slices ? ctx->grid_slices : (nq_total >= 300000 ? 4 : 8);
  GridSweepBatch gb{};
  ReduceBatch rb{};
  StepBatch sb{};
  for (int k = 0; k < n; ++k) {
    gb.p[k] = first[k];
    rb.p[k] = slot_reduce_args(act[k]);
    sb.p[k].partial = act[k]->partial;
    sb.p[k].pcount = act[k]->pcount;
    sb.p[k].nblocks = rb.p[k].nblocks;
    sb.p[k].st = act[k]->st_dev;
  }
  // params.profile = 1: ONE of the group's max_iterations + 1 batched sweeps (rotating from group to group)
  // is bracketed by two HIP events on this stream; finish_group adds the time to the group's first pair
  int timed = -1;
  ctx->batch_timed[set] = false;
  if (p->profile == 1) {
    for (hipEvent_t* e : {&ctx->batch_t0[set], &ctx->batch_t1[set]})
      if (!*e) ICPK_HIP(ctx, hipEventCreate(e));
    timed = ctx->profile_phase++ % (p->max_iterations + 1);
    ctx->batch_timed[set] = true;
  }
  auto sweep = [&](int nth, int expand) -> int {
    if (nth == timed) ICPK_HIP(ctx, hipEventRecord(ctx->batch_t0[set], ctx->stream));
    launch_nn_grid_batch(gb, n, slices, expand, ctx->stream);
    if (nth == timed) ICPK_HIP(ctx, hipEventRecord(ctx->batch_t1[set], ctx->stream));
    return ICPK_OK;
  };
  int src_ = sweep(0, 1);  // icp.cpp:98 (expanding search from element 0)
  if (src_) return src_;
  launch_assoc_reduce_batch(rb, n, p->max_nn_dist, nact, ctx->stream);
  for (int i = 0; i < p->max_iterations; ++i) {
    launch_loop_step_batch(sb, n, nsum, 0, ctx->stream);
    for (int k = 0; k < n; ++k) {  // pointer rotation only: nothing is enqueued for a chained sweep
      icpk_ctx* sl = act[k];
      NnArgs a = base_nn_args(sl);
      NnBoxes bx{};
      int recheck = 0;
      int rc = prepare_sorted_sweep(sl, ICPK_NN_GRID, a, bx, recheck);
      if (rc) {
        ctx->err = sl->err;
        return rc;
      }
      gb.p[k] = grid_sweep_args(sl, a, bx);
      after_grid_sweep(sl);
      rb.p[k].best = sl->best;
    }
    src_ = sweep(i + 1, 0);  // icp.cpp:255, K3 fused
    if (src_) return src_;
    launch_assoc_reduce_batch(rb, n, p->max_nn_dist, nact, ctx->stream);
  }
  launch_loop_step_batch(sb, n, nsum, 1, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  {  // the group's loop states: ONE copy (the slots' states are consecutive entries of the parent's pool)
    size_t lo = (size_t)-1, hi = 0;
    for (icpk_ctx* sl : act) {
      const size_t k = (size_t)(sl->st_dev - ctx->slot_states);
      lo = k < lo ? k : lo;
      hi = k > hi ? k : hi;
    }
    ICPK_HIP(ctx, hipMemcpyAsync(ctx->slot_states_host + lo, ctx->slot_states + lo, (hi - lo + 1) * sizeof(LoopState),
                                 hipMemcpyDeviceToHost, ctx->stream));
  }
  ICPK_HIP(ctx, hipEventRecord(ctx->group_ev[set], ctx->stream));
  return ICPK_OK;
}

void identity16(float* T) {
  for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.f : 0.f;
}

int align_batch_impl(icpk_ctx* ctx, int32_t n_pairs, const icpk_pair* pairs, const icpk_params* p, float* T_out,
                     icpk_stats* stats, hipMemcpyKind kind) {
  if (!ctx || n_pairs < 0 || (n_pairs > 0 && (!pairs || !T_out)) || !p) return ICPK_E_ARG;
  if (p->max_iterations < 0 || p->solve < ICPK_SOLVE_REFERENCE || p->solve > ICPK_SOLVE_POINT_TO_PLANE)
    return fail(ctx, ICPK_E_ARG, "bad params");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  int worst = ICPK_OK;
  auto note = [&](int rc) {
    if (rc < 0 && worst >= 0) worst = rc;
    if (rc > 0 && worst >= 0 && rc > worst) worst = rc;
  };
  for (int32_t b = 0; b < n_pairs; ++b) {
    identity16(T_out + 16 * (size_t)b);
    if (stats) std::memset(stats + b, 0, sizeof(icpk_stats));
  }
  auto fetch_assoc = [&](icpk_ctx* c, const icpk_pair& pr) -> int {
    if (!pr.idx_out && !pr.dist_out) return ICPK_OK;
    return icpk_get_associations(c, pr.idx_out, pr.dist_out);
  };
  if (!batch_eligible(ctx, p)) {
    // other kernels / flavours / a log callback: the pairs one after the other on this context
    for (int32_t b = 0; b < n_pairs; ++b) {
      int rc = set_target_impl(ctx, pairs[b].tx, pairs[b].ty, pairs[b].tz, pairs[b].nt, kind);
      if (rc == ICPK_OK) rc = set_source_impl(ctx, pairs[b].sx, pairs[b].sy, pairs[b].sz, pairs[b].ns, kind);
      if (rc == ICPK_OK) {
        rc = icpk_align(ctx, p, T_out + 16 * (size_t)b, stats ? stats + b : nullptr);
        if (rc >= 0) {
          const int r2 = fetch_assoc(ctx, pairs[b]);
          if (r2 < 0) rc = r2;
        }
      } else if (stats) {
        stats[b].status = rc;
      }
      note(rc);
    }
    return worst;
  }

  const int G = ctx->batch_group < 1 ? 1 : (ctx->batch_group > BATCH_MAX ? BATCH_MAX : ctx->batch_group);
  const int ngroups = (n_pairs + G - 1) / G;
  int rc = ensure_slots(ctx, ngroups > 1 ? 2 * G : (n_pairs < G ? n_pairs : G));
  if (rc) return rc;

  auto finish_group = [&](const GroupRun& g) -> int {
    bool any = false;
    for (int k = 0; k < g.count; ++k) any |= g.rc[k] == ICPK_OK;
    if (any) ICPK_HIP(ctx, hipEventSynchronize(ctx->group_ev[g.set]));
    float timed_ms = -1.f;
    if (any && ctx->batch_timed[g.set] && hipEventElapsedTime(&timed_ms, ctx->batch_t0[g.set], ctx->batch_t1[g.set]) != hipSuccess)
      timed_ms = -1.f;
    for (int k = 0; k < g.count; ++k) {
      const int b = g.first + k;
      icpk_ctx* sl = ctx->slots[(size_t)g.set * G + k];
      int r = g.rc[k];
      if (r == ICPK_OK) {
        r = device_loop_finish(sl, p, T_out + 16 * (size_t)b, stats ? stats + b : nullptr);
        if (stats && timed_ms >= 0.f) {  // the group's one timed launch, booked on its first pair (it covers ALL the group's pairs)
          stats[b].nn_ms_total = timed_ms;
          stats[b].nn_timed_launches = 1;
          timed_ms = -1.f;
        }
        if (r >= 0) {
          const int r2 = fetch_assoc(sl, pairs[b]);
          if (r2 < 0) r = r2;
        }
      } else if (r >= 100) {  // ran on the single-pair path during set-up: outputs already written
        r -= 100;
      } else if (stats) {
        stats[b].status = r;
      }
      if (r < 0) ctx->err = sl->err;
      note(r);
    }
    return ICPK_OK;
  };

  // ICPK_BATCH_TRACE=1 (diagnostic): host time per group in set-up phase 1 / phase 2 / loop
  // enqueue / waiting for the previous group, on stderr
  const bool trace = std::getenv("ICPK_BATCH_TRACE") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::micro>(b - a).count();
  };
  GroupRun prev;
  bool have_prev = false;
  std::vector<icpk_ctx*> unfinished;  // slots whose set-up stopped half-way (see the end of this function)
  // Every way out of the loop below, error or not, goes through the same tail: the previous group's results are
  // delivered, and nothing of this call is still in flight when it returns -- set-up kernels may be reading the
  // caller's device-resident clouds, uploads may be reading the caller's host buffers or a slot's staging area.
  auto drain = [&]() {
    (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t st : {ctx->setup_stream[0], ctx->setup_stream[1]})
      if (st) (void)hipStreamSynchronize(st);
    for (icpk_ctx* sl : ctx->slots) (void)hipStreamSynchronize(sl->stream);
  };
  auto bail = [&](int code) {
    if (have_prev) finish_group(prev);
    have_prev = false;
    drain();
    return code;
  };
#define ICPK_HIP_BAIL(call)                                                      \
  do {                                                                           \
    hipError_t e__ = (call);                                                     \
    if (e__ != hipSuccess) {                                                     \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e__);             \
      return bail(ICPK_E_HIP);                                                   \
    }                                                                            \
  } while (0)
  for (int gi = 0; gi < ngroups; ++gi) {
    const auto t0 = now();
        GroupRun g;
    g.first = gi * G;
    g.count = n_pairs - g.first < G ? n_pairs - g.first : G;
    g.set = ngroups > 1 ? (gi & 1) : 0;
    g.rc.assign(g.count, ICPK_OK);
    // set-up of the group's pairs: independent per slot (own buffers, own stream), so a few host
    // threads share the ~35 runtime calls per pair -- with 8 pairs per GPU (config 4 at 8 GPUs)
    // there is no previous group whose loop could hide this host time
    std::vector<GridSweepArgs> fargs(g.count);
    auto setup_one = [&](int k) {
      icpk_ctx* sl = ctx->slots[(size_t)g.set * G + k];
      const int b = g.first + k;
      int r = slot_setup_phase1(sl, pairs[b], kind);
      if (r != ICPK_OK) {
        g.rc[k] = r;
        return;
      }
      if (sl->src.n <= 0) {  // no queries: the single-pair path handles it (icp.cpp:163-182 fallback)
        r = icpk_align(sl, p, T_out + 16 * (size_t)b, stats ? stats + b : nullptr);
        g.rc[k] = r < 0 ? r : 100 + r;
        return;
      }
      r = slot_setup_phase2(sl, p, fargs[k]);
      if (r != ICPK_OK) device_loop_disarm(sl);
      g.rc[k] = r;
    };
    // Device-resident pairs: the set-up launches of the whole group are RECORDED (icpk_internal.h, SetupRecorder)
    // and issued as one launch per step on the set's set-up stream -- 13 launches instead of 13 per pair.
    std::vector<bool> recorded(g.count, false);
    // (host buffers: only when several groups follow each other -- 64 pairs 20.4 -> 17.8 ms; for a single group
    // the uploads of the pairs would queue up on the one set-up stream in front of everything else: 8 pairs
    // 3.1 -> 3.3 ms, tools/probe_host_batch.py)
    const bool batched = ctx->batch_setup != 0 && (kind == hipMemcpyDeviceToDevice || ngroups > 1 || ctx->batch_setup >= 2);
    // (host buffers: one thread -- concurrent host-to-device copies from several threads stall for
    // ~9 ms at random on this runtime, tools/one_align.py --batch under ICPK_BATCH_TRACE)
    const int nthreads = batched || kind == hipMemcpyHostToDevice ? 1 : (g.count < ctx->batch_threads ? g.count : ctx->batch_threads);
    if (batched) {
      if (!ctx->setup_stream[g.set]) {
        ICPK_HIP_BAIL(hipStreamCreateWithFlags(&ctx->setup_stream[g.set], hipStreamNonBlocking));
        ICPK_HIP_BAIL(hipEventCreateWithFlags(&ctx->setup_ev[g.set], hipEventDisableTiming));
      }
      const hipStream_t ss = ctx->setup_stream[g.set];
      std::vector<SetupRecorder> recs(g.count);
      for (int k = 0; k < g.count; ++k) {
        icpk_ctx* sl = ctx->slots[(size_t)g.set * G + k];
        if (pairs[g.first + k].ns <= 0) {  // goes down the single-pair path at once: nothing to defer
          setup_one(k);
          continue;
        }
        const hipStream_t own = sl->stream;
        sl->stream = ss;  // whatever is not recorded (first-use clears) must precede the flushed launches
        setup_recorder() = &recs[k];
        setup_one(k);
        setup_recorder() = nullptr;
        sl->stream = own;
        if (g.rc[k] == ICPK_OK && recs[k].overflow) {
          // more set-up steps than the recorder holds: the launches beyond its capacity were never issued --
          // the pair must not run on half a set-up
          device_loop_disarm(sl);
          g.rc[k] = fail(sl, ICPK_E_HIP, "frame-batch set-up recorder overflow (SETUP_MAX_CALLS)");
        }
        recorded[k] = g.rc[k] == ICPK_OK;
      }
      std::vector<SetupRecorder> ok;
      for (int k = 0; k < g.count; ++k)
        if (recorded[k]) ok.push_back(recs[k]);
      if (!ok.empty()) {
        if (ctx->batch_setup == 3 /* test hook: the pair-by-pair replay */ || !flush_setup_batches(ok.data(), (int)ok.size(), ss))
          for (const SetupRecorder& r : ok) replay_setup(r, ss);
        ICPK_HIP_BAIL(hipGetLastError());
        ICPK_HIP_BAIL(hipEventRecord(ctx->setup_ev[g.set], ss));
      }
    } else if (nthreads <= 1) {
      for (int k = 0; k < g.count; ++k) setup_one(k);
    } else {
      std::vector<std::thread> pool;
      for (int t = 1; t < nthreads; ++t)
        pool.emplace_back([&, t] {
          for (int k = t; k < g.count; k += nthreads) setup_one(k);
        });
      for (int k = 0; k < g.count; k += nthreads) setup_one(k);
      for (std::thread& th : pool) th.join();
    }
    const auto t1 = now();
    std::vector<icpk_ctx*> act;
    std::vector<GridSweepArgs> first;
    std::vector<bool> own_event;
    for (int k = 0; k < g.count; ++k) {
      if (g.rc[k] != ICPK_OK) {
        if (g.rc[k] < 0) unfinished.push_back(ctx->slots[(size_t)g.set * G + k]);
        continue;
      }
      act.push_back(ctx->slots[(size_t)g.set * G + k]);
      first.push_back(fargs[k]);
      own_event.push_back(!recorded[k]);
    }
    const auto t2 = now();
    rc = enqueue_group_loop(ctx, p, act, first, g.set, own_event);
    if (rc) {  // enqueue failed: nothing of this group can be trusted (the previous group's results still are)
      for (icpk_ctx* sl : act) device_loop_disarm(sl);
      return bail(rc);
    }
    const auto t3 = now();
    if (have_prev) finish_group(prev);
    if (trace)
      std::fprintf(stderr, "icpk batch group %d: set-up %.0f us (%d host threads), loop enqueue %.0f us, wait+finish prev %.0f us\n",
                   gi, us(t0, t1), nthreads, us(t2, t3), us(t3, now()));
    prev = g;
    have_prev = true;
  }
  const auto te0 = now();
  if (have_prev) finish_group(prev);
  const auto te1 = now();
  // host input buffers were read asynchronously: everything has landed before we return.  A pair that went
  // through the loop has: its slot's stream reached `ready_ev` before the group's loop started, and the loop
  // has been waited for.  Only slots whose set-up FAILED may still have copies in flight.
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (icpk_ctx* sl : unfinished) ICPK_HIP(ctx, hipStreamSynchronize(sl->stream));
  if (!unfinished.empty())  // (with the batched set-up their uploads went to the set's set-up stream)
    for (hipStream_t st : {ctx->setup_stream[0], ctx->setup_stream[1]})
      if (st) ICPK_HIP(ctx, hipStreamSynchronize(st));
  if (trace) std::fprintf(stderr, "icpk batch tail: wait+finish last group %.0f us, stream syncs %.0f us\n", us(te0, te1), us(te1, now()));
#undef ICPK_HIP_BAIL
  return worst;
}

}  // namespace

int icpk_align_batch(icpk_ctx* ctx, int32_t n_pairs, const icpk_pair* pairs, const icpk_params* p, float* T_out,
                     icpk_stats* stats) {
  return align_batch_impl(ctx, n_pairs, pairs, p, T_out, stats, hipMemcpyHostToDevice);
}

int icpk_align_batch_device(icpk_ctx* ctx, int32_t n_pairs, const icpk_pair* pairs, const icpk_params* p,
                            float* T_out, icpk_stats* stats) {
  return align_batch_impl(ctx, n_pairs, pairs, p, T_out, stats, hipMemcpyDeviceToDevice);
}

struct DepthFilter {  // SLAM.cpp:553-574 filterDepthImage
  int max_d, min_d, morph, ax, ay;
};
static int check_filter(icpk_ctx* ctx, int morph, int& ax, int& ay) {
  if (ax < 0) ax = 2;  // cv::dilate / cv::erode default anchor (-1,-1): the element's centre
  if (ay < 0) ay = 2;
  if (morph != 0 && (ax > 4 || ay > 4)) return icpk_host_fail(ctx, ICPK_E_ARG, "anchor outside the 5x5 element");
  return ICPK_OK;
}

// raw + filtered depth images on the device (`count` pixels each) and `ints` block-count words
// the subsample key of the next image this context back-projects (kernels_backproject.hip: bp_keep)
static unsigned long long next_subsample_key(icpk_ctx* ctx) {
  const unsigned long long k = ctx->sub_stream++;
  return ctx->sub_seed + (k + 1ull) * 0x9E3779B97F4A7C15ull;
}

// caller memory that icpk_register_host_buffer pinned: the device address of [p, p + bytes) or nullptr
static const void* registered_device_pointer(const icpk_ctx* ctx, const void* p, size_t bytes) {
  const char* c = static_cast<const char*>(p);
  for (const icpk_ctx::HostRange& r : ctx->registered)
    if (c >= r.host && c + bytes <= r.host + r.bytes) return r.dev + (c - r.host);
  return nullptr;
}

int icpk_register_host_buffer(icpk_ctx* ctx, const void* ptr, size_t bytes) {
  if (!ctx || !ptr || bytes == 0) return ICPK_E_ARG;
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  if (registered_device_pointer(ctx, ptr, bytes)) return ICPK_OK;
  ICPK_HIP(ctx, hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterMapped));
  void* dev = nullptr;
  const hipError_t e = hipHostGetDevicePointer(&dev, const_cast<void*>(ptr), 0);
  if (e != hipSuccess) {
    (void)hipHostUnregister(const_cast<void*>(ptr));
    return fail(ctx, ICPK_E_HIP, hipGetErrorString(e));
  }
  ctx->registered.push_back({static_cast<const char*>(ptr), bytes, static_cast<const char*>(dev)});
  return ICPK_OK;
}

int icpk_unregister_host_buffer(icpk_ctx* ctx, const void* ptr) {
  if (!ctx || !ptr) return ICPK_E_ARG;
  for (size_t k = 0; k < ctx->registered.size(); ++k)
    if (ctx->registered[k].host == static_cast<const char*>(ptr)) {
      ICPK_HIP(ctx, hipSetDevice(ctx->device));
      ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (nothing in flight may still read it)
      ICPK_HIP(ctx, hipHostUnregister(const_cast<void*>(ptr)));
      ctx->registered.erase(ctx->registered.begin() + k);
      return ICPK_OK;
    }
  return fail(ctx, ICPK_E_ARG, "not a registered buffer");
}

int icpk_set_subsample(icpk_ctx* ctx, int32_t factor, uint64_t seed) {
  if (!ctx || factor < 0) return ICPK_E_ARG;
  ctx->sub_factor = factor;
  ctx->sub_seed = seed;
  ctx->sub_stream = 0;
  return ICPK_OK;
}

static int ensure_depth_buffers(icpk_ctx* ctx, int count, int ints) {
  if (count > ctx->depth_cap) {
    if (ctx->depth_dev) ICPK_HIP(ctx, hipFree(ctx->depth_dev));
    if (ctx->depth_flt) ICPK_HIP(ctx, hipFree(ctx->depth_flt));
    ctx->depth_dev = ctx->depth_flt = nullptr;
    ctx->depth_cap = 0;
    ctx->frame_slot = -1;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->depth_dev, (size_t)count * sizeof(uint16_t)));
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->depth_flt, (size_t)count * sizeof(uint16_t)));
    ctx->depth_cap = count;
  }
  if (ints > ctx->bp_counts_cap) {
    if (ctx->bp_counts) ICPK_HIP(ctx, hipFree(ctx->bp_counts));
    ctx->bp_counts = nullptr;
    ctx->bp_counts_cap = 0;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->bp_counts, (size_t)ints * sizeof(int)));
    ctx->bp_counts_cap = ints;
  }
  return ICPK_OK;
}

static int backproject_impl(icpk_ctx* ctx, const uint16_t* depth, int32_t rows, int32_t cols, float fx, float cx,
                            const float offset[3], int32_t which, int normals_mode /* <0: none */,
                            const DepthFilter* flt = nullptr) {
  if (!ctx || !depth || rows <= 0 || cols <= 0 || (which != 0 && which != 1) || (int64_t)rows * cols > (1 << 28) ||
      normals_mode > ICPK_NORMALS_REFERENCE)
    return ICPK_E_ARG;
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  const int npix = rows * cols;
  const int nblocks = (npix + 1023) / 1024;
  int rc = ensure_depth_buffers(ctx, npix, nblocks + 2);
  if (rc) return rc;
  Cloud& c = which == 0 ? ctx->src0 : ctx->tgt;
  rc = ensure_cloud(ctx, c, npix);  // worst case: every pixel valid
  if (rc) return rc;
  ctx->frame_slot = -1;  // (the image buffers are shared with icpk_backproject_pair's resident frame)
  ICPK_HIP(ctx, hipMemcpyAsync(ctx->depth_dev, depth, (size_t)npix * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
  const uint16_t* dimg = ctx->depth_dev;
  if (flt) {  // SLAM.cpp:229,553-574: the frame is filtered before it is back-projected
    launch_depth_filter(ctx->depth_dev, ctx->depth_flt, rows, cols, flt->min_d, flt->max_d, flt->ax, flt->ay, flt->morph,
                        ctx->stream);
    dimg = ctx->depth_flt;
  }
  const float ox = offset ? offset[0] : 0.f, oy = offset ? offset[1] : 0.f, oz = offset ? offset[2] : 0.f;
  // the total lands in bp_counts[nblocks + 1] (device) and is read back pinned
  float *nxp = nullptr, *nyp = nullptr, *nzp = nullptr;
  if (normals_mode >= 0) {
    rc = ensure_cloud(ctx, ctx->nrm, npix);
    if (rc) return rc;
    nxp = ctx->nrm.x();
    nyp = ctx->nrm.y();
    nzp = ctx->nrm.z();
  }
  launch_backproject(dimg, rows, cols, fx, cx, ox, oy, oz, c.x(), c.y(), c.z(), nxp, nyp, nzp,
                     normals_mode < 0 ? 0 : normals_mode, ctx->bp_counts, ctx->bp_counts + nblocks + 1, next_subsample_key(ctx),
                     ctx->sub_factor, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ICPK_HIP(ctx, hipMemcpyAsync(ctx->bp_n_host, ctx->bp_counts + nblocks + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int n = *ctx->bp_n_host;
  c.n = n;
  const float pad = which == 0 ? 0.f : __builtin_inff();
  const int padded = round_up(n < 1 ? 1 : n, NN_TILE);
  launch_fill_f32(c.x() + n, padded - n, pad, ctx->stream);
  launch_fill_f32(c.y() + n, padded - n, pad, ctx->stream);
  launch_fill_f32(c.z() + n, padded - n, pad, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ctx->have_assoc = false;
  ctx->have_seed = false;
  ctx->have_qperm = false;
  if (which == 1) {
    ctx->have_dec = ctx->have_boxes = ctx->have_grid = false;
    ctx->have_normals = normals_mode >= 0;
    if (ctx->have_normals) ctx->nrm.n = n;
  }
  if (which == 0) {
    ctx->have_src = true;
    rc = copy_src0_to_src(ctx);
    if (rc) return rc;
  } else {
    ctx->have_tgt = true;
  }
  return n;  // (the depth image was consumed before the count came back: no second host wait)
}

int icpk_backproject_pair(icpk_ctx* ctx, const uint16_t* depth_source, const uint16_t* depth_target, int32_t rows,
                          int32_t cols, float fx, float cx, const float offset[3], const float R[9], const float t[3],
                          int32_t filter, int32_t max_d, int32_t min_d, int32_t morph, int32_t anchor_x, int32_t anchor_y,
                          int32_t* n_source, int32_t* n_target) {
  if (!ctx || !depth_source || rows <= 0 || cols <= 0 || (int64_t)rows * cols > (1 << 27) || (!R != !t))
    return ICPK_E_ARG;
  int ax = anchor_x, ay = anchor_y;
  int rc = filter ? check_filter(ctx, morph, ax, ay) : ICPK_OK;
  if (rc) return rc;
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  const int npix = rows * cols;
  const int nblocks = (npix + 1023) / 1024;
  const int per_image = nblocks + 2;
  const int fset[6] = {filter != 0, max_d, min_d, morph != 0, ax, ay};
  // depth_target == NULL: the previous frame is the one this context saw as `depth_source` last time (SLAM.cpp:305,
  // previous = filtered.clone()): its image -- and its filtered copy -- are still on the device
  const bool resident = depth_target == nullptr;
  if (resident && (ctx->frame_slot < 0 || ctx->frame_rows != rows || ctx->frame_cols != cols || 2 * npix > ctx->depth_cap))
    return icpk_host_fail(ctx, ICPK_E_NOT_SET, "no resident previous frame of this size (pass depth_target)");
  if (!resident && 2 * npix > ctx->depth_cap) ctx->frame_slot = -1;  // (the buffers are about to be replaced)
  rc = ensure_depth_buffers(ctx, 2 * npix, 2 * per_image + 2);
  if (rc) return rc;
  for (Cloud* c : {&ctx->src0, &ctx->src, &ctx->tgt}) {
    rc = ensure_cloud(ctx, *c, npix);  // worst case: every pixel valid
    if (rc) return rc;
  }
  // two image slots; the new frame goes where the resident one is not
  const int tslot = resident ? ctx->frame_slot : 1;
  const int sslot = 1 - tslot;
  uint16_t* const raw_s = ctx->depth_dev + (size_t)sslot * npix;
  uint16_t* const raw_t = ctx->depth_dev + (size_t)tslot * npix;
  uint16_t* const flt_s = ctx->depth_flt + (size_t)sslot * npix;
  uint16_t* const flt_t = ctx->depth_flt + (size_t)tslot * npix;
  const size_t bytes = (size_t)npix * sizeof(uint16_t);
  ctx->frame_slot = -1;  // (nothing is resident until this call has enqueued everything)
  // The images cross PCIe from the context's own pinned staging buffer, in halves: the copy engine moves one half
  // while the host copies the next one in.  (Handing the caller's pageable buffer to hipMemcpyAsync leaves the staging
  // to the runtime -- one blocking copy, then the transfer -- and was seen to take 70 us in one process and 340 us in
  // the next for the same 614 KB.)  The buffer is free again when this call returns: the counts it waits for are made
  // from the uploaded images.
  if (2 * npix > ctx->stage_depth_cap) {
    if (ctx->stage_depth) ICPK_HIP(ctx, hipHostFree(ctx->stage_depth));
    ctx->stage_depth = nullptr;
    ctx->stage_depth_cap = 0;
    ICPK_HIP(ctx, hipHostMalloc((void**)&ctx->stage_depth, (size_t)2 * npix * sizeof(uint16_t), hipHostMallocDefault));
    ctx->stage_depth_cap = 2 * npix;
  }
  // without the filter the images are not copied at all: the counting pass reads them from the staging buffer
  const bool zero_copy = !filter && ctx->zero_copy_upload;
  // (an image inside memory the caller has registered -- icpk_register_host_buffer -- is read where it lies: no copy at all)
  const uint16_t* reg_s = zero_copy ? static_cast<const uint16_t*>(registered_device_pointer(ctx, depth_source, bytes)) : nullptr;
  const uint16_t* reg_t =
      zero_copy && !resident ? static_cast<const uint16_t*>(registered_device_pointer(ctx, depth_target, bytes)) : nullptr;
  auto upload = [&](uint16_t* dev, const uint16_t* host, uint16_t* stage) -> int {
    if (zero_copy) {
      if (!(host == depth_source ? reg_s : reg_t)) std::memcpy(stage, host, bytes);
      return ICPK_OK;
    }
    const int parts = npix >= 65536 ? 2 : 1;  // (more parts cost more in copy commands than they hide)
    for (int k = 0; k < parts; ++k) {
      const size_t a0 = (size_t)npix * k / parts, a1 = (size_t)npix * (k + 1) / parts;
      std::memcpy(stage + a0, host + a0, (a1 - a0) * sizeof(uint16_t));
      ICPK_HIP(ctx, hipMemcpyAsync(dev + a0, stage + a0, (a1 - a0) * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    }
    return ICPK_OK;
  };
  if ((rc = upload(raw_s, depth_source, ctx->stage_depth))) return rc;
  if (!resident && (rc = upload(raw_t, depth_target, ctx->stage_depth + npix))) return rc;
  const uint16_t *img_s = raw_s, *img_t = raw_t;
  if (filter) {  // SLAM.cpp:229,553-574: the frames are filtered before they are back-projected
    launch_depth_filter(raw_s, flt_s, rows, cols, min_d, max_d, ax, ay, morph != 0, ctx->stream);
    // (the resident frame's filtered copy is reused when it was made with the same settings)
    if (!resident || std::memcmp(fset, ctx->frame_filter, sizeof(fset)) != 0)
      launch_depth_filter(raw_t, flt_t, rows, cols, min_d, max_d, ax, ay, morph != 0, ctx->stream);
    img_s = flt_s;
    img_t = flt_t;
  }
  if (npix > ctx->pix_cap) {
    if (ctx->pix_tidx) ICPK_HIP(ctx, hipFree(ctx->pix_tidx));
    if (ctx->pix_src) ICPK_HIP(ctx, hipFree(ctx->pix_src));
    ctx->pix_tidx = ctx->pix_src = nullptr;
    ctx->pix_cap = 0;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->pix_tidx, (size_t)npix * sizeof(int)));
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->pix_src, (size_t)npix * sizeof(int)));
    ctx->pix_cap = npix;
  }
  BpPair b;
  const uint16_t* stage_dev = nullptr;
  if (zero_copy) ICPK_HIP(ctx, hipHostGetDevicePointer((void**)&stage_dev, ctx->stage_depth, 0));
  b.im[0] = BpImage{img_s, ctx->src0.x(), ctx->src0.y(), ctx->src0.z(), ctx->src.x(), ctx->src.y(), ctx->src.z(),
                    ctx->bp_counts, 0.f, ctx->pix_src, nullptr, zero_copy ? (reg_s ? reg_s : stage_dev) : nullptr, raw_s, 0, 0, 0};
  b.im[1] = BpImage{img_t, ctx->tgt.x(), ctx->tgt.y(), ctx->tgt.z(), nullptr, nullptr, nullptr,
                    ctx->bp_counts + per_image, __builtin_inff(), nullptr, ctx->pix_tidx,
                    zero_copy && !resident ? (reg_t ? reg_t : stage_dev + npix) : nullptr, raw_t, 0, 0, 0};
  // (icp.cpp:38-39 builds the cloud of `data` first, then that of `previous`: the source draws its pattern first)
  b.im[0].sub_key = next_subsample_key(ctx);
  b.im[1].sub_key = next_subsample_key(ctx);
  b.im[0].sub_factor = b.im[1].sub_factor = ctx->sub_factor;
  Rt rt{};
  if (R) {
    std::memcpy(rt.R, R, sizeof(rt.R));
    std::memcpy(rt.t, t, sizeof(rt.t));
  }
  int* n_dev = ctx->bp_counts + 2 * per_image;
  // the one host wait: both counts (and both images consumed).  The scan writes them into pinned, mapped words as
  // well (progress words 4, 5), the host spins on those and returns while the scatter is still running -- whatever
  // comes next is stream-ordered behind it.  ICPK_RESULT_MIRROR=0: copy them back and wait for the stream.
  volatile int* const nw = ctx->progress + 4;
  nw[0] = nw[1] = -1;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  launch_backproject_pair(b, rows, cols, fx, cx, offset ? offset[0] : 0.f, offset ? offset[1] : 0.f,
                          offset ? offset[2] : 0.f, rt, R != nullptr, n_dev, ctx->result_mirror ? ctx->progress_dev + 4 : nullptr,
                          ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  if (ctx->result_mirror) {
    auto t_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
    for (unsigned spin = 1; nw[0] < 0 || nw[1] < 0; ++spin) {
      __builtin_ia32_pause();
      if ((spin & 0x3ff) != 0) continue;
      std::this_thread::yield();
      const auto now = std::chrono::steady_clock::now();
      if (now < t_query) continue;
      t_query = now + std::chrono::milliseconds(20);
      const hipError_t q = hipStreamQuery(ctx->stream);  // (a faulted kernel must end the wait)
      if (q == hipSuccess) {
        if (nw[0] < 0 || nw[1] < 0) return icpk_host_fail(ctx, ICPK_E_HIP, "back-projection ended without its counts");
        break;
      }
      if (q != hipErrorNotReady) return icpk_host_fail(ctx, ICPK_E_HIP, hipGetErrorString(q));
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    ctx->bp_n_host[0] = nw[0];
    ctx->bp_n_host[1] = nw[1];
  } else {
    ICPK_HIP(ctx, hipMemcpyAsync(ctx->bp_n_host, n_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  ctx->src0.n = ctx->src.n = ctx->bp_n_host[0];
  ctx->tgt.n = ctx->bp_n_host[1];
  ctx->src_pristine = true;  // (the scatter wrote the committed and the working copy of the source at once)
  ctx->have_pix_seed = true;  // (... and which pixel every point came from)
  ctx->pix_rows = rows;
  ctx->pix_cols = cols;
  ctx->have_src = ctx->have_tgt = true;
  ctx->have_assoc = ctx->have_seed = ctx->have_qperm = false;
  ctx->have_dec = ctx->have_boxes = ctx->have_grid = ctx->have_normals = false;
  ctx->frame_slot = sslot;
  ctx->frame_rows = rows;
  ctx->frame_cols = cols;
  std::memcpy(ctx->frame_filter, fset, sizeof(fset));
  if (n_source) *n_source = ctx->src.n;
  if (n_target) *n_target = ctx->tgt.n;
  return ICPK_OK;
}

int icpk_backproject_filtered(icpk_ctx* ctx, const uint16_t* depth, int32_t rows, int32_t cols, float fx, float cx,
                              const float offset[3], int32_t which, int32_t normals_mode, int32_t max_d,
                              int32_t min_d, int32_t morph, int32_t anchor_x, int32_t anchor_y) {
  if (!ctx) return ICPK_E_ARG;
  if (normals_mode >= 0 && which != 1) return icpk_host_fail(ctx, ICPK_E_ARG, "normals belong to the target cloud");
  int ax = anchor_x, ay = anchor_y;
  int rc = check_filter(ctx, morph, ax, ay);
  if (rc) return rc;
  const DepthFilter f{max_d, min_d, morph != 0, ax, ay};
  return backproject_impl(ctx, depth, rows, cols, fx, cx, offset, which, normals_mode < 0 ? -1 : normals_mode, &f);
}

int icpk_filter_depth_image(icpk_ctx* ctx, const uint16_t* depth_in, uint16_t* depth_out, int32_t rows, int32_t cols,
                            int32_t max_d, int32_t min_d, int32_t morph, int32_t anchor_x, int32_t anchor_y) {
  if (!ctx || !depth_in || !depth_out || rows <= 0 || cols <= 0 || (int64_t)rows * cols > (1 << 28)) return ICPK_E_ARG;
  int ax = anchor_x, ay = anchor_y;
  int rc = check_filter(ctx, morph, ax, ay);
  if (rc) return rc;
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  const int npix = rows * cols;
  rc = ensure_depth_buffers(ctx, npix, 0);
  if (rc) return rc;
  ctx->frame_slot = -1;  // (the image buffers are shared with icpk_backproject_pair's resident frame)
  ICPK_HIP(ctx, hipMemcpyAsync(ctx->depth_dev, depth_in, (size_t)npix * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
  launch_depth_filter(ctx->depth_dev, ctx->depth_flt, rows, cols, min_d, max_d, ax, ay, morph != 0, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ICPK_HIP(ctx, hipMemcpyAsync(depth_out, ctx->depth_flt, (size_t)npix * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ICPK_OK;
}

/* icp.cpp:488-515 on the context's clouds (source = the frame's key points, target = the map's) */
int icpk_associate_keypoints(icpk_ctx* ctx, int32_t nn_mode, float max_dist, int32_t* assoc_query,
                             int32_t* assoc_target, float* assoc_dist, int32_t* n_assoc, int32_t* rejected_query,
                             int32_t rejected_capacity, int32_t* n_rejected) {
  if (!ctx || !n_assoc || !n_rejected || *n_rejected < 0 || rejected_capacity < *n_rejected) return ICPK_E_ARG;
  if (!ctx->have_tgt || !ctx->have_src) return icpk_host_fail(ctx, ICPK_E_NOT_SET, "source or target cloud not set");
  // icp.cpp:490-491: an empty map returns BEFORE errors / associations are cleared: nothing is touched
  if (ctx->tgt.n <= 0) return ICPK_W_EMPTY_MAP;
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  if (int ru = ensure_unpacked(ctx)) return ru;
  const int nq = ctx->src.n;
  if (nq == 0) {  // icp.cpp:497-498: the lists are cleared, nothing is appended
    *n_assoc = 0;
    return ICPK_OK;
  }
  if (!assoc_query || !assoc_target || !assoc_dist || (!rejected_query && rejected_capacity > 0)) return ICPK_E_ARG;
  int rc = enqueue_nn(ctx, nn_mode);
  if (rc) return rc;
  const int cap = round_up(nq, NN_TILE);
  if (cap > ctx->ks_cap) {
    if (ctx->ks_buf) ICPK_HIP(ctx, hipFree(ctx->ks_buf));
    ctx->ks_buf = nullptr;
    ctx->ks_cap = 0;
    ICPK_HIP(ctx, hipMalloc((void**)&ctx->ks_buf, (size_t)4 * cap * sizeof(int32_t)));
    ctx->ks_cap = cap;
  }
  const int nblocks = (nq + 1023) / 1024;
  rc = ensure_depth_buffers(ctx, 0, nblocks + 2);
  if (rc) return rc;
  int32_t* dq = ctx->ks_buf;
  int32_t* dt = dq + ctx->ks_cap;
  float* dd = reinterpret_cast<float*>(dt + ctx->ks_cap);
  int32_t* dr = dt + 2 * (size_t)ctx->ks_cap;
  launch_assoc_split(ctx->best, nq, max_dist, ctx->bp_counts, ctx->bp_counts + nblocks + 1, dq, dt, dd, dr, ctx->stream);
  ICPK_HIP(ctx, hipGetLastError());
  ICPK_HIP(ctx, hipMemcpyAsync(ctx->bp_n_host, ctx->bp_counts + nblocks + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int na = *ctx->bp_n_host, nr = nq - na;
  if (*n_rejected + nr > rejected_capacity)
    return icpk_host_fail(ctx, ICPK_E_ARG, "rejected_capacity too small for the appended queries");
  if (na > 0) {
    ICPK_HIP(ctx, hipMemcpyAsync(assoc_query, dq, (size_t)na * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(assoc_target, dt, (size_t)na * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(assoc_dist, dd, (size_t)na * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (nr > 0)
    ICPK_HIP(ctx, hipMemcpyAsync(rejected_query + *n_rejected, dr, (size_t)nr * sizeof(int32_t), hipMemcpyDeviceToHost,
                                 ctx->stream));
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n_assoc = na;           // icp.cpp:497-498: errors / associations are rebuilt by every call
  *n_rejected += nr;       // icp.cpp:507-509: nonAssociations only ever grows
  return ICPK_OK;
}

int icpk_backproject(icpk_ctx* ctx, const uint16_t* depth, int32_t rows, int32_t cols, float fx, float cx,
                     const float offset[3], int32_t which) {
  return backproject_impl(ctx, depth, rows, cols, fx, cx, offset, which, -1);
}

int icpk_backproject_with_normals(icpk_ctx* ctx, const uint16_t* depth, int32_t rows, int32_t cols, float fx, float cx,
                                  const float offset[3], int32_t normals_mode) {
  if (normals_mode < 0) return ICPK_E_ARG;
  return backproject_impl(ctx, depth, rows, cols, fx, cx, offset, 1, normals_mode);
}

int icpk_set_target_normals(icpk_ctx* ctx, const float* nx, const float* ny, const float* nz, int32_t n) {
  if (!ctx) return ICPK_E_ARG;
  if (!ctx->have_tgt) return fail(ctx, ICPK_E_NOT_SET, "target cloud not set");
  if (n != ctx->tgt.n) return fail(ctx, ICPK_E_ARG, "normal count differs from the target size");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  int rc = upload_cloud(ctx, ctx->nrm, nx, ny, nz, n, 0.f, hipMemcpyHostToDevice);
  if (rc) return rc;
  ctx->have_normals = true;
  return ICPK_OK;
}

int icpk_get_target_normals(icpk_ctx* ctx, float* nx, float* ny, float* nz) {
  if (!ctx || !nx || !ny || !nz) return ICPK_E_ARG;
  if (!ctx->have_normals) return fail(ctx, ICPK_E_NOT_SET, "no target normals");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  const size_t b = (size_t)ctx->tgt.n * sizeof(float);
  if (b) {
    ICPK_HIP(ctx, hipMemcpyAsync(nx, ctx->nrm.x(), b, hipMemcpyDeviceToHost, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(ny, ctx->nrm.y(), b, hipMemcpyDeviceToHost, ctx->stream));
    ICPK_HIP(ctx, hipMemcpyAsync(nz, ctx->nrm.z(), b, hipMemcpyDeviceToHost, ctx->stream));
  }
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ICPK_OK;
}

int icpk_reduce_p2l(icpk_ctx* ctx, float max_dist, double* sums, int64_t* count) {
  if (!ctx || !sums) return ICPK_E_ARG;
  if (!ctx->have_assoc) return fail(ctx, ICPK_E_NOT_SET, "no nearest-neighbour sweep has run");
  if (!ctx->have_normals) return fail(ctx, ICPK_E_NOT_SET, "no target normals");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  if (int ru = ensure_unpacked(ctx)) return ru;
  if (ctx->src.n == 0) {
    std::memset(sums, 0, NP2L * sizeof(double));
    if (count) *count = 0;
    return ICPK_OK;
  }
  int rc = enqueue_reduce_p2l(ctx, max_dist);
  if (rc) return rc;
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(sums, ctx->red_host, NP2L * sizeof(double));
  if (count) std::memcpy(count, ctx->red_host + NP2L, sizeof(int64_t));
  return ICPK_OK;
}

int icpk_solve_point_to_plane(const double sums[28], double R[9], double t[3]) {
  return solve_p2l(sums, R, t) ? ICPK_OK : ICPK_W_DEGENERATE;
}

/* test hook: icp.cpp:606-620 (point3 == 0) or :595-602 (point3 == 1) on n pairs; a and b are host xyz-SoA [3][n] */
static int pair_distance_impl(icpk_ctx* ctx, const float* a, const float* b, float* out, int32_t n, int point3) {
  if (!ctx || n < 0 || (n > 0 && (!a || !b || !out))) return ICPK_E_ARG;
  if (n == 0) return ICPK_OK;
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  ICPK_HIP(ctx, hipMalloc((void**)&da, (size_t)3 * n * sizeof(float)));
  ICPK_HIP(ctx, hipMalloc((void**)&db, (size_t)3 * n * sizeof(float)));
  ICPK_HIP(ctx, hipMalloc((void**)&dout, (size_t)n * sizeof(float)));
  ICPK_HIP(ctx, hipMemcpyAsync(da, a, (size_t)3 * n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  ICPK_HIP(ctx, hipMemcpyAsync(db, b, (size_t)3 * n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  launch_pair_distance(da, db, dout, n, point3, ctx->stream);
  ICPK_HIP(ctx, hipMemcpyAsync(out, dout, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dout);
  return ICPK_OK;
}

int icpk_pair_distance(icpk_ctx* ctx, const float* a, const float* b, float* out, int32_t n) {
  return pair_distance_impl(ctx, a, b, out, n, 0);
}
int icpk_pair_distance3(icpk_ctx* ctx, const float* a, const float* b, float* out, int32_t n) {
  return pair_distance_impl(ctx, a, b, out, n, 1);
}
float icpk_distance3(const float a[3], const float b[3]) {
  return a && b ? distance3(a[0], a[1], a[2], b[0], b[1], b[2]) : __builtin_nanf("");
}

void icpk_make_rotation_matrix(float x, float y, float z, float out[9]) { make_rotation_matrix(x, y, z, out); }
int icpk_backproject_keypoints(const uint16_t* depth, int32_t rows, int32_t cols, const float* kp_xy, int32_t n, float fx,
                               float cx, float* out_xyz, int32_t* kept) {
  if (!depth || rows <= 0 || cols <= 0 || n < 0 || (n > 0 && (!kp_xy || !out_xyz))) return ICPK_E_ARG;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float fxp = kp_xy[2 * i], fyp = kp_xy[2 * i + 1];
    if (!(fxp > -1.f && fxp < (float)cols + 1.f && fyp > -1.f && fyp < (float)rows + 1.f)) continue;  // (NaN, far outside)
    const long x = std::lrint(fxp), y = std::lrint(fyp);  // cvRound: to nearest, ties to even (default rounding mode)
    if (x < 0 || x >= cols || y < 0 || y >= rows) continue;
    const uint16_t d = depth[(size_t)y * cols + x];
    if (d == 0) continue;  // pointcloud.cpp:67-70
    const float pz = ((float)d) / 5000.0f;            // pointcloud.cpp:86
    const float px = ((float)x - cx) * pz / fx;       // :87 (an int minus the float constant)
    const float py = ((float)y - cx) * pz / fx;       // :88 (CX, FX)
    out_xyz[3 * m] = px;
    out_xyz[3 * m + 1] = py;
    out_xyz[3 * m + 2] = pz;
    if (kept) kept[m] = i;
    ++m;
  }
  return m;
}
void icpk_matrix_to_quaternion(const float m[9], float q[4]) { matrix_to_quaternion(m, q); }
void icpk_quaternion_to_euler(const float q[4], float e[3]) { quaternion_to_euler(q, e); }
void icpk_solve_reference(const float M[9], float R[9]) { solve_reference(M, R); }
void icpk_solve_kabsch(int64_t n, const double sa[3], const double sb[3], const double sab[9], double R[9], double t[3]) {
  solve_kabsch(n, sa, sb, sab, R, t);
}

}  // extern "C"
